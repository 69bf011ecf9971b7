#!/bin/bash
# development aid: compile me_gemm.hip to assembly (build_tmp/me_gemm.s) and print registers / scratch of the NT kernels
cd /root/repo/midi-emotion_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 --cuda-device-only -S me_gemm.hip -o /root/repo/build_tmp/me_gemm.s -Rpass-analysis=kernel-resource-usage $@ 2>&1 | grep -E "error|${PAT:-gemm_nt8p}" -A8 | grep -E "error|Function Name|VGPRs:|ScratchSize" | sed 's/.*remark: //'
