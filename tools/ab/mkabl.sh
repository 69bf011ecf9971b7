# usage: mkabl.sh N [N ...] -> ab/lib_ablN.so
cd "$(dirname "$0")/../.." && mkdir -p ab
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-value -Wno-pass-failed -DME_ABL=$n -I include -c midi-emotion_amd/csrc/me_attn.hip -o /tmp/me_attn_abl$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_abl$n.so midi-emotion_amd/csrc/me_gemm.o midi-emotion_amd/csrc/me_elem.o /tmp/me_attn_abl$n.o && echo built abl$n &
done
wait
