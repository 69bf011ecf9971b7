# development aid: ISA of one kernel of me_attn.hip -> build_tmp/k.s + register / scratch / LDS summary.  usage: EXTRA="-D..." bash tools/isa.sh <mangled-name-fragment>
cd /root/repo
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA --offload-arch=gfx950 -S --cuda-device-only -o build_tmp/me_attn.s midi-emotion_amd/csrc/me_attn.hip -I include -I midi-emotion_amd/csrc 2>/dev/null
L=$(grep -n "^_ZN.*$1.*:" build_tmp/me_attn.s | head -1 | cut -d: -f1)
sed -n "${L},\$p" build_tmp/me_attn.s | awk '{print} /; Occupancy/{exit}' > build_tmp/k.s
grep "NumVgprs:\|Occupancy\|ScratchSize\|LDSByteSize" build_tmp/k.s
grep -n "Loop Header" build_tmp/k.s
