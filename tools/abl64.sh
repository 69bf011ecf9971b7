#!/bin/bash
# development aid: timing-only ablation builds of me_attn64.hip into abl_tmp/lib_<name>.so (MIDIEMO_LIB selects one)
# usage: tools/abl64.sh <name> "<-D flags>" [source basename, default me_attn64]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; SRC=${3:-me_attn64}
mkdir -p $R/abl_tmp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $R/midi-emotion_amd/csrc/$SRC.hip -o $R/abl_tmp/${SRC}_$NAME.o
OBJS=""
for f in me_gemm me_elem me_attn me_attn64 me_decode; do
  if [ "$f" = "$SRC" ]; then OBJS="$OBJS $R/abl_tmp/${SRC}_$NAME.o"; else OBJS="$OBJS $R/midi-emotion_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/abl_tmp/lib_$NAME.so $OBJS
echo built abl_tmp/lib_$NAME.so
