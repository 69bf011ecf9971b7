#!/bin/bash
# development aid: build a library variant with extra defines into abl_tmp/lib_<name>.so (load it with MIDIEMO_LIB=...)
# usage: tools/build_abl.sh <name> "<extra hipcc flags>"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2
mkdir -p $R/abl_tmp/obj_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1"
for f in me_gemm me_elem me_attn me_attn64 me_decode me_decode_token; do
  if [ "$f" = "${ONLY:-$f}" ] || [ ! -f $R/abl_tmp/obj_$NAME/$f.o ]; then
    if [ "$f" = "${ONLY:-$f}" ]; then X="$EXTRA"; else X=""; fi
    /opt/rocm/bin/hipcc $FLAGS $X -I$R/include -c $R/midi-emotion_amd/csrc/$f.hip -o $R/abl_tmp/obj_$NAME/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/abl_tmp/lib_$NAME.so $R/abl_tmp/obj_$NAME/*.o
ls -la $R/abl_tmp/lib_$NAME.so
