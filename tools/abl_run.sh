#!/bin/bash
# run one bench command under every abl_tmp/lib_*.so (and the in-tree build); usage: tools/abl_run.sh "<grep pattern>" <cmd...>
PAT=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
echo "== base"; "$@" 2>&1 | grep -E "$PAT"
for l in $R/abl_tmp/lib_*.so; do
  n=$(basename $l .so); echo "== ${n#lib_}"; MIDIEMO_LIB=$l "$@" 2>&1 | grep -E "$PAT"
done
