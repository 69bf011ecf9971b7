for n in 0 2; do MIDIEMO_LIB=$PWD/ab/lib_ws$n.so python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rga or attn or bwd" 2>&1 | tail -1; done
TOP=7 bash ab/run_step.sh ws0 ws1 ws2 ws0 ws1 ws2 | grep "==\|rga"
