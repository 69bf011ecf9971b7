python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rga or attn or bwd" 2>&1 | tail -2
bash ab/run_abl.sh split abl0 abl9 split abl0 abl9
