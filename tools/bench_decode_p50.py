"""decode step latency (BASELINE config 5, bf16 + f32) from bench.py's decode_bench; env switches select variants"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
for cd in (sys.argv[1:] or ["bf16"]):
    d = bench.decode_bench(cd, 2048)
    print(cd, json.dumps({k: d[k] for k in ("tokens_per_s", "step_ms_p50", "step_ms_p90", "launches_per_step", "ids_checksum")}), flush=True)
