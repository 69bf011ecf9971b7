"""development aid: p50 device step latency of the config-5 decode (bench.py's decode_bench) -- A/B runs with MIDIEMO_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
import bench
for _ in range(2):
    r = bench.decode_bench("bf16", 1024)
    print("decode bf16: p50 %.4f ms  p90 %.4f ms  %.0f tok/s" % (r["step_ms_p50"], r["step_ms_p90"], r["tokens_per_s"]))
