"""Measured box peaks for the roofline fractions (SURVEY section 5 `roofline.json`): run on the GPU box from the repo root.
  MFMA : build_tmp/ubench_mfma (tools/ubench_mfma.hip: sustained v_mfma_f32_32x32x16_bf16, zero and random operands,
         registers only / with the GEMM phase's 6 ds_read_b128 per 8 MFMAs)
  HBM  : 1 GiB torch copy / fill (a plain streaming kernel: what 'HBM-bound at the streaming rate of this part' means)
Writes gpurun_out/prof/roofline.json (copy to profiles/roofline.json).  The nominal peaks every `frac` in bench.py is
quoted against (2.5 PFLOP/s dense bf16, 8 TB/s) stay the guide's (/opt/skills/guides/MI355X_MICROARCH.md); this file records
what the box sustains so that a fraction of the ACHIEVABLE rate can be read next to it."""
import json, os, re, subprocess, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"device": torch.cuda.get_device_name(0), "nominal": {"mfma_bf16_dense_tflops": 2500.0, "hbm_tb_s": 8.0,
       "source": "/opt/skills/guides/MI355X_MICROARCH.md"}}
exe = os.path.join(ROOT, "build_tmp", "ubench_mfma")
if os.path.exists(exe):
    txt = subprocess.run([exe], capture_output=True, text=True, timeout=300).stdout
    cur, m = None, {}
    for line in txt.splitlines():
        if line.startswith("---"):
            cur = "zeros" if "zeros" in line else "random"
            m[cur] = {}
        else:
            g = re.match(r"(.+?)\s+([0-9.]+) TF/s", line)
            if g and cur:
                m[cur][g.group(1).strip()] = float(g.group(2))
    out["mfma_bf16_32x32x16_tflops"] = m
    out["mfma_sustained_random_tflops"] = max(m.get("random", {"x": 0}).values())
    out["mfma_with_gemm_lds_reads_random_tflops"] = max([v for k, v in m.get("random", {}).items() if "ds_read" in k] or [0])


def t(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


n = 2 ** 30 // 4
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
out["hbm_copy_tb_s"] = round(2 * 2 ** 30 / t(lambda: y.copy_(x)) / 1e12, 3)
out["hbm_fill_tb_s"] = round(2 ** 30 / t(lambda: y.fill_(1.0)) / 1e12, 3)
out["hbm_read_tb_s"] = round(2 ** 30 / t(lambda: x.sum()) / 1e12, 3)
os.makedirs(os.path.join(ROOT, "gpurun_out", "prof"), exist_ok=True)
p = os.path.join(ROOT, "gpurun_out", "prof", "roofline.json")
json.dump(out, open(p, "w"), indent=1)
print(json.dumps(out))
