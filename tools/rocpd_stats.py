"""Summarise a rocprofv3 rocpd sqlite database into a per-kernel table (like --stats CSV).
usage: python tools/rocpd_stats.py <results.db> [steps] [--json profiles/step_trace.json]
--json: per-family (gemm_nt, gemm_tn, attention, ...) launches / average duration / ms per step of a trace of train steps ALONE,
with the sha256 of the kernel sources it was measured on (bench.py quotes roofline.rocprof_step_trace from it only on a match)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16"))
    return name.split("(")[0][:70]


FAMILIES = [("gemm_nt", ("gemm_nt256_kernel", "gemm_nt_kernel")), ("gemm_tn", ("gemm_tn256_kernel", "gemm_tn16_kernel", "gemm_tn_kernel", "tn256_reduce_kernel")),
            ("attention", ("rga_fwd", "rga_bwd")), ("layernorm", ("resid_ln",)), ("loss", ("ce_fwd", "ce_bwd")),
            ("optimizer", ("adamw_kernel", "sumsq_kernel", "scaler_step_kernel", "cast_transpose")), ("embedding", ("embed_",))]


def main():
    argv = sys.argv[1:]
    jpath = None
    if "--json" in argv:
        i = argv.index("--json"); jpath = argv[i + 1]; del argv[i:i + 2]
    db = argv[0]
    steps = float(argv[1]) if len(argv) > 1 else None
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = con.execute("select %s, (end - start) from kernels" % namecol).fetchall()
    agg = {}
    for n, d in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("%-62s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-62s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (k[:62], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100 * a[1] / tot))
    print("total kernel time: %.3f ms" % (tot / 1e6) + (", per step %.3f ms" % (tot / 1e6 / steps) if steps else ""))
    if jpath and steps:
        import hashlib, json, os
        fam = {}
        for n, d in rows:
            for f, keys in FAMILIES:
                if any(k in n for k in keys):
                    a = fam.setdefault(f, [0, 0.0])
                    a[0] += 1; a[1] += d
                    break
        cs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "midi-emotion_amd", "csrc")
        h = hashlib.sha256()
        for n_ in sorted(os.listdir(cs)):
            if n_.endswith((".hip", ".h")):
                h.update(open(os.path.join(cs, n_), "rb").read())
        out = {"families": {f: {"launches_per_step": round(a[0] / steps, 2), "avg_us": round(a[1] / a[0] / 1e3, 2),
                                "ms_per_step": round(a[1] / 1e6 / steps, 4)} for f, a in fam.items()},
               "steps": steps, "kernel_ms_per_step": round(tot / 1e6 / steps, 4), "_source_sha256": h.hexdigest(),
               "_profile": os.environ.get("PROFILE_TAG", "profiles/")}
        json.dump(out, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
