"""Summarise a rocprofv3 rocpd sqlite database into a per-kernel table (like --stats CSV).
usage: python tools/rocpd_stats.py <results.db> [steps]"""
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


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
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


if __name__ == "__main__":
    main()
