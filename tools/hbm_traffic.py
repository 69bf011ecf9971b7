"""HBM bytes per launch and kernel from two rocprofv3 PMC passes over the same command:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -- python bench.py --steps 4 --warmup 1 --no_cpu_baseline --no_probe
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -- python bench.py ...
    python tools/hbm_traffic.py out_f/*/*.db out_w/*/*.db [--json profiles/hbm_traffic.json]
Corrections as prescribed by MI355X_MICROARCH.md (HBM section): both counters are in KiB-like units of 1 KB
(x 1024 -> bytes is NOT applied by rocprofv3: values are kilobytes); FETCH_SIZE reports half of wide coalesced
reads on gfx950 and is doubled (cross-check printed: adamw reads p, g, m, v = 4 x n_params x 4 B)."""
import collections
import json
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("_ZN12_GLOBAL__N_1"):
        import re
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
        n = int(m.group(1)); st = m.end()
        name = name[st:st + n]
    i = name.find("(")
    name = name if i < 0 else name[:i]
    j = name.find("<")
    return name if j < 0 or not name.startswith("void") else name
    

def collect(db, counter):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    q = f"""select s.kernel_name, d.id, sum(e.value)
            from {t('rocpd_pmc_event_')} e join {t('rocpd_info_pmc_')} p on e.pmc_id = p.id
            join {t('rocpd_kernel_dispatch_')} d on d.event_id = e.event_id
            join {t('rocpd_info_kernel_symbol_')} s on s.id = d.kernel_id
            where p.name = '{counter}' group by d.id"""
    out = collections.defaultdict(list)
    for k, _, v in c.execute(q):
        out[short(k)[:40]].append(v)
    return out


def main():
    args = sys.argv[1:]
    jpath = None
    if "--json" in args:
        i = args.index("--json"); jpath = args[i + 1]; del args[i:i + 2]
    f, w = collect(args[0], "FETCH_SIZE"), collect(args[1], "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) * 2 + sum(w.get(k, [0])))):
        n = max(len(f.get(k, [])), len(w.get(k, [])))
        rd = 2.0 * sum(f.get(k, [0])) / max(1, len(f.get(k, [1]))) / 1e3       # KB -> MB, x2 (gfx950)
        wr = sum(w.get(k, [0])) / max(1, len(w.get(k, [1]))) / 1e3
        rows.append((k, n, rd, wr))
    print("%-42s %9s %12s %12s" % ("kernel", "launches", "read MB", "write MB"))
    for k, n, rd, wr in rows:
        print("%-42s %9d %12.1f %12.1f" % (k, n, rd, wr))
    if jpath:
        import hashlib, os
        cs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "midi-emotion_amd", "csrc")
        h = hashlib.sha256()
        for n_ in sorted(os.listdir(cs)):
            if n_.endswith((".hip", ".h")):
                h.update(open(os.path.join(cs, n_), "rb").read())
        d = {k: {"launches": n, "read_MB": round(rd, 1), "write_MB": round(wr, 1)} for k, n, rd, wr in rows}
        d["_source_sha256"] = h.hexdigest()          # bench.py quotes roofline.traffic only for the sources it was measured on
        d["_profile"] = os.environ.get("PROFILE_TAG", "profiles/")
        json.dump(d, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
