"""Per-kernel means of hardware counters from rocprofv3 --pmc runs (rocpd sqlite output).

usage: python tools/rocpd_pmc.py <results.db> [<results.db> ...] [--match substr] [--grid N]
Counters of the same kernel collected in different passes (one db per --pmc pass) are merged by
kernel name.  Values are summed over the counter's instances per dispatch, then averaged over
dispatches; the dispatch duration (ns) is reported beside them (profiled passes clock lower)."""
import collections
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    i = name.find("(")
    return name if i < 0 else name[:i]


def main():
    args = [a for a in sys.argv[1:]]
    match, grid = None, None
    if "--match" in args:
        i = args.index("--match"); match = args[i + 1]; del args[i:i + 2]
    if "--grid" in args:
        i = args.index("--grid"); grid = int(args[i + 1]); del args[i:i + 2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for db in args:
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        def t(prefix):
            return [x for x in tabs if x.startswith(prefix)][0]
        q = f"""select s.kernel_name, d.id, d.end - d.start, d.grid_size_x, p.name, sum(e.value)
                from {t('rocpd_pmc_event_')} e join {t('rocpd_info_pmc_')} p on e.pmc_id = p.id
                join {t('rocpd_kernel_dispatch_')} d on d.event_id = e.event_id
                join {t('rocpd_info_kernel_symbol_')} s on s.id = d.kernel_id
                group by d.id, p.name"""
        seen = set()
        for kname, did, ns, gx, pname, val in c.execute(q):
            k = short(kname)
            if match and match not in k:
                continue
            if grid is not None and gx != grid:
                continue
            acc[k][pname].append(val)
            if (db, did) not in seen:
                seen.add((db, did)); dur[k].append(ns)
    for k in sorted(acc):
        print("%s   (%d dispatches, mean %.1f us)" % (k, len(dur[k]), sum(dur[k]) / len(dur[k]) / 1e3))
        for pname in sorted(acc[k]):
            v = acc[k][pname]
            print("    %-32s %16.1f" % (pname, sum(v) / len(v)))


if __name__ == "__main__":
    main()
