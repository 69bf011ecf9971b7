"""Average PMC counter values per kernel from rocprofv3 --pmc ... --output-format csv output directories (development aid).
usage: python tools/pmc_kernels.py <dir> [name-substring ...]"""
import csv, glob, sys, collections
root, keys = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if keys and not any(k in n for k in keys): continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n[:100])
    for c, v in sorted(cs.items()):
        print("    %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
