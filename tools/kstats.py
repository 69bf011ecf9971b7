"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory (development aid)."""
import csv, glob, sys
root = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:n]:
        print("%-90s %6s %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
