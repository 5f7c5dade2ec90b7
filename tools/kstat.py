"""Prints kernel name, calls and average duration (us) from a rocprofv3 --stats kernel_stats.csv found under <dir>, for names matching any of the patterns.
    python tools/kstat.py <dir> [pattern ...]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
pats = sys.argv[2:]
for r in csv.DictReader(open(f)):
    if not pats or any(p in r["Name"] for p in pats):
        print("%-46s calls %5s avg %8.1f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3))
