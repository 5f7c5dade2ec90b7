"""Prints the last kernels of a rocprofv3 --kernel-trace CSV with their durations and the gaps between them.
    python tools/trace_gaps.py <dir> [count]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0, prev = int(rows[0]["Start_Timestamp"]), None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1000 if prev else 0.0
    print("%-44s start %8.1f us  dur %6.1f us  gap %6.1f" % (r["Kernel_Name"][:44], (s - t0) / 1000, (e - s) / 1000, gap))
    prev = e
