"""Durations and gaps of the last N launches of a kernel in a rocprofv3 --kernel-trace csv tree.
usage: python tools/trace_tail.py <dir> <kernel-substring> [N]"""
import csv, glob, sys
d, pat, n = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if pat in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
rows = rows[-n:]
prev = None
for i, (s, e) in enumerate(rows):
    print("%3d  dur %7.1f us   gap %7.1f us" % (i, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
if rows:
    print("span of the last %d: %.1f us; mean period %.2f us" % (len(rows), (rows[-1][1] - rows[0][0]) / 1e3, (rows[-1][1] - rows[0][0]) / 1e3 / len(rows)))
