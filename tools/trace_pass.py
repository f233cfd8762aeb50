"""Timeline of the kernels and copies of the last passes in a rocprofv3 --kernel-trace --memory-copy-trace csv tree.
usage: python tools/trace_pass.py <dir> [rows]"""
import csv, glob, sys
d = sys.argv[1]; rows = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"), r.get("Grid_Size", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "-", ""))
ev.sort()
ev = ev[-rows:]
t0 = ev[0][0]
for s, e, n, q, g in ev:
    print("%9.1f -> %9.1f us  (%7.1f)  q=%-4s grid=%-8s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, g, n))
