"""Summarise rocprofv3 csv output dirs: python tools/prof_summary.py <dir> <kernel-substring>"""
import collections, csv, glob, sys
root, key = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(root + "/**/*_kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if key in r["Name"]:
            print("stats  %-40s calls=%s avg_ns=%s min=%s max=%s" % (r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    for k, v in sorted(agg.items()):
        print("pmc    %-24s %.5g  (n=%d)" % (k, sum(v) / len(v), len(v)))
    if meta:
        print("       VGPR=%s LDS=%s scratch=%s grid=%s wg=%s" % (meta["VGPR_Count"], meta["LDS_Block_Size"], meta["Scratch_Size"], meta["Grid_Size"], meta["Workgroup_Size"]))
