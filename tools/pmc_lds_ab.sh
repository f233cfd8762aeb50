#!/bin/bash
# LDS counters of one workload on the last commit's library (tools/_ab) and on the tree's: tools/pmc_lds_ab.sh <workload>
W=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for side in head tree; do
  O=gpurun_out/pmc_lds_$W/$side; rm -rf $O; mkdir -p $O
  if [ $side = head ]; then export PCX_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/_ab/libpcx_hip_head.so; else unset PCX_HIP_LIBRARY; fi
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O -- python3 tools/prof_fir.py $W 3 > $O/log.txt 2>&1
  echo "== $side"; python3 - $O <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if d.get("SQ_INSTS_LDS", 0) < 1e5: continue
    n = max(1, cnt[(k, "SQ_INSTS_LDS")])
    print(k, {c: round(v / n) for c, v in d.items()}, "conflict/active = %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / max(1, d["SQ_LDS_IDX_ACTIVE"])))
PY
done
