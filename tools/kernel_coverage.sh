cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cov; mkdir -p gpurun_out/cov
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cov/kt -- python3 -m pytest tests -q -x -m gpu -p no:cacheprovider --deselect tests/test_stream_gpu.py --deselect tests/test_c_client_gpu.py > gpurun_out/cov/log.txt 2>&1
python3 - <<'PY'
import csv, glob, re
names=set()
for f in glob.glob("gpurun_out/cov/kt/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        names.add(re.sub(r"[<(].*", "", n))
open("gpurun_out/cov/kernels_hit.txt","w").write("\n".join(sorted(names))+"\n")
# per-kernel totals over the whole GPU suite (calls, total ns, average ns), our kernels only
rows = {}
for f in glob.glob("gpurun_out/cov/kt/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"[<(].*", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        if not n.startswith("pcx::"):
            continue
        c, t = rows.get(n, (0, 0))
        rows[n] = (c + int(r["Calls"]), t + int(float(r["TotalDurationNs"])))
with open("gpurun_out/cov/gpu_suite_kernel_totals.csv", "w") as o:
    o.write("kernel,calls,total_ns,avg_ns\n")
    for n, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        o.write("%s,%d,%d,%d\n" % (n, c, t, t // max(c, 1)))
PY
find gpurun_out/cov/kt -name "*.csv" -delete
