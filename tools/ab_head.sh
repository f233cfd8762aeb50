# A/B of the working tree's libpcx_hip.so against a build of the last commit (tools/_ab/libpcx_hip_head.so: `git worktree add /tmp/wt
# HEAD && make -C /tmp/wt/pothoscomms_amd/csrc && cp ...`), interleaved, long runs: tools/ab_head.sh [workloads...]
O=tools/_ab/libpcx_hip_head.so; WL=${@:-fir255 fmchain}
run() { python bench.py --no-cpu --workload $2 --steps 1500 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-10s %.4f ms  frac %.4f' % ('$2', '$1', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2 3 4; do for w in $WL; do PCX_HIP_LIBRARY=$O run head $w; run tree $w; done; done
