# A/B of two builds of libpcx_hip.so on one box, interleaved: tools/ab_lib.sh <other.so> [workloads...]
O=$1; shift; WL=${@:-fir255 fmchain}
run() { python bench.py --no-cpu --workload $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-10s %.4f ms  frac %.4f' % ('$2', '$1', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2 3; do for w in $WL; do PCX_HIP_LIBRARY=$O run other $w; run product $w; done; done
