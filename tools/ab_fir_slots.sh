# A/B: the dealer (1024 persistent workgroups) against a static grid stride with more and more workgroups (PCX_OLS_SLOTS)
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
run() { python bench.py --no-cpu --workload fir255 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fir255 %-34s %.4f ms  frac %.4f' % ('$1', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
PCX_HIP_LIBRARY=$D run "dealer"
for o in 1024 4096 6144 8192 12288 16384 32768; do
PCX_HIP_LIBRARY=$D PCX_SCHED_STATIC=1 PCX_OLS_SLOTS=$o run "static, $o workgroups"
done
done
