# the FFT finding tried on the FIR: static stride with a few blocks per workgroup (PCX_OLS_VARIANT 20: H in registers, 21: H from L2)
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
run() { python bench.py --no-cpu --workload fir255 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fir255 %-40s %.4f ms  frac %.4f' % ('$1', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
PCX_HIP_LIBRARY=$D run "dealer (product)"
for v in 20 21; do for r in 2 3 4 6; do
PCX_HIP_LIBRARY=$D PCX_OLS_VARIANT=$v PCX_ROUNDS=$r run "variant $v, $r blocks per workgroup"
done; done
done
