# A/B: the dedicated 4096-point kernel (register prefetch + dealer, PCX_FFT4096_DEDICATED=1) against the radix-16 family's kernel at
# 4096 bins (the product path), with its groups-per-workgroup target swept (PCX_ROUNDS); 65,536 frames per launch
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
run() { python bench.py --no-cpu --workload fft4096 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fft4096 %-34s %.4f ms  frac %.4f' % ('$1', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2 3; do
PCX_HIP_LIBRARY=$D PCX_FFT4096_DEDICATED=1 run "dedicated persistent kernel"
PCX_HIP_LIBRARY=$D run "family (product default)"
for o in 1 2 3 4 6 8; do
PCX_HIP_LIBRARY=$D PCX_ROUNDS=$o run "family, $o frames per workgroup"
done
done
