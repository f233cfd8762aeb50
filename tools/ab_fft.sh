# (variants of the DEDICATED persistent 4096-point kernel; the product runs the radix-16 family kernel: tools/ab_fft4096_family.sh)
export PCX_FFT4096_DEDICATED=1
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
run() { python bench.py --no-cpu --workload fft4096 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %.4f ms  frac %.4f' % ('$1', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
PCX_HIP_LIBRARY=$D run "dynamic"
PCX_HIP_LIBRARY=$D PCX_FFT_PRIO=1 run "dynamic prio pass3"
PCX_HIP_LIBRARY=$D PCX_FFT_PRIO=2 run "dynamic prio pass2+3"
PCX_HIP_LIBRARY=$D PCX_SCHED_STATIC=1 PCX_FFT_SLOTS=8192 run "static 8192"
PCX_HIP_LIBRARY=$D PCX_SCHED_STATIC=1 PCX_FFT_SLOTS=8192 PCX_FFT_PRIO=1 run "static 8192 prio pass3"
PCX_HIP_LIBRARY=$D PCX_SCHED_STATIC=1 PCX_FFT_SLOTS=8192 PCX_FFT_PRIO=2 run "static 8192 prio pass2+3"
PCX_HIP_LIBRARY=$D PCX_SCHED_STATIC=1 PCX_FFT_SLOTS=4096 PCX_FFT_PRIO=1 run "static 4096 prio pass3"
done
