# A/B of the dynamic block dealing (pcx_sched.hpp) against the static grid stride, same box, diagnostic library
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
run() { python bench.py --no-cpu --workload $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-28s %.4f ms  frac %.4f' % ('$1', '$2', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
for w in fir255 fmchain fft4096 decim8 interp4; do
# (fft4096: the dedicated persistent kernel, which the product no longer uses -- tools/ab_fft4096_family.sh)
PCX_FFT4096_DEDICATED=1 PCX_HIP_LIBRARY=$D run $w "dynamic"
PCX_FFT4096_DEDICATED=1 PCX_HIP_LIBRARY=$D PCX_SCHED_STATIC=1 run $w "static"
done
done
PCX_FFT4096_DEDICATED=1 PCX_HIP_LIBRARY=$D PCX_SCHED_STATIC=1 PCX_FFT_SLOTS=8192 run fft4096 "static, 8192 slots"
