# blocks per workgroup (PCX_ROUNDS, diagnostic library) for the kernels whose grid is sized by rounds_grid: the double-precision
# overlap-save pipeline (complex_int16 / complex_float64 FIR) at 64 Mi samples per launch
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
run() { python bench.py --no-cpu --workload $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-26s %.4f ms  frac %.4f  %.1f Gsamples/s' % ('$1', '$2', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value']/1e3))"; }
for rep in 1 2; do
PCX_HIP_LIBRARY=$D run fir255_i16 "product default"
for o in 1 2 3 4 6 8 16; do
PCX_HIP_LIBRARY=$D PCX_ROUNDS=$o run fir255_i16 "$o blocks per workgroup"
done
done
