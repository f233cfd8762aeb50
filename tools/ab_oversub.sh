# A/B of the grid oversubscription factor of the persistent kernels that have no dealer of their own (diagnostic library)
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
run() { python bench.py --no-cpu --workload $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s oversub %-3s %.4f ms  frac %.4f  %.1f Gsamples/s' % ('$1', '$2', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value']/1e3))"; }
for w in decim8 interp4 fir255_i16 direct255; do
for o in 1 2 4 8; do PCX_HIP_LIBRARY=$D PCX_OVERSUB=$o run $w $o; done
done
