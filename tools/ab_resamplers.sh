# the decimating / interpolating kernels on the grid stride (product) against the round-3 dealer (PCX_SCHED_RESAMPLERS, diag library)
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
run() { python bench.py --no-cpu --workload $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-22s %.4f ms  frac %.4f' % ('$2', '$1', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do for w in decim8 interp4; do run "product (grid stride)" $w; PCX_HIP_LIBRARY=$D PCX_SCHED_RESAMPLERS=1 run "dealer" $w; done; done
