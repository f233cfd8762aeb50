# the double-precision overlap-save kernel dealt from 512 persistent workgroups against the grid-stride walk (diagnostic library, PCX_SCHED_STATIC)
run() { PCX_HIP_LIBRARY=pothoscomms_amd/libpcx_hip_diag.so $1 python bench.py --no-cpu --workload fir255_i16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %.4f ms  %.1f Gsamples/s  clock %s MHz' % ('$2', d['roofline']['avg_launch_ms'], d['value'] / 1e3, d['roofline'].get('clock_mhz_under_load')))"; }
for rep in 1 2 3; do run "env PCX_SCHED_STATIC=1" "grid stride (static)"; run "env" "dealt (product)"; done
