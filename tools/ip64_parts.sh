# the double-precision overlap-save kernel's time split (diagnostic library, WRONG outputs for PART != 0): whole / no barriers / arithmetic only
run() { PCX_HIP_LIBRARY=pothoscomms_amd/libpcx_hip_diag.so PCX_IP64_PART=$1 python bench.py --no-cpu --workload fir255_i16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PART=$1  %.4f ms  clock %s MHz' % (d['roofline']['avg_launch_ms'], d['roofline'].get('clock_mhz_under_load')))"; }
for rep in 1 2; do for p in 0 1 2; do run $p; done; done
