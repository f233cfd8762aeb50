# the product kernel of the headline workload with other cache-policy bits on its stores (libpcx_hip_diag.so, PCX_OLS_VARIANT 30..35)
L=pothoscomms_amd/libpcx_hip_diag.so
run() { PCX_HIP_LIBRARY=$L PCX_OLS_VARIANT=$1 python bench.py --no-cpu --workload fir255 --steps 1500 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant %-3s %-12s %.4f ms  frac %.4f' % ('$1', '$2', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do run -1 "nt (product)"; run 30 "none"; run 31 "sc0+nt"; run 32 "sc1+nt"; run 33 "sc0+sc1+nt"; run 34 "sc1"; run 35 "sc0+sc1"; done
