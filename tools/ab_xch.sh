# A/B of the two transform pairs of the headline kernel in ONE library (libpcx_hip_diag.so), interleaved: PCX_OLS_XCH=0 = Stockham
# both ways (eight barriers per block), 1 = second exchange inside sixteen lanes (three).  Needs `make diag`.
L=pothoscomms_amd/libpcx_hip_diag.so
run() { PCX_HIP_LIBRARY=$L PCX_OLS_XCH=$1 python bench.py --no-cpu --workload fir255 --steps 1500 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xch=$1  %.4f ms  frac %.4f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2 3 4 5; do run 0; run 1; done
