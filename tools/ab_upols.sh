# the partitioned overlap-save kernel (fir_ols_part.hip), its build variants side by side (diagnostic library: PCX_UPOLS_VARIANT, PCX_UPOLS_OVERSUB)
#   K = 4097 (2 partitions): 0 = the product (kept half window, H in registers), 1 = H from L2 in batches of 4 bin pairs, 2 = whole windows fetched
#   K = 8193 (4 partitions): 0 = the product (kept half window, H from L2 one bin pair ahead), 1 = whole windows, batches of 2
run() { PCX_HIP_LIBRARY=pothoscomms_amd/libpcx_hip_diag.so PCX_UPOLS_VARIANT=$2 PCX_UPOLS_OVERSUB=$3 python bench.py --no-cpu --workload $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 variant=$2 oversub=$3  %.4f ms  %.1f Gsamples/s' % (d['roofline']['avg_launch_ms'], d['value']/1e3))"; }
for rep in 1 2; do for v in 0 1 2; do run fir4097 $v 1; done; for v in 0 1; do run fir8193 $v 1; done; done
run fir4097 0 2; run fir8193 0 2
