# link-bound launch shapes of the host-pointer entry points (diag library): PCX_HOST_GRID = workgroups of the persistent block kernels
# (0 = the device-resident shape), PCX_HOST_MAP_GRID = blocks of the grid-stride map kernels (0 = the device-resident shape)
export PCX_HIP_LIBRARY=$PWD/pothoscomms_amd/libpcx_hip_diag.so
for g in 0 48; do for m in 0 16 32 64 128; do echo "== PCX_HOST_GRID=$g PCX_HOST_MAP_GRID=$m"; PCX_HOST_GRID=$g PCX_HOST_MAP_GRID=$m timeout 200 python tools/host_other_probe.py 2>/dev/null | cut -c1-400; [ $g = 0 ] && [ $m = 0 ] && break; done; done
