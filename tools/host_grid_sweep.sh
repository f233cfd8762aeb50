# the launch shape of a link-bound (host memory, PCIe) FIR call: PCX_HOST_GRID workgroups on the grid stride (diag library; 0 = the device-resident shape)
export PCX_HIP_LIBRARY=$PWD/pothoscomms_amd/libpcx_hip_diag.so
for g in 0 24 32 40 48 56 64 80 96; do echo "== PCX_HOST_GRID=$g"; PCX_HOST_GRID=$g timeout 200 python tools/host_slots_probe.py single 2>/dev/null | cut -c1-60; done
