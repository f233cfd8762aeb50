# interpolating cf32 FIR (tools/interp_probe.py), diagnostic library: four workgroups per CU against three (the product), with H re-read or in registers
D0=$PWD/pothoscomms_amd/libpcx_hip_diag.so; echo "product"; PCX_HIP_LIBRARY=$D0 python tools/interp_probe.py 2>/dev/null | grep "Gsamples" | head -6
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
echo "4 workgroups per CU (round-2 build)"; PCX_HIP_LIBRARY=$D PCX_INTERP_G=0 PCX_INTERP_OCC=4 python tools/interp_probe.py 2>/dev/null | grep "Gsamples" | head -6
echo "3 workgroups per CU, H re-read per block"; PCX_HIP_LIBRARY=$D PCX_INTERP_G=0 PCX_INTERP_HREG=0 python tools/interp_probe.py 2>/dev/null | grep "Gsamples" | head -6
echo "3 workgroups per CU, H in registers"; PCX_HIP_LIBRARY=$D PCX_INTERP_G=0 PCX_INTERP_HREG=1 python tools/interp_probe.py 2>/dev/null | grep "Gsamples" | head -6
for g in 1 2 3; do for h in 0 1; do
echo "short forward stage batched over 2^$g blocks (3 workgroups per CU), H in registers = $h"; PCX_HIP_LIBRARY=$D PCX_INTERP_G=$g PCX_INTERP_HREG=$h python tools/interp_probe.py 2>/dev/null | grep "Gsamples" | head -6
done; done
