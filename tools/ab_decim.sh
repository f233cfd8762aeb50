# decimating cf32 FIR (tools/decim_probe.py: 64 Mi input samples, 255 taps), diagnostic library:
#   the one-block kernel (PCX_DECIM_UNBATCHED) against the batched inverse stage at four workgroups per CU (PCX_DECIM_OCC=4, blocks per
#   group 2^PCX_DECIM_G, pass-3 constants in registers or not) and at three (the product: PCX_DECIM_G, PCX_DECIM_HREG)
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
echo "product"; PCX_HIP_LIBRARY=$D python tools/decim_probe.py 2>/dev/null | grep "Gsamples" | head -4
echo "one block per iteration (round-2 kernel)"; PCX_HIP_LIBRARY=$D PCX_DECIM_UNBATCHED=1 python tools/decim_probe.py 2>/dev/null | grep "Gsamples" | head -4
for g in 1 2 3; do for t in 0 1; do
echo "batched, 4 workgroups per CU: 2^$g blocks per group, tw3 in registers = $t"; PCX_HIP_LIBRARY=$D PCX_DECIM_OCC=4 PCX_DECIM_G=$g PCX_DECIM_TW3=$t python tools/decim_probe.py 2>/dev/null | grep "Gsamples" | head -4
done; done
for g in 1 2 3; do for h in 0 1; do
echo "batched, 3 workgroups per CU: 2^$g blocks per group, H in registers = $h"; PCX_HIP_LIBRARY=$D PCX_DECIM_G=$g PCX_DECIM_HREG=$h python tools/decim_probe.py 2>/dev/null | grep "Gsamples" | head -4
done; done
