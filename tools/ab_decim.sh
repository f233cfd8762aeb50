# decimating cf32 FIR: the one-block kernel (PCX_DECIM_UNBATCHED) against the batched inverse stage, blocks per group 2^PCX_DECIM_G,
# pass-3 constants in registers or re-read per block (PCX_DECIM_TW3); tools/decim_probe.py: 64 Mi input samples, 255 taps
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
echo "one block per iteration (round-2 kernel)"; PCX_HIP_LIBRARY=$D PCX_DECIM_UNBATCHED=1 python tools/decim_probe.py 2>/dev/null | grep "Gsamples" | head -4
for g in 1 2 3; do for t in 0 1; do
echo "batched: 2^$g blocks per group, tw3 in registers = $t"; PCX_HIP_LIBRARY=$D PCX_DECIM_BATCHED=1 PCX_DECIM_G=$g PCX_DECIM_TW3=$t python tools/decim_probe.py 2>/dev/null | grep "Gsamples" | head -4
done; done
