# mixed-radix / smooth-size FFT plans: groups per workgroup (PCX_MIXED_ROUNDS, diagnostic library) against equal shares
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
echo "equal shares (product)"; PCX_HIP_LIBRARY=$D python tools/sweep_fft_mixed.py 100 1000 1536 3000 10000 2>/dev/null | grep complex
for r in 1 2 4 8; do echo "groups per workgroup $r"; PCX_HIP_LIBRARY=$D PCX_MIXED_ROUNDS=$r python tools/sweep_fft_mixed.py 100 1000 1536 3000 10000 2>/dev/null | grep complex; done
