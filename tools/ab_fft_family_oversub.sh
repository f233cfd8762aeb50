# radix-16 family FFT kernels: groups per workgroup (PCX_ROUNDS, diagnostic library), every size, 64 Mi samples per launch
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
echo "product default"
PCX_HIP_LIBRARY=$D SWEEP_TYPES=1 python tools/sweep_fft.py 64 256 512 1024 2048 4096 8192 16384 2>/dev/null | grep complex
for o in 1 2 4 8; do
echo "groups per workgroup $o"
PCX_HIP_LIBRARY=$D PCX_ROUNDS=$o SWEEP_TYPES=1 python tools/sweep_fft.py 64 256 512 1024 2048 4096 8192 2>/dev/null | grep complex
done
