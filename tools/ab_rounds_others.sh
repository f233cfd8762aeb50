# groups / blocks per workgroup against the fixed four-per-slot grids of the Q15 FFT, the double-precision FFT family and the
# radix-16 long-tap overlap-save plans (diagnostic library)
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
echo "== complex_int16 FFT (64 Mi samples per launch)"
echo "fixed factor 4 (product)"; PCX_HIP_LIBRARY=$D python tools/sweep_fft.py 64 256 1024 4096 2>/dev/null | grep int16
for r in 1 2 4 8; do echo "groups per workgroup $r"; PCX_HIP_LIBRARY=$D PCX_Q15_ROUNDS=$r python tools/sweep_fft.py 64 256 1024 4096 2>/dev/null | grep int16; done
echo "== complex_float64 FFT (32 Mi samples per launch)"
echo "fixed factor 4 (product)"; PCX_HIP_LIBRARY=$D python tools/sweep_fft_f64.py 2>/dev/null | grep -E "N= *(64|256|1024|4096|8192) "
for r in 1 2 4 8; do echo "groups per workgroup $r"; PCX_HIP_LIBRARY=$D PCX_F64_ROUNDS=$r python tools/sweep_fft_f64.py 2>/dev/null | grep -E "N= *(64|256|1024|4096|8192) "; done
echo "== complex_float32 FIR, 4097 and 8193 taps (16 Mi samples per launch)"
echo "fixed factor 4 (product)"; PCX_HIP_LIBRARY=$D python tools/sweep_fir.py 2>/dev/null | grep -E "K= *(4097|8193) "
for r in 1 2 4 8; do echo "blocks per workgroup $r"; PCX_HIP_LIBRARY=$D PCX_R16_ROUNDS=$r python tools/sweep_fir.py 2>/dev/null | grep -E "K= *(4097|8193) "; done
