"""Memory-floor diagnostic: OLS kernel variants vs K (alignment of S = 4096-(K-1)), 64 Mi samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n = 64 * 1024 * 1024
for K in (1, 2, 129, 193, 255, 257):
    h = tp.complex_bandpass(K, 0.05, 0.05) if K > 1 else np.array([1.0 + 0j])
    x = torch.empty((n + K - 1, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(x, seed=1)
    y = torch.empty((n, 2), dtype=torch.float32, device=d)
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_algo(_lib.FIR_OLS_FFT)
    for _ in range(60): f.process_dev(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f.process_dev(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    S = 4096 - (K - 1)
    print("K=%4d S=%4d (S*8 %% 128 = %3d)  %.4f ms  %.1f Gs/s" % (K, S, (S * 8) % 128, ms, n / ms / 1e6))
