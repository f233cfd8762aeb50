"""Reference-order (EXACT) FIR kernels: rate at a few tap counts, 16 Mi complex_float32 samples (complex and real taps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device
d = torch.device("cuda", 0)
n = 16 << 20
for K in (15, 63, 255):
    for tt in ("COMPLEX", "REAL"):
        taps = (np.random.default_rng(K).normal(size=K) + (1j * np.random.default_rng(K + 1).normal(size=K) if tt == "COMPLEX" else 0)) / K
        x = torch.empty((n + K - 1, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(x, seed=3)
        y = torch.empty((n, 2), dtype=torch.float32, device=d)
        f = device.FirFilter("complex_float32", tt); f.set_taps(taps); f.set_algo(_lib.FIR_EXACT)
        for _ in range(3): f.process_dev(x, y, n + K - 1, n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f.process_dev(x, y, n + K - 1, n)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("EXACT K=%4d %-8s taps  %8.3f ms  %7.1f Gsamples/s" % (K, tt, ms, n / ms / 1e6))
