"""Resampling filters with long tap vectors, 16 Mi samples in: complex_float32 / float32, decimating and interpolating (profiles/r06/decim_long.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n = 16 << 20
for dtype, cplx in (("complex_float32", True), ("float32", False)):
    for K, M, L in ((4097, 8, 1), (4097, 64, 1), (8193, 16, 1), (2049, 8, 1), (4097, 1, 4), (8193, 1, 8), (8193, 1, 2), (16385, 1, 4), (8193, 3, 2)):
        h = tp.complex_bandpass(K, 0.4 / max(M, L), 0.02) if cplx else tp.lowpass(K, 0.4 / max(M, L))
        f = device.FirFilter(dtype, "COMPLEX" if cplx else "REAL"); f.set_taps(h); f.set_decimation(M); f.set_interpolation(L)
        Kr = f.K
        nin = n // L // M * M
        shape = (nin + Kr - 1, 2) if cplx else (nin + Kr - 1,)
        x = torch.randn(shape, device=d)
        no = nin // M * L
        y = torch.empty((no, 2) if cplx else (no,), dtype=torch.float32, device=d)
        try:
            f.process_dev(x, y); torch.cuda.synchronize()
        except Exception as e:
            print(dtype, K, M, L, "ERR", e); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): f.process_dev(x, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("%-16s ntaps=%-5d M=%-3d L=%-2d rowK=%-5d algo=%d %9.3f ms  %8.2f Gsamples/s in  %8.2f out" % (dtype, K, M, L, Kr, f.last_algo, ms, nin / ms / 1e6, no / ms / 1e6), flush=True)
