"""Every element type x (undecimated, / 8, x 4) x (255, 4097 taps): which kernel family serves a /comms/fir_filter call and at what rate
(16 Mi samples in; algo 1 = time-domain tile, 2 = frequency domain, 3 = reference order).  profiles/r06/fir_matrix.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n = 16 << 20
TT = {"float32": torch.float32, "float64": torch.float64, "int16": torch.int16, "int8": torch.int8, "int32": torch.int32, "int64": torch.int64}
for base in ("float32", "float64", "int16", "int8", "int32"):
    for cplx in (True, False):
        dtype = ("complex_" if cplx else "") + base
        for K in (255, 4097):
            row = []
            for M, L in ((1, 1), (8, 1), (1, 4)):
                h = tp.lowpass(K, 0.4 / max(M, L)) * 0.9
                f = device.FirFilter(dtype, "REAL"); f.set_taps(h); f.set_decimation(M); f.set_interpolation(L)
                Kr = f.K
                nin = n // L // M * M
                shape = (nin + Kr - 1, 2) if cplx else (nin + Kr - 1,)
                x = (torch.randn(shape, device=d) * (1 if TT[base].is_floating_point else 50)).to(TT[base])
                no = nin // M * L
                y = torch.empty((no, 2) if cplx else (no,), dtype=TT[base], device=d)
                try:
                    f.process_dev(x, y); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(2): f.process_dev(x, y)
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 2
                    row.append("M=%d L=%d: algo %d %7.1f in / %7.1f out" % (M, L, f.last_algo, nin / ms / 1e6, no / ms / 1e6))
                except Exception as e:
                    row.append("M=%d L=%d: %s" % (M, L, str(e)[:40]))
            print("%-16s K=%-5d %s" % (dtype, K, "  |  ".join(row)), flush=True)
