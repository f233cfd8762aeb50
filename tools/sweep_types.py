"""FIR throughput of the generic kernel across element types (M=L=1, 63 taps, 4 Mi samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n, K = 4 * 1024 * 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 63
TD = {"float32": torch.float32, "float64": torch.float64, "int16": torch.int16, "int32": torch.int32, "int8": torch.int8, "int64": torch.int64}
for name in ("float32", "complex_float32", "complex_float64", "complex_int16", "complex_int8", "complex_int32", "int16"):
    cplx = name.startswith("complex_")
    base = name.replace("complex_", "")
    w = 2 if cplx else 1
    for tt in (("COMPLEX", "REAL") if cplx else ("REAL",)):
        h = tp.complex_bandpass(K, 0.1, 0.05) if tt == "COMPLEX" else tp.lowpass(K, 0.1)
        shape = (n + K - 1, 2) if cplx else (n + K - 1,)
        x = (torch.rand(shape, device=d) * 200 - 100).to(TD[base])
        y = torch.empty((n, 2) if cplx else (n,), dtype=TD[base], device=d)
        f = device.FirFilter(name, tt); f.set_taps(h)
        for algo, an in ((_lib.FIR_AUTO, "auto"), (_lib.FIR_EXACT, "exact")):
            f.set_algo(algo)
            for _ in range(2): f.process_dev(x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f.process_dev(x, y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("%-16s %-8s %-6s K=%d  %8.2f Gs/s (algo %d)" % (name, tt, an, K, n / ms / 1e6, f.last_algo))
