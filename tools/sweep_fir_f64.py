"""complex_float64 FIR throughput vs tap count: sliding-window (time-domain) kernel against the double-precision
overlap-save kernel (fir_ols_f64.hip); device-resident, 16 Mi samples.  PCX_OLS64_N=1024/2048/4096/8192 forces a plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n = 16 * 1024 * 1024
Ks = [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16, 32, 63, 127, 255, 511, 1023, 2049, 4097]
for K in Ks:
    h = tp.complex_bandpass(K, 0.05, 0.05)
    lead = (-(K - 1)) % 8
    xa = (torch.rand((lead + n + K - 1, 2), dtype=torch.float64, device=d) - 0.5)
    x = xa[lead:]
    y = torch.empty((n, 2), dtype=torch.float64, device=d)
    row = []
    for algo in (_lib.FIR_DIRECT, _lib.FIR_OLS_FFT):
        if algo == _lib.FIR_DIRECT and K > 511:
            row.append("-"); continue
        f = device.FirFilter("complex_float64", "COMPLEX"); f.set_taps(h); f.set_algo(algo)
        warm, reps = (20, 20) if algo == _lib.FIR_OLS_FFT else (2, 3)
        try:
            for _ in range(warm): f.process_dev(x, y)
        except Exception as e:
            row.append("n/a"); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f.process_dev(x, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row.append("%.1f Gs/s (%.2f TB/s)" % (n / ms / 1e6, 32 * n / ms / 1e9))
    print("K=%5d  time-domain %-24s overlap-save %-24s" % (K, row[0], row[1]), flush=True)
    del xa, x, y
