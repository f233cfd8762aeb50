import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
from oracle import oracle
d = torch.device("cuda", 0)
rng = np.random.default_rng(1)
# parity
for M in (2, 4, 8, 16):
    for K in (1, 17, 255, 2049):
        h = tp.complex_bandpass(K, 0.05, 0.05) if K > 1 else np.array([0.7 - 0.2j])
        n = 3 * 4096 + 333 + K
        x = (rng.standard_normal((n, 2))).astype(np.float32)
        ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.set_decimation(M); ref.activate()
        ry, rc, rp, _ = ref.work(x, n)
        f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_decimation(M)
        y, c, p = f.process(x, n)
        err = np.max(np.abs(y - ry)) / max(np.max(np.abs(ry)), 1e-30) if rp else 0
        print("M=%2d K=%4d  consumed %d/%d produced %d/%d  err %.2e algo %d" % (M, K, c, rc, p, rp, err, f.last_algo), flush=True)
# speed
n = 64 * 1024 * 1024
for M in (2, 4, 8, 16, 10, 50, 160):
    K = 255
    h = tp.complex_bandpass(K, 0.05, 0.05)
    lead = (-(K - 1)) % 16
    xa = torch.empty((lead + n + K - 1, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(xa, seed=1)
    x = xa[lead:]
    y = torch.empty((n // M + 8, 2), dtype=torch.float32, device=d)
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_decimation(M)
    for _ in range(20): f.process_dev(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f.process_dev(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("M=%2d K=255: %.4f ms  %.1f Gsamples/s in" % (M, ms, n / ms / 1e6), flush=True)
