"""complex_float32 interpolating FIR (decimation 1): output rate of the polyphase overlap-save kernel, 255 taps per phase."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pothoscomms_amd import device, taps as tp
d = torch.device("cuda", 0)
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import oracle
rng = np.random.default_rng(2)
for L in (2, 4, 8, 16):
    for ntaps in (1, L, 5 * L + 3, 255, 1023, 2049):
        h = (rng.standard_normal(ntaps) + 1j * rng.standard_normal(ntaps)) / np.sqrt(ntaps)
        K = -(-ntaps // L)
        n = 3 * (4096 // L) + 77 + K
        x = rng.standard_normal((n, 2)).astype(np.float32)
        ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.set_interpolation(L); ref.activate()
        ry, rc, rp, _ = ref.work(x, n * L)
        f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_interpolation(L)
        try:
            y, c, p = f.process(x, n * L)
            err = np.max(np.abs(y - ry)) / max(np.max(np.abs(ry)), 1e-30) if rp else 0
            print("L=%2d ntaps=%4d  consumed %d/%d produced %d/%d  err %.2e" % (L, ntaps, c, rc, p, rp, err), flush=True)
        except Exception as e:
            print("L=%2d ntaps=%4d  %s" % (L, ntaps, e), flush=True)
n = 16 * 1024 * 1024
for L in (2, 3, 4, 8, 16):
    K = 255 if L < 8 else 2040 // L
    h = tp.complex_bandpass(K * L, 0.05 / L, 0.05 / L) * L
    xa = torch.empty((n + K - 1 + 16, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(xa, seed=1)
    x = xa[: n + K - 1]
    y = torch.empty((n * L + 8, 2), dtype=torch.float32, device=d)
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_interpolation(L)
    for _ in range(5): f.process_dev(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f.process_dev(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("L=%d taps=%d: %.4f ms  %.1f Gsamples/s out  (%.2f TB/s)" % (L, K * L, ms, n * L / ms / 1e6, (8 * n + 8 * n * L) / ms / 1e9), flush=True)
