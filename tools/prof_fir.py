"""Minimal driver for rocprofv3: runs the headline FIR workload a few times (no CPU leg)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pothoscomms_amd import _lib, device, taps as tp
from pothoscomms_amd.stream import ShardedFir

wl = sys.argv[1] if len(sys.argv) > 1 else "fir255"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = n
dev = torch.device("cuda", 0)
C = 64 * 1024 * 1024
if wl in ("fir255", "direct255"):
    sf = ShardedFir(tp.c1_taps(), C, dev, "COMPLEX", _lib.FIR_OLS_FFT if wl == "fir255" else _lib.FIR_DIRECT)
    device.fill_uniform_f32_dev(sf.buf, seed=2, offset=0)
    for _ in range(n):
        sf.step()
elif wl == "fft4096":
    nframes = 65536
    x = torch.empty((nframes * 4096, 2), dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    device.fill_uniform_f32_dev(x, seed=3)
    fft = device.Fft("complex_float32", 4096, False)
    for _ in range(n):
        fft.transform_dev(x, y, nframes)
elif wl == "fmchain":
    ch = device.FmChain(); ch.set_phase(tp.C4_PHASE); ch.set_taps(tp.c4_taps(), False)
    xa = torch.empty((2 + C + 126, 2), dtype=torch.float32, device=dev); x = xa[2:]
    y = torch.empty((C,), dtype=torch.float32, device=dev)
    device.fill_uniform_f32_dev(x, seed=5)
    for _ in range(n):
        ch.process_dev(x, y, C + 126, C)
elif wl in ("decim8", "interp4"):
    # the same configurations as bench.py --workload decim8 / interp4
    n = C if wl == "decim8" else C // 4
    M, L = (8, 1) if wl == "decim8" else (1, 4)
    h = tp.complex_bandpass(255 * L, 0.05 / max(L, M), 0.05 / max(L, M)) * L
    f = device.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(h); f.set_decimation(M); f.set_interpolation(L)
    K = f.K
    lead = (-(K - 1)) % 16
    xa = torch.empty((lead + n + K - 1, 2), dtype=torch.float32, device=dev); x = xa[lead:]
    y = torch.empty((n * L // M + 8, 2), dtype=torch.float32, device=dev)
    device.fill_uniform_f32_dev(x, seed=7, offset=0)
    for _ in range(reps):
        f.process_dev(x, y)
elif wl == "fir255_i16":
    # as bench.py --workload fir255_i16
    f = device.FirFilter("complex_int16", "COMPLEX")
    f.set_taps(tp.c1_taps() * 0.9)
    K = f.K
    x = torch.randint(-20000, 20000, (C + K - 1, 2), device=dev).to(torch.int16)
    y = torch.empty((C, 2), dtype=torch.int16, device=dev)
    for _ in range(reps):
        f.process_dev(x, y)
torch.cuda.synchronize()
print("done", wl)
