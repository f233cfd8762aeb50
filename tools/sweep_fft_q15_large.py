"""complex_int16 transforms beyond one workgroup's LDS (the stage-per-launch plan, fft_mixed.hip launch_fft_q15_global): rate by size."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from pothoscomms_amd import device
dev = torch.device("cuda", 0)
for nbins in (32768, 65536, 131072, 98304, 40000, 57344, 1 << 20):
    nframes = max(1, (16 << 20) // nbins)
    x = torch.randint(-20000, 20000, (nframes * nbins, 2), device=dev).to(torch.int16)
    y = torch.empty_like(x)
    f = device.Fft("complex_int16", nbins, False)
    for _ in range(3):
        f.transform_dev(x, y, nframes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        f.transform_dev(x, y, nframes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("complex_int16 %8d bins x %5d frames: %.3f ms  %.1f Gsamples/s  %.2f TB/s (8 B per sample)" % (nbins, nframes, dt * 1e3, nframes * nbins / dt / 1e9, 8.0 * nframes * nbins / dt / 1e12))
