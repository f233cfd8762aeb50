"""Mixed-radix (non power-of-two) FFT throughput, complex_float32 / complex_int16, device-resident, ~32 Mi samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pothoscomms_amd import device
d = torch.device("cuda", 0)
total = 32 * 1024 * 1024
sizes = [int(a) for a in sys.argv[1:]] or [12, 60, 100, 600, 1000, 1200, 1536, 1920, 3000, 6000, 10000]
for dtype, tdt, esz in (("complex_float32", torch.float32, 8), ("complex_float64", torch.float64, 16), ("complex_int16", torch.int16, 4)):
    for N in sizes:
        nframes = total // N
        x = (torch.rand((nframes * N, 2), device=d) * 2000 - 1000).to(tdt)
        y = torch.empty_like(x)
        f = device.Fft(dtype, N, False)
        for _ in range(3): f.transform_dev(x, y, nframes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f.transform_dev(x, y, nframes)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%s N=%6d  %8.1f Gs/s  %7.1f GB/s" % (dtype, N, nframes * N / ms / 1e6, 2 * esz * nframes * N / ms / 1e6), flush=True)
