import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pothoscomms_amd import device
d = torch.device("cuda", 0)
total = 32 * 1024 * 1024
for N in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 20, 1 << 21, 1 << 22):
    nframes = total // N
    x = torch.rand((nframes * N, 2), dtype=torch.float64, device=d) - 0.5
    y = torch.empty_like(x)
    f = device.Fft("complex_float64", N, False)
    for _ in range(30): f.transform_dev(x, y, nframes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f.transform_dev(x, y, nframes)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("complex_float64 N=%5d  %8.1f Gs/s  %7.1f GB/s" % (N, nframes * N / ms / 1e6, 32 * nframes * N / ms / 1e6))
