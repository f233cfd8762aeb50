"""FFT throughput vs size (device-resident, ~64 Mi samples total per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pothoscomms_amd import device
d = torch.device("cuda", 0)
total = 64 * 1024 * 1024
sizes = [int(a) for a in sys.argv[1:]] or [16, 64, 256, 1000, 1024, 2048, 4096, 8192]
for dt, esz in [("complex_float32", 8), ("complex_int16", 4)][:int(os.environ.get("SWEEP_TYPES", "2"))]:
    for N in sizes:
        nframes = total // N
        tdt = torch.float32 if esz == 8 else torch.int16
        x = torch.zeros((nframes * N, 2), dtype=tdt, device=d)
        if esz == 8: device.fill_uniform_f32_dev(x, seed=1)
        else: x.copy_((torch.rand((nframes * N, 2), device=d) * 20000 - 10000).to(torch.int16))
        y = torch.empty_like(x)
        f = device.Fft(dt, N, False)
        for _ in range(100): f.transform_dev(x, y, nframes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f.transform_dev(x, y, nframes)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print("%-16s N=%5d  %8.1f Gs/s  %7.1f GB/s" % (dt, N, nframes * N / ms / 1e6, 2 * esz * nframes * N / ms / 1e6))
