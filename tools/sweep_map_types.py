"""Element-wise blocks across element types (64 Mi complex elements, device-resident): GB/s of algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pothoscomms_amd import device
d = torch.device("cuda", 0)
n = 64 * 1024 * 1024
TD = {"float32": (torch.float32, device.F32, 4), "float64": (torch.float64, device.F64, 8), "int16": (torch.int16, device.I16, 2),
      "int8": (torch.int8, device.I8, 1), "int32": (torch.int32, device.I32, 4)}
for name, (td, sc, sb) in TD.items():
    x = (torch.rand((n, 2), device=d) * 200 - 100).to(td)
    yc = torch.empty_like(x); yr = torch.empty((n,), dtype=td, device=d)
    fd = device.FreqDemod("complex_" + name)
    cases = [("rotate", lambda: device.rotate(x, 0.7, scalar=sc, out=yc, n=n), 4 * sb), ("scale", lambda: device.scale(x, 1.5, True, scalar=sc, out=yc, n=n), 4 * sb),
             ("abs", lambda: device.abs_(x, True, scalar=sc, out=yr, n=n), 3 * sb), ("angle", lambda: device.angle(x, scalar=sc, out=yr, n=n), 3 * sb),
             ("freq_demod", lambda: fd.process_dev(x, yr, n), 3 * sb)]
    for cname, fn, bpe in cases:
        for _ in range(30): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("complex_%-8s %-11s %7.1f Gsamples/s  %7.1f GB/s" % (name, cname, n / ms / 1e6, bpe * n / ms / 1e6), flush=True)
    del x, yc, yr
