"""Element-wise block throughput (device-resident, 64 Mi complex_float32 samples): GB/s of algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pothoscomms_amd import device
d = torch.device("cuda", 0)
n = 64 * 1024 * 1024
x = torch.empty((n, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(x, seed=1)
yc = torch.empty((n, 2), dtype=torch.float32, device=d)
yr = torch.empty((n,), dtype=torch.float32, device=d)
fd = device.FreqDemod("complex_float32")
cases = [
    ("rotate", lambda: device.rotate(x, 0.7, scalar=device.F32, out=yc, n=n), 16),
    ("scale", lambda: device.scale(x, 1.5, True, scalar=device.F32, out=yc, n=n), 16),
    ("conjugate", lambda: device.conj(x, scalar=device.F32, out=yc, n=n), 16),
    ("abs", lambda: device.abs_(x, True, scalar=device.F32, out=yr, n=n), 12),
    ("angle", lambda: device.angle(x, scalar=device.F32, out=yr, n=n), 12),
    ("freq_demod", lambda: fd.process_dev(x, yr, n), 12),
]
x2 = torch.empty((n, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(x2, seed=2); x2 += 2.0
yr2 = torch.empty((n,), dtype=torch.float32, device=d)
for op in ("ADD", "MUL", "DIV"):
    cases.append(("arith " + op, (lambda op=op: device.arith(op, x, x2, True, scalar=device.F32, out=yc, n=n)), 24))
cases.append(("split", lambda: device.split_complex(x, scalar=device.F32, re=yr, im=yr2, n=n), 16))
cases.append(("combine", lambda: device.combine_complex(yr, yr2, scalar=device.F32, out=yc, n=n), 16))
for name, fn, bytes_per in cases:
    for _ in range(100): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("%-12s %.4f ms  %7.1f Gsamples/s  %7.1f GB/s" % (name, ms, n / ms / 1e6, bytes_per * n / ms / 1e6), flush=True)
