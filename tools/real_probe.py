import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n, K = 128 * 1024 * 1024, 255
x = torch.empty((n + K - 1,), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(x, seed=1)
y = torch.empty((n,), dtype=torch.float32, device=d)
f = device.FirFilter("float32", "REAL"); f.set_taps(tp.lowpass(K, 0.1))
for _ in range(150): f.process_dev(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): f.process_dev(x, y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 200
print("real f32 255 taps %s: %.4f ms  %.1f Gsamples/s  %.2f TB/s" % (os.environ.get("PCX_SCHED_STATIC", "dynamic"), ms, n / ms / 1e6, 8 * n / ms / 1e9))
