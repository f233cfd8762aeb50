import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from pothoscomms_amd import _lib, device, taps as tp
from pothoscomms_amd.stream import ShardedFir
dev = torch.device("cuda", 0)
C = 64*1024*1024
sf = ShardedFir(tp.c1_taps(), C, dev, "COMPLEX", _lib.FIR_OLS_FFT)
device.fill_uniform_f32_dev(sf.buf, seed=2)
for _ in range(5): sf.step()
torch.cuda.synchronize()
for n in (20, 100):
    t0=time.perf_counter()
    for _ in range(n): sf.step()
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(n, "steps: enqueue %.1f us/step, total %.1f us/step"%((t1-t0)/n*1e6, (t2-t0)/n*1e6))
# with events
ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(20)]
t0=time.perf_counter()
for a,b in ev:
    a.record(); sf.step(); b.record()
torch.cuda.synchronize()
print("with events total %.1f us/step; event avg %.1f us"%((time.perf_counter()-t0)/20*1e6, sum(a.elapsed_time(b) for a,b in ev)/20*1e3))
