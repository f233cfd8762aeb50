"""How the launch time of the headline FIR evolves after a device synchronisation: windows of `w` back-to-back launches,
one HIP event between windows (no event inside a window).  The driver's bench flags time 20 launches right behind a
synchronisation; this shows what such a window measures against the settled rate.
usage: python tools/transient_probe.py [window=20] [windows=24] [idle_ms=0]"""
import sys, time
import torch
sys.path.insert(0, ".")
from pothoscomms_amd import device, taps as tp, _lib
from pothoscomms_amd.stream import ShardedFir

w = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 24
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dev = torch.device("cuda", 0)
sf = ShardedFir(tp.c1_taps(), 64 << 20, dev, "COMPLEX", _lib.FIR_OLS_FFT)
device.fill_uniform_f32_dev(sf.buf, seed=2, offset=0)
for rep in range(3):
    for _ in range(150):
        sf.step()
    torch.cuda.synchronize()
    if idle:
        time.sleep(idle * 1e-3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nw + 1)]
    ev[0].record()
    for k in range(nw):
        for _ in range(w):
            sf.step()
        ev[k + 1].record()
    torch.cuda.synchronize()
    print("rep %d idle %.1f ms: per-launch us by window of %d: %s" % (rep, idle, w, " ".join("%.1f" % (ev[k].elapsed_time(ev[k + 1]) * 1e3 / w) for k in range(nw))))
