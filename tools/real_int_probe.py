"""real int16 / float64 streams on the double-precision overlap-save pipeline (fir_real_ip_kernel), 255 real taps, 128 Mi samples:
dealt from 512 persistent workgroups (product) against the grid-stride walk (diag library, PCX_SCHED_STATIC=1)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
K = 255
for dtype, tdt, n in (("int16", torch.int16, 128 << 20), ("float64", torch.float64, 64 << 20)):
    if tdt == torch.int16:
        x = torch.randint(-20000, 20000, (n + K - 1,), device=d).to(torch.int16)
    else:
        x = torch.rand((n + K - 1,), dtype=tdt, device=d) - 0.5
    y = torch.empty((n,), dtype=tdt, device=d)
    f = device.FirFilter(dtype, "REAL"); f.set_taps(tp.lowpass(K, 0.1) * 0.9)
    for _ in range(150): f.process_dev(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f.process_dev(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print("real %s 255 taps, %s: %.4f ms  %.1f Gsamples/s" % (dtype, "grid stride" if os.environ.get("PCX_SCHED_STATIC") else "dealt", ms, n / ms / 1e6))
    del x, y, f
