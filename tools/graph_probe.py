"""Which captured sequences replay correctly?  (reset + process_dev of the fused chain / FreqDemod inside a hipGraph)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import device as dev, taps as tp
d = torch.device("cuda", 0)
n = 1 << 20
x = torch.empty((n + 200, 2), dtype=torch.float32, device=d); dev.fill_uniform_f32_dev(x, seed=4)
s = torch.cuda.Stream(d)
for name in ("fmchain", "freqdemod"):
    for with_reset in (True, False):
        if name == "fmchain":
            h = dev.FmChain(); h.set_phase(0.3); h.set_taps(tp.c4_taps(), False)
            K = len(tp.c4_taps())
            y = torch.empty(n, dtype=torch.float32, device=d)
            call = lambda: h.process_dev(x, y, n + K - 1, n)
        else:
            h = dev.FreqDemod("complex_float32")
            y = torch.empty(n, dtype=torch.float32, device=d)
            call = lambda: h.process_dev(x, y, n, stream=s)
        def run():
            if with_reset: h.reset()
            call()
        with torch.cuda.stream(s):
            run()
        torch.cuda.synchronize()
        want = y.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            run()
        res = []
        for _ in range(3):
            y.fill_(float("nan")); g.replay(); torch.cuda.synchronize()
            res.append((bool(torch.equal(y, want)), float(y[0]), float(want[0]), int((y != want).sum())))
        print(name, "reset in graph" if with_reset else "no reset", res)
