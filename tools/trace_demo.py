"""A few calls with ROCTx ranges on (pcx_trace): run under
    rocprofv3 --marker-trace --kernel-trace --output-format csv -d <dir> -- python3 tools/trace_demo.py
and the marker trace names the C-ABI call around each kernel (tests/test_trace_gpu.py checks exactly that)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def main():
    import torch
    from pothoscomms_amd import _lib, device as dev, taps as tp
    L = _lib.load()
    _lib.check(L.pcx_trace(1))
    d = torch.device("cuda", 0)
    n = 1 << 20
    x = torch.empty((n + 254, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x, seed=1)
    y = torch.empty((n, 2), dtype=torch.float32, device=d)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps())
    f.process_dev(x, y, n + 254, n)
    t = dev.Fft("complex_float32", 4096, False)
    t.transform_dev(x, y, n // 4096)
    ch = dev.FmChain(); ch.set_phase(0.3); ch.set_taps(tp.c4_taps(), False)
    ch.process_dev(x, y.view(-1)[:n], n + 126, n)
    dev.conj(x[:n], scalar=dev.F32, out=y, n=n)
    torch.cuda.synchronize()
    host = np.random.default_rng(0).standard_normal((5000, 2)).astype(np.float32)
    dev.FreqDemod("complex_float32").process(host)
    _lib.check(L.pcx_trace(0))
    f.process_dev(x, y, n + 254, n)          # no range around this one
    torch.cuda.synchronize()
    print("trace demo done")


if __name__ == "__main__":
    main()
