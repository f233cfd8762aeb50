"""A/B of a build variant of the overlap-save FIR (diagnostic library: PCX_OLS_VARIANT, PCX_OLS_SLOTS, PCX_UPOLS_VARIANT ..., read once
per process): parity against the oracle on a short stream and the per-pass time at 64 Mi samples, for a few tap counts (AB_KS).  Run
once per setting:
    PCX_HIP_LIBRARY=pothoscomms_amd/libpcx_hip_diag.so PCX_OLS_VARIANT=9 python tools/ab_ols.py
(Rounds 2-5 compared block SIZES with it -- PCX_OLS_N, the radix-16 family plans: removed in round 6, see pcx_fir_api.hip fir_ols_partitions.)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
from oracle import oracle

d = torch.device("cuda", 0)
N = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("PCX_") and k != "PCX_HIP_LIBRARY") or "product"
Ks = [int(k) for k in os.environ.get("AB_KS", "16,64,127,255,511").split(",")]
n = 64 * 1024 * 1024
REAL = os.environ.get("AB_TYPE", "complex_float32") == "float32"     # real stream, real taps
for K in Ks:
    h = tp.lowpass(K, 0.1) if REAL else tp.complex_bandpass(K, 0.05, 0.05)
    f = device.FirFilter("float32" if REAL else "complex_float32", "REAL" if REAL else "COMPLEX"); f.set_taps(h); f.set_algo(_lib.FIR_OLS_FFT)
    # parity on 100,003 outputs (ragged tail) against the reference loop restated in oracle/
    m = 100003
    xs = oracle.fill_uniform_f32(m + K - 1, 7) if REAL else oracle.fill_uniform_f32(2 * (m + K - 1), 7).reshape(-1, 2)
    o = oracle.Fir(oracle.F32, not REAL, not REAL); o.set_taps(h); o.activate()
    want = o.work(xs, m)[0]
    xd = torch.from_numpy(xs).to(d); yd = torch.empty((m,) if REAL else (m, 2), dtype=torch.float32, device=d)
    try:
        f.process_dev(xd, yd)
    except Exception as e:
        print("N=%s K=%d: %s" % (N, K, e)); continue
    got = yd.cpu().numpy()
    err = float(np.abs(got - want).max() / np.abs(want).max())
    # AB_SHARD_ALIGN=1: place the stream so that the first sample AFTER the K-1 history (the shard) sits on a
    # 128-byte line, instead of the history itself
    lead = (-(K - 1)) % (32 if REAL else 16) if os.environ.get("AB_SHARD_ALIGN") else 0
    xa = torch.empty((lead + n + K - 1,) if REAL else (lead + n + K - 1, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(xa, seed=1)
    x = xa[lead:]
    y = torch.empty((n,) if REAL else (n, 2), dtype=torch.float32, device=d)
    for _ in range(150): f.process_dev(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f.process_dev(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print("N=%s K=%4d  err %.2e  %.4f ms  %.1f Gsamples/s" % (N, K, err, ms, n / ms / 1e6), flush=True)
    del x, xa, y
