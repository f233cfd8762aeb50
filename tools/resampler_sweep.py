"""Configurations of the resampling cf32 FIR kernels, each in a process of its own (the diagnostic library reads its switches once), settled
before it is timed (300 untimed launches, 600 timed):  python tools/resampler_sweep.py            -- the sweep
                                                        python tools/resampler_sweep.py decim 8  -- one timing, environment as given"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(kind, fac):
    import torch
    from pothoscomms_amd import device, taps as tp
    d = torch.device("cuda", 0)
    n = 64 * 1024 * 1024 if kind == "decim" else 64 * 1024 * 1024 // fac
    M, L = (fac, 1) if kind == "decim" else (1, fac)
    h = tp.complex_bandpass(255 * L, 0.05 / max(L, M), 0.05 / max(L, M)) * L
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_decimation(M); f.set_interpolation(L)
    K = f.K
    lead = (-(K - 1)) % 16
    xa = torch.empty((lead + n + K - 1, 2), dtype=torch.float32, device=d); x = xa[lead:]
    y = torch.empty((n * L // M + 8, 2), dtype=torch.float32, device=d)
    device.fill_uniform_f32_dev(x, seed=7, offset=0)
    for _ in range(300): f.process_dev(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(600): f.process_dev(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 600
    print("%.4f ms  %.1f Gsamples/s %s" % (ms, (n if kind == "decim" else n * L) / ms / 1e6, "in" if kind == "decim" else "out"))


if len(sys.argv) > 2:
    one(sys.argv[1], int(sys.argv[2]))
    sys.exit(0)

diag = os.path.join(ROOT, "pothoscomms_amd", "libpcx_hip_diag.so")
cases = []
for M in (2, 4, 8, 16):
    cfgs = [{}]
    for g in (1, 2, 3):
        for hreg in (0, 1):
            cfgs.append({"PCX_DECIM_G": g, "PCX_DECIM_HREG": hreg})
    cases += [("decim", M, c) for c in cfgs]
for L in (2, 4, 8, 16):
    cfgs = [{}]
    for g in (0, 1, 2):
        for hreg in (0, 1):
            cfgs.append({"PCX_INTERP_G": g, "PCX_INTERP_HREG": hreg})
    cases += [("interp", L, c) for c in cfgs]
for rep in range(2):
    for kind, fac, cfg in cases:
        env = dict(os.environ, PCX_HIP_LIBRARY=diag, **{k: str(v) for k, v in cfg.items()})
        out = subprocess.run([sys.executable, __file__, kind, str(fac)], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print("%-6s %2d  %-40s %s" % (kind, fac, " ".join("%s=%s" % kv for kv in cfg.items()) or "(product)", out[-1] if out else "failed"), flush=True)
