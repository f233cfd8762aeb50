#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json, measured on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1: plain python)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

metric  : Msamples/s of the 255-tap complex_float32 FIR (BASELINE.json configs[1]: a 64 Mi-sample
          stream per GPU; at N GPUs the stream is N x 64 Mi samples, overlap-save sharded with the
          254-sample halo moved between neighbours by RCCL send/recv -- configs[3] at N=8).
step    : one pass of /comms/fir_filter over the rank's 64 Mi-sample shard, input and output
          resident in HBM, including the halo exchange when N > 1.
value   : samples filtered by ALL ranks / wall time of the K timed steps (max over ranks).
roofline: algorithmic bytes of the FIR kernel (16 B per sample: 8 read + 8 written)
          / its average launch duration from HIP events on the launch stream, against the
          8 TB/s HBM3E peak (MI355X_MICROARCH.md); `traffic` is the PMC-measured HBM bytes per
          launch from profiles/ when a matching measurement is committed.
cpu_baseline: the oracle's single-thread restatement of FIRFilter.cpp:286-302 timed on this
          host on a bounded slice of the same stream (rank 0, N=1 only).

Other workloads (--workload fft4096 | fmchain | rotate | direct255 | decim8 | interp4 | fir255_i16) print the same kind of line
for the secondary configs; the driver uses the default.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
SHARD = 64 * 1024 * 1024       # samples per GPU (configs[1])
PREWARM = 100                  # untimed setup passes before the W warm-up steps (clock settling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="fir255", choices=["fir255", "direct255", "fft4096", "fmchain", "rotate", "decim8", "interp4", "fir255_i16"])
    ap.add_argument("--shard", type=int, default=SHARD, help="samples per GPU (default 64 Mi)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    return ap.parse_args()


def cpu_baseline_fir(taps, seed, nsamples):
    """Single-thread oracle FIR (reference accumulation order) on `nsamples` of the stream."""
    import numpy as np

    from oracle import oracle as o      # CPU baseline leg: the checker, timed as the baseline
    K = len(taps)
    x = o.fill_uniform_f32(2 * (nsamples + K - 1), seed, 0).reshape(-1, 2)
    blk = o.Fir(o.F32, True, True)
    blk.set_taps(taps)
    blk.activate()
    t0 = time.perf_counter()
    _, c, p, _ = blk.work(x, nsamples)
    dt = time.perf_counter() - t0
    assert p == nsamples
    # all host cores: static chunking with K-1 overlap (not reference behaviour, reported beside)
    import threading
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 64))        # bounded: the leg must stay within a few tens of seconds
    per = nsamples // 8
    outs = [np.zeros((per, 2), np.float32) for _ in range(ncores)]
    L = o.lib()

    def work(i):
        L.orc_fir_cf32_chunk(blk.h, x.ctypes.data, outs[i].ctypes.data, per)

    th = [threading.Thread(target=work, args=(i,)) for i in range(ncores)]
    t1 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt_all = time.perf_counter() - t1
    return {
        "value": round(nsamples / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": "%d-sample slice of the same stream, %d taps, oracle/pcx_oracle.c fir_loop_f32 (reference "
                  "accumulation order, -O2, no FMA), single thread" % (nsamples, K),
        "all_cores": {"value": round(ncores * per / dt_all / 1e6, 3), "cores": ncores,
                      "note": "same loop on every host core, %d samples each (parallelised restatement, not reference behaviour)" % per},
    }


def load_traffic(workload):
    """PMC-measured HBM bytes per launch for this workload, if a measurement is committed."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return None
    try:
        with open(p) as f:
            t = json.load(f)
        e = t.get(workload)
        return e.get("hbm_bytes_per_launch") if e else None
    except (OSError, ValueError):
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist

    from pothoscomms_amd import _lib, device, taps as tp
    from pothoscomms_amd.stream import ShardedFir

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # one rank per GPU on the node the driver gives us
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        # "nccl" is RCCL on ROCm.  PCX_BENCH_BACKEND=gloo exists only to rehearse the multi-rank
        # control flow on a single-GPU box (ranks then share cuda:0 and the halo goes through gloo).
        backend = os.environ.get("PCX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    C = args.shard
    wl = args.workload
    roof_bytes = None
    kernel_name = None

    if wl in ("fir255", "direct255"):
        h = tp.c1_taps()
        algo = _lib.FIR_OLS_FFT if wl == "fir255" else _lib.FIR_DIRECT
        sf = ShardedFir(h, C, dev, "COMPLEX", algo)
        K = sf.K
        # the node-wide stream starts K-1 samples before shard 0 (rank 0's history); every rank
        # fills [its halo | its shard] from the same counter-hash stream, then the timed steps
        # overwrite the halo through RCCL
        device.fill_uniform_f32_dev(sf.buf, seed=2, offset=2 * rank * C)
        units = C
        roof_bytes = 16.0 * C
        kernel_name = "fir_cf32_ols4096_kernel" if wl == "fir255" else "fir_cf32_direct_kernel"

        def step():
            sf.step()
        desc = {"workload": "255-tap complex_float32 FIR (/comms/fir_filter, COMPLEX taps, M=L=1), %d-sample shard per GPU, "
                            "%s" % (C, "frequency-domain overlap-save (4096-pt Stockham)" if wl == "fir255" else "LDS-tiled direct form"),
                "taps": 255, "shard_samples": C, "halo_samples": K - 1, "setup_passes": PREWARM,
                "parallelism": "overlap-save shards x%d, RCCL send/recv halo" % world if world > 1 else "single GPU"}
        metric = "Msamples/s complex_float32 255-tap FIR"
    elif wl == "fft4096":
        nframes = 65536
        x = torch.empty((nframes * 4096, 2), dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        device.fill_uniform_f32_dev(x, seed=3, offset=2 * rank * nframes * 4096)
        fft = device.Fft("complex_float32", 4096, False)
        units = nframes * 4096
        roof_bytes = 16.0 * units
        kernel_name = "fft4096_kernel"

        def step():
            fft.transform_dev(x, y, nframes)
        desc = {"workload": "4096-pt complex_float32 FFT (/comms/fft), 65536 frames per GPU", "frames": nframes}
        metric = "Msamples/s complex_float32 4096-pt FFT"
    elif wl == "fmchain":
        n = C
        ch = device.FmChain()
        ch.set_phase(tp.C4_PHASE)
        ch.set_taps(tp.c4_taps(), False)
        # [lead | 126-sample history | n samples] with the samples (not the history) on a 128-byte line,
        # the same placement ShardedFir uses
        xa = torch.empty((2 + n + 126, 2), dtype=torch.float32, device=dev)
        x = xa[2:]
        y = torch.empty((n,), dtype=torch.float32, device=dev)
        device.fill_uniform_f32_dev(x, seed=5, offset=0)
        units = n
        roof_bytes = 12.0 * n
        kernel_name = "fmchain_cf32_ols4096_kernel"

        def step():
            ch.process_dev(x, y, n + 126, n)
        if world > 1:
            # the stream sharded over the ranks: K-sample halo from the left neighbour (stream.ShardedFmChain)
            from pothoscomms_amd.stream import ShardedFmChain
            sc = ShardedFmChain(tp.c4_taps(), tp.C4_PHASE, n, dev)
            device.fill_uniform_f32_dev(sc.buf, seed=5, offset=2 * rank * n)
            del xa, x, y
            step = sc.step
        desc = {"workload": "fused Rotate->FIR(127 real taps)->FreqDemod in one frequency-domain kernel, complex_float32 -> float32, %d samples" % n}
        if world > 1:
            desc["parallelism"] = "overlap-save shards x%d, RCCL send/recv halo of 127 samples" % world
        metric = "Msamples/s fused FM-demod chain"
    elif wl in ("decim8", "interp4"):
        # resampling complex_float32 FIR, 255 taps (per polyphase row when interpolating); independent replicas per rank
        n = C if wl == "decim8" else C // 4
        M, L = (8, 1) if wl == "decim8" else (1, 4)
        h = tp.complex_bandpass(255 * L, 0.05 / max(L, M), 0.05 / max(L, M)) * L
        f = device.FirFilter("complex_float32", "COMPLEX")
        f.set_taps(h); f.set_decimation(M); f.set_interpolation(L)
        K = f.K
        lead = (-(K - 1)) % 16
        xa = torch.empty((lead + n + K - 1, 2), dtype=torch.float32, device=dev)
        x = xa[lead:]
        y = torch.empty((n * L // M + 8, 2), dtype=torch.float32, device=dev)
        device.fill_uniform_f32_dev(x, seed=7, offset=0)
        units = n if wl == "decim8" else n * L            # decimator: input samples; interpolator: output samples
        roof_bytes = 8.0 * n + 8.0 * (n * L // M)
        kernel_name = "fir_cf32_ols4096_decim_kernel" if wl == "decim8" else "fir_cf32_ols4096_interp_kernel"

        def step():
            f.process_dev(x, y)
        desc = {"workload": "255-tap complex_float32 FIR, %s, %d input samples per GPU" %
                            ("decimation 8 folded into the spectrum (input rate)" if wl == "decim8"
                             else "interpolation 4 from the replicated spectrum, 255 taps per phase (output rate)", n),
                "decimation": M, "interpolation": L}
        metric = "Msamples/s complex_float32 %s FIR" % ("decimating (in)" if wl == "decim8" else "interpolating (out)")
    elif wl == "fir255_i16":
        # complex_int16 255-tap FIR: bit-exact on the double-precision overlap-save pipeline
        n = C
        h = tp.c1_taps() * 0.9
        f = device.FirFilter("complex_int16", "COMPLEX")
        f.set_taps(h)
        K = f.K
        x = torch.randint(-20000, 20000, (n + K - 1, 2), device=dev).to(torch.int16)
        y = torch.empty((n, 2), dtype=torch.int16, device=dev)
        units = n
        roof_bytes = 8.0 * n
        kernel_name = "fir_cf64_ols_kernel"

        def step():
            f.process_dev(x, y)
        desc = {"workload": "255-tap complex_int16 FIR (bit-exact, double-precision overlap-save), %d samples per GPU" % n, "taps": 255}
        metric = "Msamples/s complex_int16 255-tap FIR"
    else:
        n = C
        x = torch.empty((n, 2), dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        device.fill_uniform_f32_dev(x, seed=6, offset=0)
        units = n
        roof_bytes = 16.0 * n
        kernel_name = "map_kernel<rotate>"

        def step():
            device.rotate(x, 0.7, scalar=device.F32, out=y, n=n)
        desc = {"workload": "/comms/rotate complex_float32, %d samples" % n}
        metric = "Msamples/s complex_float32 rotate"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup: let the clocks settle.  The first ~50 back-to-back launches after an idle period run
    # through a DVFS transient on this part (230 us -> 320 us -> 245 us per launch, profiles/r01);
    # these untimed passes are part of setup, not of the W warm-up steps or the timed region.
    for _ in range(PREWARM):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events on the launch stream (torch's current stream is the stream every pcx_*_dev call
    # gets): ONE pair around the K timed steps, so no event packet sits between two launches
    # (a pair per step costs ~12 us of gap per step on this stack); avg launch = span / K.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [ev0.elapsed_time(ev1) / args.steps]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        value = world * units * args.steps / elapsed / 1e6
        avg_ms = float(np.mean(kern_ms))
        achieved = roof_bytes / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": metric, "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if wl == "fir255_i16" else "f32",
            "data": "synthetic", "config": desc,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(wl),
                         "kernel": kernel_name, "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_bytes_per_launch": roof_bytes,
                         "bytes_counted": "algorithmic read + write (SURVEY 8d); the read stream alone is the smaller share"},
        }
        if world == 1 and not args.no_cpu and wl in ("fir255", "direct255"):
            out["cpu_baseline"] = cpu_baseline_fir(tp.c1_taps(), 2, min(C, 64 * 1024 * 1024))   # the whole 64 Mi-sample shard: ~11 s on one core
        elif world == 1 and not args.no_cpu:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
