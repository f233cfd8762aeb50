#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json, measured on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launched plainly with N > 1 (no WORLD_SIZE in the environment) this process does NOT touch the GPU: it
starts N rank processes through `python -m torch.distributed.run` (one per GPU, rendezvous on 127.0.0.1),
relays rank 0's JSON line and fails if a rank fails or the line does not carry n_gpus == N.  Every line
reports the world size torch.distributed actually formed (`n_gpus`, `config.world_size_observed`) and the
device each rank bound (`config.rank_devices`); a rank refuses to run when WORLD_SIZE != --gpus or when the
node has fewer GPUs than ranks (RCCL backend).

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
cpu_baseline: the oracle (-O3 restatement, kind "port") or oracle/_ref (the reference's own kissfft, kind
          "reference") timed on this host on a bounded sample of the same workload, median of 3
          (rank 0, N=1 only); the FIR line also carries the C0 case (63 taps, 1 Mi samples).

Other workloads (--workload fft4096 | fmchain | rotate | abs | freq_demod | direct255 | decim8 | interp4 | fir255_i16) print the same kind of line
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
# the host driver of this pool supports dmabuf IPC only: without this RCCL's peer setup fails with "hipIpcGetMemHandle: invalid argument".
# Already exported on the boxes; set here as well (before torch / the HIP runtime load) so that a rank started from a bare environment works.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # datasheet FP64 vector: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
SHARD = 64 * 1024 * 1024       # samples per GPU (configs[1])
PREWARM = 400                  # untimed setup passes before the W warm-up steps (clock settling: tools/transient_probe.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="fir255", choices=["fir255", "direct255", "fft4096", "fmchain", "rotate", "abs", "freq_demod", "decim8", "interp4", "fir255_i16", "fir4097", "fir8193", "fir4097_real"])
    ap.add_argument("--shard", type=int, default=SHARD, help="samples per GPU (default 64 Mi)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--driver", default="ranks", choices=["ranks", "native"],
                    help="ranks: one process per GPU over torch.distributed (the bench contract); native: ONE process driving every GPU "
                         "through the C ABI's pcx_shard_* (RCCL send/recv loaded by the library itself) -- what a Pothos block that owns "
                         "a pcx_shard does")
    ap.add_argument("--native-devices", default="",
                    help="native driver only: comma-separated device ordinals, one per shard (default 0..N-1).  Repeating an ordinal "
                         "puts several shards on one device over peer copies -- the one-GPU rehearsal; the line then says so")
    ap.add_argument("--sustain", type=float, default=1.0,
                    help="seconds of extra launches behind the timed region whose last half is reported as roofline.sustained (0: off)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default workload only: skip the `secondary` object (configs[2] 4096-pt FFT, configs[4] fused chain, the element-wise blocks, "
                         "the integer FIR, the resamplers, the long-tap FIRs, configs[3] on one device, the host path: ~1 s of GPU each)")
    ap.add_argument("--rehearse-rccl-rank", action="store_true",
                    help="N = 1 only, fir255 / fmchain: time the pass a MIDDLE rank of an RCCL world runs -- side stream, one grouped RCCL send + receive "
                         "of the halo (to the rank itself: one GPU is enough), gate signal, ONE gated launch on the slots such a rank takes -- instead "
                         "of the plain single-GPU launch; the line says it is a rehearsal")
    ap.add_argument("--native-transport", choices=["auto", "rccl", "peer"], default="auto",
                    help="--driver native: how the halo travels between DISTINCT devices (auto: RCCL; peer: hipMemcpyPeerAsync)")
    ap.add_argument("--no-autotune", action="store_true",
                    help="ranks over RCCL: do not time the pipelined against the unpipelined pass during setup, take the pipelined one")
    ap.add_argument("--native-submit-threads", action="store_true",
                    help="--driver native: one thread per device queues that device's share of a pass (pcx_shard_set_submit_threads)")
    ap.add_argument("--native-pingpong", action="store_true",
                    help="--driver native: two handles driven double-buffered (pcx_shard_post_exchange / pcx_shard_compute)")
    ap.add_argument("--no-pingpong", dest="pingpong", action="store_false",
                    help="fir255 / fmchain over ranks: ONE input buffer, every pass waits for its own exchange at its tail.  Default: two input "
                         "buffers, the halo of batch k+1 exchanged while batch k is filtered (stream.PingPongFir)")
    ap.add_argument("--rehearse-slots", type=int, default=0,
                    help="with --rehearse-rccl-rank: resident workgroups of the gated launch (a multiple of 128; default: what a rank of an RCCL world takes)")
    ap.add_argument("--no-cold", action="store_true", help="skip roofline.cold (three bursts of 20 launches behind 5 ms of idle)")
    ap.add_argument("--settle", type=int, default=PREWARM,
                    help="untimed setup passes before the W warm-up steps (clock settling after idle; reported as config.setup_passes)")
    return ap.parse_args()


def _flush_c_stdio():
    """RCCL prints a version banner through C stdio when its first communicator comes up; flushed only at exit it would land BEHIND the
    JSON line this program prints with Python's own (unbuffered-on-flush) stdout.  The result line must be the last thing on stdout."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def _median_time(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def _host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


def cpu_baseline_fir(taps, seed, nsamples, real_taps=False, real_stream=False):
    """Single-thread oracle FIR (reference accumulation order) on `nsamples` of the stream, median of 3;
    beside it the same loop on all host cores, the stream statically chunked with K-1 overlap (complex streams)."""
    import threading

    import numpy as np

    from oracle import oracle as o      # CPU baseline leg: the checker, timed as the baseline
    K = len(taps)
    x = o.fill_uniform_f32(nsamples + K - 1, seed, 0) if real_stream else o.fill_uniform_f32(2 * (nsamples + K - 1), seed, 0).reshape(-1, 2)
    blk = o.Fir(o.F32, not real_stream, not (real_taps or real_stream))
    blk.set_taps(taps)
    blk.activate()

    def one():
        _, c, p, _ = blk.work(x, nsamples)
        assert p == nsamples
    dt = _median_time(one)
    if real_stream:
        return {"value": round(nsamples / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": "%d-sample slice of the same REAL stream, %d real taps, oracle/pcx_oracle.c fir_loop_f32 (reference accumulation "
                          "order, -O3, no FMA), single thread, median of 3" % (nsamples, K)}
    # all host cores: thread i filters samples [i*per, (i+1)*per) of the SAME slice (its K-1 history is the
    # tail of chunk i-1: static chunking with overlap) -- a parallelised restatement, not reference behaviour
    ncores = _host_cores()
    per = nsamples // ncores
    y = np.zeros((ncores * per, 2), np.float32)
    L = o.lib()

    def work(i):
        L.orc_fir_cf32_chunk(blk.h, x[i * per:].ctypes.data, y[i * per:].ctypes.data, per)

    def all_cores():
        th = [threading.Thread(target=work, args=(i,)) for i in range(ncores)]
        [t.start() for t in th]
        [t.join() for t in th]
    dt_all = _median_time(all_cores) if not real_taps else None
    out = {
        "value": round(nsamples / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": "%d-sample slice of the same stream, %d taps, oracle/pcx_oracle.c fir_loop_f32 (reference "
                  "accumulation order, -O3, no FMA, taps pre-narrowed to float), single thread, median of 3" % (nsamples, K),
    }
    if dt_all:
        out["all_cores"] = {"value": round(ncores * per / dt_all / 1e6, 3), "cores": ncores,
                            "note": "the same slice statically chunked over every host core with K-1 overlap, %d samples each, "
                                    "median of 3 (parallelised restatement, not reference behaviour)" % per}
    return out


def cpu_baseline_c0():
    """BASELINE.json configs[0]: 63 complex taps, 1 Mi-sample BufferChunk, one scheduler thread."""
    from pothoscomms_amd import taps as tp
    r = cpu_baseline_fir(tp.c0_taps(), 1, 1 << 20)
    r.pop("all_cores", None)
    r["sample"] = "configs[0]: 63 taps, 1048576 samples, " + r["sample"].split(", ", 2)[2]
    return r


def cpu_baseline_fft(nframes):
    """4096-point forward transforms: the reference's kissfft<float> compiled from its own source
    (oracle/_ref) when present, else the oracle's restatement of it."""
    from oracle import oracle as o
    x = o.fill_uniform_f32(2 * nframes * 4096, 3, 0).reshape(-1, 2)
    have_ref = o.ref() is not None
    fn = (lambda: o.ref_fft(x, 4096, False)) if have_ref else (lambda: o.fft(x, 4096, False))
    dt = _median_time(fn)
    return {"value": round(nframes * 4096 / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1,
            "kind": "reference" if have_ref else "port",
            "sample": "%d frames of 4096 bins of the same stream, %s, single thread, median of 3 (%.0f frames/s)"
                      % (nframes, "fft/kissfft.hh compiled as oracle/_ref (-O3)" if have_ref else "oracle/pcx_oracle.c kissfft restatement (-O3)",
                         nframes / dt)}


def cpu_baseline_fmchain(n):
    """Rotate -> FIR(127 real taps) -> FreqDemod as three oracle blocks on n samples of the C4 stream."""
    from oracle import oracle as o
    from pothoscomms_amd import taps as tp
    h = tp.c4_taps()
    K = len(h)
    x = o.fill_uniform_f32(2 * (n + K - 1), 5, 0).reshape(-1, 2)
    fir = o.Fir(o.F32, True, False)
    fir.set_taps(h)
    fir.activate()

    def chain():
        r = o.rotate(x, tp.C4_PHASE)
        y, c, p, _ = fir.work(r, n)
        dm = o.FreqDemod(o.F32)
        d = dm.work(y)
        assert d.shape[0] == n
    dt = _median_time(chain)
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d samples of the same stream through the oracle's rotate, FIR (127 real taps) and freq_demod loops "
                      "(-O3), one after the other on one thread, median of 3" % n}


def cpu_baseline_rotate(n):
    from oracle import oracle as o
    x = o.fill_uniform_f32(2 * n, 6, 0).reshape(-1, 2)
    dt = _median_time(lambda: o.rotate(x, 0.7))
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d samples, oracle arrayRotate loop (-O3), single thread, median of 3" % n}


def cpu_baseline_map(wl, n):
    """the oracle's loop of one element-wise block (rotate / abs / freq_demod) on n complex_float32 samples"""
    from oracle import oracle as o
    x = o.fill_uniform_f32(2 * n, 8, 0).reshape(-1, 2)
    if wl == "rotate":
        fn, what = (lambda: o.rotate(x, 0.7)), "arrayRotate loop"
    elif wl == "abs":
        fn, what = (lambda: o.abs_(x, True)), "getAbs (std::abs of std::complex<float>) loop"
    else:
        fd = o.FreqDemod(o.F32)
        fn, what = (lambda: fd.work(x)), "FreqDemod::work loop (complex multiply + std::arg)"
    dt = _median_time(fn)
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d samples, oracle %s (-O3), single thread, median of 3" % (n, what)}


def cpu_baseline_resampler(h, M, L, n):
    from oracle import oracle as o
    blk = o.Fir(o.F32, True, True)
    blk.set_taps(h); blk.set_decimation(M); blk.set_interpolation(L)
    blk.activate()
    K = blk.K
    x = o.fill_uniform_f32(2 * (n + K - 1), 7, 0).reshape(-1, 2)

    def one():
        _, c, p, _ = blk.work(x, n * L // M)
        assert c == n, (c, n)
    dt = _median_time(one)
    units = n if L == 1 else n * L
    return {"value": round(units / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d input samples, decimation %d, interpolation %d, oracle polyphase loop (-O3), single thread, median of 3" % (n, M, L)}


def cpu_baseline_fir_i16(h, n):
    import numpy as np

    from oracle import oracle as o
    blk = o.Fir(o.I16, True, True)
    blk.set_taps(h)
    blk.activate()
    K = blk.K
    rng = np.random.default_rng(11)
    x = rng.integers(-20000, 20000, (n + K - 1, 2)).astype(np.int16)

    def one():
        _, c, p, _ = blk.work(x, n)
        assert p == n
    dt = _median_time(one)
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d complex_int16 samples, 255 taps, oracle integer loop (-O3; Q-format as restated, DESIGN.md 2), single thread, median of 3" % n}


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks ourselves.

    This parent never imports torch and never touches the GPU; the rank processes torch.distributed.run starts are SUPERVISORS
    (bench_supervisor.py: each runs the real rank as a child, watches it, and the N of them fall back together to a conservative
    form of the pass in FRESH children when an attempt hangs or dies).  Their stdout is captured so the one JSON line can be checked
    (n_gpus must equal --gpus) before it is relayed; stderr passes through as it is written.  The launcher itself is watched too:
    still running after the sum of the attempts' budgets, its process group is told to stop (the supervisors stop their children)."""
    import signal
    import socket
    import subprocess
    import threading

    import bench_supervisor as sup

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    overall = float(os.environ.get("PCX_BENCH_OVERALL_S", len(sup.ATTEMPTS) * (sup.ATTEMPT_BUDGET_S + 120.0)))
    for attempt in range(4):
        # a port that is free NOW; another process may take it before the rendezvous binds it (eight test workers at once did):
        # that failure is recognised and the launch repeated on another port
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        # the ranks' stderr is passed on as it comes (a hung run must not be silent) and kept for the port check
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, start_new_session=True)
        err_lines, out_lines = [], []

        def relay(pipe=proc.stderr, keep=err_lines):
            for ln in pipe:
                keep.append(ln)
                sys.stderr.write(ln)
                sys.stderr.flush()

        def collect(pipe=proc.stdout, keep=out_lines):
            for ln in pipe:
                keep.append(ln)
        ts = [threading.Thread(target=relay, daemon=True), threading.Thread(target=collect, daemon=True)]
        for t in ts:
            t.start()
        try:
            proc.wait(timeout=overall)
        except subprocess.TimeoutExpired:
            print("bench.py: the launcher is still running after %.0f s: stopping its process group" % overall, file=sys.stderr, flush=True)
            for sig, grace in ((signal.SIGTERM, 20), (signal.SIGKILL, 10)):
                try:
                    os.killpg(proc.pid, sig)               # the launcher's own session: the processes this parent started
                    proc.wait(timeout=grace)
                    break
                except (ProcessLookupError, PermissionError):
                    break
                except subprocess.TimeoutExpired:
                    continue
        for t in ts:
            t.join(timeout=5)
        stdout = "".join(out_lines)
        err = "".join(err_lines)
        taken = any(m in err for m in ("EADDRINUSE", "Address already in use", "errno: 98", "address already in use"))
        if proc.returncode != 0 and taken and attempt < 3 and '"metric"' not in stdout:
            print("bench.py: port %d was taken before the rendezvous could bind it, starting the ranks again" % port, file=sys.stderr)
            continue
        break
    line = None
    for ln in stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is None:
        raise SystemExit("bench.py: the ranks printed no result line (torch.distributed.run exit code %r)" % proc.returncode)
    got = json.loads(line)
    if got.get("n_gpus") != args.gpus:
        raise SystemExit("bench.py: asked for %d GPUs, the ranks report n_gpus=%r" % (args.gpus, got.get("n_gpus")))
    if proc.returncode != 0:
        # the line is complete (written behind the timed region, the max-over-ranks clock and the seam check); what failed is a teardown
        print("bench.py: torch.distributed.run exit code %r behind the result line (a rank's teardown)" % proc.returncode, file=sys.stderr)
    print(line, flush=True)


def source_hashes(files):
    """sha256 (first 16 hex digits) of kernel source files, paths relative to the repository root."""
    import hashlib
    out = {}
    for rel in files:
        with open(os.path.join(ROOT, rel), "rb") as f:
            out[rel] = hashlib.sha256(f.read()).hexdigest()[:16]
    return out


def load_profile(workload, kernel_name):
    """The committed PMC measurement of this workload's kernel (profiles/traffic.json: HBM bytes per launch, VALU instruction counts)
    -- only when it was taken on THIS kernel:
    the entry of profiles/traffic.json names the kernel symbol and carries the hashes of the source files the kernel is
    built from (tools/collect_profiles.py stamps them when it files the PMC summary); any mismatch, and the line says null."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return None
    try:
        with open(p) as f:
            t = json.load(f)
        e = t.get(workload)
        if not e or not e.get("kernel") or not e.get("sources"):
            return None
        if e["kernel"] not in kernel_name and kernel_name not in e["kernel"]:
            return None
        if source_hashes(sorted(e["sources"])) != e["sources"]:
            return None
        return e
    except (OSError, ValueError, KeyError):
        return None


def run_native(args):
    """One process, every GPU: pcx_shard_* behind the C ABI (include/pcx.h).  Same workload, same JSON line; `n_gpus` and
    `world_size_observed` are the number of DISTINCT devices that carry a shard, and the run refuses fewer GPUs than shards unless
    --native-devices names the rehearsal layout explicitly."""
    import ctypes as Cc

    import numpy as np
    import torch

    from pothoscomms_amd import _lib, device, taps as tp
    if args.workload not in ("fir255", "fmchain"):
        raise SystemExit("--driver native runs the sharded workloads: fir255, fmchain")
    L = _lib.load()
    ndev = torch.cuda.device_count()
    G = args.gpus
    if args.native_devices:
        devs = [int(d) for d in args.native_devices.split(",")]
        if len(devs) != G:
            raise SystemExit("--native-devices names %d shards, --gpus %d" % (len(devs), G))
    else:
        if ndev < G:
            raise SystemExit("--gpus %d needs %d GPUs on this node, %d visible (--native-devices 0,0,... rehearses on fewer)" % (G, G, ndev))
        devs = list(range(G))
    distinct = len(set(devs))
    transport = device.NodeStream.RCCL if distinct == G else device.NodeStream.PEER_COPY
    if args.native_transport == "peer":
        # hipMemcpyPeerAsync between the devices instead of ncclSend / ncclRecv: a copy has no kernel that must find a slot beside the gated
        # launch (RCCL's finds one only when that launch's first workgroups exit: DESIGN.md 6) -- for the comparison on a real node
        transport = device.NodeStream.PEER_COPY
    elif args.native_transport == "rccl" and distinct != G:
        raise SystemExit("--native-transport rccl: RCCL takes one shard per device, %d shards on %d device(s)" % (G, distinct))
    C = args.shard
    chain = args.workload == "fmchain"
    h = tp.c4_taps() if chain else tp.c1_taps()
    K = len(h)
    # --native-pingpong: two handles, two consecutive batches of the stream; the halos of batch k+1 are exchanged while batch k is filtered
    # (pcx_shard_post_exchange / pcx_shard_compute) -- one exchange and one pass per step, as without
    handles = []
    for b in range(2 if args.native_pingpong else 1):
        ns = device.NodeStream(devs, transport)
        ns.set_submit_threads(args.native_submit_threads)
        if chain:
            ns.set_chain(True, tp.C4_PHASE)
        ns.set_taps(h, complex_taps=not chain)
        ns.configure(C)
        for g in range(G):
            i, o, st, d = ns.buffers(g)
            _lib.check(L.pcx_fill_uniform_f32_dev(Cc.c_void_p(i), 2 * (K - 1 + C), 5 if chain else 2, 2 * (b * G + g) * C, Cc.c_void_p(st)))
        handles.append(ns)
    ns = handles[0]
    turn = [0]
    if args.native_pingpong:
        handles[0].post_exchange()

        def step():
            cur, nxt = handles[turn[0] & 1], handles[(turn[0] + 1) & 1]
            turn[0] += 1
            cur.compute()
            nxt.post_exchange()
    else:
        step = ns.step

    def sync():
        for x in handles:
            x.sync()
    for _ in range(args.settle + args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    per = elapsed / args.steps
    bytes_per = (12.0 if chain else 16.0) * C           # per shard and pass
    value = G * C * args.steps / elapsed / 1e6
    # per-DEVICE roofline: each device streams its shards' bytes in the time of one pass
    achieved = bytes_per * (G / distinct) / per / 1e9
    out = {
        "metric": "Msamples/s fused FM-demod chain" if chain else "Msamples/s complex_float32 255-tap FIR",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": distinct, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(per * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": ("fused Rotate->FIR(127 real taps)->FreqDemod" if chain else "255-tap complex_float32 FIR (/comms/fir_filter, COMPLEX taps, M=L=1)")
                               + ", %d-sample shard x %d, one process, pcx_shard_* behind the C ABI" % (C, G),
                   "taps": K, "shard_samples": C, "halo_samples": K if chain else K - 1, "setup_passes": args.settle,
                   "driver": "native", "shards": G, "shard_devices": devs, "world_size_observed": distinct,
                   "halo_transport": "rccl send/recv (ncclCommInitAll, one process)" if transport == device.NodeStream.RCCL
                                     else "peer copies (REHEARSAL: %d shards on %d device(s))" % (G, distinct) if distinct != G
                                     else "peer copies (hipMemcpyPeerAsync between the devices, one process)",
                   "parallelism": "overlap-save shards x%d, one gated launch per shard and pass" % G
                                  + ("; double-buffered over two handles: the halos of batch k+1 exchanged while batch k is filtered" if args.native_pingpong else "")
                                  + ("; a submit thread per device" if args.native_submit_threads else "")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "fmchain_cf32_ols4096_kernel" if chain else "fir_cf32_ols4096_kernel", "avg_launch_ms": round(per * 1e3, 4),
                     "algorithmic_bytes_per_launch": bytes_per,
                     "bytes_counted": "per device: its shards' algorithmic read + write bytes over the wall time of a pass (host clock "
                                      "around the K steps, all streams synchronised on both sides)"},
    }
    for x in handles:
        x.close()
    _flush_c_stdio()
    print(json.dumps(out), flush=True)


# ---- measured ceilings next to the datasheet ones ---------------------------------------------------------------------------------
FP32_FMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector peak with packed v_pk_fma_f32 (256 CUs x 4 SIMDs x 16 lanes x 2 x 2 flop x 2.4 GHz)
N_SIMDS = 1024                 # 256 CUs x 4
# A wave64 VALU instruction holds its SIMD (16 lanes) for four cycles (transcendentals and f64 longer: the share below is a floor).
VALU_CYCLES_PER_INST = 4


def measured_f64_issue_rate():
    """G wave-instructions/s of the FP64 pipe at two waves per SIMD on an FFT's instruction mix (tools/f64_lab.hip, the committed
    profiles/*/f64_lab.txt), or None"""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "f64_lab.txt"))):
        for m in re.finditer(r"fft mix\s+2 wave\(s\)/SIMD:[^\n]*?([0-9.]+) G wave-inst/s", open(f).read()):
            best = (float(m.group(1)), os.path.relpath(f, ROOT))
    return best


def ctr_insts(entry):
    ctr = (entry or {}).get("counters") or {}
    return float(ctr["SQ_INSTS_VALU"]) if ctr.get("SQ_INSTS_VALU") else None


def measured_fma_rate():
    """the packed-FMA rate the committed microbenchmark reached on this part (profiles/*/ubench_roofs.txt, TFLOP/s), or None"""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "ubench_roofs.txt"))):
        for m in re.finditer(r"pk_fma[^\n]*?([0-9.]+)\s*TFLOP", open(f).read()):
            best = (float(m.group(1)), os.path.relpath(f, ROOT))
    return best


class Workload:
    """One bench workload: device buffers, the step closure and what the roofline of its dominant kernel is priced on."""
    step = None
    units = 0                  # samples per step and GPU (the metric's unit)
    roof_bytes = 0.0           # algorithmic HBM bytes per launch
    read_bytes = 0.0
    kernel_name = ""
    metric = ""
    desc = None
    dtype = "f32"
    bound = "hbm"
    flops_per_unit = None      # set where another roof than HBM binds (direct255: 8 K flop per sample)
    limiter = None             # what the profile says holds the kernel below the HBM roof, when it is not the memory system
    blocks = None              # overlap-save blocks per launch (for valu.per_wave_block)
    owner = None               # the ShardedFir / ShardedFmChain, when the stream is sharded over ranks
    inputs = ()                # the device tensors the step reads (tools/floor_table.py zeroes them for the "kernel on zeros" row)


PIPELINED = ("; software-pipelined over two input buffers: the halo of batch k+1 is exchanged while batch k is filtered -- one exchange and one "
             "pass per step, as without (stream.PingPongFir)")


def build_workload(wl, C, dev, rank, world, args):
    import torch

    from pothoscomms_amd import _lib, device, taps as tp
    from pothoscomms_amd.stream import ShardedFir
    W = Workload()
    W.name = wl
    if wl in ("fir255", "direct255"):
        h = tp.c1_taps()
        algo = _lib.FIR_OLS_FFT if wl == "fir255" else _lib.FIR_DIRECT
        pingpong = bool(getattr(args, "pingpong", True)) and world > 1 and wl == "fir255"
        if pingpong:
            from pothoscomms_amd.stream import PingPongFir
            pp = PingPongFir(h, C, dev, "COMPLEX", algo)
            sf = pp.halves[0]
            # the second buffer holds the NEXT batch of the node-wide stream
            device.fill_uniform_f32_dev(pp.halves[1].buf, seed=2, offset=2 * (world + rank) * C)
        else:
            sf = ShardedFir(h, C, dev, "COMPLEX", algo)
        K = sf.K
        # the node-wide stream starts K-1 samples before shard 0 (rank 0's history); every rank
        # fills [its halo | its shard] from the same counter-hash stream, then the timed steps
        # overwrite the halo through RCCL
        device.fill_uniform_f32_dev(sf.buf, seed=2, offset=2 * rank * C)
        W.owner = pp if pingpong else sf
        W.inputs = (sf.buf, pp.halves[1].buf) if pingpong else (sf.buf,)
        W.units = C
        W.roof_bytes = 16.0 * C
        W.read_bytes = 8.0 * C
        W.kernel_name = "fir_cf32_ols4096_kernel" if wl == "fir255" else "fir_cf32_direct_kernel"
        W.step = pp.step if pingpong else sf.step
        W.desc = {"workload": "255-tap complex_float32 FIR (/comms/fir_filter, COMPLEX taps, M=L=1), %d-sample shard per GPU, "
                              "%s" % (C, "frequency-domain overlap-save (4096-pt radix-16 passes)" if wl == "fir255" else "LDS-tiled direct form"),
                  "taps": 255, "shard_samples": C, "halo_samples": K - 1,
                  "parallelism": "overlap-save shards x%d, RCCL send/recv halo" % world + (PIPELINED if pingpong else "") if world > 1 else "single GPU"}
        W.metric = "Msamples/s complex_float32 255-tap FIR"
        if wl == "direct255":
            # the time-domain form is NOT HBM-bound -- 8 K flop per sample against 16 B (127 flop/B, machine balance 19.7)
            W.bound = "fp32-fma"
            W.flops_per_unit = 8.0 * K
        else:
            W.blocks = -(-C // (4096 - (K - 1 + 15) // 16 * 16))
    elif wl == "fft4096":
        nframes = 65536
        x = torch.empty((nframes * 4096, 2), dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        device.fill_uniform_f32_dev(x, seed=3, offset=2 * rank * nframes * 4096)
        fft = device.Fft("complex_float32", 4096, False)
        W.units = nframes * 4096
        W.roof_bytes = 16.0 * W.units
        W.read_bytes = 8.0 * W.units
        W.kernel_name = "fft_r16_kernel<12>"
        W.step = lambda: fft.transform_dev(x, y, nframes)
        W.inputs = (x,)
        W.desc = {"workload": "4096-pt complex_float32 FFT (/comms/fft), 65536 frames per GPU", "frames": nframes}
        W.metric = "Msamples/s complex_float32 4096-pt FFT"
    elif wl == "fmchain":
        n = C
        K = len(tp.c4_taps())
        if world > 1:
            # the stream sharded over the ranks: K-sample halo from the left neighbour (stream.ShardedFmChain)
            from pothoscomms_amd.stream import PingPongFmChain, ShardedFmChain
            if getattr(args, "pingpong", True):
                sc = PingPongFmChain(tp.c4_taps(), tp.C4_PHASE, n, dev)
                for b, half in enumerate(sc.halves):       # two consecutive batches of the node-wide stream
                    device.fill_uniform_f32_dev(half.buf, seed=5, offset=2 * (b * world + rank) * n)
                W.inputs = tuple(half.buf for half in sc.halves)
            else:
                sc = ShardedFmChain(tp.c4_taps(), tp.C4_PHASE, n, dev)
                device.fill_uniform_f32_dev(sc.buf, seed=5, offset=2 * rank * n)
                W.inputs = (sc.buf,)
            W.owner = sc
            W.step = sc.step
        else:
            ch = device.FmChain()
            ch.set_phase(tp.C4_PHASE)
            ch.set_taps(tp.c4_taps(), False)
            # [lead | 126-sample history | n samples] with the samples (not the history) on a 128-byte line,
            # the same placement ShardedFir uses
            xa = torch.empty((2 + n + 126, 2), dtype=torch.float32, device=dev)
            x = xa[2:]
            y = torch.empty((n,), dtype=torch.float32, device=dev)
            device.fill_uniform_f32_dev(x, seed=5, offset=0)
            W.step = lambda: ch.process_dev(x, y, n + 126, n)
            W.inputs = (x,)
        W.units = n
        W.roof_bytes = 12.0 * n
        W.read_bytes = 8.0 * n
        W.kernel_name = "fmchain_cf32_ols4096_kernel"
        W.desc = {"workload": "fused Rotate->FIR(127 real taps)->FreqDemod in one frequency-domain kernel, complex_float32 -> float32, %d samples" % n}
        if world > 1:
            W.desc["parallelism"] = "overlap-save shards x%d, RCCL send/recv halo of 127 samples" % world + (PIPELINED if getattr(args, "pingpong", True) else "")
        W.metric = "Msamples/s fused FM-demod chain"
        # 998 VALU instructions per wave and block (661 of them the two transforms) fill ~0.7 of all SIMD issue time at the ~1.95 GHz the
        # power cap leaves on real data; on all-zero input (2.37 GHz) the same kernel reads 0.62 (profiles/r04/floor_table.txt)
        W.limiter = "valu issue at the power-capped clock"
        W.blocks = -(-n // (4096 - (K + 31) // 32 * 32))
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
        W.units = n if wl == "decim8" else n * L            # decimator: input samples; interpolator: output samples
        W.roof_bytes = 8.0 * n + 8.0 * (n * L // M)
        W.read_bytes = 8.0 * n
        W.kernel_name = "fir_cf32_ols4096_decim_batched_kernel" if wl == "decim8" else "fir_cf32_ols4096_interp_batched_kernel"
        W.step = lambda: f.process_dev(x, y)
        W.inputs = (x,)
        W.desc = {"workload": "255-tap complex_float32 FIR, %s, %d input samples per GPU" %
                              ("decimation 8 folded into the spectrum (input rate)" if wl == "decim8"
                               else "interpolation 4 from the replicated spectrum, 255 taps per phase (output rate)", n),
                  "decimation": M, "interpolation": L}
        W.metric = "Msamples/s complex_float32 %s FIR" % ("decimating (in)" if wl == "decim8" else "interpolating (out)")
        # not VALU (0.36 / 0.44 of SIMD issue time): the blocks' LDS exchanges at THREE workgroups per CU -- the registers hold H and the
        # pass-3 constants, a fourth workgroup would have to re-read both per block and measures slower (profiles/r04/floor_table.txt,
        # floor_lab_sweep.txt, decim8_occupancy_variants.txt)
        W.limiter = "LDS exchanges exposed at three workgroups per CU"
    elif wl == "fir255_i16":
        # complex_int16 255-tap FIR: bit-exact on the double-precision overlap-save pipeline
        n = C
        h = tp.c1_taps() * 0.9
        f = device.FirFilter("complex_int16", "COMPLEX")
        f.set_taps(h)
        K = f.K
        x = torch.randint(-20000, 20000, (n + K - 1, 2), device=dev).to(torch.int16)
        y = torch.empty((n, 2), dtype=torch.int16, device=dev)
        W.units = n
        W.roof_bytes = 8.0 * n
        W.read_bytes = 4.0 * n
        W.kernel_name = "fir_cf64_ip_kernel"
        W.step = lambda: f.process_dev(x, y)
        W.inputs = (x,)
        W.desc = {"workload": "255-tap complex_int16 FIR (bit-exact, double-precision overlap-save), %d samples per GPU" % n, "taps": 255}
        W.metric = "Msamples/s complex_int16 255-tap FIR"
        W.dtype = "f64"
        # NOT an HBM-bound kernel: 8 B per sample against ~100 double-precision flop per sample.  Per 4096-sample block and lane (256
        # lanes): six plain 16-point transforms at 128 additions + 8 constant complex multiplies, 60 per-lane factor multiplies, 16 for
        # H -- a complex multiply counted as 6 real flop -- = 1,512 flop; S = 4096 - 256 outputs per block at 255 taps
        W.bound = "fp64"
        S = 4096 - (K - 1 + 15) // 16 * 16
        W.blocks = -(-n // S)
        W.flops_per_unit = 256 * 1512.0 / S
        W.limiter = ("FP64 issue at two waves per SIMD (the 64 KB image of a block caps the occupancy): the arithmetic alone is 0.79 of the launch "
                     "(tools/ip64_parts.sh), and at that occupancy the pipe issues ~380 G wave-instructions/s chip-wide on an add / multiply / "
                     "FMA mix like an FFT's (tools/f64_lab.hip), not the 614 G/s the datasheet's 78.6 TFLOP/s stand for")
    elif wl == "fir4097_real":
        # a REAL float32 stream through a 4097-tap real filter: the call's two halves side by side as one complex stream through the
        # partitioned kernel (fir_ols_part.hip); 4 B read + 4 B written per sample
        n = C
        K = 4097
        h = tp.lowpass(K, 0.1)
        f = device.FirFilter("float32", "REAL")
        f.set_taps(h)
        xa = torch.empty(((n + K - 1 + 1) // 2, 2), dtype=torch.float32, device=dev)
        device.fill_uniform_f32_dev(xa, seed=9, offset=0)
        x = xa.view(-1)[:n + K - 1]
        y = torch.empty((n,), dtype=torch.float32, device=dev)
        W.units = n
        W.roof_bytes = 8.0 * n
        W.read_bytes = 4.0 * n
        W.kernel_name = "fir_cf32_upols_kernel"
        W.step = lambda: f.process_dev(x, y)
        W.inputs = (x,)
        W.desc = {"workload": "%d-tap float32 FIR (real stream, real taps), two halves of the call as one complex stream through the "
                              "partitioned kernel, %d samples per GPU" % (K, n), "taps": K}
        W.metric = "Msamples/s float32 %d-tap FIR" % K
        W.blocks = -(-n // 4096)
        W.limiter = ("the complex kernel's transform pair per 4096 real outputs at two waves per SIMD (DESIGN.md 4.8); until the end of round "
                     "6 this filter ran on the time-domain tile at 5 Gsamples/s")
    elif wl in ("fir4097", "fir8193"):
        # the long-tap plans of the complex_float32 FIR (2049 < K <= 8193): 4096-sample blocks advancing by 2048, the taps in
        # 2 / 4 partitions of 2048 against the spectra of the previous windows (fir_ols_part.hip) -- one transform pair per 2048
        # outputs whatever K, every input sample fetched once
        n = C
        K = 4097 if wl == "fir4097" else 8193
        h = tp.complex_bandpass(K, 0.05, 0.05)
        f = device.FirFilter("complex_float32", "COMPLEX")
        f.set_taps(h)
        lead = (-(K - 1)) % 16
        xa = torch.empty((lead + n + K - 1, 2), dtype=torch.float32, device=dev)
        x = xa[lead:]
        y = torch.empty((n, 2), dtype=torch.float32, device=dev)
        device.fill_uniform_f32_dev(x, seed=9, offset=0)
        W.units = n
        W.roof_bytes = 16.0 * n
        W.read_bytes = 8.0 * n
        W.kernel_name = "fir_cf32_upols_kernel"
        W.step = lambda: f.process_dev(x, y)
        W.inputs = (x,)
        W.desc = {"workload": "%d-tap complex_float32 FIR, 4096-sample overlap-save blocks with the taps in %d partitions, %d samples per GPU"
                              % (K, (K - 1 + 2047) // 2048, n), "taps": K}
        W.metric = "Msamples/s complex_float32 %d-tap FIR" % K
        W.blocks = -(-n // 2048)
        W.limiter = ("a transform pair per 2048 outputs: 1.9 x the headline's arithmetic per output, plus one multiply-add per bin and "
                     "partition against spectra held in registers (which is what sets the occupancy: 3 workgroups per CU at 2 "
                     "partitions, 2 at 3 and 4) and the partitions' spectra re-read from L2 in every block")
    elif wl in ("abs", "freq_demod"):
        # complex_float32 in, float32 out: 8 B read + 4 B written per sample, one launch per step
        n = C
        x = torch.empty((n, 2), dtype=torch.float32, device=dev)
        y = torch.empty((n,), dtype=torch.float32, device=dev)
        device.fill_uniform_f32_dev(x, seed=8, offset=0)
        W.units = n
        W.roof_bytes = 12.0 * n
        W.read_bytes = 8.0 * n
        if wl == "abs":
            W.kernel_name = "map_kernel<abs>"
            W.step = lambda: device.abs_(x, True, scalar=device.F32, out=y, n=n)
        else:
            fd = device.FreqDemod("complex_float32")
            W.kernel_name = "freqdemod_kernel"
            W.step = lambda: fd.process_dev(x, y, n)
            W.owner_keepalive = fd
        W.inputs = (x,)
        W.desc = {"workload": "/comms/%s complex_float32 -> float32, %d samples" % (wl, n)}
        W.metric = "Msamples/s complex_float32 %s" % wl
    else:
        n = C
        x = torch.empty((n, 2), dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        device.fill_uniform_f32_dev(x, seed=6, offset=0)
        W.units = n
        W.roof_bytes = 16.0 * n
        W.read_bytes = 8.0 * n
        W.kernel_name = "map_kernel<rotate>"
        W.step = lambda: device.rotate(x, 0.7, scalar=device.F32, out=y, n=n)
        W.inputs = (x,)
        W.desc = {"workload": "/comms/rotate complex_float32, %d samples" % n}
        W.metric = "Msamples/s complex_float32 rotate"
    W.desc["setup_passes"] = args.settle
    return W


_BREAK_EXCHANGE = [False]


def seam_check(owner, rank_has_halo):
    """Behind the timed region, on every rank (the exchanges are a collective step of the ring): does a pass still see its halo?  The
    bench's buffers are static -- every exchange delivers the bytes the slot already holds -- so a halo that arrived late, or bytes of a
    peer not yet visible behind the gate, would go unnoticed in the timed passes.  Here the halo slot of the buffer whose exchange has
    not been posted yet is POISONED (NaN), the driver stepped until that buffer has been exchanged and filtered, and the outputs at the
    front of the shard compared with a plain head call on the completed buffer (no NaN; equal within 1e-3 of the largest output: the two
    walk different block boundaries, and the check is for a missing halo, not for float parity -- tests/ do that).  Returns "" or what
    is wrong."""
    import torch
    halves = getattr(owner, "halves", None)
    pipelined = bool(halves) and owner.pipelined
    # (pipelined: the current buffer's exchange is already on its way, the upcoming one's is posted by the next step)
    target = owner.upcoming if pipelined else owner.current if halves else owner
    halo = target.ring.halo
    if rank_has_halo:
        target.buf[:halo] = float("nan")                   # (`buf` fences the input: behind the send that may still read the tail)
    out = None
    for _ in range(2 if pipelined else 1):
        out = owner.step()
    torch.cuda.synchronize()
    n = min(target.head, 4 * halo + 64)
    got = out[:n].clone()
    if not rank_has_halo:
        return ""
    if bool(torch.isnan(target.buf[:halo]).any()):
        return "the halo slot still holds the poison: the exchange did not deliver"
    if bool(torch.isnan(got).any()):
        return "outputs at the front of the shard were computed on the poisoned halo (the pass did not wait for its exchange)"
    ref = target.head_reference(n)
    torch.cuda.synchronize()
    err = float((got - ref).abs().max() / ref.abs().max())
    return "" if err <= 1e-3 else "outputs at the front of the shard differ from a plain head call by %.3g of the largest" % err


def time_launches(step, n):
    """n back-to-back steps between ONE pair of HIP events on the launch stream (torch's current stream is the stream every
    pcx_*_dev call gets; a pair per step costs ~12 us of gap per step on this stack).  Returns ms per step."""
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def cold_launches(step, n=20, idle_ms=5.0, reps=3):
    """What a bursty topology sees: the first n launches behind idle_ms of idle (the clocks have dropped; tools/transient_probe.py shows
    the ramp launch by launch).  Median of `reps` bursts; ms per launch."""
    import torch
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        time.sleep(idle_ms * 1e-3)
        out.append(time_launches(step, n))
    out.sort()
    return out[len(out) // 2]


def clock_under_load(step, launches=80, spin_us=2000):
    """The shader clock the workload runs at: one probe wave (pcx_clock_probe_dev) on a side stream spins for spin_us beside `launches`
    back-to-back steps that are already settled.  MHz, or None if the probe did not run."""
    import torch

    from pothoscomms_amd import device
    out = torch.zeros((16,), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for k in range(launches):
        if k == 8:
            device.clock_probe(out, spin_us, side)
        step()
    torch.cuda.synchronize()
    mhz = float(out[0].item())
    return round(mhz, 1) if mhz > 0 else None


def roofline_of(W, avg_ms, sustained=None, cold=None, sustain_s=1.0, clk=None):
    """The roofline object of a workload's dominant kernel from its measured average launch duration."""
    sec = avg_ms * 1e-3
    hbm = W.roof_bytes / sec / 1e9
    entry = load_profile(W.name, W.kernel_name)
    traffic = entry.get("hbm_bytes_per_launch") if entry else None
    bytes_note = ("frac: algorithmic read + write (SURVEY 8d); read_only_frac: the read stream alone (north_star words its target "
                  "on reads: a kernel that writes every sample back cannot put more than its read share of the pins into reads)")
    if W.bound == "fp32-fma":
        tf = W.flops_per_unit * W.units / sec / 1e12
        r = {"bound": "fp32-fma", "achieved": round(tf, 2), "peak": FP32_FMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(tf / FP32_FMA_PEAK_TFLOPS, 4), "traffic": traffic, "kernel": W.kernel_name, "avg_launch_ms": round(avg_ms, 4),
             "algorithmic_flops_per_launch": W.flops_per_unit * W.units,
             "flops_counted": "8 K = %d flop per output sample (K complex multiply-adds, SURVEY 8d) x %d samples; peak = the datasheet "
                              "packed-FP32 vector rate" % (int(W.flops_per_unit), W.units),
             "hbm": {"achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes_per_launch": W.roof_bytes,
                     "note": "the other ceiling: 127 flop per byte puts this kernel on the FMA roof, not here"}}
        m = measured_fma_rate()
        if m:
            r["frac_of_measured_fma_rate"] = round(tf / m[0], 4)
            r["measured_fma_rate"] = {"TFLOP/s": m[0], "source": m[1]}
    elif W.bound == "fp64":
        tf = W.flops_per_unit * W.units / sec / 1e12
        r = {"bound": "fp64", "achieved": round(tf, 2), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(tf / FP64_VECTOR_PEAK_TFLOPS, 4), "traffic": traffic, "kernel": W.kernel_name, "avg_launch_ms": round(avg_ms, 4),
             "algorithmic_flops_per_launch": W.flops_per_unit * W.units,
             "flops_counted": "per 4096-sample block 256 lanes x 1,512 flop (six 16-point transforms at 128 additions + 8 constant complex "
                              "multiplies, 60 per-lane factor multiplies, 16 for H; a complex multiply = 6 flop) = %.1f flop per output sample x "
                              "%d samples; peak = the datasheet FP64 VECTOR rate, which counts every instruction as an FMA (2 flop): an "
                              "FFT's instructions are 61 %% additions, so 0.6 is the most this fraction can reach" % (W.flops_per_unit, W.units),
             "hbm": {"achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes_per_launch": W.roof_bytes,
                     "note": "the other ceiling, kept beside: 8 B per sample -- this kernel is nowhere near it and never will be"}}
        m = measured_f64_issue_rate()
        if m and W.blocks:
            insts = (ctr_insts(entry) or 1350.0 * 4 * W.blocks)
            r["issue"] = {"wave_insts_per_launch": insts, "G_wave_insts_per_s": round(insts / sec / 1e9, 1),
                          "measured_roof_G_wave_insts_per_s": m[0], "frac_of_measured_issue_roof": round(insts / sec / 1e9 / m[0], 4),
                          "source": m[1],
                          "note": "what the FP64 pipe issues at this kernel's occupancy (two waves per SIMD) on independent add / multiply / FMA "
                                  "instructions in an FFT's proportions, measured on this part; wave_insts_per_launch from the PMC pass when one "
                                  "is committed for this kernel, else 1,350 per wave and block from the ISA"}
    else:
        r = {"bound": "hbm", "achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4),
             "read_only_frac": round(W.read_bytes / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
             "kernel": W.kernel_name, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": W.roof_bytes,
             "bytes_counted": bytes_note}
    if W.limiter:
        r["limiter"] = W.limiter
    ctr = (entry or {}).get("counters") or {}
    if ctr.get("SQ_INSTS_VALU"):
        insts = float(ctr["SQ_INSTS_VALU"])
        v = {"insts_per_launch": insts, "wave_insts_per_sample": round(insts / W.units, 4),
             "source": entry.get("source"),
             "note": "wave64 VALU instructions per launch from the PMC pass; a wave instruction holds one of the %d SIMDs for %d cycles, "
                     "so simd_cycle_share = insts x %d / (%d x launch time x clock) is the part of all SIMD issue time the arithmetic "
                     "alone takes at that clock" % (N_SIMDS, VALU_CYCLES_PER_INST, VALU_CYCLES_PER_INST, N_SIMDS)}
        if ctr.get("SQ_WAVES"):
            v["per_wave"] = round(insts / float(ctr["SQ_WAVES"]), 1)
        if W.blocks:
            v["per_wave_block"] = round(insts / (4.0 * W.blocks), 1)
        if clk:
            v["simd_cycle_share"] = round(insts * VALU_CYCLES_PER_INST / (N_SIMDS * sec * clk * 1e6), 3)
            v["at_clock_mhz"] = clk
        v["simd_cycle_share_at_boost_2400_MHz"] = round(insts * VALU_CYCLES_PER_INST / (N_SIMDS * sec * 2400e6), 3)
        r["valu"] = v
    if clk:
        r["clock_mhz_under_load"] = clk
    if sustained is not None:
        r["sustained"] = {"avg_launch_ms": round(sustained[0], 4),
                          "frac": round((hbm if W.bound == "hbm" else r["achieved"]) * avg_ms / sustained[0] / r["peak"], 4),
                          "launches": sustained[1],
                          "note": "the same launches kept up for ~%.1f s behind the timed region, the last half timed: "
                                  "the clock sags under the package power cap" % sustain_s}
    if cold is not None:
        r["cold"] = {"avg_launch_ms": round(cold, 4), "frac": round(r["achieved"] * avg_ms / cold / r["peak"], 4), "launches": 20, "idle_ms": 5.0,
                     "note": "the first 20 launches behind 5 ms of idle, median of 3 bursts: what a topology with gaps between work() "
                             "calls sees (the clocks ramp over ~200 launches, tools/transient_probe.py)"}
    return r


def cpu_baseline_of(wl, C):
    from pothoscomms_amd import taps as tp
    cpu_n = min(C, 16 * 1024 * 1024)          # bounded sample: 10-30 s of CPU work for the whole leg
    if wl in ("fir255", "direct255"):
        cb = cpu_baseline_fir(tp.c1_taps(), 2, cpu_n)
        cb["c0"] = cpu_baseline_c0()
        return cb
    if wl == "fft4096":
        return cpu_baseline_fft(4096)
    if wl == "fmchain":
        return cpu_baseline_fmchain(cpu_n)
    if wl == "rotate":
        return cpu_baseline_rotate(cpu_n)
    if wl in ("abs", "freq_demod"):
        return cpu_baseline_map(wl, cpu_n)
    if wl == "decim8":
        return cpu_baseline_resampler(tp.complex_bandpass(255, 0.05 / 8, 0.05 / 8), 8, 1, cpu_n)
    if wl == "interp4":
        return cpu_baseline_resampler(tp.complex_bandpass(255 * 4, 0.05 / 4, 0.05 / 4) * 4, 1, 4, cpu_n // 4)
    if wl in ("fir4097", "fir8193"):
        K = 4097 if wl == "fir4097" else 8193
        return cpu_baseline_fir(tp.complex_bandpass(K, 0.05, 0.05), 9, 1024 * 1024)
    if wl == "fir4097_real":
        return cpu_baseline_fir(tp.lowpass(4097, 0.1), 9, 1024 * 1024, real_stream=True)
    return cpu_baseline_fir_i16(tp.c1_taps() * 0.9, cpu_n // 2)


def cpu_baseline_secondary(wl):
    """the CPU leg of a secondary workload: SHORT slices (about a second each: the driver's whole run stays well under a minute)"""
    from pothoscomms_amd import taps as tp
    if wl == "fft4096":
        return cpu_baseline_fft(2048)
    if wl == "fmchain":
        return cpu_baseline_fmchain(4 * 1024 * 1024)
    if wl in ("rotate", "abs", "freq_demod"):
        return cpu_baseline_map(wl, 4 * 1024 * 1024)
    if wl == "decim8":
        return cpu_baseline_resampler(tp.complex_bandpass(255, 0.05 / 8, 0.05 / 8), 8, 1, 2 * 1024 * 1024)
    if wl == "interp4":
        return cpu_baseline_resampler(tp.complex_bandpass(255 * 4, 0.05 / 4, 0.05 / 4) * 4, 1, 4, 256 * 1024)
    if wl in ("fir4097", "fir8193"):
        K = 4097 if wl == "fir4097" else 8193
        return cpu_baseline_fir(tp.complex_bandpass(K, 0.05, 0.05), 9, (64 if K == 4097 else 32) * 1024)
    if wl == "fir4097_real":
        return cpu_baseline_fir(tp.lowpass(4097, 0.1), 9, 128 * 1024, real_stream=True)
    return cpu_baseline_fir_i16(tp.c1_taps() * 0.9, 512 * 1024)


def guarded(name, fn, *a):
    """a secondary measurement must not cost the primary line: whatever it raises -- SystemExit included -- becomes {"error": ...}"""
    try:
        return fn(*a)
    except BaseException as e:      # noqa: BLE001 (KeyboardInterrupt is passed on)
        if isinstance(e, KeyboardInterrupt):
            raise
        sys.stderr.write("bench.py: secondary %s failed: %s: %s\n" % (name, type(e).__name__, e))
        return {"error": "%s: %s" % (type(e).__name__, e)}


def measure_secondary(wl, dev, args):
    """BASELINE configs[2] / configs[4] and the remaining rows of SURVEY 8 behind the headline's timed region, in the same process and
    under the same event protocol (settling passes, one event pair around the timed launches); outside `value`."""
    import gc

    import torch
    W = build_workload(wl, SHARD, dev, 0, 1, args)
    for _ in range(args.settle // 2 + args.warmup):
        W.step()
    torch.cuda.synchronize()
    probe = time_launches(W.step, 20)
    n = max(50, min(2000, int(0.6 / max(probe * 1e-3, 1e-6))))       # about 0.6 s of launches
    t0 = time.perf_counter()
    avg_ms = time_launches(W.step, n)
    wall = time.perf_counter() - t0
    clk = clock_under_load(W.step)
    cold = cold_launches(W.step)
    out = {"metric": W.metric, "value": round(W.units / (avg_ms * 1e-3) / 1e6, 1), "unit": "Msamples/s", "steps": n,
           "ms_per_step": round(avg_ms, 4), "wall_ms_per_step": round(wall / n * 1e3, 4), "dtype": W.dtype, "config": W.desc,
           "roofline": roofline_of(W, avg_ms, cold=cold, clk=clk)}
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_secondary(wl)
    del W
    gc.collect()
    torch.cuda.empty_cache()
    return out


def _hb(phase):
    """A heartbeat for the supervisor of this rank (bench_supervisor.py): where the rank is.  Silent outside a supervised run."""
    if os.environ.get("PCX_BENCH_CHILD"):
        print("bench.py[hb] rank %s: %s" % (os.environ.get("RANK", "0"), phase), file=sys.stderr, flush=True)


def _test_fault(point, rank):
    """Tests only (tests/test_stream_cpu.py): PCX_BENCH_TEST_HANG / PCX_BENCH_TEST_DIE = "<rank>[:<attempt>[:<point>]]" make that rank of
    that attempt (default 1) hang or die at that point (default "setup")."""
    for var, act in (("PCX_BENCH_TEST_HANG", "hang"), ("PCX_BENCH_TEST_DIE", "die")):
        v = os.environ.get(var)
        if not v:
            continue
        f = v.split(":")
        if int(f[0]) != rank or int(f[1] if len(f) > 1 else 1) != int(os.environ.get("PCX_BENCH_ATTEMPT", "1")) or (f[2] if len(f) > 2 else "setup") != point:
            continue
        if act == "die":
            raise SystemExit("bench.py: rank %d dies at %r (PCX_BENCH_TEST_DIE)" % (rank, point))
        while True:
            time.sleep(3600)


def measure_c3_one_device(args, plain_ms):
    """BASELINE configs[3] rehearsed where the driver runs: 8 shards of 64 Mi samples through pcx_shard_* (the native, one-process driver
    behind the C ABI) on DEVICE 0, the 254-sample halos moved by peer copies, one gated launch per shard and pass.  Not a multi-GPU
    number -- the line says so -- but the whole mechanism of a pass (exchange, gate, eight launches) under the driver's own clock:
    ms per pass against 8 x the plain single launch, and every seam checked behind the timed passes (halos poisoned first)."""
    import ctypes as Cc

    import numpy as np

    from pothoscomms_amd import _lib, device, taps as tp
    L = _lib.load()
    G, C = 8, SHARD
    h = tp.c1_taps()
    K = len(h)
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
    ns.set_taps(h, complex_taps=True)
    ns.configure(C)
    bufs = [ns.buffers(g) for g in range(G)]
    for g, (i, o, st, d) in enumerate(bufs):
        _lib.check(L.pcx_fill_uniform_f32_dev(Cc.c_void_p(i), 2 * (K - 1 + C), 4, 2 * g * C, Cc.c_void_p(st)))
    ns.sync()
    for _ in range(40):
        ns.step()
    ns.sync()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        ns.step()
    ns.sync()
    per_ms = (time.perf_counter() - t0) / n * 1e3
    # the seams: poison every halo slot, one more pass, then shard by shard -- the slot holds the left neighbour's tail again, the first
    # 4096 outputs are finite and equal a plain call on the completed buffer (1e-3 of the largest: a missing halo, not float parity)
    nan = np.full((K - 1, 2), np.nan, np.float32)
    for g in range(1, G):
        _lib.check(L.pcx_memcpy_h2d(Cc.c_void_p(bufs[g][0]), nan.ctypes.data_as(Cc.c_void_p), nan.nbytes, Cc.c_void_p(bufs[g][2])))
    ns.sync()
    ns.step()
    ns.sync()
    f = device.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(h)
    wrong = []
    m = 4096
    for g in range(1, G):
        xin = np.empty((K - 1 + m, 2), np.float32)
        y = np.empty((m, 2), np.float32)
        tail = np.empty((K - 1, 2), np.float32)
        _lib.check(L.pcx_memcpy_d2h(xin.ctypes.data_as(Cc.c_void_p), Cc.c_void_p(bufs[g][0]), xin.nbytes, None))
        _lib.check(L.pcx_memcpy_d2h(y.ctypes.data_as(Cc.c_void_p), Cc.c_void_p(bufs[g][1]), y.nbytes, None))
        _lib.check(L.pcx_memcpy_d2h(tail.ctypes.data_as(Cc.c_void_p), Cc.c_void_p(bufs[g - 1][0] + 8 * C), tail.nbytes, None))
        _lib.check(L.pcx_stream_sync(None))
        if not np.array_equal(xin[:K - 1], tail):
            wrong.append("shard %d: the halo slot does not hold shard %d's tail" % (g, g - 1))
            continue
        ref, c, p = f.process(xin, m)
        if not np.isfinite(y).all() or float(np.max(np.abs(y - ref)) / np.max(np.abs(ref))) > 1e-3:
            wrong.append("shard %d: the outputs at its front do not match a plain call on the completed buffer" % g)
    ns.close()
    out = {"metric": "Msamples/s complex_float32 255-tap FIR", "value": round(G * C / (per_ms * 1e-3) / 1e6, 1), "unit": "Msamples/s",
           "n_gpus": 1, "steps": n, "ms_per_pass": round(per_ms, 4),
           "config": {"workload": "BASELINE configs[3] REHEARSED ON ONE DEVICE: 255-tap complex_float32 FIR, 8 shards x %d samples through pcx_shard_* "
                                  "(one process, C ABI), all on device 0, halos by peer copies, one gated launch per shard and pass" % C,
                      "shards": G, "shard_samples": C, "halo_samples": K - 1, "driver": "native"},
           "ratio_to_8_plain_launches": round(per_ms / (8 * plain_ms), 4) if plain_ms else None,
           "plain_launch_ms": round(plain_ms, 4) if plain_ms else None,
           "seam_check": ("halo slots poisoned behind the timed passes, one more pass: all 7 seams hold the left neighbour's tail and the shard fronts "
                          "match a plain call") if not wrong else "FAILED: " + "; ".join(wrong),
           "roofline": {"bound": "hbm", "achieved": round(16.0 * G * C / (per_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(16.0 * G * C / (per_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                        "kernel": "fir_cf32_ols4096_kernel (gated), 8 launches per pass",
                        "clock": "wall clock around %d passes, every stream synchronised on both sides" % n}}
    if wrong:
        # not a measurement (the multi-rank path refuses its line in the same situation): the figures go, the reason stays
        out["value"] = None
        out["ratio_to_8_plain_launches"] = None
        out["roofline"] = None
    return out


def measure_host_path(args, cpu):
    """SURVEY 8d's "one end-to-end number including H2D/D2H": /comms/fir_filter work() -- the block a Pothos topology loads -- on host port
    buffers, transfers INSIDE the timed region (every call returns with its result in the output buffer).  Priced on the PCIe roof of
    THIS box, measured in this run (pcx_pcie_probe: copy engines, both directions at once).  Two inputs: the block's own page-locked
    port slabs, and the framework's pageable double-mapped "circular" buffer (what the FIR is handed inside Pothos,
    filter/FIRFilter.cpp:196-199), which the block page-locks where it lies -- window placed across the wrap.  Native call loop
    (pcxb_work_loop): what a C++ scheduler pays, no Python per call."""
    import ctypes as Cc

    import numpy as np

    from pothoscomms_amd import _lib, blocks, device, taps as tp
    L = _lib.load()
    up, down, both = Cc.c_double(), Cc.c_double(), Cc.c_double()
    _lib.check(L.pcx_pcie_probe(128 << 20, 3, Cc.byref(up), Cc.byref(down), Cc.byref(both)))
    # The roof of a direction is what the link carries in that direction ALONE (PCIe is full duplex: the physical ceiling of each
    # direction does not depend on the other being busy).  What the copy engines reach with BOTH directions busy is reported beside
    # it: in a process on torch's bundled HIP runtime that is half of the sum (28 GB/s: the two transfers run one after the other),
    # in a plain C process on the system runtime 48 (examples/c_pcie_probe.c, profiles/r05/pcie_lab.txt) -- the in-place kernel,
    # which drives both directions at once itself, is above the first and below the second.
    # (the FASTER of the two: the link is symmetric, and a copy engine that lags in one direction -- 30 GB/s D2H was seen on one box in one
    # run, 57 in the next -- says nothing about what the link carries)
    roof = max(up.value, down.value)
    if not roof > 0:
        raise RuntimeError("pcx_pcie_probe measured no transfer rate (h2d %r, d2h %r GB/s)" % (up.value, down.value))
    K = 255
    calls = {}
    probe = blocks.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    slab = probe.call("getPortSlabBytes")                # the block's DEFAULT port slab: what a topology gets without setting anything
    probe.close()
    n_default = slab // 8
    for n in sorted({1 << 20, n_default, 1 << 24}):
        for mode in ("pinned_port_buffers", "circular_input_page_locked_in_place"):
            blk = blocks.make("/comms/fir_filter", "complex_float32", "COMPLEX")
            blk.call("setTaps", tp.c1_taps())
            blk.activate()
            yout, _ = blk.port_buffer(1, (n, 2), np.float32)
            circ = None
            if mode == "pinned_port_buffers":
                xin, _ = blk.port_buffer(0, (n + K - 1, 2), np.float32)
            else:
                circ = blocks.CircularBuffer(2 * (n + K) * 8)
                cap = circ.size // 8
                xin = circ.view((cap - n // 2) * 8, (n + K - 1) * 8, np.float32).reshape(-1, 2)
            half = xin.shape[0] // 2       # (filled on the device's generator would need a device buffer; a host fill of 128 MiB costs ~0.1 s)
            xin[:half] = np.random.default_rng(3).uniform(-1, 1, (half, 2)).astype(np.float32)
            xin[half:] = xin[:xin.shape[0] - half]
            blk.work_loop(xin.ctypes.data, n + K - 1, yout.ctypes.data, n, 3)
            reps = max(5, (1 << 25) // n)
            best = None
            for _ in range(3):
                t, c, p = blk.work_loop(xin.ctypes.data, n + K - 1, yout.ctypes.data, n, reps)
                if (c, p) != (n, n):
                    raise SystemExit("bench.py: host_path: work() consumed %d produced %d of %d" % (c, p, n))
                best = t / reps if best is None else min(best, t / reps)
            gbs = 8.0 * n / best / 1e9
            calls.setdefault("%d_samples_per_call" % n, {})[mode] = {"ms_per_call": round(best * 1e3, 4), "Msamples_per_s": round(n / best / 1e6, 1),
                                                                     "GB_per_s_each_way": round(gbs, 2), "frac_of_pcie_roof": round(gbs / roof, 4)}
            blk.close()
            if circ is not None:
                circ.close()
    head = calls["%d_samples_per_call" % n_default]["pinned_port_buffers"]
    out = {"metric": "Msamples/s complex_float32 255-tap FIR, host buffers in and out (PCIe inside the timed region)",
           "value": head["Msamples_per_s"], "unit": "Msamples/s", "ms_per_step": head["ms_per_call"],
           "config": {"workload": "/comms/fir_filter work() on host port buffers, 255 complex taps, complex_float32, %d samples per call "
                                  "(what the block's DEFAULT %d MiB port slabs carry; setPortSlabBytes moves it: `calls` has 1 Mi and 16 Mi samples "
                                  "per call beside it); every call returns with its result in the output buffer" % (n_default, slab >> 20),
                      "samples_per_call": n_default, "port_slab_bytes": slab, "call_loop": "native (pcxb_work_loop)"},
           "roofline": {"bound": "pcie", "achieved": head["GB_per_s_each_way"], "peak": round(roof, 2), "unit": "GB/s", "frac": head["frac_of_pcie_roof"],
                        "traffic": None,
                        "peak_measured": {"h2d_alone": round(up.value, 2), "d2h_alone": round(down.value, 2),
                                          "h2d_and_d2h_at_once_per_direction": round(both.value, 2),
                                          "how": "pcx_pcie_probe in this run: copy engines, 128 MiB of page-locked memory each way, two streams, "
                                                 "3 transfers back to back behind a warm-up one"},
                        "bytes_counted": "8 B per sample in and 8 B out over the wall time of a call; achieved = GB/s in EACH direction (both are busy "
                                         "at once); peak = the link's per-direction rate measured ALONE, the faster of the two directions (symmetric link, full duplex: each direction's ceiling)",
                        "note": "the kernel reads and writes the host buffers in place over PCIe (zero-copy both ways); a copy-engine pipeline was measured "
                                "and is slower below ~100 MiB per call (profiles/r05/pcie_lab.txt, drain_ab.txt)"},
           "calls": calls}
    # the same loop from a plain C process on the SYSTEM HIP runtime -- what a Pothos process is (this one runs on torch's bundled runtime):
    # examples/c_block_path, built by __graft_entry__.build(); a child process, started and waited for
    exe = os.path.join(ROOT, "examples", "c_block_path")
    if os.path.exists(exe):
        import subprocess
        try:
            r = subprocess.run([exe, "brief"], capture_output=True, text=True, timeout=120)
            rows = [ln.split() for ln in r.stdout.splitlines() if len(ln.split()) == 3]
            if r.returncode == 0 and rows:
                out["plain_c_process"] = {
                    "%d_samples_per_call" % int(n): {"pinned_port_buffers": {"ms_per_call": round(float(a) * 1e3, 4), "Msamples_per_s": round(int(n) / float(a) / 1e6, 1)},
                                                     "circular_input_page_locked_in_place": {"ms_per_call": round(float(b) * 1e3, 4), "Msamples_per_s": round(int(n) / float(b) / 1e6, 1)}}
                    for n, a, b in rows}
                out["plain_c_process"]["note"] = "examples/c_block_path.c: the same work() loop in a process on the system HIP runtime (no Python, no torch)"
        except (OSError, subprocess.SubprocessError, ValueError):
            pass
    if cpu:
        out["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cpu}
        if "all_cores" in cpu:
            out["cpu_baseline"]["all_cores"] = cpu["all_cores"]
    return out


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.driver == "native":
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--driver native is ONE process: do not start it under torch.distributed.run")
        return run_native(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)      # before torch is imported or the GPU touched
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to measure a different job than the one asked for" % (args.gpus, world))
    if world > 1 and not os.environ.get("PCX_BENCH_CHILD"):
        # a rank process of torch.distributed.run: it SUPERVISES the real rank, a child process (bench_supervisor.py) -- before torch is
        # imported or the GPU touched
        import bench_supervisor
        raise SystemExit(bench_supervisor.supervise(sys.argv[1:], os.path.abspath(__file__)))
    _hb("started")
    if os.environ.get("PCX_BENCH_TEST_RANK_ENV"):          # tests only: "rank:KEY=VALUE" -- ONE rank sees another environment than its peers
        r_, kv = os.environ["PCX_BENCH_TEST_RANK_ENV"].split(":", 1)
        if int(r_) == rank and int(os.environ.get("PCX_BENCH_ATTEMPT", "1")) == 1:
            os.environ[kv.split("=", 1)[0]] = kv.split("=", 1)[1]

    import numpy as np
    import torch
    import torch.distributed as dist

    standin = os.environ.get("PCX_BENCH_TEST_STANDIN")
    if standin:
        # tests only: a CPU stand-in for the device and the workload (tests/bench_standin.py), so that the ranks' control flow -- process
        # group, fall-backs, seam check, the line -- runs under gloo on a box without a GPU.  The line it yields says what it is.
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_standin", standin)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.install(globals())
    _hb("torch imported")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = torch.cuda.device_count()
    # "nccl" is RCCL on ROCm.  PCX_BENCH_BACKEND=gloo: the halo staged through the host -- the supervisors' last resort on a node whose
    # RCCL does not come up (every rank on its own GPU), and the rehearsal of the multi-rank control flow on a single-GPU box (ranks then
    # share cuda:0); the line says so.
    backend = os.environ.get("PCX_BENCH_BACKEND", "nccl") if world > 1 else "none"
    if (backend == "nccl" or os.environ.get("PCX_BENCH_OWN_GPU") == "1") and ndev < world:
        raise SystemExit("--gpus %d needs %d GPUs on this node, %d visible" % (args.gpus, world, ndev))
    dev_index = local_rank % ndev          # one rank per GPU on the node the driver gives us
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ctl = dev if backend == "nccl" else "cpu"      # where small control values are reduced (a host-driven backend reduces them on the host)
    rank_devices = [dev_index]
    if world > 1:
        if backend == "nccl":
            # NO device_id: with it torch creates the communicator eagerly and binds it to the device, and on this stack the ranks' pass --
            # grouped isend / irecv on a side stream beside the gated launch -- then takes 225 us instead of 194 (tools/host_step_probe.py
            # 768 devid, profiles/r04/rccl_pass_slots.txt).  The device is the current one (set above); barriers name it.
            dist.init_process_group("nccl")
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, dev_index, torch.cuda.get_device_properties(dev).name))
        _flush_c_stdio()
        rank_devices = [g[1] for g in sorted(gathered)]
        world = dist.get_world_size()      # what the line reports is what the collective layer formed
        _hb("process group up")
    _test_fault("setup", rank)

    C = args.shard
    wl = args.workload
    rehearsal = False
    if args.rehearse_rccl_rank:
        if world != 1 or wl not in ("fir255", "fmchain"):
            raise SystemExit("--rehearse-rccl-rank: one process, --workload fir255 or fmchain")
        # a process group of ONE over RCCL; the workload is then built as a rank of an RCCL world would build it (its slots), and its ring
        # is replaced by one whose neighbours are both the rank itself: a send AND a receive in one RCCL group, like any interior rank
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        if os.environ.get("PCX_BENCH_REHEARSE_DEVICE_ID") == "1":      # (A/B of the finding above: the communicator bound to the device)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", rank=0, world_size=1)      # (no device_id: as the ranks of a real world, above)
        rehearsal = True
    W = build_workload(wl, C, dev, rank, 2 if rehearsal else world, args)
    if rehearsal:
        from pothoscomms_amd import stream as _stream

        class SelfRing(_stream.HaloRing):
            def __init__(self, halo, owner):
                self.halo, self.group, self.rank, self.world, self.owner = halo, None, 1, 3, owner

            def start(self, buf):
                # (tests only: the seam check must notice an exchange that delivers nothing -- "1": in the one-launch form only, the
                # two-launch form works and the line is re-timed in it; "2": in either form, no line)
                if _BREAK_EXCHANGE[0] == "2" or (_BREAK_EXCHANGE[0] == "1" and not self.owner.two_launch):
                    return []
                return dist.batch_isend_irecv([dist.P2POp(dist.isend, buf[buf.shape[0] - self.halo:], 0), dist.P2POp(dist.irecv, buf[:self.halo], 0)])
        W.owner.ring = SelfRing(W.owner.ring.halo, W.owner)
        want_slots = args.rehearse_slots or ((_stream.RCCL_SLOTS or 0) if wl == "fir255" else 0)
        if not want_slots and args.pingpong:
            from pothoscomms_amd.stream import PINGPONG_SLOTS as want_slots
        if want_slots:                                     # (the world of one it was built in did not ask for them)
            W.owner.set_slots(want_slots)
        W.desc["parallelism"] = ("REHEARSAL on one GPU: the pass of a MIDDLE rank of an RCCL world -- grouped RCCL send + receive of the halo (to the rank "
                                 "itself), gate signal, one gated launch on %s resident workgroups%s" % (W.owner.slots or 1024,
                                 "; two input buffers, the exchange of batch k+1 posted in front of the launch of batch k" if args.pingpong else ""))
    step, desc = W.step, W.desc
    if world > 1 and W.owner is not None:
        # Every rank must take the same branches below (each is a collective step of the ring: the launch-stream probe, the autotune, the
        # switches between the forms of the pass), and some of what selects them comes from each rank's OWN environment and arguments
        # (PCX_STREAM_TWO_LAUNCH, PCX_BENCH_NO_STREAM_PICK, --no-autotune, --no-pingpong): agree once, refuse a mismatch.
        mine = [int(bool(getattr(W.owner, "two_launch", False))), int(bool(getattr(W.owner, "pipelined", False))), int(bool(args.no_autotune)),
                int(os.environ.get("PCX_BENCH_NO_STREAM_PICK") == "1"), int(hasattr(W.owner, "halves")), args.steps, args.warmup, args.settle]
        lo = torch.tensor(mine, dtype=torch.int64, device=ctl)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if lo.tolist() != hi.tolist():
            raise SystemExit("bench.py: the ranks disagree about the form of the pass (two_launch, pipelined, no_autotune, no_stream_pick, "
                             "two buffers, steps, warmup, settle): rank %d has %r, the ring's minimum is %r and maximum %r" % (rank, mine, lo.tolist(), hi.tolist()))
    if W.owner is not None and (rehearsal or (world > 1 and backend == "nccl")) and os.environ.get("PCX_BENCH_NO_STREAM_PICK") != "1":
        # the passes are launched on a stream whose hardware queue the exchange does not share (stream.py, HARDWARE QUEUES): collective,
        # every rank runs the same number of probe exchanges
        from pothoscomms_amd.stream import pick_launch_stream
        torch.cuda.synchronize()
        torch.cuda.set_stream(pick_launch_stream(W.owner))
        _hb("launch stream picked")

    tuned = None

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # setup: let the clocks settle.  The launches behind an idle period run through a DVFS transient on this part: after
    # 5 ms of idle the headline kernel takes 209-229 us per launch for the first 80, 200-206 for the next 100 and reaches
    # its settled 196-199 only after ~200 launches (tools/transient_probe.py, profiles/r03/transient_probe.txt); a
    # synchronisation WITHOUT idle time (the barrier below) costs nothing.  These untimed passes are part of setup, not
    # of the W warm-up steps or the timed region; their number is reported as config.setup_passes, and what the launches
    # behind an idle period cost is reported as roofline.cold.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fell_back = None       # why the ranks switched to the two-launch pass, if they did

    def switch_to_two_launch(why):
        nonlocal fell_back
        fell_back = why
        W.owner.two_launch = True

    def gate_check_and_fall_back(where):
        """A gate timeout during the setup passes is reported and cleared, so that the check behind the timed region speaks for the timed
        passes alone -- and it is ACTED on: the one-launch pass rests on assumptions that one GPU cannot prove (a peer's bytes visible
        behind the gate's acquire, RCCL's kernel finding room beside a persistent launch).  If ANY rank saw its gate time
        out, EVERY rank switches to the two-launch pass (body, halo, head: stream.py two_launch) from here on, and the line says so
        (config.halo_scheme).  Every rank calls this at the same points."""
        if world == 1 or W.owner is None:
            return
        timed_out = 0
        try:
            W.owner.check_gate()
        except RuntimeError as e:
            timed_out = 1
            print("bench.py: rank %d, %s (cleared): %s" % (rank, where, e), file=sys.stderr, flush=True)
        if os.environ.get("PCX_BENCH_TEST_GATE_TIMEOUT") == str(rank):      # tests only: this rank reports a timeout it did not have
            timed_out = 1
        flag = torch.tensor([timed_out], dtype=torch.int32, device=ctl)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and not W.owner.two_launch:
            switch_to_two_launch("a gated launch timed out waiting for its halo %s" % where)

    for k in range(args.settle):
        step()
        if k == 2:
            gate_check_and_fall_back("after the first three setup passes")     # early: 400 passes of two-second timeouts would be a quarter of an hour
            _hb("first passes done")
    if (W.owner is not None and hasattr(W.owner, "halves") and (rehearsal or (world > 1 and backend == "nccl")) and W.owner.pipelined
            and not args.no_autotune):
        # MEASURE, ON THE HARDWARE THIS RUNS ON, which form of the pass is faster -- pipelined on PINGPONG_SLOTS resident workgroups, or every
        # pass behind its own exchange on all 1024 -- and take that one: the pipelined default rests on one-GPU rehearsals.
        # Behind the settling passes (in front of them the clocks are still ramping and whichever form is timed first loses), the two forms
        # in turn, twice, the better time of each.  Collective: every rank times both, the decision is made on the slowest rank's times.
        from pothoscomms_amd.stream import PINGPONG_SLOTS
        slots_p = args.rehearse_slots or W.owner.slots or PINGPONG_SLOTS
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = [[], []]
        for rnd in range(2):
            for which, (pipe, sl) in enumerate(((True, slots_p), (False, 1024))):
                W.owner.pipeline = pipe
                W.owner.set_slots(sl)
                for _ in range(60):
                    step()
                e0.record()
                for _ in range(100):
                    step()
                e1.record()
                torch.cuda.synchronize()
                times[which].append(e0.elapsed_time(e1) / 100)
                _hb("autotune %d/4" % (2 * rnd + which + 1))
        t = torch.tensor([min(times[0]), min(times[1])], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        best = [float(v) for v in t.tolist()]
        pipe = best[0] <= best[1]
        W.owner.pipeline = pipe
        W.owner.set_slots(slots_p if pipe else 1024)
        tuned = {"pipelined_ms": round(best[0], 4), "unpipelined_ms": round(best[1], 4), "chosen": "pipelined" if pipe else "unpipelined",
                 "note": "100 passes of either form, twice in turn, behind the settling passes (best of each, slowest rank); --no-autotune takes the pipelined form unmeasured"}
        for _ in range(60):
            step()
    for _ in range(args.warmup):
        step()
    before = fell_back
    gate_check_and_fall_back("during the setup passes")
    if fell_back and not before:
        for _ in range(args.settle // 4 + args.warmup):      # the setup once more, in the form the timed region will run
            step()
    _hb("setup done")
    _test_fault("timed", rank)

    def timed_region():
        """EXACTLY K steps between barrier + synchronize on both sides -> (wall seconds, max over ranks; ms per launch from ONE pair of HIP
        events on the launch stream around the K steps: no event packet between two launches -- a pair per step costs ~12 us of gap per
        step on this stack)."""
        # (torch creates the HIP event behind an Event object at its FIRST record(): created here, not between the clock reading and the
        # first timed launch -- at the driver's --steps 20 the two creations read as 1-3 % of the region, tools/wall_vs_events.py)
        ev0.record()
        ev1.record()
        barrier()
        t0 = time.perf_counter()
        ev0.record()
        for k in range(args.steps):
            step()
        ev1.record()
        barrier()
        wall = time.perf_counter() - t0
        span_ms = ev0.elapsed_time(ev1) / args.steps
        if world > 1:
            t = torch.tensor([wall], dtype=torch.float64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, span_ms

    elapsed, span_ms = timed_region()
    kern_ms = [span_ms]
    _hb("timed region done")
    # Behind the timed region, outside `value`: the same launches kept up for about a second, the LAST half of them timed.  The
    # package power cap lets the clock sag over the first seconds of a run (profiles/r03/README.md: a 3 s loop reads 3-4 % below a
    # 0.13 s one on the same box), so the line carries both numbers; single GPU only.
    sustained = None
    cold = None
    if world == 1 and args.sustain > 0:
        n_s = max(200, int(args.sustain / max(kern_ms[0] * 1e-3, 1e-6)))
        es0, es1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(n_s):
            if k == n_s // 2:
                es0.record()
            step()
        es1.record()
        torch.cuda.synchronize()
        sustained = (es0.elapsed_time(es1) / (n_s - n_s // 2), n_s)
    clk = None
    if world == 1 and not args.no_cold:
        clk = clock_under_load(step)
        cold = cold_launches(step)
    seams = None
    retimed = None
    if W.owner is not None and (world > 1 or rehearsal) and hasattr(W.owner, "ring"):
        _BREAK_EXCHANGE[0] = os.environ.get("PCX_BENCH_TEST_BREAK_SEAM", "")

        pretend = [os.environ.get("PCX_BENCH_TEST_GATE_TIMEOUT") == "%d:timed" % rank]    # tests only: ONE pretended timeout behind the timed region

        def gate_timed_out():
            """a pass whose gated launch ran without its halo (the bounded wait of pcx_fir_process_dev_gated) is not a measurement"""
            try:
                W.owner.check_gate()
            except RuntimeError as e:
                return str(e)
            if pretend[0]:
                pretend[0] = False
                return "a gate timed out (pretended: PCX_BENCH_TEST_GATE_TIMEOUT)"
            return ""

        def seams_wrong():
            """the gate check and the seam check on every rank -> "" or what is wrong on SOME rank (this rank's own finding first)"""
            # all three on EVERY rank, whatever the first finds: seam_check is a step of the ring (owner.step() posts the send / recv with the
            # neighbours), and a rank that skipped it because its own gate had timed out left its neighbours waiting in it while it
            # waited for them in the all_reduce below (`a or b or c` short-circuits)
            g1 = gate_timed_out()
            sc = seam_check(W.owner, W.owner.ring.rank > 0)
            g2 = gate_timed_out()
            wrong = g1 or sc or g2
            bad = torch.tensor([1 if wrong else 0], dtype=torch.int32, device=ctl if world > 1 else "cpu")
            if world > 1:
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if wrong:
                print("bench.py: rank %d, seam check: %s" % (rank, wrong), file=sys.stderr, flush=True)
            if int(bad.item()) and world > 1:              # (every rank knows `bad`: every rank gathers) whose seam, and what it saw
                found = [None] * world
                dist.all_gather_object(found, wrong)
                return "; ".join("rank %d: %s" % (r, w) for r, w in enumerate(found) if w)
            return wrong

        wrong = seams_wrong()
        if wrong and not W.owner.two_launch:
            # The one-launch pass did not see its halo on this hardware.  Not the end: every rank switches to two launches per pass
            # (body, halo, head -- no gate, no pipelining), settles, and the K steps are timed AGAIN in that form and checked again;
            # the line then carries the second timing and says why (config.fallback_reason, config.retimed).
            retimed = {"first_form": "one gated launch per pass" + (", pipelined" if getattr(W.owner, "pipelined", False) else ""),
                       "first_form_ms_per_step": round(elapsed / args.steps * 1e3, 4), "first_form_seam_check": wrong}
            switch_to_two_launch("the seam check behind the timed region failed in the one-launch form (%s)" % wrong)
            for _ in range(args.settle // 4 + args.warmup):
                step()
            _hb("re-timing in the two-launch form")
            elapsed, span_ms = timed_region()
            kern_ms = [span_ms]
            retimed["ms_per_step"] = round(elapsed / args.steps * 1e3, 4)
            wrong = seams_wrong()
        if wrong:
            raise SystemExit("bench.py: the seam check behind the timed region failed on some rank (%s): not a measurement" % wrong)
        seams = "halo slots poisoned behind the timed region, one more pass per buffer: every rank's outputs at the shard front match a plain call on the completed buffer"
        _hb("seam check done")

    if rank == 0:
        value = world * W.units * args.steps / elapsed / 1e6
        avg_ms = float(np.mean(kern_ms))
        desc["world_size_observed"] = world
        desc["rank_devices"] = rank_devices
        if world > 1:
            desc["halo_backend"] = "rccl" if backend == "nccl" else backend + (" (halo staged through the host)" if len(set(rank_devices)) == world else " (rehearsal: ranks share a GPU)")
        if world > 1 or rehearsal:
            if getattr(W.owner, "two_launch", False):
                desc["halo_scheme"] = ("two launches per pass (body, halo, head): FALLBACK, " + fell_back if fell_back
                                       else "two launches per pass (body, halo, head): PCX_STREAM_TWO_LAUNCH")
            elif backend != "nccl" and not rehearsal:
                desc["halo_scheme"] = "two launches per pass (body, halo, head): a host-driven backend opens no gate"
            else:
                desc["halo_scheme"] = "one gated launch per pass"
            # which attempt of the supervisors this line comes from, and why the earlier ones (or the one-launch form) were given up
            desc["attempt"] = int(os.environ.get("PCX_BENCH_ATTEMPT", "1"))
            desc["attempt_mode"] = os.environ.get("PCX_BENCH_ATTEMPT_MODE", "as asked")
            reasons = json.loads(os.environ.get("PCX_BENCH_FALLBACK_REASON", "[]"))
            if fell_back:
                reasons.append("within attempt %d: two launches per pass, because %s" % (desc["attempt"], fell_back))
            desc["fallback_reason"] = reasons or None
            if retimed:
                desc["retimed"] = retimed
        if standin:
            desc["TEST_STAND_IN"] = "CPU stand-in for the device and the workload (tests/bench_standin.py): NOT a measurement"
        if W.owner is not None and (backend == "nccl" or rehearsal):
            from pothoscomms_amd.stream import exchange_shares_queue
            # (True would mean every exchange ran BEHIND the pass it should run beside: stream.py, HARDWARE QUEUES)
            desc["rccl_stream_shares_the_launch_queue"] = exchange_shares_queue(dev_index)
        if seams:
            desc["seam_check"] = seams
        if tuned:
            desc["halo_exchange_form"] = tuned
        if PIPELINED in str(desc.get("parallelism", "")) and not getattr(W.owner, "pipelined", False):
            # two buffers were built, but this run did not pipeline them (a host-driven backend, or the two-launch fall-back)
            desc["parallelism"] = desc["parallelism"].replace(PIPELINED, "; two input buffers taken in turn, every pass behind its own exchange (not pipelined: a host-driven backend, the two-launch fall-back, or the unpipelined form measured faster -- halo_backend / halo_scheme / halo_exchange_form)")
        if W.owner is not None and hasattr(W.owner, "slots"):
            desc["resident_workgroups_per_launch"] = W.owner.slots or 1024
        roof = roofline_of(W, avg_ms, sustained, cold, args.sustain, clk)
        # ONE line, TWO clocks, both stated: `value` / `ms_per_step` are the wall clock between the barriers (the contract: whole-job
        # throughput, launch gaps and the exchange included); `roofline.achieved` / `frac` are the dominant kernel's average launch
        # duration from HIP events on the launch stream.  roofline.wall_clock re-prices the same bytes on the wall clock, so that
        # value x bytes per sample / peak can be reproduced from the line.
        wall_ms = elapsed / args.steps * 1e3
        roof["clock"] = "HIP events on the launch stream around the K timed steps (avg_launch_ms); value and ms_per_step use the wall clock between the barriers"
        roof["wall_clock"] = {"ms_per_step": round(wall_ms, 4), "frac": round(roof["frac"] * avg_ms / wall_ms, 4),
                              "note": "the same algorithmic bytes (or flops) over the wall-clock step: = value x bytes per sample / n_gpus / peak"}
        out = {
            "metric": W.metric, "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall_ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": W.dtype,
            "data": "synthetic" if not standin else "TEST STAND-IN (no GPU): not a measurement", "config": desc,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu:
            _hb("cpu baseline")
            out["cpu_baseline"] = cpu_baseline_of(wl, C)
        if world == 1 and wl == "fir255" and not args.no_secondary and not rehearsal:
            # BASELINE.json configs[2] and configs[4], measured by the same command so that the driver's own run covers them
            del W, step
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            _hb("secondary workloads")
            t_sec = time.perf_counter()
            sec = {"fft4096": guarded("fft4096", measure_secondary, "fft4096", dev, args),
                   "fmchain": guarded("fmchain", measure_secondary, "fmchain", dev, args)}
            # ... the other rows of SURVEY 8: the element-wise blocks (one launch each), the bit-exact integer FIR on the double-precision
            # pipeline (its roof is FP64 issue, not HBM), the resampling FIRs
            sec["elementwise"] = {w: guarded(w, measure_secondary, w, dev, args) for w in ("rotate", "abs", "freq_demod")}
            sec["fir255_i16"] = guarded("fir255_i16", measure_secondary, "fir255_i16", dev, args)
            sec["resamplers"] = {w: guarded(w, measure_secondary, w, dev, args) for w in ("decim8", "interp4")}
            sec["long_taps"] = {w: guarded(w, measure_secondary, w, dev, args) for w in ("fir4097", "fir8193", "fir4097_real")}   # taps in partitions
            # ... configs[3] rehearsed on this one device through the native driver, and the end-to-end number of SURVEY 8d (PCIe inside)
            sec["c3_one_device"] = guarded("c3_one_device", measure_c3_one_device, args, avg_ms)
            sec["host_path"] = guarded("host_path", measure_host_path, args, out.get("cpu_baseline"))
            sec["seconds"] = round(time.perf_counter() - t_sec, 1)
            out["secondary"] = sec
        result_line = json.dumps(out)
        if world > 1:
            # a supervised rank: the line goes out BEFORE the process group is torn down -- the measurement is complete (timed region,
            # max-over-ranks clock, gate and seam checks), and a teardown that hangs on some rank must not cost it (bench_supervisor.py)
            _flush_c_stdio()
            print(result_line, flush=True)
    _test_fault("teardown", rank)
    if rehearsal:
        W.owner.check_gate()
        dist.destroy_process_group()
    if world > 1:
        barrier()
        dist.destroy_process_group()
    if rank == 0 and world == 1:
        _flush_c_stdio()            # whatever RCCL printed through C stdio goes out first: the result line is the last thing on stdout
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
