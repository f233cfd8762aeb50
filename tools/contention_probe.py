"""Several processes on ONE GPU, each creating handles and running small host-pointer calls in a loop: does every call
return the same answer as the first one?  (The fuzz soak with eight pytest workers on one device found calls whose output
had holes; a single process never shows them.)  Variants separate the suspects:

  pageable   FirFilter.process on numpy buffers: H2D staging copy, kernel, D2H staging copy, all on the handle's stream
  pinned     the same call on page-locked buffers: the kernel reads and writes them in place, no copies
  dev        process_dev on torch tensors on a torch stream, synchronised, then read back by torch
  demod      a fresh FreqDemod handle on pageable buffers (state zeroed at create)

Run:  python tools/contention_probe.py [procs] [iterations]
"""
import ctypes as C
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def pinned_array(lib, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    assert lib.pcx_host_alloc(C.byref(p), n) == 0
    buf = (C.c_char * n).from_address(p.value)
    return np.frombuffer(buf, dtype=dtype).reshape(shape), p


def holes(got, want):
    bad = np.flatnonzero(np.any(got != want, axis=-1) if got.ndim > 1 else got != want)
    if bad.size == 0:
        return None
    zero = int(np.sum(np.all(got[bad] == 0, axis=-1) if got.ndim > 1 else got[bad] == 0))
    runs = np.split(bad, np.flatnonzero(np.diff(bad) > 1) + 1)
    return "%d bad of %d (%d of them zero) in %d runs: %s" % (bad.size, got.shape[0], zero, len(runs),
                                                              ", ".join("%d..%d" % (r[0], r[-1]) for r in runs[:6]))


def worker(rank, iters, q):
    try:
        run(rank, iters, q)
    except Exception as e:     # the parent must not wait for a worker that died
        q.put((rank, {"error": ({"error": 1}, [repr(e)[:300]])}))


def run(rank, iters, q):
    import torch
    from pothoscomms_amd import _lib, device as dev
    lib = _lib.load()
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    out = {}
    for K in (8194, 255):
        n_in = K - 1 + 1317 if K > 8000 else 40000
        n_out = n_in - K + 1
        taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / np.sqrt(K)
        x = rng.standard_normal((n_in, 2)).astype(np.float32)
        first = None
        xp, _ = pinned_array(lib, (n_in, 2), np.float32)
        yp, _ = pinned_array(lib, (n_out, 2), np.float32)
        xp[:] = x
        xd = torch.from_numpy(x).to(d)
        counts = {"pageable": 0, "pinned": 0, "dev": 0}
        notes = []
        for it in range(iters):
            f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(taps)
            got, c, p = f.process(x, n_out)
            if first is None:
                first = got.copy()
            h = holes(got, first)
            if h:
                counts["pageable"] += 1; notes.append("pageable it %d: %s" % (it, h))
            yp[:] = 0
            cc, pp = C.c_size_t(), C.c_size_t()
            _lib.check(lib.pcx_fir_process(f._h, xp.ctypes.data_as(C.c_void_p), n_in, yp.ctypes.data_as(C.c_void_p), n_out, C.byref(cc), C.byref(pp)))
            h = holes(yp, first)
            if h:
                counts["pinned"] += 1; notes.append("pinned it %d: %s" % (it, h))
            yd = torch.zeros((n_out, 2), dtype=torch.float32, device=d)
            s = torch.cuda.Stream(device=d)
            s.wait_stream(torch.cuda.current_stream())
            f.process_dev(xd, yd, n_in, n_out, stream=s)
            s.synchronize()
            h = holes(yd.cpu().numpy(), first)
            if h:
                counts["dev"] += 1; notes.append("dev it %d: %s" % (it, h))
            del f
        out["K=%d" % K] = (counts, notes[:6])
    x = (rng.standard_normal((6000, 2)) + 0.1).astype(np.float32)
    first, bad, notes = None, 0, []
    for it in range(iters * 3):
        blk = dev.FreqDemod("complex_float32")
        a = blk.process(x[:3000]); b = blk.process(x[3000:])
        got = np.concatenate([a, b])
        if first is None:
            first = got.copy()
        h = holes(got, first)
        if h:
            bad += 1; notes.append("demod it %d: %s" % (it, h))
        del blk
    out["demod"] = ({"demod": bad}, notes[:6])
    q.put((rank, out))


def run_reused(rank, iters, q):
    """handles created ONCE, then many host-pointer calls of a few sizes: a difference here cannot be a create-time race"""
    from pothoscomms_amd import _lib, device as dev
    lib = _lib.load()
    rng = np.random.default_rng(11 + rank)
    out = {}
    for K in (8194, 255, 31):
        taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / np.sqrt(K)
        f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(taps)
        sizes = [K - 1 + m for m in (1317, 5000, 20011)]
        xs = [rng.standard_normal((n, 2)).astype(np.float32) for n in sizes]
        first = [None] * len(sizes)
        bad, notes = 0, []
        for it in range(iters):
            j = it % len(sizes)
            got, c, p = f.process(xs[j], sizes[j] - K + 1)
            if first[j] is None:
                first[j] = got.copy()
            h = holes(got, first[j])
            if h:
                bad += 1; notes.append("reused it %d size %d: %s" % (it, sizes[j], h))
        out["K=%d" % K] = ({"reused": bad}, notes[:6])
    q.put((rank, out))


def run_fresh(rank, iters, q):
    """a NEW handle for every call (tables uploaded, state zeroed, first kernel at once): create-time races only"""
    from pothoscomms_amd import _lib, device as dev
    _lib.load()
    rng = np.random.default_rng(23 + rank)
    out = {}
    for K in (63, 300):
        taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / np.sqrt(K)
        n = K - 1 + 3000
        x = rng.standard_normal((n, 2)).astype(np.float32)
        first, bad, notes = None, 0, []
        for it in range(iters):
            f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(taps)
            got, c, p = f.process(x, 3000)
            del f
            if first is None:
                first = got.copy()
            h = holes(got, first)
            if h:
                bad += 1; notes.append("fresh it %d: %s" % (it, h))
        out["K=%d" % K] = ({"fresh": bad}, notes[:6])
    q.put((rank, out))


def worker_fresh(rank, iters, q):
    try:
        run_fresh(rank, iters, q)
    except Exception as e:
        q.put((rank, {"error": ({"error": 1}, [repr(e)[:300]])}))


def run_hetero(rank, iters, q):
    """what the fuzz suite does: a new handle of a random kind and size per iteration, memory of many sizes allocated and
    freed in between, one host-pointer call -- then THE SAME CALL twice more on the same handle.  first != second with
    second == third says the first call of a handle saw something unfinished; a lasting difference says a table is wrong."""
    from pothoscomms_amd import _lib, device as dev
    _lib.load()
    rng = np.random.default_rng(101 + rank)
    counts = {"first_call_differs": 0, "later_call_differs": 0}
    notes = []
    for it in range(iters):
        K = int(rng.choice([1, 2, 17, 63, 255, 300, 1023, 2049, 2050, 4097, 8193, 8194]))
        ctaps = bool(rng.integers(0, 2))
        kind = int(rng.integers(0, 3))
        m = int(rng.integers(1, 30000))
        taps = (rng.normal(size=K) + (1j * rng.normal(size=K) if ctaps else 0)) / np.sqrt(K)
        if kind == 0:
            f = dev.FirFilter("complex_float32", "COMPLEX" if ctaps else "REAL"); f.set_taps(taps)
            x = rng.standard_normal((K - 1 + m, 2)).astype(np.float32)
            call = lambda: f.process(x, m)[0]
        elif kind == 1:
            f = dev.FirFilter("complex_float64", "COMPLEX" if ctaps else "REAL"); f.set_taps(taps)
            x = rng.standard_normal((K - 1 + m, 2))
            call = lambda: f.process(x, m)[0]
        else:
            f = dev.FirFilter("complex_int16", "COMPLEX" if ctaps else "REAL"); f.set_taps(taps * 0.3)
            x = rng.integers(-3000, 3000, (K - 1 + m, 2)).astype(np.int16)
            call = lambda: f.process(x, m)[0]
        a = call(); b = call(); c = call()
        if not np.array_equal(b, c):
            counts["later_call_differs"] += 1
            notes.append("it %d kind %d K %d m %d: second vs third %s" % (it, kind, K, m, holes(b, c)))
        if not np.array_equal(a, c):
            counts["first_call_differs"] += 1
            notes.append("it %d kind %d K %d m %d ctaps %d algo %d: first vs third %s" % (it, kind, K, m, ctaps, f.last_algo, holes(a, c)))
        if it % 3 == 0:
            d = dev.FreqDemod("complex_float32"); d.process(x[:100].astype(np.float32)); del d
        del f
    q.put((rank, {"hetero": (counts, notes[:10])}))


def worker_hetero(rank, iters, q):
    try:
        run_hetero(rank, iters, q)
    except Exception as e:
        q.put((rank, {"error": ({"error": 1}, [repr(e)[:300]])}))


def run_streams(rank, iters, q):
    """consecutive chunks of one FreqDemod stream on ALTERNATING streams with no host synchronisation: the handle orders them
    with an event (pcx_api.hip ctx_enter); the carried _prev makes any overtaking visible at the chunk's first sample"""
    import torch
    from pothoscomms_amd import _lib, device as dev
    _lib.load()
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(301 + rank)
    n = 1 << 18
    ph = np.cumsum(rng.uniform(-1.0, 1.0, n))
    xh = np.stack([np.cos(ph), np.sin(ph)], 1).astype(np.float32)
    x = torch.from_numpy(xh).to(d)
    streams = [torch.cuda.Stream(d) for _ in range(3)]
    first, bad, notes = None, 0, []
    for it in range(iters):
        y = torch.zeros(n, dtype=torch.float32, device=d)
        torch.cuda.synchronize()
        dm = dev.FreqDemod("complex_float32")
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(1, n, 40)]))
        if first is not None:
            cuts = first[1]
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            dm.process_dev(x[a:b], y[a:b], b - a, stream=streams[i % 3])
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        if first is None:
            first = (got.copy(), cuts)
        h = holes(got, first[0])
        if h:
            bad += 1; notes.append("streams it %d: %s" % (it, h))
        del dm
    q.put((rank, {"streams": ({"streams": bad}, notes[:6])}))


def worker_streams(rank, iters, q):
    try:
        run_streams(rank, iters, q)
    except Exception as e:
        q.put((rank, {"error": ({"error": 1}, [repr(e)[:300]])}))


def worker_threads(rank, iters, q):
    """the Pothos picture: one process, one actor thread per block -- eight threads here, each with its own handles"""
    import queue
    import threading
    tq = queue.Queue()
    ts = [threading.Thread(target=worker_hetero, args=(rank * 100 + t, iters, tq)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    merged = {}
    notes = []
    while not tq.empty():
        _, out = tq.get()
        for key, (counts, ns) in out.items():
            for k, v in counts.items():
                merged[k] = merged.get(k, 0) + v
            notes += ns
    q.put((rank, {"threads": (merged, notes[:10])}))


def worker_reused(rank, iters, q):
    try:
        run_reused(rank, iters, q)
    except Exception as e:
        q.put((rank, {"error": ({"error": 1}, [repr(e)[:300]])}))


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    global worker
    if len(sys.argv) > 3 and sys.argv[3] == "reused":
        worker = worker_reused
    if len(sys.argv) > 3 and sys.argv[3] == "threads":
        worker = worker_threads
    if len(sys.argv) > 3 and sys.argv[3] == "streams":
        worker = worker_streams
    if len(sys.argv) > 3 and sys.argv[3] == "hetero":
        worker = worker_hetero
    if len(sys.argv) > 3 and sys.argv[3] == "fresh":
        worker = worker_fresh
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, iters, q)) for r in range(procs)]
    for p in ps:
        p.start()
    res = []
    for _ in ps:
        try:
            res.append(q.get(timeout=400))
        except Exception:
            print("a worker did not report")
            break
    for p in ps:
        p.join(timeout=5)
        if p.is_alive():
            p.terminate()
    total = {}
    for rank, out in sorted(res):
        for key, (counts, notes) in out.items():
            for k, v in counts.items():
                total[(key, k)] = total.get((key, k), 0) + v
            for n in notes:
                print("rank %d %s %s" % (rank, key, n))
    print("%d processes x %d iterations, calls that differed from the process's first call:" % (procs, iters))
    for (key, k), v in sorted(total.items()):
        print("  %-8s %-9s %d" % (key, k, v))


if __name__ == "__main__":
    main()
