"""Two /comms/fir_filter blocks on two scheduler threads, each calling work() on its own pinned port buffers (what two
Pothos actors do).  Every handle owns a stream (pcx_host.hpp ExecCtx), so their kernels overlap instead of serialising
on the legacy default stream.

    python tools/two_blocks.py                       aggregate rate, one thread vs two
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/two -- python3 tools/two_blocks.py trace
    python tools/two_blocks.py summarize gpurun_out/two      overlap found in the kernel trace
"""
import csv, glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(reps, nthreads, n=1 << 20):
    import numpy as np
    from pothoscomms_amd import blocks, taps as tp
    K = 255
    blks = []
    for _ in range(nthreads):
        b = blocks.make("/comms/fir_filter", "complex_float32", "COMPLEX")
        b.call("setTaps", tp.c1_taps()); b.activate()
        x, _ = b.port_buffer(0, (n + K - 1, 2), np.float32)
        y, _ = b.port_buffer(1, (n, 2), np.float32)
        x[:] = np.random.default_rng(1).uniform(-1, 1, x.shape).astype(np.float32)
        b.work(x, n, outbuf=y)
        blks.append((b, x, y))

    def loop(b, x, y):
        for _ in range(reps):
            b.work(x, n, outbuf=y)
    th = [threading.Thread(target=loop, args=t) for t in blks]
    t0 = time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    return nthreads * reps * n / dt / 1e9


def summarize(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "fir_cf32" in r["Kernel_Name"]]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
    overlap_ns, pairs = 0, 0
    for (s0, e0, q0), (s1, e1, q1) in zip(ev[:-1], ev[1:]):
        if s1 < e0 and q0 != q1:
            pairs += 1
            overlap_ns += min(e0, e1) - s1
    busy = sum(e - s for s, e, _ in ev)
    print("%d FIR kernel launches on %d streams/queues; %d consecutive pairs from different streams overlap in time, "
          "%.1f %% of the summed kernel time runs concurrently with the neighbour launch" % (len(ev), len(set(q for _, _, q in ev)), pairs, 100.0 * overlap_ns / max(busy, 1)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "summarize":
        summarize(sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "trace":
        run(40, 2)
    else:
        one = run(200, 1)
        two = run(200, 2)
        print("one block, one thread : %.2f Gsamples/s" % one)
        print("two blocks, two threads: %.2f Gsamples/s aggregate (x%.2f)" % (two, two / one))
