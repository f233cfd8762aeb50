"""pcx_shard_* on ONE device: G peer-copy shards of 64 Mi / G samples each against one 64 Mi-sample pass.
What it shows: the cost of the native driver's head/body split, events and halo copies (no second GPU needed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from pothoscomms_amd import _lib, device, taps as tp

if len(sys.argv) < 2:
    # one FRESH process per shard count: a process that has created and destroyed streams maps its next streams onto the device's
    # four hardware queues differently, and two shards whose streams share a queue run one after the other
    # (same G = 2, same library: 0.1994 ms in a fresh process, 0.2457 behind a G = 1 run in the same process)
    import subprocess
    for G in (1, 2, 4, 8):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(G)], check=False)
    sys.exit(0)
L = _lib.load()
total = int(os.environ.get("PCX_PROBE_TOTAL", 64 * 1024 * 1024))     # PCX_PROBE_TOTAL=536870912 with G = 8: configs[3]'s shards on one device
h = tp.c1_taps()
for G in [int(a) for a in sys.argv[1:]]:
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY if G > 1 else device.NodeStream.RCCL)
    threads = bool(os.environ.get("PCX_PROBE_THREADS"))     # pcx_shard_set_submit_threads: a thread per device queues its share of a pass
    ns.set_submit_threads(threads)
    ns.set_taps(h)
    ns.configure(total // G)
    for g in range(G):
        i, o, s, d = ns.buffers(g)
        _lib.check(L.pcx_fill_uniform_f32_dev(C.c_void_p(i), 2 * (len(h) - 1 + total // G), 2, 2 * g * (total // G), C.c_void_p(s)))
    import os as _os
    short = bool(_os.environ.get("PCX_PROBE_SHORT"))
    for _ in range(20 if short else max(40, 400 * (64 << 20) // total)):
        ns.step()
    ns.sync()
    t0 = time.perf_counter()
    n = 10 if short else max(30, 300 * (64 << 20) // total)
    for _ in range(n):
        ns.step()
    host = (time.perf_counter() - t0) / n        # what the host spends queueing one pass
    ns.sync()
    dt = (time.perf_counter() - t0) / n
    print("G=%d shards on device 0 (%s): %.4f ms per pass over %d samples = %.1f Gsamples/s  (host: %.4f ms to queue a pass)" %
          (G, "rccl comm of one" if G == 1 else "peer copies" + (", SUBMIT THREADS" if threads else ""), dt * 1e3, total, total / dt / 1e9, host * 1e3), flush=True)
    ns.close()
