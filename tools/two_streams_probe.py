"""Two half-size FIR launches on two streams of ONE device against one full-size launch: what two shards on one device can reach
at best (no halo, no events) -- the floor under tools/shard_probe.py's G = 2 row."""
import sys, time
import torch
sys.path.insert(0, ".")
from pothoscomms_amd import device, taps as tp

dev = torch.device("cuda", 0)
h = tp.c1_taps(); K = len(h)
total = 64 << 20
def mk(n):
    lead = (-(K - 1)) % 16
    xa = torch.empty((lead + K - 1 + n, 2), dtype=torch.float32, device=dev)
    device.fill_uniform_f32_dev(xa[lead:], seed=2, offset=0)
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
    return f, xa[lead:], torch.empty((n, 2), dtype=torch.float32, device=dev)
for G in (1, 2, 4):
    parts = [mk(total // G) for _ in range(G)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    def step():
        for (f, x, y), s in zip(parts, streams):
            f.process_dev(x, y, stream=s)
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("G=%d independent launches on %d streams: %.4f ms per pass over %d samples" % (G, G, dt * 1e3, total))
