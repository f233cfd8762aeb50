"""pcx_shard peer-copy passes (scatter, step, gather with nothing in between) on pageable memory: is the FIRST pass right?
Run with AMD_DIRECT_DISPATCH=0 as well: that mode showed a zero shard in the first pass until every bounce buffer was allocated
before the first transfer was queued."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pothoscomms_amd import device, taps as tp
from oracle import oracle as o
h = tp.c1_taps(); K = len(h); G, Cs = 3, 3000
x = o.fill_uniform_f32(2 * (K - 1 + G * Cs), 2, 0).reshape(-1, 2)
blk = o.Fir(o.F32, True, True); blk.set_taps(h); blk.activate()
ref = blk.work(x, G * Cs)[0]
ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
ns.set_taps(h); ns.configure(Cs)
for p in range(2):
    ns.scatter(x); ns.step(); got = ns.gather()
    print("pass", p, "zero rows", int(np.sum(np.all(got == 0, axis=1))), "max err", float(np.abs(got - ref).max()))
