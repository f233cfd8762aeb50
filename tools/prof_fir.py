"""Minimal driver for rocprofv3: runs ONE bench workload a few times (no settling, no CPU leg) -- the same buffers, taps and calls as
bench.py --workload <wl> (bench.build_workload), so that the PMC passes measure the kernel the bench line times.
    rocprofv3 --kernel-trace --stats -- python3 tools/prof_fir.py fir255 5"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "fir255"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
W = bench.build_workload(wl, bench.SHARD, dev, 0, 1, types.SimpleNamespace(settle=0))
for _ in range(n):
    W.step()
torch.cuda.synchronize()
print("done", wl)
