"""The product kernels next to their floors (tools/floor_lab): for fir255, fmchain, decim8 and interp4 the kernel on the bench's data,
the same kernel on ALL-ZERO input (same instructions without the switching energy: the power cap lets go of the clock), and the shader
clock each runs at -- on the same box, in the same call as tools/floor_lab.
    python tools/floor_table.py [seconds-per-row]"""
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print("%-38s %-8s %10s %8s %8s %8s" % ("workload", "row", "ms/launch", "GB/s", "of 8 TB/s", "MHz"))
for wl in ("fir255", "fmchain", "decim8", "interp4"):
    W = bench.build_workload(wl, bench.SHARD, dev, 0, 1, types.SimpleNamespace(settle=0))
    for row in ("data", "zeros"):
        if row == "zeros":
            for t in W.inputs:
                t.zero_()
        for _ in range(400):
            W.step()
        torch.cuda.synchronize()
        probe = bench.time_launches(W.step, 50)
        ms = bench.time_launches(W.step, max(100, int(secs / (probe * 1e-3))))
        clk = bench.clock_under_load(W.step)
        gbs = W.roof_bytes / ms / 1e6
        print("%-38s %-8s %10.4f %8.0f %8.4f %8.0f" % (W.kernel_name, row, ms, gbs, gbs / 8000.0, clk or 0), flush=True)
    del W
    torch.cuda.empty_cache()
lab = os.path.join(ROOT, "tools", "floor_lab")
if os.path.exists(lab):
    sys.stdout.flush()
    subprocess.run([lab, str(secs)], check=False)
