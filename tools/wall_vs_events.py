"""Where do the ~6 us per step go that bench.py's wall clock (value, ms_per_step) reads above its HIP events at the driver's --steps 20?
20 settled steps between synchronisation points, wall clock against one event pair, ten times; argument `spin`: hipSetDeviceFlags(
hipDeviceScheduleSpin) before the device is initialised (the host spins in hipDeviceSynchronize instead of sleeping on an interrupt)."""
import ctypes, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "spin" in sys.argv[1:]:
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(1))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
W = bench.build_workload("fir255", bench.SHARD, dev, 0, 1, types.SimpleNamespace(settle=0))
for _ in range(600):
    W.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for K in (20, 200):
    rows = []
    for rep in range(10):
        for _ in range(5):
            W.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K):
            W.step()
        e1.record()
        tq = time.perf_counter()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rows.append(((t1 - t0) / K * 1e6, e0.elapsed_time(e1) / K * 1e3, (tq - t0) * 1e6, (t1 - t0) * 1e6 - e0.elapsed_time(e1) * 1e3))
    rows.sort()
    m = rows[len(rows) // 2]
    print("K = %3d: wall %.2f us per step, events %.2f; host took %.0f us to queue the K steps; wall - events = %.0f us per region (median of 10 by wall)" % ((K,) + m))
