"""Does a gate signal queued AFTER the gated launch, on another stream, always get through?  HIP maps a process's streams onto four
hardware queues and every packet carries the barrier bit: a signal whose stream shares the launch's hardware queue waits for the
launch, which waits for the signal (until the gate's 2 s bound).  tools/gate_queue_probe.py [priority]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pothoscomms_amd import device, taps as tp

prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
h = tp.c1_taps(); K = len(h); n = 2100 * 3840
dev = torch.device("cuda", 0)
lead = (-(K - 1)) % 16
xa = torch.empty((lead + K - 1 + n, 2), dtype=torch.float32, device=dev); x = xa[lead:]
device.fill_uniform_f32_dev(x, seed=9, offset=0)
f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
got = torch.empty((n, 2), dtype=torch.float32, device=dev)
gate = torch.zeros((64,), dtype=torch.int32, device=dev)
value = 0
for k in range(10):
    side = torch.cuda.Stream(device=dev, priority=prio)
    value += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c, p, gated = f.process_dev_gated(x, got, gate, value)
    time.sleep(0.02)
    device.gate_signal(gate, value, side.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("side stream %d (priority %d): %.3f s  gate[1]=%#x" % (k, prio, dt, int(gate[1].item()) & 0xffffffff))
    gate[1] = 0
