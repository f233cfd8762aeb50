"""How long does a 2 KB RCCL send + receive of a rank to ITSELF take on an otherwise idle GPU?  (tools/host_step_probe.py sees the
receive kernel resident for ~150 us under the FIR: the load's doing, or RCCL's own?)"""
import os, sys, time
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
a = torch.ones((254, 2), device="cuda"); b = torch.zeros((254, 2), device="cuda")
def xchg():
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):
        w.wait()
for _ in range(20): xchg()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(100): xchg()
    e1.record(); torch.cuda.synchronize()
    print("idle GPU: %.1f us per self send+receive (100 back to back)" % (e0.elapsed_time(e1) * 10))
dist.destroy_process_group()
