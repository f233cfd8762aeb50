"""Which streams share the NULL stream's hardware queue?  HIP maps a process's streams onto a few HSA queues (4 per priority); two streams on
one queue run in order, whatever the program says.  A one-thread sleep kernel (~2 ms) on the null stream, then a tiny kernel + event on the
candidate: if the candidate's event completes while the sleeper still runs, the two are on different queues.
usage: queue_map_probe.py [rccl|rccl-hp]    (rccl: also where RCCL's own stream lands, through a 2 KB exchange with the rank itself)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
word = torch.zeros((64,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def shares_queue_with_current(fn):
    """fn() queues something on some other stream and returns an event behind it"""
    end = torch.cuda.Event()
    torch.cuda._sleep(4_000_000)          # ~2 ms of one thread
    end.record()
    ev = fn()
    ev.synchronize()
    concurrent = not end.query()
    torch.cuda.synchronize()
    return not concurrent


def on_stream(s):
    def fn():
        with torch.cuda.stream(s):
            word.zero_()
            ev = torch.cuda.Event(); ev.record()
        return ev
    return fn


for prio, name in ((0, "normal"), (-1, "high")):
    ss = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(12)]
    print("%-6s priority, 12 streams from torch's pool: shares the null stream's queue: %s" % (name, " ".join("Y" if shares_queue_with_current(on_stream(s)) else "." for s in ss)))
# libpcx's own streams (hipStreamNonBlocking, created by the handles) shift the assignment: create a few handles and look again
from pothoscomms_amd import device, taps as tp
hs = [device.FmChain() for _ in range(3)]
ss = [torch.cuda.Stream(device=dev) for _ in range(12)]
print("normal priority, 12 more after three FmChain handles:                 %s" % " ".join("Y" if shares_queue_with_current(on_stream(s)) else "." for s in ss))

if len(sys.argv) > 1 and sys.argv[1].startswith("rccl"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
    if sys.argv[1] == "rccl-hp":
        dist.init_process_group("nccl", rank=0, world_size=1, pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
    buf = torch.zeros((1024,), dtype=torch.float32, device=dev)
    side = next(s for s in ss if not shares_queue_with_current(on_stream(s)))

    def exchange():
        with torch.cuda.stream(side):
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf[512:], 0), dist.P2POp(dist.irecv, buf[:512], 0)]):
                w.wait()
            ev = torch.cuda.Event(); ev.record()
        return ev
    exchange().synchronize()              # (the communicator is created here)
    torch.cuda.synchronize()
    for pre in (0, 1, 2, 3, 5):
        print("%s: RCCL's stream shares the null stream's queue: %s" % (sys.argv[1], "Y" if shares_queue_with_current(exchange) else "."))
    dist.destroy_process_group()
