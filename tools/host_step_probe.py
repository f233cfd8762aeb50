"""Host time to QUEUE one pass of the rank driver's RCCL path (ShardedFir._step_gated) -- it must stay below the pass's device time
(0.19 ms) or N > 1 becomes host-bound.  One GPU, one rank: the halo is sent to the rank ITSELF through RCCL (send + receive in one
group), everything else exactly as a middle rank of a node queues it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from pothoscomms_amd import taps as tp
from pothoscomms_amd.stream import ShardedFir, ShardedFmChain, HaloRing

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hp = len(sys.argv) > 2 and sys.argv[2] == "hp"          # RCCL's own streams at high priority
if hp:
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", rank=0, world_size=1, pg_options=opts)
elif "devid" in sys.argv[2:]:                             # eager communicator bound to the device, as bench.py's ranks create it
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("nccl", rank=0, world_size=1)


class SelfRing(HaloRing):
    def __init__(self, halo):
        self.halo = halo; self.group = None; self.rank = 1; self.world = 3      # a middle rank: sends AND receives, has a gate

    def start(self, buf):
        ops = [dist.P2POp(dist.isend, buf[buf.shape[0] - self.halo:], 0), dist.P2POp(dist.irecv, buf[:self.halo], 0)]
        return dist.batch_isend_irecv(ops)


C = 64 * 1024 * 1024
slots = int(sys.argv[1]) if len(sys.argv) > 1 else None      # resident workgroups the gated launch takes (default: all 1024)
chain = "chain" in sys.argv[2:]                            # the fused chain's pass instead of the FIR's (halo of K samples)
if chain:
    sf = ShardedFmChain(tp.c4_taps(), tp.C4_PHASE, C, dev, slots=slots)
    sf.ring = SelfRing(sf.K)
else:
    sf = ShardedFir(tp.c1_taps(), C, dev, slots=slots)
    sf.ring = SelfRing(sf.K - 1)
from pothoscomms_amd import device
device.fill_uniform_f32_dev(sf.buf, seed=2, offset=0)     # DATA, not zeros: on all-zero input the power cap lets go of the clock and every pass reads ~8 % faster
for _ in range(50):
    sf.step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    sf.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(("high-priority RCCL streams, " if hp else "") + ("device_id given, " if "devid" in sys.argv[2:] else "") + ("fused chain, " if chain else "") + "slots %s  host: %.1f us to queue a pass; device: %.1f us per pass (queue + drain of %d passes)" % (slots, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, n))
dist.destroy_process_group()
