"""Which PART of a middle rank's RCCL pass makes it ~10 % longer than the plain launch?  One GPU, a process group of one over RCCL, the halo sent
to the rank itself; on DATA; each variant 600 settling passes + 1500 timed (HIP events on the launch stream), the list run twice:
  plain        the plain launch (what rank 0 runs)
  gate         the gated launch + the gate signal on the side stream, NO exchange
  rccl         the exchange on the side stream beside a PLAIN launch that does not wait for it
  rccl+gate    the pass as shipped (ShardedFir.step)
  pingpong     two buffers, the exchange of batch k+1 beside the pass of batch k (PingPongFir.step)
each at 1024 resident workgroups and at the slots given on the command line (default 896 768).  First argument `chain`: the fused chain's pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from pothoscomms_amd import device, taps as tp
from pothoscomms_amd.stream import ShardedFir, ShardedFmChain, PingPongFir, PingPongFmChain, HaloRing

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
from pothoscomms_amd import stream as _st
HP_RCCL, HP_SIDE = os.environ.get("PROBE_HP_RCCL") == "1", os.environ.get("PROBE_HP_SIDE") == "1"      # A/B: stream priorities (profiles/r04/prio_matrix.txt)
dist.init_process_group("nccl", rank=0, world_size=1, pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True) if HP_RCCL else None)


def hp_side(owner):
    if HP_SIDE:
        for h in getattr(owner, "halves", [owner]):
            h._gate_setup()
            h._side = torch.cuda.Stream(device=dev, priority=-1)


class SelfRing(HaloRing):
    def __init__(self, halo):
        self.halo = halo; self.group = None; self.rank = 1; self.world = 3

    def start(self, buf):
        return dist.batch_isend_irecv([dist.P2POp(dist.isend, buf[buf.shape[0] - self.halo:], 0), dist.P2POp(dist.irecv, buf[:self.halo], 0)])


class NullRing(SelfRing):
    def start(self, buf):
        return []


C = 64 * 1024 * 1024
chain = len(sys.argv) > 1 and sys.argv[1] == "chain"
slot_list = [None] + [int(a) for a in (sys.argv[2 if chain else 1:] or ["896", "768"])]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(step, n_settle=600, n=1500):
    for _ in range(n_settle):
        step()
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
    for slots in slot_list:
        sf = ShardedFmChain(tp.c4_taps(), tp.C4_PHASE, C, dev, slots=slots) if chain else ShardedFir(tp.c1_taps(), C, dev, slots=slots)
        device.fill_uniform_f32_dev(sf.buf, seed=2, offset=0)
        K = sf.K + 1 if chain else sf.K                    # (the chain's halo is K samples)
        row = {}
        sf.ring = SelfRing(K - 1)
        hp_side(sf)
        plain = (lambda: sf._run(sf._chains[0], 1, C, 1)) if chain else (lambda: sf._run(0, C))
        sf._run0 = plain
        row["plain"] = timed(plain)
        sf.ring = NullRing(K - 1)
        row["gate"] = timed(sf.step)
        sf.ring = SelfRing(K - 1)

        def rccl_beside():
            cur = torch.cuda.current_stream(dev)
            sf._side.wait_stream(cur)
            with torch.cuda.stream(sf._side):
                sf.ring.finish(sf.ring.start(sf._buf))
            sf._run0()
        row["rccl"] = timed(rccl_beside)
        row["rccl+gate"] = timed(sf.step)
        sf.check_gate()
        del sf
        pp = PingPongFmChain(tp.c4_taps(), tp.C4_PHASE, C, dev) if chain else PingPongFir(tp.c1_taps(), C, dev)
        for h in pp.halves:
            device.fill_uniform_f32_dev(h.buf, seed=2, offset=0)
        pp.set_slots(slots or 1024)
        pp.ring = SelfRing(K - 1)
        hp_side(pp)
        row["pingpong"] = timed(pp.step)
        pp.check_gate()
        del pp
        print("slots %-5s " % (slots or 1024) + "  ".join("%s %.1f us" % kv for kv in row.items()), flush=True)
print("RCCL stream %s, side stream %s priority; exchange shares the launch queue: %s" % ("high" if HP_RCCL else "normal", "high" if HP_SIDE else "normal", _st._QUEUES_CHECKED))
dist.destroy_process_group()
