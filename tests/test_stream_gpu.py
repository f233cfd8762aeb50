"""GPU suite: the multi-rank pass end to end with the real kernels -- two processes share cuda:0
(the test box has one GPU) and exchange the halo over gloo; on the 8-GPU node the same code runs
one rank per GPU over RCCL.  The concatenated shard outputs must equal the single-stream oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.util import TOL, nerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, C, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pothoscomms_amd import device, taps as tp
    from pothoscomms_amd.stream import ShardedFir
    torch.cuda.set_device(0)
    sf = ShardedFir(tp.c1_taps(), C, torch.device("cuda", 0))
    K = sf.K
    # every rank fills [halo | shard] positions from the node-wide stream; ranks > 0 then poison
    # the halo so only a working exchange can make the result right
    device.fill_uniform_f32_dev(sf.buf, seed=2, offset=2 * rank * C)
    if rank > 0:
        sf.buf[:K - 1] = float("nan")
    out = sf.step()
    torch.cuda.synchronize()
    q.put((rank, out.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_kernels_no_seam(oracle):
    from pothoscomms_amd import taps as tp
    world, C = 2, 40000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, q)) for r in range(world)]
    [p.start() for p in procs]
    parts = dict(q.get(timeout=300) for _ in range(world))
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    got = np.concatenate([parts[r] for r in range(world)])
    h = tp.c1_taps()
    K = len(h)
    stream = oracle.fill_uniform_f32(2 * (K - 1 + world * C), 2, 0).reshape(-1, 2)
    ref_blk = oracle.Fir(oracle.F32, True, True); ref_blk.set_taps(h); ref_blk.activate()
    ref, _, p, _ = ref_blk.work(stream, world * C)
    assert p == world * C and not np.isnan(got).any()
    assert nerr(got, ref) <= TOL


def _gated_worker(rank, world, port, C, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pothoscomms_amd import device, taps as tp
    from pothoscomms_amd.stream import ShardedFir
    torch.cuda.set_device(0)
    ShardedFir.gate_host_driven = True            # the one-launch step (what the RCCL backend runs) with gloo carrying the halo
    sf = ShardedFir(tp.c1_taps(), C, torch.device("cuda", 0))
    K = sf.K
    device.fill_uniform_f32_dev(sf.buf, seed=2, offset=2 * rank * C)
    seam = []
    for rep in range(2):
        if rank > 0:
            sf.buf[:K - 1] = float("nan")
        out = sf.step()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        if rank > 0:
            assert int(sf._gate[0].item()) == rep + 1 and int(sf._gate[1].item()) == 0      # signalled, never timed out
        seam.append(out[:6000].cpu().numpy())
    # the shard against a plain call on the completed buffer: bit for bit
    want = torch.empty_like(sf.out)
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps())
    assert f.process_dev(sf.buf, want) == (C, C)
    torch.cuda.synchronize()
    same = bool(torch.equal(sf.out, want))
    q.put((rank, seam[1], same, sf.buf[:K - 1 + 6000].cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gated_launch_per_pass(oracle):
    """ShardedFir's one-launch step -- the halo exchange beside ONE gated launch over the shard, pcx_fir_process_dev_gated -- on
    shards long enough for the dealt kernel (> 2048 blocks), two processes on this box's one GPU, gloo carrying the halo: poisoned
    halos, two passes, the seam against the oracle and the whole shard bit-identical to a plain call"""
    from pothoscomms_amd import taps as tp
    world, C = 2, 2080 * 3840
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gated_worker, args=(r, world, port, C, q)) for r in range(world)]
    [p.start() for p in procs]
    parts = {}
    for _ in range(world):
        rank, seam, same, xin = q.get(timeout=300)
        parts[rank] = (seam, same, xin)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    h = tp.c1_taps()
    for rank in range(world):
        seam, same, xin = parts[rank]
        assert same, "rank %d" % rank
        ref_blk = oracle.Fir(oracle.F32, True, True); ref_blk.set_taps(h); ref_blk.activate()
        ref, _, p, _ = ref_blk.work(xin, 6000)
        assert p == 6000 and nerr(seam, ref) <= TOL


def _chain_worker(rank, world, port, C, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pothoscomms_amd import taps as tp
    from pothoscomms_amd.stream import ShardedFmChain
    torch.cuda.set_device(0)
    d = torch.device("cuda", 0)
    h = tp.c4_taps()
    K = len(h)
    sc = ShardedFmChain(h, tp.C4_PHASE, C, d)
    # node-wide stream: K-1 history samples, then world*C samples; rank r owns [K-1 + r*C, K-1 + (r+1)*C)
    x = tp.fm_test_signal(K - 1 + world * C)
    xs = torch.from_numpy(np.ascontiguousarray(x).view(np.float32).reshape(-1, 2)).to(d)
    sc.shard.copy_(xs[K - 1 + rank * C:K - 1 + (rank + 1) * C])
    if rank == 0:
        sc.buf[1:K] = xs[:K - 1]                       # the stream's own history; buf[0] is unused on rank 0
    else:
        sc.buf[:K] = float("nan")                      # must come from the left neighbour
    out = sc.step()
    torch.cuda.synchronize()
    q.put((rank, out.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("C", [40000, 3000])
def test_two_ranks_fm_chain_no_seam(oracle, C):
    """configs[4] sharded: halo of K samples (FIR history + the demodulator's predecessor), one dropped output"""
    from pothoscomms_amd import taps as tp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chain_worker, args=(r, world, port, C, q)) for r in range(world)]
    [p.start() for p in procs]
    parts = dict(q.get(timeout=300) for _ in range(world))
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    got = np.concatenate([parts[r] for r in range(world)])
    h = tp.c4_taps()
    K = len(h)
    x = tp.fm_test_signal(K - 1 + world * C)
    xr = oracle.rotate(x, tp.C4_PHASE)
    fir = oracle.Fir(oracle.F32, True, False); fir.set_taps(h); fir.activate()
    y, _, p, _ = fir.work(xr, world * C)
    ref = oracle.FreqDemod(oracle.F32).work(y)
    assert p == world * C and got.shape == ref.shape and not np.isnan(got).any()
    d = (got.astype(np.float64) - ref + np.pi) % (2 * np.pi) - np.pi
    bad = np.flatnonzero(np.abs(d) / np.pi > TOL)
    assert bad.size == 0, (C, bad.size, bad[:8], bad[-3:], got[bad[:4]], ref[bad[:4]])


def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)     # "nccl" is RCCL on ROCm: what bench.py opens (no device_id: bench.py says why)
    from pothoscomms_amd import device, taps as tp
    from pothoscomms_amd.stream import ShardedFir
    sf = ShardedFir(tp.c1_taps(), 1 << 16, dev)
    device.fill_uniform_f32_dev(sf.buf, seed=2, offset=0)
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    out = sf.step()
    torch.cuda.synchronize()
    t = torch.tensor([3.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                 # bench.py's max-over-ranks timing reduction
    dist.barrier(device_ids=[0])
    q.put((float(t.item()), sf.buf.cpu().numpy(), out.cpu().numpy()))
    dist.destroy_process_group()


def test_rccl_group_of_one_runs_the_bench_control_flow(oracle):
    """The 8-GPU run opens an RCCL group (bench.py: init_process_group("nccl"), barrier(device_ids), all_reduce MAX
    around the sharded step).  One GPU cannot host two RCCL ranks, so the seam itself is covered over gloo above;
    this holds the RCCL side of the control flow -- the library loads, the communicator comes up on this image,
    barrier / all_reduce run on the device, and the sharded step works inside an initialised nccl group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    tmax, buf, out = got
    assert tmax == 3.5
    from pothoscomms_amd import taps as tp
    ref = oracle.Fir(oracle.F32, True, True)
    ref.set_taps(tp.c1_taps()); ref.activate()
    want, _, _, _ = ref.work(buf, 1 << 16)
    assert nerr(out, want) <= TOL


def _run_bench(extra_env, *argv):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True, text=True, timeout=900)


def test_bench_spawns_its_own_ranks_and_reports_the_world_it_saw():
    """`python bench.py --gpus 2` with no rank environment must start two ranks itself and say n_gpus: 2
    (the halo goes over gloo here: one GPU on the test box; RCCL on the 8-GPU node)."""
    import json
    r = _run_bench({"PCX_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--shard", "1048576", "--steps", "3", "--warmup", "1", "--no-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size_observed"] == 2
    assert len(out["config"]["rank_devices"]) == 2
    assert out["config"]["halo_backend"].startswith("gloo")
    assert out["value"] > 0
    # a host-driven backend has nothing to pipeline, and the line does not claim it; the seams were checked all the same
    assert "software-pipelined" not in out["config"]["parallelism"] and "taken in turn" in out["config"]["parallelism"]
    assert "match a plain call" in out["config"]["seam_check"]


def test_bench_falls_back_to_two_launches_when_a_rank_reports_a_gate_timeout_in_setup():
    """the first real multi-GPU run will be the driver's: if ANY rank's gated launch times out waiting for its halo during the setup
    passes, every rank switches to the two-launch pass for the timed region and the line says so (here the timeout is pretended, on
    rank 1 of two gloo ranks; the all-reduce of the flag, the switch on BOTH ranks and the annotation are real)"""
    import json
    r = _run_bench({"PCX_BENCH_BACKEND": "gloo", "PCX_BENCH_TEST_GATE_TIMEOUT": "1"}, "--gpus", "2", "--shard", "1048576", "--steps", "3", "--warmup", "1",
                   "--settle", "8", "--no-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and "FALLBACK" in out["config"]["halo_scheme"] and out["value"] > 0
    r = _run_bench({"PCX_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--shard", "1048576", "--steps", "3", "--warmup", "1", "--settle", "8", "--no-cpu")
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "FALLBACK" not in out["config"]["halo_scheme"] and "host-driven" in out["config"]["halo_scheme"]


def test_bench_refuses_more_rccl_ranks_than_gpus():
    """With the RCCL backend every rank needs its own GPU: asking for more than the node has must fail, not fold."""
    import torch
    n = torch.cuda.device_count() + 1
    r = _run_bench({}, "--gpus", str(n), "--shard", "1048576", "--steps", "1", "--warmup", "0", "--no-cpu")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_native_driver_rehearsed_on_one_gpu_and_refused_without_enough_gpus():
    """`bench.py --driver native`: ONE process, pcx_shard_* behind the C ABI.  On this one-GPU box the layout has to be named
    (--native-devices 0,0: two shards on device 0 over peer copies) and the line says it is a rehearsal and counts ONE gpu;
    without the override two shards on one GPU are refused."""
    import json
    import torch
    r = _run_bench({}, "--driver", "native", "--gpus", "2", "--native-devices", "0,0", "--shard", str(2080 * 3840), "--steps", "5", "--warmup", "2",
                   "--settle", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["world_size_observed"] == 1 and line["config"]["shards"] == 2
    assert line["config"]["driver"] == "native" and "REHEARSAL" in line["config"]["halo_transport"]
    assert line["value"] > 0 and line["roofline"]["frac"] > 0.1
    if torch.cuda.device_count() < 2:
        r = _run_bench({}, "--driver", "native", "--gpus", "2", "--steps", "2", "--warmup", "1", "--settle", "1")
        assert r.returncode != 0 and "needs 2 GPUs" in r.stderr
    r = _run_bench({}, "--driver", "native", "--gpus", "2", "--native-devices", "0,0", "--workload", "fmchain", "--shard", str(2080 * 3968),
                   "--steps", "3", "--warmup", "1", "--settle", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["metric"].startswith("Msamples/s fused FM-demod chain") and line["config"]["halo_samples"] == 127


# ---- the RCCL branch of the rank driver, for real, on ONE GPU: a group of one rank whose ring sends the halo to the rank itself ----

def _rccl_self_worker(port, C, chain, q, two_launch=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from pothoscomms_amd import device, taps as tp
    from pothoscomms_amd.stream import HaloRing, ShardedFir, ShardedFmChain

    class SelfRing(HaloRing):
        """a MIDDLE rank of three whose neighbours are both itself: one RCCL group with a send and a receive, like any interior rank"""
        def __init__(self, halo):
            self.halo = halo; self.group = None; self.rank = 1; self.world = 3

        def start(self, buf):
            return dist.batch_isend_irecv([dist.P2POp(dist.isend, buf[buf.shape[0] - self.halo:], 0),
                                           dist.P2POp(dist.irecv, buf[:self.halo], 0)])

    dev = torch.device("cuda", 0)
    res = []
    if chain:
        sc = ShardedFmChain(tp.c4_taps(), tp.C4_PHASE, C, dev, two_launch=two_launch)
        sc.ring = SelfRing(sc.K)
        import numpy as np
        x = torch.from_numpy(np.ascontiguousarray(tp.fm_test_signal(C + sc.K)).view(np.float32).reshape(-1, 2)).to(dev)
        for p in range(2):
            sc.buf.copy_(torch.roll(x, 1000 * p, 0))
            tail = sc.buf[sc.buf.shape[0] - sc.K:].clone()
            sc.buf[:sc.K] = float("nan")               # only the exchange can make the front of the shard right
            out = sc.step()
            torch.cuda.synchronize()
            assert torch.equal(sc.buf[:sc.K], tail)
            res.append((sc.buf.cpu().numpy(), out.cpu().numpy()))
    else:
        sf = ShardedFir(tp.c1_taps(), C, dev, two_launch=two_launch, slots=768 if two_launch else None)      # (also on fewer resident workgroups than all)
        sf.ring = SelfRing(sf.K - 1)
        K = sf.K
        f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps())
        want = torch.empty((C, 2), dtype=torch.float32, device=dev)
        for p in range(2):
            device.fill_uniform_f32_dev(sf.buf, seed=3 + p, offset=0)
            tail = sf.buf[sf.buf.shape[0] - (K - 1):].clone()
            sf.buf[:K - 1] = float("nan")
            out = sf.step()
            torch.cuda.synchronize()
            assert torch.equal(sf.buf[:K - 1], tail)                     # the halo arrived, in place
            assert f.process_dev(sf.buf, want) == (C, C)                 # a plain call on the completed buffer
            torch.cuda.synchronize()
            timed_out = int(sf._gate[1].item()) if getattr(sf, "_gate", None) is not None else 0       # (the two-launch scheme has no gate)
            assert (getattr(sf, "_gate", None) is None) == two_launch
            # one gated launch walks the same blocks as the plain call: the same bits.  Body + head are two calls with their own block
            # boundaries: the same stream within the float bar
            same = bool(torch.equal(out, want)) if not two_launch else float((out - want).abs().max() / want.abs().max()) <= 1e-5
            res.append((same, timed_out, sf.buf[:K - 1 + 8192].cpu().numpy(), out[:8192].cpu().numpy()))
    q.put(res)
    dist.destroy_process_group()


@pytest.mark.parametrize("two_launch", [False, True], ids=["gated", "two-launch"])
@pytest.mark.parametrize("chain", [False, True])
def test_the_rccl_pass_of_a_middle_rank_on_one_gpu(oracle, chain, two_launch):
    """ShardedFir / ShardedFmChain._step_gated over the RCCL backend -- side stream, grouped isend / irecv, the gate signal behind them,
    ONE gated launch -- had only ever run in its host-driven variant (gloo).  One GPU is enough to run it for real: a group of one rank
    whose ring sends the halo to the rank itself.  Poisoned halo slot, two passes with different data: bit-identical to a plain call on
    the completed buffer (FIR), the chain against the oracle chain; no gate time-out.  two-launch: the same pass with the round-2 scheme the
    ranks driver falls back to under PCX_STREAM_TWO_LAUNCH=1 (body, wait for the halo, head) -- the bypass must stay correct."""
    from pothoscomms_amd import taps as tp
    C = 2100 * 3840 if not chain else 2100 * 3968
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_self_worker, args=(_free_port(), C, chain, q, two_launch))
    p.start()
    res = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    if not chain:
        h = tp.c1_taps()
        for same, timed_out, xin, got in res:
            assert same and timed_out == 0
            ref_blk = oracle.Fir(oracle.F32, True, True); ref_blk.set_taps(h); ref_blk.activate()
            ref, _, _, _ = ref_blk.work(xin, 8192)
            assert nerr(got, ref) <= TOL
    else:
        from tests.util import ang_err
        h = tp.c4_taps()
        for xbuf, got in res:
            n = 20000                                      # the front of the shard: what the halo feeds
            xr = oracle.rotate(xbuf[:n + len(h)], tp.C4_PHASE)
            fir = oracle.Fir(oracle.F32, True, False); fir.set_taps(h); fir.activate()
            y, _, _, _ = fir.work(xr, n + 1)
            ref = oracle.FreqDemod(oracle.F32).work(y)[1:n + 1]
            assert not np.isnan(got).any()
            assert ang_err(got[:n], ref) <= TOL


def _rccl_pingpong_worker(port, C, q):
    try:
        _rccl_pingpong_body(port, C, q)
    except BaseException:                                  # (the parent must not wait ten minutes for a worker that died on an assert)
        import traceback
        q.put(("error", traceback.format_exc()))
        raise


def _rccl_pingpong_body(port, C, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from pothoscomms_amd import device, taps as tp
    from pothoscomms_amd.stream import HaloRing, PingPongFir

    class SelfRing(HaloRing):
        def __init__(self, halo):
            self.halo = halo; self.group = None; self.rank = 1; self.world = 3

        def start(self, buf):
            return dist.batch_isend_irecv([dist.P2POp(dist.isend, buf[buf.shape[0] - self.halo:], 0),
                                           dist.P2POp(dist.irecv, buf[:self.halo], 0)])

    dev = torch.device("cuda", 0)
    pp = PingPongFir(tp.c1_taps(), C, dev)
    pp.ring = SelfRing(pp.K - 1)
    K = pp.K
    # the passes on a stream whose hardware queue the exchange does not share (stream.py, HARDWARE QUEUES)
    from pothoscomms_amd.stream import exchange_shares_queue, pick_launch_stream
    torch.cuda.set_stream(pick_launch_stream(pp))
    assert exchange_shares_queue(0) is False
    from pothoscomms_amd.stream import PINGPONG_SLOTS
    pp.set_slots(PINGPONG_SLOTS)                           # room for RCCL's workgroup beside the persistent launch (what a rank of a real world gets)
    assert pp.slots == 896 and all(h.slots == 896 for h in pp.halves)
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps())
    want = torch.empty((C, 2), dtype=torch.float32, device=dev)

    def load(half, seed):
        device.fill_uniform_f32_dev(half.buf, seed=seed, offset=0)
        half.buf[:K - 1] = float("nan")                    # only the exchange can make the front of the shard right

    load(pp.current, 20)
    res = []
    for k in range(6):
        load(pp.upcoming, 21 + k)                          # batch k+1 into the other buffer, then the step: its exchange rides beside batch k's pass
        cur = pp.current
        out = pp.step()
        torch.cuda.synchronize()
        tail = cur.buf[cur.buf.shape[0] - (K - 1):]
        halo_ok = bool(torch.equal(cur.buf[:K - 1], tail))
        assert f.process_dev(cur.buf, want) == (C, C)
        torch.cuda.synchronize()
        res.append((halo_ok, bool(torch.equal(out, want)), int(cur._gate[1].item()), 20 + k, float(out.abs().sum().item())))
    pp.check_gate()
    q.put(res)
    dist.destroy_process_group()


def test_two_input_buffers_the_next_batchs_halo_exchanged_beside_this_batchs_pass(oracle):
    """stream.PingPongFir over RCCL on one GPU (the ring sends the halo to the rank itself): six batches through two buffers, every halo
    poisoned before its exchange; each pass bit-identical to a plain call on the completed buffer, each batch different, no gate time-out."""
    C = 2100 * 3840
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_pingpong_worker, args=(_free_port(), C, q))
    p.start()
    res = q.get(timeout=600)
    p.join(120)
    assert not (len(res) == 2 and res[0] == "error"), res[1]
    assert p.exitcode == 0
    assert len(res) == 6
    for halo_ok, same, timed_out, _, _ in res:
        assert halo_ok and same and timed_out == 0
    assert len({r[4] for r in res}) == 6                   # six different batches went through


@pytest.mark.parametrize("workload", ["fir255", "fmchain"])
@pytest.mark.parametrize("flags", [(), ("--no-pingpong",)], ids=["pipelined", "unpipelined"])
def test_bench_poisons_the_halo_behind_the_timed_region_and_refuses_a_line_whose_seam_is_wrong(flags, workload):
    """bench.py's seam check (its buffers are static, so the timed passes cannot tell a late halo from a timely one): the middle-rank
    rehearsal on one GPU carries config.seam_check, and with the exchange made to deliver nothing for the check pass the run ends
    without a result line."""
    import json
    common = ("--rehearse-rccl-rank", "--workload", workload, "--shard", str(2100 * (3840 if workload == "fir255" else 3968)), "--steps", "5", "--warmup", "2",
              "--settle", "5", "--no-cpu", "--no-cold", "--no-secondary", "--sustain", "0") + tuple(flags)
    r = _run_bench({}, *common)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "match a plain call" in line["config"]["seam_check"]
    assert line["config"]["rccl_stream_shares_the_launch_queue"] is False
    assert ("two input buffers" in line["config"]["parallelism"]) == (not flags)
    form = line["config"].get("halo_exchange_form")
    if flags:                                              # --no-pingpong: one buffer, nothing to choose between
        assert form is None and line["config"]["resident_workgroups_per_launch"] == 1024
    else:                                                  # both forms were timed during setup and the faster one ran (one GPU: the pipelined one, by a margin)
        assert form["chosen"] in ("pipelined", "unpipelined") and form["pipelined_ms"] > 0 and form["unpipelined_ms"] > 0
        assert (form["chosen"] == "pipelined") == (form["pipelined_ms"] <= form["unpipelined_ms"])
        assert line["config"]["resident_workgroups_per_launch"] == (896 if form["chosen"] == "pipelined" else 1024)
    # the exchange delivers nothing in the one-launch form only: the seam check notices, the pass is switched to two launches, timed AGAIN
    # and checked again -- one line, carrying both timings and the reason
    r = _run_bench({"PCX_BENCH_TEST_BREAK_SEAM": "1"}, *common)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "seam check" in r.stderr and "poison" in r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["config"]
    assert "FALLBACK" in c["halo_scheme"] and "poison" in c["retimed"]["first_form_seam_check"]
    assert c["retimed"]["first_form_ms_per_step"] > 0 and c["retimed"]["ms_per_step"] == line["ms_per_step"]
    assert "match a plain call" in c["seam_check"] and any("seam check" in w for w in c["fallback_reason"])
    # ... in either form: no line
    r = _run_bench({"PCX_BENCH_TEST_BREAK_SEAM": "2"}, *common)
    assert r.returncode != 0
    assert "seam check" in r.stderr and "poison" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _rccl_pingpong_chain_body(port, C, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from pothoscomms_amd import device, taps as tp
    from pothoscomms_amd.stream import HaloRing, PingPongFmChain, PINGPONG_SLOTS, pick_launch_stream

    class SelfRing(HaloRing):
        def __init__(self, halo):
            self.halo = halo; self.group = None; self.rank = 1; self.world = 3

        def start(self, buf):
            return dist.batch_isend_irecv([dist.P2POp(dist.isend, buf[buf.shape[0] - self.halo:], 0),
                                           dist.P2POp(dist.irecv, buf[:self.halo], 0)])

    dev = torch.device("cuda", 0)
    pp = PingPongFmChain(tp.c4_taps(), tp.C4_PHASE, C, dev)
    pp.ring = SelfRing(pp.K)
    pp.set_slots(PINGPONG_SLOTS)
    torch.cuda.set_stream(pick_launch_stream(pp))
    K = pp.K
    x = torch.from_numpy(np.ascontiguousarray(tp.fm_test_signal(C + K)).view(np.float32).reshape(-1, 2)).to(dev)
    plain = device.FmChain(); plain.set_phase(tp.C4_PHASE); plain.set_taps(tp.c4_taps(), False)
    want = torch.empty((C + 1,), dtype=torch.float32, device=dev)

    def load(half, k):
        half.buf.copy_(torch.roll(x, 1000 * k, 0) * (1.0 + 0.25 * k))      # (batches differ by shift and amplitude, not by phase jumps)
        half.buf[:K] = float("nan")

    load(pp.current, 0)
    res = []
    for k in range(4):
        load(pp.upcoming, k + 1)
        cur = pp.current
        out = pp.step()
        torch.cuda.synchronize()
        halo_ok = bool(torch.equal(cur.buf[:K], cur.buf[cur.buf.shape[0] - K:]))
        plain.reset()
        assert plain.process_dev(cur.buf, want, C + 1 + K - 1, C + 1) == (C + 1, C + 1)      # one plain call over halo + shard: output -1 first
        torch.cuda.synchronize()
        same = bool(torch.equal(out, want[1:]))
        res.append((halo_ok, same, int(cur._gate[1].item()), cur.buf[:20000 + K].cpu().numpy(), out[:20000].cpu().numpy()))
    pp.check_gate()
    q.put(res)
    dist.destroy_process_group()


def _rccl_pingpong_chain_worker(port, C, q):
    try:
        _rccl_pingpong_chain_body(port, C, q)
    except BaseException:
        import traceback
        q.put(("error", traceback.format_exc()))
        raise


def test_two_input_buffers_fused_chain(oracle):
    """stream.PingPongFmChain over RCCL on one GPU: four batches through two buffers, halos poisoned; every pass bit-identical to ONE plain
    fused call over [halo | shard] (the gated launch walks the same blocks) and, at the shard front -- what the halo feeds -- within the
    angle bar of the oracle's Rotate -> FIR -> FreqDemod."""
    from pothoscomms_amd import taps as tp
    from tests.util import ang_err
    C = 2100 * 3968
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_pingpong_chain_worker, args=(_free_port(), C, q))
    p.start()
    res = q.get(timeout=600)
    p.join(120)
    assert not (len(res) == 2 and res[0] == "error"), res[1]
    assert p.exitcode == 0 and len(res) == 4
    h = tp.c4_taps()
    for halo_ok, same, timed_out, xbuf, got in res:
        assert halo_ok and same and timed_out == 0
        n = 20000
        xr = oracle.rotate(xbuf[:n + len(h)], tp.C4_PHASE)
        fir = oracle.Fir(oracle.F32, True, False); fir.set_taps(h); fir.activate()
        y, _, _, _ = fir.work(xr, n + 1)
        ref = oracle.FreqDemod(oracle.F32).work(y)[1:n + 1]
        assert not np.isnan(got).any()
        assert ang_err(got[:n], ref) <= TOL
