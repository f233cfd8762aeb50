"""CPU suite: the multi-GPU sharding logic, world_size 2 over gloo.

The data path needs no collective except the neighbour halo exchange (stream.HaloRing);
here two CPU processes each own a shard, exchange the K-1 halo with torch.distributed (gloo)
and filter their shard with the ORACLE (test infrastructure standing in for the HIP kernel,
which needs a GPU) -- the concatenated shard outputs must equal the single-stream result,
i.e. the sharding introduces no seam."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, C, K, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from pothoscomms_amd.stream import HaloRing
    # the node-wide stream; every rank can regenerate it (counter-hash), keeps only its part
    stream = o.fill_uniform_f32(2 * (K - 1 + world * C), 2, 0).reshape(-1, 2)
    buf = torch.zeros((K - 1 + C, 2), dtype=torch.float32)
    buf[K - 1:] = torch.from_numpy(stream[K - 1 + rank * C:K - 1 + (rank + 1) * C])
    if rank == 0:
        buf[:K - 1] = torch.from_numpy(stream[:K - 1])      # rank 0 owns the stream's own history
    else:
        buf[:K - 1] = float("nan")                         # must be overwritten by the exchange
    HaloRing(K - 1).exchange(buf)
    rng = np.random.default_rng(0)
    taps = rng.normal(size=K) + 1j * rng.normal(size=K)
    blk = o.Fir(o.F32, True, True)
    blk.set_taps(taps)
    blk.activate()
    y, c, p, _ = blk.work(buf.numpy(), C)
    assert (c, p) == (C, C)
    q.put((rank, y))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,world", [(255, 2), (2, 2), (255, 4), (255, 8)])
def test_overlap_save_sharding_has_no_seam(K, world):
    """world 2 as the contract asks; world 4 also exercises interior ranks, which both send and receive"""
    from oracle import oracle as o
    C = 5000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, K, q)) for r in range(world)]
    [p.start() for p in procs]
    parts = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    got = np.concatenate([parts[r] for r in range(world)])
    stream = o.fill_uniform_f32(2 * (K - 1 + world * C), 2, 0).reshape(-1, 2)
    rng = np.random.default_rng(0)
    taps = rng.normal(size=K) + 1j * rng.normal(size=K)
    blk = o.Fir(o.F32, True, True)
    blk.set_taps(taps)
    blk.activate()
    ref, c, p, _ = blk.work(stream, world * C)
    assert p == world * C
    assert np.array_equal(got, ref)     # same arithmetic order per output -> bit-identical


def test_halo_ring_single_rank_is_a_noop():
    from pothoscomms_amd.stream import HaloRing
    b = torch.arange(20, dtype=torch.float32).reshape(10, 2).clone()
    HaloRing(3).exchange(b)
    assert torch.equal(b, torch.arange(20, dtype=torch.float32).reshape(10, 2))


def _chain_worker(rank, world, port, C, K, q):
    """configs[4] sharded: the halo is K samples (FIR history + the demodulator's predecessor); a rank other
    than the first computes one extra output in front and drops it (stream.ShardedFmChain does the same)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from pothoscomms_amd import taps as tp
    from pothoscomms_amd.stream import HaloRing
    x = tp.fm_test_signal(K - 1 + world * C)
    xs = np.ascontiguousarray(x).view(np.float32).reshape(-1, 2)
    buf = torch.zeros((K + C, 2), dtype=torch.float32)
    buf[K:] = torch.from_numpy(xs[K - 1 + rank * C:K - 1 + (rank + 1) * C])
    if rank == 0:
        buf[1:K] = torch.from_numpy(xs[:K - 1])
    else:
        buf[:K] = float("nan")
    HaloRing(K).exchange(buf)
    taps = tp.lowpass(K, 0.1)
    first = rank == 0
    xin = buf.numpy()[1:] if first else buf.numpy()
    xr = o.rotate(xin, 0.7)
    fir = o.Fir(o.F32, True, False); fir.set_taps(taps); fir.activate()
    n_out = C if first else C + 1
    y, c, p, _ = fir.work(xr, n_out)
    assert p == n_out
    dm = o.FreqDemod(o.F32).work(y)
    q.put((rank, dm if first else dm[1:]))
    dist.barrier()
    dist.destroy_process_group()


def test_fm_chain_sharding_has_no_seam():
    from oracle import oracle as o
    from pothoscomms_amd import taps as tp
    world, C, K = 3, 4000, 31
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chain_worker, args=(r, world, port, C, K, q)) for r in range(world)]
    [p.start() for p in procs]
    parts = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    got = np.concatenate([parts[r] for r in range(world)])
    x = tp.fm_test_signal(K - 1 + world * C)
    xr = o.rotate(x, 0.7)
    fir = o.Fir(o.F32, True, False); fir.set_taps(tp.lowpass(K, 0.1)); fir.activate()
    y, _, p, _ = fir.work(xr, world * C)
    ref = o.FreqDemod(o.F32).work(y)
    assert p == world * C
    assert np.array_equal(got, ref)      # same arithmetic per output: bit-identical across the seams


def test_bench_refuses_a_world_size_that_is_not_what_was_asked():
    """bench.py under a rank environment of 4 with --gpus 2 must refuse before doing anything (no GPU needed)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    assert not r.stdout.strip()


def test_bench_parent_fails_loudly_when_its_ranks_fail():
    """No GPU here: the two ranks bench.py starts itself die on their GPU assertion; the parent must exit
    non-zero and print no result line (a one-GPU number under an N-GPU label is the failure this guards)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_check_gate_raises_once_on_a_timed_out_gate():
    """stream.py check_gate: the word behind the gate word says a gated launch gave up waiting; reported once, then cleared"""
    import types

    import torch

    from pothoscomms_amd import stream
    owner = types.SimpleNamespace(_gate=torch.zeros((64,), dtype=torch.int32), _pass=7, ring=types.SimpleNamespace(rank=3), _side=None)
    stream._check_gate(owner)                       # clean
    owner._gate[1] = stream.GATE_TIMED_OUT
    with pytest.raises(stream.GateTimeout, match="rank 3.*pass 7"):
        stream._check_gate(owner)
    stream._check_gate(owner)                       # cleared by having been reported
    stream._check_gate(types.SimpleNamespace())     # no gate yet: nothing to check


def test_two_launch_switch():
    from pothoscomms_amd import stream
    assert stream._two_launch_forced(True) and not stream._two_launch_forced(False)
    old = os.environ.pop("PCX_STREAM_TWO_LAUNCH", None)
    try:
        assert not stream._two_launch_forced(None)
        os.environ["PCX_STREAM_TWO_LAUNCH"] = "1"
        assert stream._two_launch_forced(None)
    finally:
        os.environ.pop("PCX_STREAM_TWO_LAUNCH", None)
        if old is not None:
            os.environ["PCX_STREAM_TWO_LAUNCH"] = old


def test_ping_pong_posts_the_next_batchs_exchange_in_front_of_this_batchs_pass(monkeypatch):
    """stream.PingPongFir.step: which half's exchange and which half's pass a step queues, with stand-ins for the two buffers (no GPU):
    batch k's pass always follows batch k's exchange by one step, every step posts exactly one exchange except the first (two), the two
    halves share ONE side stream, and with two_launch set (bench.py's fall-back) the step is the plain step of the current half."""
    import types

    from pothoscomms_amd import stream
    log, afters = [], []

    class Half:
        def __init__(self, name):
            self.name, self.two_launch, self.slots = name, False, None
            self.ring = types.SimpleNamespace(world=3, rank=1, group=None)
            self._buf = types.SimpleNamespace(is_cuda=True)
            self._side = None

        def _gate_setup(self):
            if self._side is None:
                self._side = object()

        def post_exchange(self, after=None):
            log.append(("x", self.name))
            afters.append(after)

        def compute(self):
            log.append(("c", self.name))
            return self.name

        def step(self):
            log.append(("s", self.name))
            return self.name

        def check_gate(self):
            log.append(("g", self.name))

    monkeypatch.setattr(stream, "_rccl_world", lambda ring: True)
    pp = object.__new__(stream.PingPongFir)
    pp.halves, pp.k, pp._primed = [Half("A"), Half("B")], 0, False
    pp._mark = lambda half: "mark in front of %s's pass" % half.name
    assert pp.current.name == "A" and pp.upcoming.name == "B"
    assert [pp.step() for _ in range(4)] == ["A", "B", "A", "B"]
    # the host queues the pass first; the exchange of the next batch is ordered behind a mark taken IN FRONT of that pass
    assert log == [("x", "A"), ("c", "A"), ("x", "B"), ("c", "B"), ("x", "A"), ("c", "A"), ("x", "B"), ("c", "B"), ("x", "A")]
    assert afters == [None, "mark in front of A's pass", "mark in front of B's pass", "mark in front of A's pass", "mark in front of B's pass"]
    assert pp.halves[0]._side is pp.halves[1]._side and pp.halves[0]._side is not None
    # every pass ran behind its own half's latest exchange
    for i, (what, name) in enumerate(log):
        if what == "c":
            assert ("x", name) in log[:i] and ("c", name) not in log[max(j for j in range(i) if log[j] == ("x", name)):i]
    del log[:]
    pp.two_launch = True
    assert all(h.two_launch for h in pp.halves)
    assert [pp.step() for _ in range(2)] == ["A", "B"] and log == [("s", "A"), ("s", "B")]
    pp.two_launch = False                                  # back: the pipeline is primed again from the current half
    del log[:]
    pp.step()
    assert log == [("x", "A"), ("c", "A"), ("x", "B")]
    pp.check_gate()
    assert log[-2:] == [("g", "A"), ("g", "B")]


# ---- bench.py --gpus N: a line comes out whatever fails (bench_supervisor.py) -------------------------------------------------------------
# The ranks are real processes under gloo; the device and the workload are the CPU stand-in of tests/bench_standin.py (the line says so).
# What is under test is everything else: launcher, supervisors, watchdog, fresh children in the conservative form, the in-rank re-timing.

def _bench_standin(extra_env, *argv, launcher="parent", gpus=2, timeout=600):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"PCX_BENCH_TEST_STANDIN": os.path.join(ROOT, "tests", "bench_standin.py"), "PCX_BENCH_BACKEND": "gloo",
                "PCX_BENCH_WATCHDOG_S": "15", "PCX_BENCH_RESULT_GRACE_S": "3"})
    env.update(extra_env)
    args = ["--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--settle", "4", "--shard", "8192", "--no-cpu"] + list(argv)
    if launcher == "parent":                  # `python bench.py --gpus N`: bench.py starts torch.distributed.run itself
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:                                     # what the driver runs at N > 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    import json
    return r, [json.loads(ln) for ln in lines]


def _is_flagged_standin_line(out, gpus):
    assert out["n_gpus"] == gpus and out["config"]["world_size_observed"] == gpus and out["value"] > 0
    assert "STAND-IN" in out["data"] and "TEST_STAND_IN" in out["config"]                 # never mistaken for a measurement
    assert "match a plain call" in out["config"]["seam_check"]


@pytest.mark.parametrize("launcher", ["parent", "torchrun"])
def test_supervised_bench_first_attempt_clean(launcher):
    r, lines = _bench_standin({}, launcher=launcher, gpus=2 if launcher == "parent" else 3)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    out = lines[0]
    _is_flagged_standin_line(out, 2 if launcher == "parent" else 3)
    assert out["config"]["attempt"] == 1 and out["config"]["attempt_mode"] == "as asked" and out["config"]["fallback_reason"] is None
    assert "retimed" not in out["config"]
    # both clocks stated; the wall-clock fraction reproduces from value
    roof = out["roofline"]
    assert "HIP events" in roof["clock"] and roof["wall_clock"]["ms_per_step"] == out["ms_per_step"]


def test_supervised_bench_a_hung_rank_costs_an_attempt_not_the_line():
    """rank 1 of attempt 1 stops making progress behind the process group's setup (the others wait for it in a collective): the
    watchdog stops the attempt, FRESH children run the conservative form, and the line says so"""
    r, lines = _bench_standin({"PCX_BENCH_TEST_HANG": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    out = lines[0]
    _is_flagged_standin_line(out, 2)
    assert out["config"]["attempt"] == 2 and out["config"]["attempt_mode"].startswith("conservative")
    assert len(out["config"]["fallback_reason"]) == 1 and "silent for 15 s" in out["config"]["fallback_reason"][0]
    assert "PCX_STREAM_TWO_LAUNCH" in out["config"]["halo_scheme"]
    assert "starting fresh rank processes" in r.stderr


def test_supervised_bench_a_rank_that_dies_in_setup_costs_an_attempt_not_the_line():
    r, lines = _bench_standin({"PCX_BENCH_TEST_DIE": "1"}, launcher="torchrun")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    out = lines[0]
    _is_flagged_standin_line(out, 2)
    assert out["config"]["attempt"] == 2
    why = out["config"]["fallback_reason"][0]
    assert "rank 1: exit code 1" in why and "PCX_BENCH_TEST_DIE" in why          # the root cause is named, first-hand


def test_supervised_bench_a_broken_exchange_in_the_one_launch_form_is_retimed_in_the_two_launch_form():
    """PCX_BENCH_TEST_BREAK_SEAM=1: the exchange delivers nothing while the ranks run the one-launch form.  The seam check behind the
    timed region notices (poisoned halo), every rank switches to two launches per pass, the K steps are timed again and checked again:
    one line, from the FIRST attempt, carrying both timings and the reason"""
    r, lines = _bench_standin({"PCX_BENCH_TEST_BREAK_SEAM": "1"}, gpus=3)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    out = lines[0]
    _is_flagged_standin_line(out, 3)
    c = out["config"]
    assert c["attempt"] == 1 and "FALLBACK" in c["halo_scheme"]
    assert c["retimed"]["first_form_ms_per_step"] > 0 and c["retimed"]["ms_per_step"] == out["ms_per_step"]
    assert "rank 1: the halo slot still holds the poison" in c["retimed"]["first_form_seam_check"]
    assert "rank 2" in c["retimed"]["first_form_seam_check"] and "rank 0" not in c["retimed"]["first_form_seam_check"]
    assert any("seam check" in w for w in c["fallback_reason"])


def test_supervised_bench_no_line_when_every_form_fails():
    """PCX_BENCH_TEST_BREAK_SEAM=2: the exchange delivers nothing in any form -- every attempt ends in a failed seam check, the run
    exits non-zero and prints NO line (a number with wrong seams is not a measurement)"""
    # (up to two goes: on a loaded box a rendezvous of one of the six child groups this run starts has been seen to fail for reasons of
    # its own -- once in a dozen full suites --, which ends the run just as non-zero and line-less but by another road)
    for go in range(2):
        r, lines = _bench_standin({"PCX_BENCH_TEST_BREAK_SEAM": "2"})
        assert r.returncode != 0 and not lines
        if "attempt 2" in r.stderr and "every attempt failed" in r.stderr:
            break
    assert "attempt 2" in r.stderr and "every attempt failed" in r.stderr, r.stderr[-3000:]


def test_supervised_bench_a_teardown_that_hangs_behind_the_line_does_not_cost_it():
    r, lines = _bench_standin({"PCX_BENCH_TEST_HANG": "1:1:teardown"}, launcher="torchrun")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["config"]["attempt"] == 1
    _is_flagged_standin_line(lines[0], 2)


def test_bench_and_its_supervisor_never_replace_the_process():
    """os.exec* from a process that has touched the GPU takes the box down on this pool: children are started with subprocess only"""
    import re
    for f in ("bench.py", "bench_supervisor.py"):
        src = open(os.path.join(ROOT, f)).read()
        code = "\n".join(ln.split("#")[0] for ln in src.splitlines() if not ln.lstrip().startswith(('"', "#")))
        assert not re.search(r"\bos\.exec[lv]p?e?\s*\(|\bos\.spawn|\bexecv\s*\(", code), f


def test_supervised_bench_ranks_that_disagree_about_the_form_of_the_pass_abort_the_attempt():
    """part of what selects the collective branches comes from each rank's OWN environment.  Rank 1 alone sees
    PCX_STREAM_TWO_LAUNCH=1 in attempt 1: the ranks compare the form of the pass before their first step, refuse the mismatch (instead of
    hanging in a collective only some of them enter), and the conservative attempt -- where every rank has the same switches -- yields the line"""
    r, lines = _bench_standin({"PCX_BENCH_TEST_RANK_ENV": "1:PCX_STREAM_TWO_LAUNCH=1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    out = lines[0]
    _is_flagged_standin_line(out, 2)
    assert out["config"]["attempt"] == 2
    assert "disagree about the form of the pass" in out["config"]["fallback_reason"][0]


def test_supervised_bench_a_gate_timeout_in_setup_switches_every_rank_to_two_launches():
    """a gate timeout reported by ONE rank during the setup passes (pretended here: PCX_BENCH_TEST_GATE_TIMEOUT) is all-reduced and every rank
    switches to the two-launch pass for the timed region, inside the first attempt; the line says so"""
    r, lines = _bench_standin({"PCX_BENCH_TEST_GATE_TIMEOUT": "1"}, gpus=3, launcher="torchrun")
    assert r.returncode == 0, r.stderr[-3000:]
    out = lines[0]
    _is_flagged_standin_line(out, 3)
    c = out["config"]
    assert c["attempt"] == 1 and "FALLBACK" in c["halo_scheme"] and "timed out" in c["halo_scheme"]
    assert any("a gated launch timed out" in w for w in c["fallback_reason"]) and "retimed" not in c


def test_supervised_bench_a_gate_timeout_behind_the_timed_region_is_retimed_not_deadlocked():
    """ONE rank > 0 finds a gate timeout behind the timed region.  It still takes part in the seam check (a collective step of
    the ring), every rank learns of the finding, all switch to the two-launch form, the K steps are timed again and the line says so --
    inside the first attempt, without the watchdog"""
    r, lines = _bench_standin({"PCX_BENCH_TEST_GATE_TIMEOUT": "1:timed"}, gpus=3, launcher="torchrun")
    assert r.returncode == 0, r.stderr[-3000:]
    out = lines[0]
    _is_flagged_standin_line(out, 3)
    c = out["config"]
    assert c["attempt"] == 1 and "FALLBACK" in c["halo_scheme"]
    assert c["retimed"] and "pretended" in c["retimed"]["first_form_seam_check"] and "rank 1" in c["retimed"]["first_form_seam_check"]
    assert "watchdog" not in r.stderr
