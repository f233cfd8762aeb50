"""Test infrastructure: a CPU stand-in for the device and the workload of bench.py's multi-rank path.

bench.py loads this file only when PCX_BENCH_TEST_STANDIN names it (tests/test_stream_cpu.py does).  It lets the ranks' CONTROL FLOW --
supervisors and children, process group, the agreement on the form of the pass, gate / seam fall-backs, the re-timing, the line --
run under gloo on a box without a GPU.  Nothing here is a measurement and the line says so (config.TEST_STAND_IN, data).

install(g): g is bench.py's module namespace.  torch.cuda's handful of entry points bench.py uses are replaced by host stand-ins
and build_workload by one whose pass is a numpy convolution over a small shard with the real HaloRing (gloo) in front of it.
"""
import os
import time
import types

import numpy as np
import torch


class _Event:
    def __init__(self, enable_timing=False):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3

    def synchronize(self):
        pass

    def query(self):
        return True


class StandInFir:
    """What bench.py needs of stream.ShardedFir, on host tensors: [halo | C samples], two launches per pass (body, halo, head)."""

    def __init__(self, h, C):
        from pothoscomms_amd import stream
        self.h = np.asarray(h, dtype=np.complex64)
        self.K, self.C = len(h), int(C)
        self.ring = stream.HaloRing(self.K - 1)
        self._buf = torch.zeros((self.K - 1 + self.C, 2), dtype=torch.float32)
        self.out = torch.zeros((self.C, 2), dtype=torch.float32)
        self.two_launch = stream._two_launch_forced(None)
        self.head = min(self.C, 1024)
        self.slots = None

    buf = property(lambda self: self._buf)

    def _run(self, first, n):
        x = self._buf[first:first + n + self.K - 1].numpy().view(np.complex64).ravel()
        y = np.convolve(x, self.h, "valid").astype(np.complex64)
        self.out[first:first + n] = torch.from_numpy(y.view(np.float32).reshape(-1, 2))

    def head_reference(self, n):
        self._run(0, self.head)
        return self.out[:n]

    def check_gate(self):
        pass

    def set_slots(self, slots):
        self.slots = slots

    def step(self):
        # PCX_BENCH_TEST_BREAK_SEAM (bench.py): "1" the exchange delivers nothing in the one-launch form, "2" in either form
        brk = os.environ.get("PCX_BENCH_TEST_BREAK_SEAM", "")
        broken = brk == "2" or (brk == "1" and not self.two_launch)
        reqs = [] if broken else self.ring.start(self._buf)
        if self.C > self.head:
            self._run(self.head, self.C - self.head)
        self.ring.finish(reqs)
        self._run(0, self.head)
        return self.out


def build_workload(wl, C, dev, rank, world, args):
    import bench
    W = bench.Workload()
    W.name = wl
    rng = np.random.default_rng(7)
    h = (rng.normal(size=31) + 1j * rng.normal(size=31)) / 31
    sf = StandInFir(h, C)
    g = torch.Generator().manual_seed(100 + rank)
    sf._buf[:] = torch.rand(sf._buf.shape, generator=g) * 2 - 1
    W.owner = sf
    W.inputs = (sf._buf,)
    W.units = C
    W.roof_bytes, W.read_bytes = 16.0 * C, 8.0 * C
    W.kernel_name = "TEST-STAND-IN"
    W.step = sf.step
    W.metric = "Msamples/s complex_float32 255-tap FIR"
    W.desc = {"workload": "TEST STAND-IN: %d-sample shard per rank, 31 taps, numpy on the host" % C, "taps": 31, "shard_samples": C,
              "halo_samples": sf.K - 1, "parallelism": "overlap-save shards x%d" % world, "setup_passes": args.settle}
    return W


def install(g):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: world
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.Event = _Event
    torch.cuda.get_device_properties = lambda d: types.SimpleNamespace(name="cpu stand-in")
    torch.cuda.empty_cache = lambda: None
    g["build_workload"] = build_workload
