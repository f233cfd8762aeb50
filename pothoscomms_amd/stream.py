"""Overlap-save sharding of ONE sample stream across the GPUs of a node.

The reference has no multi-device story (a Pothos block runs on one scheduler thread); the
FIR loop (filter/FIRFilter.cpp:286-302) makes output n depend on inputs n .. n+K-1 only, so a
stream of G*C samples splits into G contiguous shards of C samples, shard g needing the LAST
K-1 samples of shard g-1 as its front halo -- exactly the history the reference keeps in its
circular input buffer between work() calls (FIRFilter.cpp:305-307).  Rank 0 keeps its own
K-1 history.  One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm), one grouped send/recv of (K-1)*elem bytes per shard boundary per pass (2,032 B for 255
taps) and no other data-path collective.

Buffer layout on every rank: one contiguous tensor [halo | C samples] -- the halo sits directly
in front of the shard so the kernel sees the same "history at the front" buffer as a single-GPU
call (pcx_fir_process_dev), and the received bytes land in place (no staging copy).  The shard,
not the halo, is placed on a 128-byte line (see ShardedFir.__init__).

Latency hiding: a 2 KB message is pure latency (tens of microseconds against a ~0.2 ms pass).
Only the first block of a shard reads the halo, so a pass is ONE kernel launch over the whole
shard that walks its blocks back to front and holds the first one behind a gate word
(pcx_fir_process_dev_gated, include/pcx.h): the exchange is posted on a side stream, a one-thread
kernel behind it sets the gate, and everything else of the shard is filtered meanwhile.  (Round 2
ran a body launch, waited, and ran a head launch: the second launch's start-up, ragged end and
kernel boundary cost 5.6 % of a pass.)  Configurations without a gated kernel, and host-driven
backends (the gloo rehearsal), keep the two launches.

RCCL's send/recv kernel finds no slot beside a launch that fills the device (its waves need 136 VGPRs,
the launch's workgroups leave 128 per SIMD) and runs when the launch's first workgroups exit, so a
pass that waits for its OWN exchange ends late (+9-11 %).  Streaming use is therefore double-buffered
(PingPongFir / PingPongFmChain): the halo of batch k+1 is exchanged while batch k is filtered.  Two
things that needs are in this file too: room for RCCL's workgroup (PINGPONG_SLOTS) and a hardware
queue of its own for the exchange (HARDWARE QUEUES, pick_launch_stream).
"""
import os

import torch
import torch.distributed as dist

GATE_TIMED_OUT = 0xDEAD      # pcx_sched.hpp kGateTimedOut: what a gated launch leaves in gate[1] when its halo never came


class GateTimeout(RuntimeError):
    """A pass's gated launch gave up waiting for its halo (two seconds) and ran its first block on a stale halo slot."""


def _check_gate(owner):
    """Wait for the owner's pass, then read the word behind its gate word; raise (once) if a launch timed out."""
    gate = getattr(owner, "_gate", None)
    if gate is None:
        return
    buf = getattr(owner, "_buf", None)
    if gate.is_cuda or (buf is not None and buf.is_cuda):      # (a host-driven gate lives in page-locked HOST memory: the pass still runs on the device)
        dev = gate.device if gate.is_cuda else buf.device
        # the stream the last gated launch went out on (compute() notes it): a caller who launches its passes on another stream than
        # the one current HERE (pick_launch_stream's, inside a `with torch.cuda.stream(...)`) must still have the newest pass waited for
        launched_on = getattr(owner, "_launch_stream", None)
        if launched_on is not None:
            launched_on.synchronize()
        torch.cuda.current_stream(dev).synchronize()
        if getattr(owner, "_side", None) is not None:
            owner._side.synchronize()
    if int(gate[1].item()) == GATE_TIMED_OUT:
        gate[1] = 0
        raise GateTimeout("rank %d: the halo did not arrive within two seconds of pass %d (or an earlier one since the last check); "
                          "the outputs at the front of the shard were computed on a stale halo" % (owner.ring.rank, owner._pass))


# (Round 4 first gave a rank of an RCCL world 768 of the 1024 resident slots, on a probe whose input was all ZERO -- without the power cap's
# grip the pass read 206 us at 1024 slots and 194 at 768.  On DATA the slots do not matter: 212-215 us at 1024, 210-211 at 896, 214 at 768,
# 224 at 640 against 193 for the single-GPU launch (bench.py --rehearse-rccl-rank --rehearse-slots N, profiles/r04/rccl_pass_slots.txt).
# RCCL's protocol kernel costs 9-11 % of a pass whatever room it is given; the default stays 1024.)
RCCL_SLOTS = None
# The software-pipelined pass (PingPongFir) is another matter.  Its launch never waits for RCCL, but RCCL's workgroup (136 VGPRs per wave,
# 20 KB of LDS) has no room beside the launch's 128-VGPR, 36 KB workgroups -- not beside four of them on a CU and not beside three: it is
# placed when workgroups of the launch EXIT.  With three on a CU one exit is enough, with four it takes two on the same CU: at 896 slots
# (half the CUs carry three) the exchange of batch k+1 gets in at the start of pass k's tail and the pass takes 199-200 us against 195-198
# for the plain launch and 206-209 unpipelined; at 1024 it tends to lose the race against pass k+1's own workgroups and then waits for
# THAT pass's tail: 234-240 us (tools/rccl_cost_probe.py, profiles/r04/rccl_cost_probe.txt).
PINGPONG_SLOTS = 896


def _rccl_world(ring):
    return ring.world > 1 and dist.is_initialized() and dist.get_backend(ring.group) == "nccl"


# HARDWARE QUEUES.  HIP maps a process's streams onto four hardware queues, by creation order, and two streams on one queue run in order
# whatever the program says.  When RCCL's own stream (drawn from torch's pool when the communicator is created) or the side stream lands on
# the queue of the stream the passes are launched on -- it did in one process out of a few, depending on how many streams the handles of
# the process had created before -- the exchange runs BEHIND the pass it was meant to run beside: a pass of 239 us instead of 196
# (profiles/r04/pingpong_queues.txt).  Stream priorities do not help: a high-priority side stream or RCCL stream beside a normal-priority
# launch stream made the pass 250-530 us in one process and left it alone in another (profiles/r04/prio_matrix.txt).  So the drivers LOOK:
# _exchange_shares_queue() below, once behind the first exchange (a warning), and pick_launch_stream() for a caller who can choose the
# stream the passes run on (bench.py does).
_QUEUES_CHECKED = {}


def _exchange_shares_queue(owner, tries=2):
    """Does the exchange (side stream, RCCL's stream) share the hardware queue of the CURRENT stream?  A one-thread sleeper (~2 ms) on the
    current stream, an exchange beside it: if the exchange is done while the sleeper still runs, it does not.  COLLECTIVE: every rank
    runs exactly `tries` exchanges (a late neighbour looks like a shared queue, so one free try decides; the exchange rewrites the halo
    with the bytes it already holds)."""
    dev = owner._buf.device
    shared = True
    for _ in range(tries):
        end = torch.cuda.Event()
        torch.cuda._sleep(4_000_000)
        end.record()
        with torch.cuda.stream(owner._side):
            owner.ring.finish(owner.ring.start(owner._buf))
            ev = torch.cuda.Event()
            ev.record()
        ev.synchronize()
        shared = shared and end.query()
        torch.cuda.current_stream(dev).synchronize()
    return shared


def _halves(owner):
    return getattr(owner, "halves", [owner])


def pick_launch_stream(owner, candidates=4):
    """COLLECTIVE (every rank of the ring calls it, with the same arguments, before its first step): the stream to launch this driver's
    passes on -- the current one if the exchange can run beside it, else the first of `candidates` fresh streams it can run beside.
    Make the result current (torch.cuda.set_stream / `with torch.cuda.stream(...)`) for every step().  Every rank runs the same number
    of exchanges whatever it finds."""
    hs = _halves(owner)
    if not (_rccl_world(hs[0].ring) and hs[0]._buf.is_cuda and hasattr(torch.cuda, "_sleep")):
        return torch.cuda.current_stream()
    dev = hs[0]._buf.device
    for h in hs:
        h._gate_setup()
        h._side = hs[0]._side
    h = hs[0]
    with torch.cuda.stream(h._side):                       # the communicator and its connections come up in the first exchange
        h.ring.finish(h.ring.start(h._buf))
    h._side.synchronize()
    for x in hs:
        x._exchanged_once = True
    torch.cuda.current_stream(dev).synchronize()
    first = torch.cuda.current_stream(dev)
    chosen = None
    for cand in [first] + [torch.cuda.Stream(device=dev) for _ in range(candidates)]:
        with torch.cuda.stream(cand):
            shared = _exchange_shares_queue(h)
        if chosen is None and not shared:
            chosen = cand
    _QUEUES_CHECKED[dev.index] = chosen is None
    if chosen is None:
        import warnings
        warnings.warn("pothoscomms_amd.stream: no stream found whose hardware queue the exchange does not share; every exchange will run behind the pass it should run beside")
        chosen = first
    return chosen


def _first_exchange(owner):
    """The FIRST exchange of a driver is waited for on the host before the gated launch that depends on it is queued: RCCL sets its
    point-to-point connections up lazily, inside the first send / receive, and that can take longer than the two seconds a gated launch
    waits for its halo -- the first pass of a run must not be the one that times out.  Every rank passes here in its first step.
    Once per process and device it is also checked that the exchange can run BESIDE a launch on the current stream (HARDWARE QUEUES above)."""
    if not getattr(owner, "_exchanged_once", False):
        owner._side.synchronize()
        owner._exchanged_once = True
        key = owner._buf.device.index
        if hasattr(torch.cuda, "_sleep") and _rccl_world(owner.ring):
            # The probe is COLLECTIVE (two ring exchanges): whether it runs must not depend on what THIS process happens to have looked at
            # before (another driver on the device, another subgroup) or the ranks' send / receive counts stop pairing.  The ring agrees:
            # if any rank of it has not looked yet, every rank probes.
            need = torch.tensor([0 if key in _QUEUES_CHECKED else 1], dtype=torch.int32, device=owner._buf.device)
            dist.all_reduce(need, op=dist.ReduceOp.MAX, group=owner.ring.group)
            if not int(need.item()):
                return
            torch.cuda.current_stream(owner._buf.device).synchronize()
            _QUEUES_CHECKED[key] = _exchange_shares_queue(owner)
            if _QUEUES_CHECKED[key]:
                import warnings
                warnings.warn("pothoscomms_amd.stream: the halo exchange (side stream, RCCL's stream) shares the hardware queue of the stream the passes "
                              "are launched on -- every exchange will run behind the pass it should run beside.  Launch the passes on the stream "
                              "pothoscomms_amd.stream.pick_launch_stream(driver) returns.")


def exchange_shares_queue(device_index=0):
    """What the first step (or pick_launch_stream) found (None before): True if the exchange shares the launch stream's hardware queue."""
    return _QUEUES_CHECKED.get(device_index)


def _two_launch_forced(arg):
    """The RCCL path takes ONE gated launch per pass by default; two_launch=True (or PCX_STREAM_TWO_LAUNCH=1) forces the round-2
    scheme -- body launch, wait for the halo, head launch -- so that the two can be compared, and the gated one bypassed, on a node
    where it misbehaves (it rests on a peer's halo bytes being visible behind a system-scope acquire and on RCCL's receive kernel
    finding room beside a persistent launch: rehearsed on one GPU only so far)."""
    if arg is not None:
        return bool(arg)
    return os.environ.get("PCX_STREAM_TWO_LAUNCH", "0") not in ("", "0")


class HaloRing:
    """Neighbour exchange of the tap-length halo (rank r -> r+1), in place.

    Works on any backend/device (the CPU tests run it over gloo): it only moves the last
    `halo` elements of each rank's buffer into the first `halo` elements of the next rank's.
    """

    def __init__(self, halo, group=None):
        self.halo = int(halo)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def start(self, buf):
        """Post the exchange; returns the requests to pass to finish()."""
        if self.world == 1 or self.halo == 0:
            return []
        if buf.is_cuda and dist.get_backend(self.group) != "nccl":
            # RCCL orders its send behind the current stream's earlier work; a host-driven backend (the one-GPU gloo rehearsal)
            # reads the tail of buf whenever it gets to it, so whatever filled the shard has to be complete first
            torch.cuda.current_stream(buf.device).synchronize()
        ops = []
        if self.rank + 1 < self.world:
            ops.append(dist.P2POp(dist.isend, buf[buf.shape[0] - self.halo:], self.rank + 1, self.group))
        if self.rank > 0:
            ops.append(dist.P2POp(dist.irecv, buf[:self.halo], self.rank - 1, self.group))
        return dist.batch_isend_irecv(ops)

    @staticmethod
    def finish(reqs):
        """Make the current stream (or the host, on CPU backends) wait for the halo."""
        for w in reqs:
            w.wait()

    def exchange(self, buf):
        """buf: [halo + C, ...] contiguous.  Returns after the halo is usable on the current stream."""
        self.finish(self.start(buf))


class ShardedFir:
    """A 255-tap-style complex_float32 FIR over one shard of a node-wide stream (M = L = 1)."""

    # outputs computed after the halo has arrived: one overlap-save block's worth is the
    # minimum (outputs 0 .. 4096-K); rounding up to 4096 keeps the split independent of K
    HEAD = 4096
    gate_host_driven = False     # tests set it: the gated launch with a host-driven exchange (gloo)

    def __init__(self, taps, shard_len, device, taps_type="COMPLEX", algo=None, group=None, slots=None, two_launch=None):
        from . import device as dv   # the HIP path; raises if libpcx_hip.so is missing
        self.two_launch = _two_launch_forced(two_launch)
        self.fir = dv.FirFilter("complex_float32", taps_type)
        self.fir.set_taps(taps)
        if algo is not None:
            self.fir.set_algo(algo)
        self.K = self.fir.K
        self.C = int(shard_len)
        self.ring = HaloRing(self.K - 1, group)
        if slots is None and RCCL_SLOTS and _rccl_world(self.ring):
            slots = RCCL_SLOTS       # (off: see RCCL_SLOTS above)
        if slots is not None:        # several ranks on ONE device (a rehearsal): each takes its share of the resident slots
            self.fir.set_slots(slots)
        self.slots = slots
        # layout in HBM: [lead | halo (K-1) | shard (C)] with the SHARD on a 128-byte line (the halo is
        # right-aligned against it).  The overlap-save kernel rounds its block overlap up to 16 samples,
        # so with this placement every 2 KiB row it loads AND every row it stores starts on a line
        # (measured on MI355X: 0.2245 -> 0.2187 ms per 64 Mi samples against a line-aligned halo).
        lead = (-(self.K - 1)) % 16
        self._alloc = torch.zeros((lead + self.K - 1 + self.C, 2), dtype=torch.float32, device=device)
        self._buf = self._alloc[lead:]
        self.out = torch.empty((self.C, 2), dtype=torch.float32, device=device)
        # the two-launch fallback's split: the body (outputs head .. C-1) must not read the halo slot buf[0 : K-1]
        self.head = min(self.C, max(self.HEAD, -(-(self.K - 1) // self.HEAD) * self.HEAD))

    # WRITING THE INPUT.  `buf` ([halo | C samples]) and `shard` (the C samples) are handed out as views, and taking either one fences
    # the input (fence_input): whoever asks is about to write.  Do NOT keep a view across step(): a view taken before a pass and
    # written after it is not ordered behind that pass's send of the shard's tail, which runs on a side stream (the C driver orders
    # every later writer itself, pcx_shard_step part 4; this one leaves the 13 us cross-stream wait out of the pass and puts it
    # here, where a writer pays it).  Take the view again for every refill.
    @property
    def buf(self):
        """[halo (K-1) | C samples], a view; fences the input like `shard`."""
        self.fence_input()
        return self._buf

    @property
    def shard(self):
        """The C samples this rank owns (a view behind the halo).  Taking the view fences the input (fence_input): whoever asks for it is
        about to write samples."""
        self.fence_input()
        return self._buf[self.K - 1:]

    def fence_input(self):
        """Make the current stream wait until the last pass's exchange has READ the shard's tail (the send to the right neighbour runs on
        a side stream).  Only a WRITER of the shard needs that -- the next pass only reads it -- so step() does not pay for the
        cross-stream wait (13 us of a 190 us pass, tools/host_step_probe.py); call this, or take `shard` / `buf` again, before anything
        queued on the current stream overwrites samples."""
        if getattr(self, "_side", None) is not None and self._buf.is_cuda:
            torch.cuda.current_stream(self._buf.device).wait_stream(self._side)

    def set_slots(self, slots):
        """resident workgroups the shard's launches take (pcx_fir_set_slots)"""
        self.fir.set_slots(slots)
        self.slots = slots

    def check_gate(self):
        """Synchronise, then raise GateTimeout if a pass since the last check ran its first block without its halo (the gated launch's
        wait is bounded, pcx.h): call it wherever results leave the device.  bench.py and the tests do, after the timed region."""
        _check_gate(self)

    def _run(self, first_out, n_out):
        # outputs [first_out, first_out + n_out) read buf[first_out : first_out + n_out + K - 1]
        c, p = self.fir.process_dev(self._buf[first_out:], self.out[first_out:], n_out + self.K - 1, n_out)
        assert c == n_out and p == n_out, (c, p, n_out)

    def head_reference(self, n):
        """The first n outputs (n <= head) computed again by a PLAIN call on the buffer as it stands -- halo in place -- into `out`; what a
        seam check compares a pass's shard front with (bench.py seam_check)."""
        self._run(0, self.head)
        return self.out[:n]

    def _gate_setup(self):
        # the gate word (holds the pass number) and the side stream the exchange is posted on
        if getattr(self, "_gate", None) is None:
            # A host-driven backend opens the gate BEHIND the gated launch.  A signal kernel queued that late can land in the launch's own
            # hardware queue and wait for it (include/pcx.h, the gate's contract), so the word lives in page-locked host memory and the
            # host stores to it.  RCCL queues exchange and signal before the launch: device word, side stream.
            host_driven = not (dist.is_initialized() and dist.get_backend(self.ring.group) == "nccl")
            if host_driven:
                self._gate = torch.zeros((64,), dtype=torch.int32).pin_memory()
            else:
                self._gate = torch.zeros((64,), dtype=torch.int32, device=self._buf.device)
            self._side = torch.cuda.Stream(device=self._buf.device)
            self._pass = 0
            self._sent = None

    def step(self):
        """One pass: the halo from the left neighbour in flight while the shard is filtered -> C outputs."""
        if self.ring.world == 1:
            self._run(0, self.C)
            return self.out
        gated_ok = dist.is_initialized() and ((dist.get_backend(self.ring.group) == "nccl" and not self.two_launch) or self.gate_host_driven)
        if self._buf.is_cuda and gated_ok:
            return self._step_gated()
        # host tensors (the CPU tests of the exchange logic) and host-driven backends on device tensors (the gloo rehearsal of the
        # multi-rank control flow on one GPU: two PROCESSES time-share the device and the exchange is a host round trip, so the
        # one-launch scheme has nothing to hide behind -- measured 0.25-0.27 ms per step against 0.22 for this one): body, wait, head
        reqs = self.ring.start(self._buf)
        if self.C > self.head:
            self._run(self.head, self.C - self.head)      # does not touch the halo
        self.ring.finish(reqs)
        self._run(0, self.head)
        return self.out

    def post_exchange(self, after=None):
        """RCCL: the exchange of THIS buffer's halo and the gate signal behind it, on the side stream, ordered behind everything the current
        stream holds so far (the samples are in place; the previous pass on this buffer has read its halo) -- or behind the event
        `after`, recorded where that was true.  compute() runs the pass."""
        from . import device as dv
        self._gate_setup()
        self._pass += 1
        cur = torch.cuda.current_stream(self._buf.device)
        if after is None:
            self._side.wait_stream(cur)
        else:
            self._side.wait_event(after)
        with torch.cuda.stream(self._side):
            # (posting the exchange from the CURRENT stream instead -- RCCL's stream then waits for it directly, one cross-stream hop
            # in front of the exchange instead of two -- measured slower, 212 against 205 us per pass, tools/host_step_probe.py)
            self.ring.finish(self.ring.start(self._buf))   # RCCL orders the exchange behind the side stream and the side stream behind it
            if self.ring.rank > 0:
                dv.gate_signal(self._gate, self._pass, self._side.cuda_stream)
        _first_exchange(self)

    def compute(self):
        """The pass whose exchange post_exchange() queued: ONE launch over the shard, its first block behind the gate."""
        cur = torch.cuda.current_stream(self._buf.device)
        self._launch_stream = cur                          # (check_gate waits for THIS stream, whatever is current where it is called)
        if self.ring.rank == 0:
            self._run(0, self.C)
            return self.out
        c, p, gated = self.fir.process_dev_gated(self._buf, self.out, self._gate, self._pass, self.C + self.K - 1, self.C)
        if gated:
            assert c == self.C and p == self.C, (c, p)
        else:                                             # no gated kernel for this configuration: the body, the halo, the head
            if self.C > self.head:
                self._run(self.head, self.C - self.head)
            cur.wait_stream(self._side)
            self._run(0, self.head)
        return self.out

    def _step_gated(self):
        """ONE launch over the shard; the exchange beside it, a one-thread kernel behind it opens the gate.
        RCCL: the exchange and the signal are queued on a side stream (post_exchange), then the launch (compute).  A host-driven backend
        (the gloo rehearsal on one GPU): the exchange is started, the launch queued, and the host waits for the halo before it opens the
        gate with a store to a page-locked word."""
        from . import device as dv
        if dist.get_backend(self.ring.group) == "nccl":
            self.post_exchange()
            return self.compute()
        self._gate_setup()
        self._pass += 1
        self._launch_stream = torch.cuda.current_stream(self._buf.device)
        reqs = self.ring.start(self._buf)                  # (drains the current stream first: HaloRing.start)
        gated = True
        c, p, gated = self.fir.process_dev_gated(self._buf, self.out, self._gate, self._pass, self.C + self.K - 1, self.C) if self.ring.rank > 0 else (self.C, self.C, True)
        if self.ring.rank == 0:
            self._run(0, self.C)
        elif gated:
            assert c == self.C and p == self.C, (c, p)
        elif self.C > self.head:
            self._run(self.head, self.C - self.head)      # no gated kernel for this configuration: the body now, the head below
        self.ring.finish(reqs)                            # the host waits for the halo ...
        if self.ring.rank > 0:
            dv.gate_signal_host(self._gate, self._pass)   # ... and opens the gate (a store to the host word)
        if not gated:
            torch.cuda.current_stream(self._buf.device).wait_stream(self._side)
            self._run(0, self.head)
        # (a later WRITER of the shard has to wait for the send that is still reading its tail: fence_input)
        return self.out


class PingPongFir:
    """Two input buffers, software-pipelined: while batch k is filtered, the halo of batch k+1 -- already in place in the OTHER buffer -- is
    exchanged, so that RCCL's protocol kernel (it gets a slot only when a workgroup of the launch beside it exits, ~150 us after it was queued:
    it made a pass 9-11 % longer when the pass had to wait for it, profiles/r04/rccl_pass_slots.txt) has a whole pass to finish in and the gate of batch k+1 is open long
    before its first block -- the last one computed -- asks.  The streaming order of a rank:
        fill(`upcoming.shard`)         batch k+1; behind the pass that last read that buffer (batch k-1): taking the view fences that
        step()                         runs batch k's pass (whose exchange the previous step posted) and posts batch k+1's exchange
    One exchange and one pass per step, as before; only their pairing moved.  Without an RCCL world (one rank, gloo) step() is the plain
    ShardedFir.step() of the current buffer.

    PRIMING AND THE END OF A STREAM (the contract of the pipelined form):
      * the FIRST step() posts the exchange of `current` (batch 0) itself and, like every step, the exchange of `upcoming` (batch 1): BOTH
        buffers must hold their batch before the first step, not only `upcoming`;
      * every step posts an exchange for the batch AFTER the one it filters.  The exchange is a collective step of the ring, so all
        ranks must call step() the same number of times; behind the last step one exchange is outstanding whose batch never comes --
        it rewrites the halo slot of `upcoming` with the neighbour's current tail, harmlessly, and is complete once the side stream
        has drained (check_gate() waits for it).  A rank that stops one step early leaves its right neighbour's last exchange
        unanswered: RCCL blocks there."""

    def __init__(self, taps, shard_len, device, taps_type="COMPLEX", algo=None, group=None, slots=None, two_launch=None):
        self.halves = [ShardedFir(taps, shard_len, device, taps_type, algo, group, slots, two_launch) for _ in range(2)]
        self._default_slots(slots)

    ring = property(lambda self: self.halves[0].ring)

    @ring.setter
    def ring(self, r):
        for h in self.halves:
            h.ring = r

    slots = property(lambda self: self.halves[0].slots)

    def set_slots(self, slots):
        for h in self.halves:
            h.set_slots(slots)

    def _default_slots(self, slots):
        self.K, self.C = self.halves[0].K, self.halves[0].C
        self.k = 0
        self._primed = False
        if slots is None and _rccl_world(self.halves[0].ring):
            self.set_slots(PINGPONG_SLOTS)

    two_launch = property(lambda self: self.halves[0].two_launch)

    @two_launch.setter
    def two_launch(self, v):
        for h in self.halves:
            h.two_launch = bool(v)

    @property
    def current(self):
        """the ShardedFir whose batch the next step() filters"""
        return self.halves[self.k & 1]

    @property
    def upcoming(self):
        """the ShardedFir the batch AFTER that goes into (fill its `shard` / `buf` before the next step())"""
        return self.halves[(self.k + 1) & 1]

    def check_gate(self):
        for h in self.halves:
            h.check_gate()

    pipeline = True      # False: step() is the plain step of the current buffer (every pass behind its own exchange) -- for A/B, and for a
                         # caller that measured the pipelined pass slower on its hardware (bench.py does measure, at N > 1)

    def _pipelined(self):
        h = self.halves[0]
        return self.pipeline and _rccl_world(h.ring) and not h.two_launch and h._buf.is_cuda

    pipelined = property(lambda self: self._pipelined(), doc="does step() pair batch k's pass with batch k+1's exchange (an RCCL world, one-launch passes)?")

    def step(self):
        cur, nxt = self.current, self.upcoming
        self.k += 1
        if not self._pipelined():
            self._primed = False
            return cur.step()
        if not self._primed:                               # the very first batch: its exchange has not been posted by a previous step
            # ONE side stream for both buffers: the exchanges are serial anyway, and every further stream is a further chance that HIP maps
            # two of them onto one hardware queue (4 per process), where a wait for RCCL then holds up whatever else that queue carries --
            # with a side stream per buffer the pass came out at 195-202 us in some processes and 213-216 in others
            # (profiles/r04/rccl_cost_probe.txt)
            for h in self.halves:
                h._gate_setup()
            nxt._side = cur._side
            cur.post_exchange()
            self._primed = True
        # batch k+1's exchange waits for what the current stream holds NOW (its samples are in place, the pass that last read that buffer is
        # queued), but the HOST queues batch k's pass first: RCCL's enqueue is ~90 us of host time, and behind a synchronisation point
        # (the first step of a timed region) the device would sit idle for it
        ev = self._mark(cur)
        out = cur.compute()
        nxt.post_exchange(after=ev)
        return out

    def _mark(self, half):
        """an event on the current stream, here"""
        if getattr(self, "_events", None) is None:
            self._events = [torch.cuda.Event(), torch.cuda.Event()]
        ev = self._events[self.k & 1]
        ev.record(torch.cuda.current_stream(half._buf.device))
        return ev


class ShardedFmChain:
    """Rotate -> FIR -> FreqDemod (BASELINE configs[4]) over one shard of a node-wide stream.

    FreqDemod needs the FIR output just before the shard (demod/FreqDemod.cpp:63-65 carries it in
    `_prev`), so the halo is K samples -- the FIR's K-1 plus one -- and a rank other than the
    first computes one extra output at the front, whose only purpose is to be that predecessor, and drops
    it.  Rank 0 starts from the reset state exactly as a single-device run does.  Buffer per rank:
    [K halo | C samples], the halo filled in place by the left neighbour.  As in ShardedFir the body of
    the shard is processed while the halo is in flight and the head (first 4096 outputs) afterwards.
    """
    HEAD = 4096

    def __init__(self, taps, phase, shard_len, device, complex_taps=False, algo=None, group=None, two_launch=None, slots=None):
        from . import device as dv   # the HIP path; raises if libpcx_hip.so is missing
        self.two_launch = _two_launch_forced(two_launch)
        self._chains = []
        for _ in range(2):           # head and body calls each start from the reset state
            ch = dv.FmChain()
            ch.set_phase(phase)
            ch.set_taps(taps, complex_taps)
            if algo is not None:
                ch.set_algo(algo)
            self._chains.append(ch)
        self.K = len(taps)
        self.C = int(shard_len)
        self.ring = HaloRing(self.K, group)
        if slots is not None:
            for ch in self._chains:
                ch.set_slots(slots)
        self.slots = slots
        lead = (-(self.K - 1)) % 16      # the sample behind the FIR history of the head call on a 128-byte line
        self._alloc = torch.zeros((lead + self.K + self.C, 2), dtype=torch.float32, device=device)
        self._buf = self._alloc[lead:]
        self._out = torch.empty((self.C + 1,), dtype=torch.float32, device=device)
        self.head = min(self.HEAD, self.C)

    @property
    def buf(self):
        """[K halo | C samples], a view; fences the input.  Views must not be kept across step() (ShardedFir)."""
        self.fence_input()
        return self._buf

    @property
    def shard(self):
        """The C samples this rank owns; taking the view fences the input (ShardedFir.fence_input)."""
        self.fence_input()
        return self._buf[self.K:]

    def fence_input(self):
        if getattr(self, "_side", None) is not None and self._buf.is_cuda:
            torch.cuda.current_stream(self._buf.device).wait_stream(self._side)

    def set_slots(self, slots):
        for ch in self._chains:
            ch.set_slots(slots)
        self.slots = slots

    def check_gate(self):
        """ShardedFir.check_gate"""
        _check_gate(self)

    @property
    def out(self):
        """The C demodulated samples of this shard."""
        return self._out[1:]

    def _run(self, ch, first_in, n_out, out_at):
        ch.reset()
        c, p = ch.process_dev(self._buf[first_in:], self._out[out_at:], n_out + self.K - 1, n_out)
        assert c == n_out and p == n_out, (c, p, n_out)

    def step(self):
        first = self.ring.rank == 0
        if self.ring.world > 1 and self._buf.is_cuda and dist.is_initialized() and dist.get_backend(self.ring.group) == "nccl" and not self.two_launch:
            return self._step_gated()
        reqs = self.ring.start(self._buf)
        if self.C > self.head:
            # body: FIR outputs head-1 .. C-1; the first one only seeds the demodulator and lands on
            # _out[head], which the head call overwrites
            self._run(self._chains[1], self.head, self.C - self.head + 1, self.head)
        self.ring.finish(reqs)
        if first:
            self._run(self._chains[0], 1, self.head, 1)             # stream start: reset state, no extra output
        else:
            self._run(self._chains[0], 0, self.head + 1, 0)         # extra output -1 from the halo, dropped
        return self.out

    def head_reference(self, n):
        """ShardedFir.head_reference (a rank behind a halo: the head call's extra output -1 is dropped by `out`)"""
        self._run(self._chains[0], 0, self.head + 1, 0)
        return self.out[:n]

    def _gate_setup(self):
        if getattr(self, "_gate", None) is None:
            self._gate = torch.zeros((64,), dtype=torch.int32, device=self._buf.device)
            self._side = torch.cuda.Stream(device=self._buf.device)
            self._pass = 0

    def post_exchange(self, after=None):
        """RCCL: the exchange of THIS buffer's halo and the gate signal on the side stream (ShardedFir.post_exchange)."""
        from . import device as dv
        self._gate_setup()
        self._pass += 1
        cur = torch.cuda.current_stream(self._buf.device)
        if after is None:
            self._side.wait_stream(cur)
        else:
            self._side.wait_event(after)
        with torch.cuda.stream(self._side):
            self.ring.finish(self.ring.start(self._buf))
            if self.ring.rank > 0:
                dv.gate_signal(self._gate, self._pass, self._side.cuda_stream)
        _first_exchange(self)

    def compute(self):
        """The pass whose exchange post_exchange() queued: ONE launch over the shard on the current stream."""
        cur = torch.cuda.current_stream(self._buf.device)
        self._launch_stream = cur
        ch = self._chains[0]
        if self.ring.rank == 0:
            self._run(ch, 1, self.C, 1)                             # stream start: reset state, no extra output
        else:
            ch.reset()
            c, p, gated = ch.process_dev_gated(self._buf, self._out, self._gate, self._pass, self.C + 1 + self.K - 1, self.C + 1)
            if not gated:                                           # long filters, short shards: the halo first, then the shard
                cur.wait_stream(self._side)
                self._run(ch, 0, self.C + 1, 0)
            else:
                assert c == self.C + 1 and p == self.C + 1, (c, p)
        return self.out

    def _step_gated(self):
        """RCCL: the exchange and the gate signal on a side stream, ONE launch over the shard on the current stream (ShardedFir)."""
        self.post_exchange()
        return self.compute()


class PingPongFmChain(PingPongFir):
    """PingPongFir for the fused chain: two ShardedFmChain buffers."""

    def __init__(self, taps, phase, shard_len, device, complex_taps=False, algo=None, group=None, two_launch=None, slots=None):
        self.halves = [ShardedFmChain(taps, phase, shard_len, device, complex_taps, algo, group, two_launch, slots) for _ in range(2)]
        self._default_slots(slots)
