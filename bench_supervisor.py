"""bench_supervisor.py -- `bench.py --gpus N` must end with a result line whatever the first contact with real peers does.

Every rank process torch.distributed.run starts (bench.py with WORLD_SIZE > 1 in its environment) becomes a SUPERVISOR: it never
imports torch, never touches the GPU, and runs the actual rank as a CHILD process (a fresh interpreter of bench.py, PCX_BENCH_CHILD=1,
its own process group).  The N supervisors of a node agree through small files in one directory under /tmp (one node: the filesystem
is shared; no torch store, no collective that could hang with the thing it watches).  Per attempt:

  1. rank 0's supervisor picks a free rendezvous port and publishes it; every supervisor starts its child on it;
  2. the child reports where it is with heartbeat lines on stderr (`bench.py[hb] rank R: <phase>`); a child that stays silent for
     longer than the phase allows, or exits non-zero, fails the attempt -- its supervisor says so in a status file, the other
     supervisors see that file and stop their own children (which would otherwise wait in a collective for the dead rank);
  3. rank 0's child prints its result line BEFORE it tears the process group down; from then on ("result in hand") a hang or a crash
     in any rank's teardown no longer fails the attempt;
  4. all attempts' children are fresh processes: nothing of a failed attempt (a wedged communicator, a stuck stream) survives.

The attempts (ATTEMPTS below): as asked; then CONSERVATIVE -- two launches per pass (no gate), one input buffer (no pipelining), no
autotune, no launch-stream probe: none of the mechanisms that have only ever met a rank sending to itself; then the halo staged
through the host over gloo, RCCL not used at all.  The line of a later attempt says which attempt it is and why the earlier ones
failed (config.attempt, config.fallback_reason).  When every attempt has failed the supervisors exit non-zero and no line is printed.

Nothing here uses os.exec*; children are started with subprocess and killed by their own process-group id only.
"""
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

HB = "bench.py[hb]"

CONSERVATIVE_ENV = {"PCX_STREAM_TWO_LAUNCH": "1", "PCX_BENCH_NO_STREAM_PICK": "1"}
CONSERVATIVE_ARGS = ["--no-pingpong", "--no-autotune"]
ATTEMPTS = [
    {"mode": "as asked", "env": {}, "args": []},
    {"mode": "conservative: two launches per pass (no gate), one input buffer (no pipelining), no autotune, no launch-stream probe",
     "env": dict(CONSERVATIVE_ENV), "args": list(CONSERVATIVE_ARGS)},
    {"mode": "halo staged through the host over gloo (RCCL not used), two launches per pass, one input buffer",
     # (PCX_BENCH_OWN_GPU: a rank refuses to share a device -- the rehearsal a user asks for with PCX_BENCH_BACKEND=gloo may fold ranks
     # onto one GPU, a FALLBACK must not turn "fewer GPUs than ranks" into a line that says n_gpus = N)
     "env": dict(CONSERVATIVE_ENV, PCX_BENCH_BACKEND="gloo", PCX_BENCH_OWN_GPU="1"), "args": list(CONSERVATIVE_ARGS)},
]

# seconds of SILENCE (no heartbeat, no other stderr line) a child is allowed, by the phase its last heartbeat named.  The first import of
# torch on a fresh box pages the image in (1-2 minutes for one process; eight at once share the disk); RCCL brings its communicator and
# its peer connections up inside the first collective.  PCX_BENCH_WATCHDOG_S overrides every limit (the tests use a few seconds).
# A healthy run passes every phase in seconds (the communicator of eight ranks: well under a minute); the limits are what a HUNG attempt costs
# before the next one starts, so they are kept as short as a slow box allows: three hung attempts in a row end inside ~12 minutes.
SILENCE_LIMITS = {None: 420.0, "started": 420.0, "torch imported": 150.0, "process group up": 150.0}
SILENCE_DEFAULT = 90.0
ATTEMPT_BUDGET_S = 900.0            # an attempt that is still running after this long is stopped whatever it says
RESULT_GRACE_S = 20.0               # teardown time a child gets once rank 0's line is in hand
PEER_FAILED_GRACE_S = 3.0           # time a child gets to finish by itself once another rank's failure is known


def _limit(phase):
    o = os.environ.get("PCX_BENCH_WATCHDOG_S")
    if o:
        # (the tests' few seconds apply once torch is imported: the import itself can take a minute on a box whose page cache is cold)
        return max(float(o), 120.0) if phase in (None, "started") else float(o)
    return SILENCE_LIMITS.get(phase, SILENCE_DEFAULT)


def _proc_start_time(pid):
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[19]      # field 22: start time in clock ticks since boot
    except (OSError, IndexError):
        return "0"


def session_dir():
    """One directory per launch, the same on every rank: keyed by the launcher (the parent of all rank processes: its pid and start
    time) and the rendezvous port it handed out."""
    base = os.environ.get("PCX_BENCH_SESSION_DIR")
    if base:
        return base
    ppid = os.getppid()
    key = "%d_%s_%s" % (ppid, _proc_start_time(ppid), os.environ.get("MASTER_PORT", "0"))
    # the first of these that can be written to (the same answer on every rank of a node: they share the file system and the user)
    for root in (os.environ.get("TMPDIR"), "/tmp", "/dev/shm", os.path.dirname(os.path.abspath(__file__))):
        if root and os.path.isdir(root) and os.access(root, os.W_OK | os.X_OK):
            return os.path.join(root, "pcx_bench_" + key)
    return os.path.join("/tmp", "pcx_bench_" + key)


def _write_atomic(path, text):
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _wait_for(paths, timeout):
    """-> {path: text} of those that appeared within `timeout` seconds (all of them, or fewer)"""
    got = {}
    end = time.monotonic() + timeout
    while True:
        for p in paths:
            if p not in got:
                t = _read(p)
                if t is not None:
                    got[p] = t
        if len(got) == len(paths) or time.monotonic() >= end:
            return got
        time.sleep(0.05)


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


class _Child:
    """One rank's child process: started in its own session (so that it can be killed as a group and nothing else is), its stderr relayed
    line by line as it comes, its heartbeats and its stdout kept."""

    def __init__(self, cmd, env, rank):
        self.rank = rank
        self.proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, start_new_session=True)
        self.phase = None
        self.last_sign = time.monotonic()
        self.last_err = ""
        self.stdout_lines = []
        self.result_line = None
        self._lock = threading.Lock()
        self._threads = [threading.Thread(target=self._pump_err, daemon=True), threading.Thread(target=self._pump_out, daemon=True)]
        for t in self._threads:
            t.start()

    def _pump_err(self):
        for ln in self.proc.stderr:
            with self._lock:
                self.last_sign = time.monotonic()
                if ln.startswith(HB):
                    self.phase = ln.split(":", 1)[1].strip() if ":" in ln else ln.strip()
                elif ln.strip():
                    self.last_err = ln.strip()[-400:]
            sys.stderr.write(ln)
            sys.stderr.flush()

    def _pump_out(self):
        for ln in self.proc.stdout:
            with self._lock:
                self.last_sign = time.monotonic()
                if ln.startswith("{") and '"metric"' in ln:
                    self.result_line = ln.strip()
                else:
                    self.stdout_lines.append(ln)
            if not (ln.startswith("{") and '"metric"' in ln):
                sys.stderr.write(ln)                    # (RCCL's banner and the like: never on the supervisor's stdout)
                sys.stderr.flush()

    def silent_for(self):
        with self._lock:
            return time.monotonic() - self.last_sign

    def kill(self):
        if self.proc.poll() is None:
            try:
                os.killpg(self.proc.pid, signal.SIGKILL)          # its own session: exactly the processes this supervisor started
            except (ProcessLookupError, PermissionError):
                pass
        try:
            self.proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass

    def drain(self):
        for t in self._threads:
            t.join(timeout=2)


def _run_attempt(a, att, rank, world, sdir, argv, reasons, script, run_no=None):
    """-> (ok, status text of this rank, result line or None).  a: the attempt (which mode), run_no: the run (names the files)"""
    tag = os.path.join(sdir, "a%d" % (a if run_no is None else run_no))
    if rank == 0:
        _write_atomic(tag + ".port", str(_free_port()))
    got = _wait_for([tag + ".port"], 120.0)
    if not got:
        return False, "no rendezvous port from rank 0's supervisor within 120 s", None
    port = got[tag + ".port"].strip()
    env = dict(os.environ)
    env.update(att["env"])
    env.update({"PCX_BENCH_CHILD": "1", "PCX_BENCH_ATTEMPT": str(a + 1), "PCX_BENCH_ATTEMPT_MODE": att["mode"],
                "PCX_BENCH_FALLBACK_REASON": json.dumps(reasons), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port,
                "RANK": str(rank), "WORLD_SIZE": str(world)})
    env.setdefault("LOCAL_RANK", str(rank))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the child's own rank 0 hosts the rendezvous store on the fresh port: the launcher's agent store belongs to the supervisors' world
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    child = _Child([sys.executable, script] + argv + att["args"], env, rank)
    _CURRENT[0] = child
    t0 = time.monotonic()
    grace = float(os.environ.get("PCX_BENCH_RESULT_GRACE_S", RESULT_GRACE_S))
    status, ok = None, False
    peer_failed_at = None
    result_at = None
    while True:
        rc = child.proc.poll()
        now = time.monotonic()
        have_result = os.path.exists(tag + ".result") or (rank == 0 and child.result_line is not None)
        if rank == 0 and child.result_line is not None and not os.path.exists(tag + ".result"):
            _write_atomic(tag + ".result", child.result_line)
        if have_result and result_at is None:
            result_at = now
        if rc is not None:
            child.drain()
            if rc == 0 or have_result:
                ok, status = True, "ok" if rc == 0 else "ok (exit code %d in teardown, behind the result line)" % rc
            else:
                status = "exit code %d in phase %r: %s" % (rc, child.phase, child.last_err or "(nothing on stderr)")
            break
        if have_result and now - result_at > grace:
            child.kill()
            ok, status = True, "ok (stopped in teardown %.0f s behind the result line)" % grace
            break
        if not have_result:
            if child.silent_for() > _limit(child.phase):
                child.kill()
                status = "silent for %.0f s in phase %r: stopped by the watchdog" % (_limit(child.phase), child.phase)
                break
            if now - t0 > ATTEMPT_BUDGET_S:
                child.kill()
                status = "still running after %.0f s (phase %r): stopped" % (ATTEMPT_BUDGET_S, child.phase)
                break
            if peer_failed_at is None:
                for r in range(world):
                    if r != rank:
                        st = _read("%s.rank%d" % (tag, r))
                        if st is not None and not json.loads(st)["ok"]:
                            peer_failed_at = now
                            peer = (r, json.loads(st)["status"])
                            break
            elif now - peer_failed_at > PEER_FAILED_GRACE_S:
                child.kill()
                status = "stopped: rank %d failed (%s)" % peer
                break
        time.sleep(0.1)
    _CURRENT[0] = None
    _write_atomic("%s.rank%d" % (tag, rank), json.dumps({"ok": ok, "status": status, "phase": child.phase, "t": time.time()}))
    # every supervisor waits for every verdict of this attempt: they move on (or stop) together
    paths = ["%s.rank%d" % (tag, r) for r in range(world)]
    # (a peer learns of a failure within PEER_FAILED_GRACE_S and of a result at once; a peer whose supervisor is gone takes the launcher down)
    got = _wait_for(paths, ATTEMPT_BUDGET_S + 60.0)
    verdicts = {}
    for r, p in enumerate(paths):
        verdicts[r] = json.loads(got[p]) if p in got else {"ok": False, "status": "its supervisor never reported"}
    result = _read(tag + ".result")
    if result is not None:
        # the measurement is complete (the line is written behind the timed region, the max-over-ranks clock and the seam check)
        return True, status, result.strip()
    # the root cause first: the rank whose failure was noticed FIRST (the others' errors -- a peer's connection closed, a collective that
    # never returned -- follow it); a rank that was stopped because another had failed is a consequence and goes last
    failed = sorted(((v.get("t", float("inf")), r, v) for r, v in verdicts.items() if not v["ok"]), key=lambda x: ("stopped: rank" in x[2]["status"], x[0], x[1]))
    bad = ["rank %d: %s" % (r, v["status"]) for _, r, v in failed]
    if not bad:
        bad = ["every rank ended cleanly but rank 0 printed no result line"]
    return False, "; ".join(bad[:3]) + (" (+%d more)" % (len(bad) - 3) if len(bad) > 3 else ""), None


_CURRENT = [None]


def _on_signal(signum, frame):
    c = _CURRENT[0]
    if c is not None:
        c.kill()
    sys.exit(128 + signum)


def supervise(argv, script):
    """The body of a rank process started by torch.distributed.run.  argv: bench.py's own arguments (sys.argv[1:])."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ["WORLD_SIZE"])
    # ONE node: the verdict and port files live under this node's /tmp, the rendezvous is 127.0.0.1, LOCAL_RANK defaults to RANK.  Under a
    # multi-node launch every supervisor off node 0 would wait two minutes per run for files that never appear: refused up front, by every rank
    nodes = int(os.environ.get("GROUP_WORLD_SIZE", "1") or "1")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)) or world)
    if nodes > 1 or local_world != world:
        if int(os.environ.get("LOCAL_RANK", str(rank)) or rank) == 0:
            print("bench.py: the supervised bench runs the ranks of ONE node (--nnodes=1): this launch has %d node group(s), %d of %d ranks "
                  "on this node" % (nodes, local_world, world), file=sys.stderr, flush=True)
        return 2
    sdir = session_dir()
    os.makedirs(sdir, exist_ok=True)
    signal.signal(signal.SIGTERM, _on_signal)
    signal.signal(signal.SIGINT, _on_signal)
    attempts = list(ATTEMPTS)
    limit = os.environ.get("PCX_BENCH_MAX_ATTEMPTS")
    if limit:
        attempts = attempts[:max(1, int(limit))]
    if os.environ.get("PCX_BENCH_BACKEND", "nccl") != "nccl":
        attempts = [x for x in attempts if "PCX_BENCH_BACKEND" not in x["env"]]      # already host-staged: nothing further to fall back to
    reasons = []
    code = 1
    run_no = 0
    for a, att in enumerate(attempts):
        # (the rendezvous port is free when rank 0's supervisor picks it and may be taken when the child binds it: that failure -- every
        # supervisor reads the same verdict files, so they agree on it -- repeats the SAME attempt on another port, twice at most)
        for again in range(3):
            ok, status, line = _run_attempt(a, att, rank, world, sdir, argv, reasons, script, run_no)
            run_no += 1
            if ok or not any(m in status for m in ("EADDRINUSE", "Address already in use", "errno: 98", "address already in use")):
                break
            if rank == 0:
                print("bench.py: the rendezvous port was taken before the ranks could bind it, the same attempt again", file=sys.stderr, flush=True)
        if ok:
            # rank 0 holds the line: it decides whether the job produced one, and every rank leaves with ITS verdict
            final = os.path.join(sdir, "final.run%d" % run_no)
            if rank == 0:
                got = json.loads(line)
                if got.get("n_gpus") != world:
                    print("bench.py: asked for %d GPUs, the ranks report n_gpus=%r: no result line" % (world, got.get("n_gpus")), file=sys.stderr, flush=True)
                    _write_atomic(final, "mismatch")
                    break
                print(line, flush=True)
                _write_atomic(final, "ok")
                code = 0
                break
            _wait_for([final], 30.0)
            code = 0 if (_read(final) or "ok").strip() == "ok" else 1
            break
        reasons.append("attempt %d (%s): %s" % (a + 1, att["mode"].split(":")[0], status))
        if rank == 0:
            print("bench.py: %s" % reasons[-1], file=sys.stderr, flush=True)
            if a + 1 < len(attempts):
                print("bench.py: starting fresh rank processes, %s" % attempts[a + 1]["mode"], file=sys.stderr, flush=True)
    else:
        if rank == 0:
            print("bench.py: every attempt failed, no result line", file=sys.stderr, flush=True)
    # leave the directory to rank 0, once everybody is through
    _write_atomic(os.path.join(sdir, "done.rank%d" % rank), "")
    if rank == 0 and not os.environ.get("PCX_BENCH_SESSION_DIR"):
        _wait_for([os.path.join(sdir, "done.rank%d" % r) for r in range(world)], 30.0)
        import shutil
        shutil.rmtree(sdir, ignore_errors=True)
    return code
