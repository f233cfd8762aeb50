"""Shared helpers for the parity tests."""
import numpy as np

from oracle import oracle as o

SCALARS = [o.F64, o.F32, o.I64, o.I32, o.I16, o.I8]
NAMES = {o.F64: "float64", o.F32: "float32", o.I64: "int64", o.I32: "int32", o.I16: "int16", o.I8: "int8"}
TOL = 1e-5  # north_star: within 1e-5 relative for float32, normalised by max|ref|


def rand_stream(rng, scalar, n, is_complex, amp=None):
    """Random stream of `n` elements as (n,2) pairs or (n,) reals of the scalar type."""
    dt = o.NP_SCALAR[scalar]
    shape = (n, 2) if is_complex else (n,)
    if np.issubdtype(dt, np.floating):
        return rng.uniform(-1, 1, shape).astype(dt)
    info = np.iinfo(dt)
    if amp is None:
        return rng.integers(info.min, info.max + 1, shape, dtype=dt)
    return rng.integers(-amp, amp + 1, shape).astype(dt)


def nerr(got, ref):
    """max|got-ref| / max|ref| per buffer."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if ref.size == 0:
        return 0.0
    den = np.max(np.abs(ref))
    return float(np.max(np.abs(got - ref)) / (den if den > 0 else 1.0))


def ang_err(got, ref):
    """max |wrap_pi(got-ref)| / pi (FreqDemod metric: +pi and -pi are the same angle)."""
    d = np.asarray(got, np.float64) - np.asarray(ref, np.float64)
    d = (d + np.pi) % (2 * np.pi) - np.pi
    return float(np.max(np.abs(d)) / np.pi) if d.size else 0.0


def diff_note(got, want):
    """where two arrays differ, for an assertion message: count, extent, how many of the differing rows are all zero"""
    g, w = np.asarray(got), np.asarray(want)
    rows = np.any(g != w, axis=tuple(range(1, g.ndim))) if g.ndim > 1 else g != w
    bad = np.flatnonzero(rows)
    if bad.size == 0:
        return "equal"
    zero = int(np.sum(np.all(g[bad] == 0, axis=tuple(range(1, g.ndim))) if g.ndim > 1 else g[bad] == 0))
    runs = np.split(bad, np.flatnonzero(np.diff(bad) > 1) + 1)
    return "%d rows of %d differ (%d of them zero) in %d runs: %s" % (
        bad.size, g.shape[0], zero, len(runs), ", ".join("%d..%d" % (r[0], r[-1]) for r in runs[:8]))


def h2d(a, d):
    """numpy -> device tensor.  With PCX_TEST_PINNED=1 through page-locked memory, so that under a runtime-mode variation
    (profiles/r02/contention.md section 4) the test's own transfers stay off the runtime's pageable copy path and a difference the
    test reports is the library's; the plain path otherwise (page-locking every case makes the soak five times longer)."""
    import os

    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.pin_memory() if os.environ.get("PCX_TEST_PINNED") else t).to(d)


def d2h(t):
    """device tensor -> numpy (page-locked bounce under PCX_TEST_PINNED=1)"""
    import os

    import torch
    if not os.environ.get("PCX_TEST_PINNED"):
        return t.cpu().numpy()
    out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    out.copy_(t)
    torch.cuda.current_stream(t.device).synchronize()
    return out.numpy().copy()
