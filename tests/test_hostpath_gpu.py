"""GPU suite: the host-pointer FIR entry point on large PAGE-LOCKED buffers, in both of its forms.

  * the product's: the kernel reads and writes the caller's buffers in place over PCIe;
  * the DRAINED form of the diagnostic library (pcx_api.hip drain_*: chunk c's kernel writes a device workspace, a copy engine moves
    the chunk out on a second stream -- measured slower on this platform, profiles/r05/drain_ab.txt, and kept for re-measuring):
    test_the_drained_form_under_the_diagnostic_library runs this file's other tests once more under libpcx_hip_diag.so with
    PCX_DRAIN_FROM set, where every call here goes in two to eight chunks.

Either way a call must return exactly what one device-resident call over everything returns."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from pothoscomms_amd import _lib, device, taps as tp
from tests.util import TOL, nerr

pytestmark = pytest.mark.gpu


class Pinned:
    """a numpy array over a pcx_host_alloc slab"""

    def __init__(self, shape, dtype):
        self.L = _lib.load()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.p = C.c_void_p()
        _lib.check(self.L.pcx_host_alloc(C.byref(self.p), nbytes))
        self.a = np.ctypeslib.as_array((C.c_char * nbytes).from_address(self.p.value)).view(dtype).reshape(shape)

    def free(self):
        if self.p:
            self.a = None
            _lib.check(self.L.pcx_host_free(self.p))
            self.p = None


def _fir_host(f, x, y, n_in, n_out_cap):
    c, p = C.c_size_t(), C.c_size_t()
    _lib.check(_lib.load().pcx_fir_process(f._h, x.ctypes.data, n_in, y.ctypes.data, n_out_cap, C.byref(c), C.byref(p)))
    return c.value, p.value


@pytest.mark.parametrize("K", [255, 257, 63, 1000])
def test_pinned_fir_call_is_bit_identical_to_one_device_call(oracle, K):
    """complex_float32, M = L = 1 on the plain overlap-save plan: the chunks are whole blocks and a chunk's first block reads the
    previous chunk's samples in front of it, so every output is the uncut call's, bit for bit (K = 255: the window of a block
    starts 2 samples before its first input, K = 257: none)"""
    import torch
    rng = np.random.default_rng(K)
    h = (rng.normal(size=K) + 1j * rng.normal(size=K)) / np.sqrt(K)
    n = (1 << 20) + 12345                                  # 8 MiB of output: four chunks
    xin, yout = Pinned((n + K - 1, 2), np.float32), Pinned((n, 2), np.float32)
    try:
        xin.a[:] = rng.uniform(-1, 1, xin.a.shape).astype(np.float32)
        yout.a[:] = np.nan
        f = device.FirFilter("complex_float32", "COMPLEX")
        f.set_taps(h)
        assert _fir_host(f, xin.a, yout.a, n + K - 1, n) == (n, n)
        xd = torch.from_numpy(xin.a).cuda()
        yd = torch.empty((n, 2), dtype=torch.float32, device="cuda")
        assert f.process_dev(xd, yd) == (n, n)
        torch.cuda.synchronize()
        assert np.array_equal(yout.a, yd.cpu().numpy())
        # and against the oracle, around the chunk seams (a quarter of the call each, rounded to whole blocks) and at both ends
        S = 4096 - (K - 1 + 15) // 16 * 16
        Nc = -(-(-(-n // 4)) // S) * S
        ref = oracle.Fir(1, True, True)
        ref.set_taps(h)
        for at in (0, Nc, 2 * Nc, 3 * Nc, n - 3000):
            lo = max(0, at - 1500)
            m = min(3000, n - lo)
            ref.activate()
            want, _, p, _ = ref.work(xin.a[lo:lo + m + K - 1], m)
            assert p == m and nerr(yout.a[lo:lo + m], want) <= TOL, at
        # a pageable INPUT with a page-locked output: staged in, drained out -- the same bits
        yout.a[:] = np.nan
        assert _fir_host(f, np.array(xin.a), yout.a, n + K - 1, n) == (n, n)
        assert np.array_equal(yout.a, yd.cpu().numpy())
    finally:
        xin.free(); yout.free()


@pytest.mark.parametrize("K", [2050, 4097, 6000, 8193])
def test_long_tap_fir_does_not_depend_on_who_computes_which_block(oracle, K):
    """the partitioned plan (fir_ols_part.hip): block b is a function of windows b, b - 1, ... alone -- a pinned-host call (48
    workgroups, long runs) and a device call (512 workgroups) return the same bits; a call cut in two agrees within the bar (its first
    blocks see zeros where the uncut call's windows hold samples that meet no tap)"""
    import torch
    rng = np.random.default_rng(K)
    h = (rng.normal(size=K) + 1j * rng.normal(size=K)) / np.sqrt(K)
    n = (1 << 20) + 777
    xin, yout = Pinned((n + K - 1, 2), np.float32), Pinned((n, 2), np.float32)
    try:
        xin.a[:] = rng.uniform(-1, 1, xin.a.shape).astype(np.float32)
        yout.a[:] = np.nan
        f = device.FirFilter("complex_float32", "COMPLEX")
        f.set_taps(h)
        assert _fir_host(f, xin.a, yout.a, n + K - 1, n) == (n, n)
        xd = torch.from_numpy(xin.a).cuda()
        yd = torch.empty((n, 2), dtype=torch.float32, device="cuda")
        assert f.process_dev(xd, yd) == (n, n)
        torch.cuda.synchronize()
        whole = yd.cpu().numpy()
        if os.environ.get("PCX_HOSTPATH_INNER"):           # the drained form (below) CUTS the host call: within the bar, see the docstring
            assert nerr(yout.a, whole) <= TOL
        else:
            assert np.array_equal(yout.a, whole)
        cut = 2048 * 137
        y2 = torch.full((n, 2), float("nan"), dtype=torch.float32, device="cuda")
        assert f.process_dev(xd[:cut + K - 1], y2[:cut]) == (cut, cut)
        assert f.process_dev(xd[cut:], y2[cut:]) == (n - cut, n - cut)
        torch.cuda.synchronize()
        assert nerr(y2.cpu().numpy(), whole) <= TOL and np.array_equal(y2[:cut - 8192].cpu().numpy(), whole[:cut - 8192])
        ref = oracle.Fir(1, True, True)
        ref.set_taps(h)
        for lo in (0, cut - 700, n - 1500):
            ref.activate()
            want, _, p, _ = ref.work(xin.a[lo:lo + 1500 + K - 1], 1500)
            assert p == 1500 and nerr(whole[lo:lo + 1500], want) <= TOL, lo
    finally:
        xin.free(); yout.free()


@pytest.mark.parametrize("dtype,M,L", [("complex_float32", 8, 1), ("complex_float32", 1, 4), ("complex_int16", 1, 1), ("complex_int16", 2, 1),
                                       ("float32", 1, 1), ("complex_float64", 1, 1)])
def test_pinned_fir_call_other_plans(oracle, dtype, M, L):
    """resampling, integer, real and double streams through the drained call: consume / produce totals are the reference's, the
    integer results bit-exact, the float ones within the bar against the oracle across the chunk seams"""
    from tests.util import rand_stream
    scalar, cplx = device.parse_dtype(dtype)
    rng = np.random.default_rng(M * 10 + L)
    K = 127
    h = (rng.normal(size=K * L) + (1j * rng.normal(size=K * L) if cplx else 0)) / np.sqrt(K)
    if scalar == 4:
        h = h * 0.2
    esz = device.NP_SCALAR[scalar]().itemsize * (2 if cplx else 1)
    n = max(1 << 20, (4 << 20) // esz * M // L)            # at least 4 MiB of output: the drained form, two chunks or more
    n_out_cap = n * L // M + 8
    shape_in = (n + K - 1, 2) if cplx else (n + K - 1,)
    shape_out = (n_out_cap, 2) if cplx else (n_out_cap,)
    xin, yout = Pinned(shape_in, device.NP_SCALAR[scalar]), Pinned(shape_out, device.NP_SCALAR[scalar])
    try:
        xin.a[:] = rand_stream(rng, scalar, n + K - 1, cplx, amp=3000)
        f = device.FirFilter(dtype, "COMPLEX" if cplx else "REAL")
        f.set_taps(h); f.set_decimation(M); f.set_interpolation(L)
        c, p = _fir_host(f, xin.a, yout.a, n + K - 1, n_out_cap)
        ref = oracle.Fir(scalar, cplx, cplx)
        ref.set_taps(h); ref.set_decimation(M); ref.set_interpolation(L); ref.activate()
        m = 300000                                          # the oracle on the first 300k inputs: covers the first chunk seam (a quarter)
        want, rc, rp, _ = ref.work(xin.a[:m + K - 1], m * L // M)
        assert c == (n // M) * M and p == (n // M) * L
        assert p * esz >= (2 << 20)                         # (large enough for two chunks or more in the drained form)
        got = yout.a[:rp]
        if scalar in (4, 5):
            assert np.array_equal(got, want)
        else:
            assert nerr(got, want) <= (TOL if scalar == 1 else 1e-12)
        # the tail of the call against the oracle as well (last chunk)
        ref.activate()
        t0 = (n - 200000) // M * M
        want, rc, rp, _ = ref.work(xin.a[t0:n + K - 1], 200000 * L // M + 8)
        got = yout.a[t0 // M * L:t0 // M * L + rp]
        if scalar in (4, 5):
            assert np.array_equal(got, want)
        else:
            assert nerr(got, want) <= (TOL if scalar == 1 else 1e-12)
    finally:
        xin.free(); yout.free()


@pytest.mark.parametrize("scalar", [1, 4, 0])
def test_link_bound_launch_shapes_return_the_same_bits(scalar):
    """A host-pointer call on PAGE-LOCKED memory takes the link-bound launch shape (pcx_internal.hpp LINK-BOUND LAUNCHES: 32 / 64 blocks for
    the grid-stride maps, 48 workgroups for the persistent block kernels); the same call on pageable memory is staged and takes the
    device-resident shape.  Same kernels, same per-element arithmetic: every entry point must return the same bits either way -- odd
    lengths, so that the last block of either grid is ragged."""
    from tests.util import rand_stream
    rng = np.random.default_rng(100 + scalar)
    dt = device.NP_SCALAR[scalar]
    n = (1 << 20) + 12347
    x = rand_stream(rng, scalar, n, True, amp=20000)
    x2 = rand_stream(rng, scalar, n, True, amp=20000)
    px, px2 = Pinned((n, 2), dt), Pinned((n, 2), dt)
    po, po1 = Pinned((n, 2), dt), Pinned((n,), dt)
    pre, pim = Pinned((n,), dt), Pinned((n,), dt)
    try:
        px.a[:] = x
        px2.a[:] = x2
        for name, pinned_call, plain_call in (
                ("rotate", lambda: device.rotate(px.a, 0.7, out=po.a), lambda: device.rotate(x, 0.7)),
                ("scale", lambda: device.scale(px.a, 0.3337, True, out=po.a), lambda: device.scale(x, 0.3337, True)),
                ("conj", lambda: device.conj(px.a, out=po.a), lambda: device.conj(x)),
                ("abs", lambda: device.abs_(px.a, True, out=po1.a), lambda: device.abs_(x, True)),
                ("angle", lambda: device.angle(px.a, out=po1.a), lambda: device.angle(x)),
                ("arith MUL", lambda: device.arith("MUL", px.a, px2.a, True, out=po.a), lambda: device.arith("MUL", x, x2, True))):
            got = np.array(pinned_call())
            want = plain_call()
            assert np.array_equal(got, want, equal_nan=True), name
        re, im = device.split_complex(px.a, re=pre.a, im=pim.a)
        wre, wim = device.split_complex(x)
        assert np.array_equal(re, wre) and np.array_equal(im, wim)
        assert np.array_equal(device.combine_complex(pre.a, pim.a, out=po.a), device.combine_complex(wre, wim))
        # carried state across two calls each way
        a, b = device.FreqDemod("complex_" + dt.__name__), device.FreqDemod("complex_" + dt.__name__)
        h = n // 2
        got = np.concatenate([np.array(a.process(px.a[:h], out=po1.a[:h])), np.array(a.process(px.a[h:], out=po1.a[h:]))])
        want = np.concatenate([b.process(x[:h]), b.process(x[h:])])
        assert np.array_equal(got, want), "freq_demod"
        if scalar == 1:
            ch = device.FmChain(); ch.set_phase(0.7); ch.set_taps(tp.c4_taps(), False)
            c, p = C.c_size_t(), C.c_size_t()
            L = _lib.load()
            m = n - 126
            _lib.check(L.pcx_fmchain_process(ch._h, px.a.ctypes.data, n, po1.a.ctypes.data, m, C.byref(c), C.byref(p)))
            assert (c.value, p.value) == (m, m)
            got = np.array(po1.a[:m])
            ch.reset()
            y = np.empty((m,), np.float32)
            _lib.check(L.pcx_fmchain_process(ch._h, x.ctypes.data, n, y.ctypes.data, m, C.byref(c), C.byref(p)))
            assert np.array_equal(got, y), "fm chain"
            fft = device.Fft("complex_float32", 4096, False)
            nf = n // 4096
            _lib.check(L.pcx_fft_transform(fft._h, px.a.ctypes.data, po.a.ctypes.data, nf))
            assert np.array_equal(po.a[:nf * 4096], fft.transform(x[:nf * 4096])), "fft"
    finally:
        for b in (px, px2, po, po1, pre, pim):
            b.free()


def test_the_drained_form_under_the_diagnostic_library():
    """the same tests, every call drained in chunks (libpcx_hip_diag.so reads PCX_DRAIN_FROM / PCX_DRAIN_CHUNK; the product never drains)"""
    if os.environ.get("PCX_HOSTPATH_INNER"):
        pytest.skip("this IS the inner run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "pothoscomms_amd", "libpcx_hip_diag.so")
    assert os.path.exists(diag), "make -C pothoscomms_amd/csrc diag"
    env = dict(os.environ, PCX_HIP_LIBRARY=diag, PCX_DRAIN_FROM=str(1 << 20), PCX_DRAIN_CHUNK=str(2 << 20), PCX_HOSTPATH_INNER="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-p", "no:xdist"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
