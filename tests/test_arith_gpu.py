"""GPU parity for /comms/arithmetic, split_complex, combine_complex through the C ABI (pcx_arith*,
pcx_split_complex*, pcx_combine_complex*) against the oracle.  Bars: bit-exact for every integer
type; float results are bit-identical too for finite normal-range operands (same operation order,
no contraction), asserted as such with a 1e-5 fallback stated for complex division.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))
TYPES = ["int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float32", "float64"]
OPS = ["ADD", "SUB", "MUL", "DIV"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def operands(rng, dt, shape, for_div):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return (rng.standard_normal(shape) * 100).astype(dt), (rng.standard_normal(shape) * 10).astype(dt)
    info = np.iinfo(dt)
    a = rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
    b = rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
    if for_div:   # zero divisors trap and MIN/-1 is undefined in the reference: separate test below
        small = rng.integers(-9 if info.min < 0 else 1, 10, size=shape)
        b = np.where(small == 0, 3, small).astype(dt)
        a = (a // 4).astype(dt)
    return a, b


@pytest.mark.parametrize("name", TYPES)
@pytest.mark.parametrize("opname", OPS)
def test_reference_test_vectors(dev, name, opname):
    """math/TestArithmeticBlocks.cpp:47-245 through the device path, ports folded as work() does."""
    ins = []
    while "arith_%s_%s_in%d" % (opname, name, len(ins)) in GOLD.files:
        ins.append(GOLD["arith_%s_%s_in%d" % (opname, name, len(ins))])
    acc = ins[0]
    for a in ins[1:]:
        acc = dev.arith(opname, acc, a, False)
    assert np.array_equal(bits(acc), bits(GOLD["arith_%s_%s_exp" % (opname, name)]))
    acc = ins[0].reshape(50, 2)
    for a in ins[1:]:
        acc = dev.arith(opname, acc, a.reshape(50, 2), True)
    assert np.array_equal(bits(acc), bits(GOLD["arith_%s_c%s_exp" % (opname, name)]))


@pytest.mark.parametrize("name", TYPES)
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("opname", OPS)
def test_arith_vs_oracle(oracle, dev, name, cplx, opname):
    # a stable digest, not hash(): str hashes are salted per process and a failure must be reproducible
    import zlib
    rng = np.random.default_rng(zlib.crc32(("%s/%d/%s" % (name, int(cplx), opname)).encode()))
    for n in (1, 7, 1000, 70001):      # ragged tails behind the 16-byte vectors
        a, b = operands(rng, name, (n, 2) if cplx else (n,), opname == "DIV")
        got = dev.arith(opname, a, b, cplx)
        ref = oracle.arith(getattr(oracle, opname), a, b, cplx)
        if np.dtype(name).kind == "f" and cplx and opname == "DIV":
            # bit-identical in practice; the stated bar for float work is 1e-5 of max|ref|
            assert np.max(np.abs(got - ref)) <= 1e-5 * np.max(np.abs(ref))
        assert np.array_equal(bits(got), bits(ref)), (n,)


def test_integer_division_corner_cases(oracle, dev):
    for dt in (np.int8, np.int16, np.int32, np.int64):
        info = np.iinfo(dt)
        a = np.array([7, -7, info.min, 5, info.min, 0], dt)
        b = np.array([0, 2, -1, -1, 1, 0], dt)
        assert np.array_equal(dev.arith("DIV", a, b, False), oracle.arith(oracle.DIV, a, b, False))
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
        a = np.array([7, 200, np.iinfo(dt).max, 0], dt)
        b = np.array([0, 3, 1, 0], dt)
        assert np.array_equal(dev.arith("DIV", a, b, False), oracle.arith(oracle.DIV, a, b, False))


def test_unsupported_args(dev):
    from pothoscomms_amd import _lib
    with pytest.raises(_lib.InvalidArgument):
        dev.arith("POW", np.zeros(4, np.float32), np.zeros(4, np.float32), False)


def test_device_buffers_in_place_fold(oracle, dev):
    """Device-resident operands; out aliases in0 and a third port folds in place (Arithmetic.cpp:217-224)."""
    import torch
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    n = 1 << 20
    xs = [(rng.standard_normal((n, 2)) * 10).astype(np.float32) for _ in range(3)]
    ts = [torch.from_numpy(x).to(d) for x in xs]
    out = ts[0].clone()
    dev.arith("MUL", out, ts[1], True, scalar=dev.F32, out=out, n=n)      # out == in0
    dev.arith("MUL", out, ts[2], True, scalar=dev.F32, out=out, n=n)
    ref = oracle.arith(oracle.MUL, oracle.arith(oracle.MUL, xs[0], xs[1], True), xs[2], True)
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    out2 = ts[1].clone()
    dev.arith("SUB", ts[0], out2, True, scalar=dev.F32, out=out2, n=n)    # out == in1
    assert np.array_equal(bits(out2.cpu().numpy()), bits(oracle.arith(oracle.SUB, xs[0], xs[1], True)))
    # misaligned views take the scalar path
    a, b, o = ts[0][1:], ts[1][1:], torch.empty((n - 1, 2), dtype=torch.float32, device=d)
    dev.arith("ADD", a.reshape(-1)[1:-1], b.reshape(-1)[1:-1], False, scalar=dev.F32, out=o.reshape(-1)[1:-1], n=2 * (n - 1) - 2)
    want = oracle.arith(oracle.ADD, xs[0][1:].reshape(-1)[1:-1], xs[1][1:].reshape(-1)[1:-1], False)
    assert np.array_equal(bits(o.reshape(-1)[1:-1].cpu().numpy()), bits(want))


@pytest.mark.parametrize("name", ["int8", "int16", "int32", "int64", "float32", "float64"])
def test_split_combine(oracle, dev, name):
    dt = np.dtype(name)
    rng = np.random.default_rng(4)
    for n in (0, 1, 5, 1000, 65539):
        re = (rng.standard_normal(n) * 100).astype(dt)
        im = (rng.standard_normal(n) * 100).astype(dt)
        z = dev.combine_complex(re, im)
        assert np.array_equal(bits(z), bits(oracle.combine_complex(re, im)))
        r2, i2 = dev.split_complex(z)
        o_re, o_im = oracle.split_complex(z)
        assert np.array_equal(bits(r2), bits(o_re)) and np.array_equal(bits(i2), bits(o_im))
        assert np.array_equal(bits(r2), bits(re)) and np.array_equal(bits(i2), bits(im))   # utility/TestComplex.cpp


def test_full_size_round_trip(dev):
    """64 Mi complex_float32 elements: split -> combine is the identity; (a*b)/b ~ a; a-b+b == a for ints."""
    import torch
    d = torch.device("cuda", 0)
    n = 64 * 1024 * 1024
    x = torch.empty((n, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x, seed=11)
    re = torch.empty((n,), dtype=torch.float32, device=d)
    im = torch.empty((n,), dtype=torch.float32, device=d)
    y = torch.empty_like(x)
    dev.split_complex(x, scalar=dev.F32, re=re, im=im, n=n)
    assert torch.equal(re, x[:, 0]) and torch.equal(im, x[:, 1])
    dev.combine_complex(re, im, scalar=dev.F32, out=y, n=n)
    assert torch.equal(x, y)
    b = torch.empty_like(x)
    dev.fill_uniform_f32_dev(b, seed=12)
    b += 2.0                                   # keep |b| away from zero
    dev.arith("MUL", x, b, True, scalar=dev.F32, out=y, n=n)
    dev.arith("DIV", y, b, True, scalar=dev.F32, out=y, n=n)
    assert float((y - x).abs().max()) <= 1e-5 * float(x.abs().max()) * 4
    xi = (x * 30000).to(torch.int16)
    bi = (b * 1000).to(torch.int16)
    yi = torch.empty_like(xi)
    dev.arith("SUB", xi, bi, True, scalar=dev.I16, out=yi, n=n)
    dev.arith("ADD", yi, bi, True, scalar=dev.I16, out=yi, n=n)
    assert torch.equal(xi, yi)
