"""CPU suite for the /comms/arithmetic, split_complex, combine_complex row: pins
oracle/pcx_oracle.c's restatement of the element operators against

  * math/TestArithmeticBlocks.cpp:47-245's own vectors (tests/golden: inputs + the expectations the
    test computes, generated with the C++ operators themselves),
  * seeded random operands evaluated by std::complex / the C++ operators (tests/golden, and
    oracle/_ref directly when it is present).
"""
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))
TYPES = ["int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float32", "float64"]
OPS = ["ADD", "SUB", "MUL", "DIV"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.mark.parametrize("name", TYPES)
@pytest.mark.parametrize("opname", OPS)
def test_reference_test_vectors(oracle, name, opname):
    """Arithmetic::work folds the ports left to right (Arithmetic.cpp:217-224)."""
    op = getattr(oracle, opname)
    ins = []
    while "arith_%s_%s_in%d" % (opname, name, len(ins)) in GOLD.files:
        ins.append(GOLD["arith_%s_%s_in%d" % (opname, name, len(ins))])
    assert len(ins) == (3 if opname == "ADD" else 2)
    acc = ins[0]
    for a in ins[1:]:
        acc = oracle.arith(op, acc, a, False)
    assert np.array_equal(bits(acc), bits(GOLD["arith_%s_%s_exp" % (opname, name)]))
    acc = ins[0].reshape(50, 2)
    for a in ins[1:]:
        acc = oracle.arith(op, acc, a.reshape(50, 2), True)
    assert np.array_equal(bits(acc), bits(GOLD["arith_%s_c%s_exp" % (opname, name)]))


def test_add_closed_form():
    """TestArithmeticBlocks.cpp:66-73 states the ADD expectation in closed form."""
    for name in TYPES:
        f = GOLD["arith_ADD_formula_unsigned" if name.startswith("uint") else "arith_ADD_formula_signed"]
        assert np.array_equal(GOLD["arith_ADD_%s_exp" % name].astype(np.float64), f.astype(np.float64))


@pytest.mark.parametrize("name", TYPES)
@pytest.mark.parametrize("cplx", [0, 1])
def test_random_operands(oracle, name, cplx):
    a, b = GOLD["arith_rand_%s_%d_a" % (name, cplx)], GOLD["arith_rand_%s_%d_b" % (name, cplx)]
    for opname in ("ADD", "SUB", "MUL"):
        got = oracle.arith(getattr(oracle, opname), a, b, bool(cplx))
        assert np.array_equal(bits(got), bits(GOLD["arith_rand_%s_%d_%s" % (name, cplx, opname)])), opname
    ad, bd = GOLD["arith_rand_%s_%d_ad" % (name, cplx)], GOLD["arith_rand_%s_%d_bd" % (name, cplx)]
    got = oracle.arith(oracle.DIV, ad, bd, bool(cplx))
    assert np.array_equal(bits(got), bits(GOLD["arith_rand_%s_%d_DIV" % (name, cplx)]))


def test_against_cxx_operators(oracle):
    """Wider random sweep against std::complex / C++ operators when oracle/_ref is at hand."""
    if oracle.ref() is None or not hasattr(oracle.ref(), "ref_std_arith"):
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    for code, dt in oracle.NP_SCALAR.items():
        for cplx in (False, True):
            shape = (5000, 2) if cplx else (5000,)
            if np.issubdtype(dt, np.floating):
                a = (rng.standard_normal(shape) * 100).astype(dt)
                b = (rng.standard_normal(shape) * 10).astype(dt)
                ad, bd = a, b
            else:
                info = np.iinfo(dt)
                a = rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
                b = rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
                small = rng.integers(-9 if info.min < 0 else 1, 10, size=shape)
                bd = np.where(small == 0, 3, small).astype(dt)
                ad = (a // 4).astype(dt)
            for op in (oracle.ADD, oracle.SUB, oracle.MUL):
                assert np.array_equal(bits(oracle.arith(op, a, b, cplx)), bits(oracle.ref_arith(op, a, b, cplx))), (dt, cplx, op)
            assert np.array_equal(bits(oracle.arith(oracle.DIV, ad, bd, cplx)), bits(oracle.ref_arith(oracle.DIV, ad, bd, cplx))), (dt, cplx)


def test_division_corner_cases(oracle):
    """x/0 traps in the reference and INT_MIN/-1 is undefined: the restatement's stated results."""
    a = np.array([7, -7, -128, 5], np.int8)
    b = np.array([0, 2, -1, -1], np.int8)
    assert oracle.arith(oracle.DIV, a, b, False).tolist() == [0, -3, -128, -5]   # -128/-1 = 128 -> narrows to -128
    a = np.array([np.iinfo(np.int32).min, 9], np.int32)
    b = np.array([-1, 0], np.int32)
    assert oracle.arith(oracle.DIV, a, b, False).tolist() == [np.iinfo(np.int32).min, 0]


def test_unsupported(oracle):
    with pytest.raises(ValueError):
        oracle.arith(7, np.zeros(4, np.float32), np.zeros(4, np.float32), False)


@pytest.mark.parametrize("name", ["int8", "int16", "int32", "int64", "float32", "float64"])
def test_split_combine(oracle, name):
    """utility/TestComplex.cpp:13-60: combine then split returns both planes unchanged."""
    dt = np.dtype(name)
    rng = np.random.default_rng(3)
    re = (rng.standard_normal(1000) * 100).astype(dt)
    im = (rng.standard_normal(1000) * 100).astype(dt)
    z = oracle.combine_complex(re, im)
    assert z.shape == (1000, 2) and np.array_equal(z[:, 0], re) and np.array_equal(z[:, 1], im)
    r2, i2 = oracle.split_complex(z)
    assert np.array_equal(bits(r2), bits(re)) and np.array_equal(bits(i2), bits(im))
