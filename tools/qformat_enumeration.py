"""Which semantics of Pothos::Util::floatToQ / fromQ survive the evidence inside the reference tree?

PothosCore's include/Pothos/Util/QFormat.hpp is not under /root/reference.  What the tree does pin:

  call sites   floatToQ<QType>(x) and fromQ<Type>(q) are called WITHOUT a shift argument (math/Rotate.cpp:21,74,
               math/Scale.cpp:21,73, filter/FIRFilter.cpp:300,348): the fractional bit count is a default that depends on
               a type only; utility/SignalProbe.cpp:141,148,157 calls fromQ<ProbeType>(x, 0) with ProbeType = double or
               complex<double>: a second parameter exists, it is the fractional bit count, and 0 means "plain conversion";
  Q types      int8 -> int16, int16 -> int32, int32 -> int64, int64 -> int64 (rotateFactory Rotate.cpp:143-157,
               scaleFactory Scale.cpp:142-160, FIRFilterFactory FIRFilter.cpp:377-382);
  arithmetic   tmp = coefficient * QType(in[i]) in the Q type (wrapping), out = fromQ<Type>(tmp) (Rotate.cpp:15-23,
               Scale.cpp:15-23);
  tests        math/TestRotate.cpp:28-53 (13 points (10 i, -20 i), phases k pi/2, |got - Type(in * polar(1, phase))| <= 1)
               and math/TestScale.cpp:28-52 (13 points 10 i, factors -1 .. 1 step 0.5, |got - Type(in * factor)| <= 1).

This script runs every candidate
    n_to   = fractional bits floatToQ gives its result      in {4*sizeof(Q), 4*sizeof(T), 8*sizeof(Q)-1, 8*sizeof(Q)-2}
    n_from = fractional bits fromQ removes                   in the same set
    floatToQ rounding  in {truncate (cast of ldexp), nearest}
    fromQ rounding     in {arithmetic shift (floor), division toward zero, nearest}
through both tests with the reference's own integer arithmetic and prints the survivors.

Run:  python tools/qformat_enumeration.py   (CPU only)
"""
import itertools
import math

TYPES = {"int8": (8, 16), "int16": (16, 32), "int32": (32, 64), "int64": (64, 64)}   # element bits, Q bits
NUM_POINTS = 13


def wrap(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def c_cast(x, bits):
    """C++ static_cast<intN>(double): truncation toward zero; out-of-range is undefined -- x86 gives INT_MIN for 32/64-bit
    targets, and for narrower targets the int32 conversion result truncated.  Flag it so candidates relying on it are visible."""
    t = int(x)   # toward zero
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    if t < lo or t > hi:
        return wrap(t, bits), True
    return t, False


def n_rule(rule, ebits, qbits):
    return {"Q/2": qbits // 2, "T/2": ebits // 2, "Q-1": qbits - 1, "Q-2": qbits - 2}[rule]


def float_to_q(x, n, qbits, rounding):
    v = math.ldexp(x, n)
    if rounding == "nearest":
        v = math.floor(v + 0.5)
    return c_cast(v, qbits)


def from_q(q, n, ebits, rounding):
    if rounding == "floor":
        r = q >> n
    elif rounding == "zero":
        r = int(q / (1 << n)) if abs(q) < (1 << 52) else (abs(q) >> n) * (1 if q >= 0 else -1)
    else:
        r = (q + (1 << (n - 1))) >> n if n > 0 else q
    return wrap(r, ebits)


def rotate_ok(name, cand):
    ebits, qbits = TYPES[name]
    nt, nf = n_rule(cand[0], ebits, qbits), n_rule(cand[1], ebits, qbits)
    for k in range(4):
        phase = k * math.pi / 2
        pr, pi = math.cos(phase), math.sin(phase)          # std::polar(1.0, phase)
        (qr, o1), (qi, o2) = float_to_q(pr, nt, qbits, cand[2]), float_to_q(pi, nt, qbits, cand[2])
        if o1 or o2:
            return False
        for i in range(NUM_POINTS):
            re, im = wrap(10 * i, ebits), wrap(-20 * i, ebits)
            # std::complex<QType> product, wrapping in the Q type
            tr = wrap(qr * re - qi * im, qbits)
            ti = wrap(qr * im + qi * re, qbits)
            gr, gi = from_q(tr, nf, ebits, cand[3]), from_q(ti, nf, ebits, cand[3])
            er, _ = c_cast(re * pr - im * pi, ebits)
            ei, _ = c_cast(re * pi + im * pr, ebits)
            if math.hypot(gr - er, gi - ei) > 1:             # POTHOS_TEST_CLOSE on complex: |a - b| <= 1
                return False
    return True


def scale_ok(name, cand):
    ebits, qbits = TYPES[name]
    nt, nf = n_rule(cand[0], ebits, qbits), n_rule(cand[1], ebits, qbits)
    for k in range(5):
        factor = k / 2.0 - 1.0
        q, over = float_to_q(factor, nt, qbits, cand[2])
        if over:
            return False
        for i in range(NUM_POINTS):
            x = wrap(10 * i, ebits)
            g = from_q(wrap(q * x, qbits), nf, ebits, cand[3])
            e, _ = c_cast(x * factor, ebits)
            if abs(g - e) > 1:
                return False
    return True


def main():
    rules = ["Q/2", "T/2", "Q-1", "Q-2"]
    cands = list(itertools.product(rules, rules, ["truncate", "nearest"], ["floor", "zero", "nearest"]))
    print("%d candidates (n_to, n_from, floatToQ rounding, fromQ rounding); survivors of TestRotate + TestScale on all four integer types:" % len(cands))
    surv = []
    for c in cands:
        fails = [t for t in TYPES if not (rotate_ok(t, c) and scale_ok(t, c))]
        if not fails:
            surv.append(c)
    for c in surv:
        print("  n_to=%-4s n_from=%-4s floatToQ=%-9s fromQ=%-8s" % c)
    print("%d survive.  Every survivor has n_to = n_from (a coefficient of 1.0 must come back as the input)." % len(surv))
    by_rule = sorted(set(c[0] for c in surv))
    print("fractional-bit rules that survive: %s" % ", ".join(by_rule))
    print("Rounding is NOT pinned: the test coefficients (0, +-0.5, +-1) times the test inputs (multiples of 10) are exact "
          "in every surviving format, so truncation, floor and nearest all return the same integers.")
    ours = ("Q/2", "Q/2", "truncate", "floor")
    print("the restatement in oracle/pcx_oracle.c and the device kernels (ldexp + cast, arithmetic >> of half the Q word): %s"
          % ("survives" if ours in surv else "DOES NOT SURVIVE"))


if __name__ == "__main__":
    main()
