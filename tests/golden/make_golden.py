#!/usr/bin/env python3
"""Generate tests/golden/golden.npz -- the committed golden vectors for the hot path.

Run in the BUILD container only (needs /root/reference through oracle/_ref):
    make -C oracle all _ref && python tests/golden/make_golden.py

Two kinds of vectors, all plain data (inputs + expected outputs):

 1. The known-answer vectors the reference's OWN tests hold for this path, transcribed as
    data with the expectation formula each test states:
      fft/TestFFT.cpp:14-29,64-80      4-point float KAT (+ inverse = input*N)
      fft/TestFFT.cpp:95-105,131-132   int16 KAT (forward = result/N), :155-156 inverse
      math/TestRotate.cpp:28-32,50-53  13 points (10i, -20i), phase in {0, pi/2, pi, 3pi/2}
      math/TestScale.cpp:28-31,49-52   13 points 10i, factor in {-1,-.5,0,.5,1}
      math/TestAbs.cpp:28-31,63-66     100 values i-50 (complex: re-typed as 50 pairs)
      math/TestConjugate.cpp:30-34     150 pairs (the test uses unseeded rand()%100; a
                                       seeded stand-in of the same range is stored)
      math/TestAngle.cpp:30-35,53-65   13 points mag*polar(1, i*pi/5) (pins getAngle, which
                                       FreqDemod shares)
      math/TestArithmeticBlocks.cpp:47-245  ADD/SUB/MUL/DIV vectors for all 20 element types
 2. Outputs of the compiled reference (oracle/_ref: kissfft.hh, kiss_fft.c -DFIXED_POINT=16,
    fxpt_atan2.cpp, FxptHelpers.hpp built from /root/reference where they lie) on seeded
    random inputs -- these pin the oracle bit-for-bit on the GPU box, where the reference
    sources do not exist.
 3. /comms/freq_demod: demod/FreqDemod.cpp itself needs <Pothos/Framework.hpp> and cannot be compiled here, but its loop
    (:60-67) is nothing but three operations the compiled reference DOES export --
        diff = in[i] * _prev      std::complex<T> operator*            (ref_std_arith, MUL, the toolchain's <complex>/libgcc)
        angle = getAngle(diff)    functions/FxptHelpers.hpp:14-29      (ref_angle_*, compiled from the reference header)
        _prev = std::conj(in[i])  sign flip of the imaginary part
    -- with _prev = 0 at activate() (:44-47).  `freqdemod_out_*` = ref_angle(ref_std_arith(MUL, x[i], conj(x[i-1]))), x[-1] = 0:
    the compiled pieces composed, no restatement of either in between.  The integer inputs cover the wrap of the product in
    complex<intN> followed by getAngle's truncation to int16.
 4. /comms/fir_filter, floating point: filter/FIRFilter.cpp needs <Pothos/Framework.hpp> too, but for float element types its loop
    (:294-300) is `y_n += _interpTaps[j][k] * QType(x[n-k])` in std::complex<float> / <double> -- the toolchain's operator* and
    operator+= again (fromQ / floatToQ are plain casts for floating-point Q types) -- k ascending, y_n starting at 0.  `fir_*_out` =
    that sum composed of ref_std_arith MUL and ADD, tap by tap (REAL taps: `T * complex<T>` scales both parts, two real
    multiplies), on the BASELINE tap sets (63 complex taps of configs[0], 255 of configs[1], 127 real taps of configs[4]) narrowed
    to the element type once, as :348 does.  Pins the ORDER and the operators of the oracle's FIR loop, and the device's EXACT kernel,
    bit for bit; the integer element types go through the un-vendored Q-format and stay with the parametrised restatement.
 5. /comms/rotate and /comms/scale, floating point: `phasor * in` and `factor * in` through the same compiled operators, the phasor
    from glibc's sincos as an optimised std::polar evaluates it (section 6 of main()).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz")
INT_TYPES = {"int8": np.int8, "int16": np.int16, "int32": np.int32, "int64": np.int64}
ALL_TYPES = dict(INT_TYPES, float32=np.float32, float64=np.float64)


def main():
    assert o.ref() is not None, "oracle/_ref is not built (needs /root/reference)"
    g = {}
    rng = np.random.default_rng(20240101)

    # ---- 1. reference test vectors ------------------------------------------------
    kat_in = np.array([[0.4, 0.6], [-0.7, 0.6], [-0.2, 0.8], [0.9, 0.2]])
    kat_out = np.array([[0.4, 2.2], [1.0, 1.4], [0.0, 0.6], [0.2, -1.8]])   # numpy.fft.fft of kat_in
    g["fft_kat_in"], g["fft_kat_out"] = kat_in, kat_out

    i13 = np.arange(13, dtype=np.float64)
    for name, dt in ALL_TYPES.items():
        # TestRotate: pIn[i] = (Type(10*i), Type(-20*i)) -- integer types wrap like the C++ cast
        xin = np.stack([10 * i13, -20 * i13], 1).astype(np.int64).astype(dt)
        g["rotate_in_" + name] = xin
        for pi_, phase in enumerate([0.0, np.pi / 2, np.pi, 3 * np.pi / 2]):
            z = (xin[:, 0].astype(np.float64) + 1j * xin[:, 1].astype(np.float64)) * np.exp(1j * phase)
            # expected = std::complex<Type>(input * polar(1, phase)): C++ double->integer casts truncate
            exp = np.stack([z.real, z.imag], 1)
            exp = np.trunc(exp) if name.startswith("int") else exp
            g["rotate_exp_%s_%d" % (name, pi_)] = exp   # float64, compared with tolerance 1 (TestRotate.cpp:53)
        sin = (10 * i13).astype(np.int64).astype(dt)
        g["scale_in_" + name] = sin
        for fi, factor in enumerate([-1.0, -0.5, 0.0, 0.5, 1.0]):
            e = sin.astype(np.float64) * factor
            g["scale_exp_%s_%d" % (name, fi)] = np.trunc(e) if name.startswith("int") else e
        # TestAbs: T(i) - T(50); expected = getAbs (the compiled reference)
        av = (np.arange(100) - 50).astype(dt)
        g["abs_in_" + name] = av
        g["abs_real_exp_" + name] = o.ref_abs(av, False)
        g["abs_cplx_exp_" + name] = o.ref_abs(av.reshape(50, 2), True)
        # TestAngle inputs (complex<Type>(mag*polar(1, angle))); expected std::arg of the typed input
        ang = i13 * (np.pi / 5)
        z = (i13 * 1000) * np.exp(1j * ang)
        zin = np.stack([z.real, z.imag], 1)
        zin = np.trunc(zin).astype(np.int64).astype(dt) if name.startswith("int") else zin.astype(dt)
        g["angle_in_" + name] = zin
        g["angle_ref_" + name] = o.ref_angle(zin)      # what getAngle returns (compiled reference)
    g["conj_in"] = rng.integers(0, 100, (150, 2)).astype(np.float32)

    # ---- 2. compiled-reference outputs on seeded inputs ----------------------------
    for n in (2, 3, 4, 5, 8, 9, 15, 16, 20, 64, 100, 210, 256, 1024, 4096):
        for inv in (0, 1):
            nf = 2 if n < 256 else 1
            x32 = rng.uniform(-1, 1, (nf * n, 2)).astype(np.float32)
            x64 = rng.uniform(-1, 1, (nf * n, 2))
            x16 = rng.integers(-32768, 32768, (nf * n, 2)).astype(np.int16)
            g["fft_f32_in_%d_%d" % (n, inv)] = x32
            g["fft_f32_out_%d_%d" % (n, inv)] = o.ref_fft(x32, n, bool(inv))
            if n <= 256:
                g["fft_f64_in_%d_%d" % (n, inv)] = x64
                g["fft_f64_out_%d_%d" % (n, inv)] = o.ref_fft(x64, n, bool(inv))
            g["fft_i16_in_%d_%d" % (n, inv)] = x16
            g["fft_i16_out_%d_%d" % (n, inv)] = o.ref_fft(x16, n, bool(inv))
    yx = rng.integers(-32768, 32768, (2048, 2)).astype(np.int16)
    yx[:64] = rng.integers(-3, 4, (64, 2))
    yx[64:72] = [[0, 0], [1, 1], [-1, -1], [-32768, -32768], [32767, 32767], [-32768, 32767], [0, -32768], [-32768, 0]]
    R = o.ref()
    g["atan2_in"] = yx
    g["atan2_out"] = np.array([R.ref_fxpt_atan2(int(y), int(x)) for y, x in yx], dtype=np.uint16)
    for name, dt in ALL_TYPES.items():
        if name.startswith("int"):
            info = np.iinfo(dt)
            z = rng.integers(info.min, info.max + 1, (512, 2), dtype=dt)
            z[:64] = rng.integers(-100, 100, (64, 2))
        else:
            z = (rng.normal(size=(512, 2)) * np.exp(rng.uniform(-20, 20, (512, 1)))).astype(dt)
        g["rand_in_" + name] = z
        g["rand_angle_" + name] = o.ref_angle(z)
        g["rand_abs_cplx_" + name] = o.ref_abs(z, True)
        g["rand_abs_real_" + name] = o.ref_abs(np.ascontiguousarray(z[:, 0]), False)

    # ---- 3. /comms/arithmetic ---------------------------------------
    # math/TestArithmeticBlocks.cpp:47-245: 100-element closed-form vectors for all 20 element types
    # (the complex cases re-type the same bytes as 50 pairs, :80-91,116-127, and recompute MUL/DIV
    # expectations with the std::complex operators, :146-151,201-206 -- here through ref_std_arith).
    arith_types = dict(ALL_TYPES, uint8=np.uint8, uint16=np.uint16, uint32=np.uint32, uint64=np.uint64)
    e = np.arange(100, dtype=np.int64)
    for name, dt in arith_types.items():
        signed = not name.startswith("uint")
        sgn = -1 if signed else 1
        cast = lambda v: np.asarray(v, dtype=np.int64).astype(dt)   # static_cast<T>(size_t expression)
        ins = {
            "ADD": [cast(e), cast(e // 2) * dt(sgn), cast(e // 4) * dt(sgn)],
            "SUB": [cast(e), cast(e * 2) if signed else cast(e // 2)],
            "MUL": [cast(e), cast((e % 2) + 1) * dt(sgn)],
            "DIV": [cast(e), cast((e % 2) + 1) * dt(sgn)],
        }
        for opname, arrs in ins.items():
            arrs = [np.ascontiguousarray(a.astype(dt)) for a in arrs]
            for k, a in enumerate(arrs):
                g["arith_%s_%s_in%d" % (opname, name, k)] = a
            op = {"ADD": o.ADD, "SUB": o.SUB, "MUL": o.MUL, "DIV": o.DIV}[opname]
            # real: the expectation formulas of the test, evaluated by the C++ operators (ref_std_arith)
            acc = arrs[0]
            for a in arrs[1:]:
                acc = o.ref_arith(op, acc, a, False)
            g["arith_%s_%s_exp" % (opname, name)] = acc
            acc = arrs[0].reshape(50, 2)
            for a in arrs[1:]:
                acc = o.ref_arith(op, acc, a.reshape(50, 2), True)
            g["arith_%s_c%s_exp" % (opname, name)] = acc
    # closed forms the test states for the real ADD case (:66-73), as an independent check of the above
    g["arith_ADD_formula_signed"] = e - e // 2 - e // 4
    g["arith_ADD_formula_unsigned"] = e + e // 2 + e // 4
    # std::complex / C++ operators on seeded random operands: pins oracle/pcx_oracle.c's restatement
    # where libstdc++/libgcc_s are not at hand
    rng3 = np.random.default_rng(20240303)
    for name, dt in arith_types.items():
        for cplx in (0, 1):
            shape = (192, 2) if cplx else (192,)
            if name.startswith("float"):
                a = (rng3.standard_normal(shape) * 100).astype(dt)
                b = (rng3.standard_normal(shape) * 10).astype(dt)
                bd = b
            else:
                info = np.iinfo(dt)
                a = rng3.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
                b = rng3.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
                # divisors: small and non-zero (x/0 traps, MIN/-1 is undefined in the reference)
                small = rng3.integers(-9 if info.min < 0 else 1, 10, size=shape)
                bd = np.where(small == 0, 3, small).astype(dt)
            g["arith_rand_%s_%d_a" % (name, cplx)] = a
            g["arith_rand_%s_%d_b" % (name, cplx)] = b
            g["arith_rand_%s_%d_bd" % (name, cplx)] = bd
            ad = a if name.startswith("float") else (a // 4).astype(dt)   # keeps MIN/-1 out of the quotients
            g["arith_rand_%s_%d_ad" % (name, cplx)] = ad
            for opname, op in (("ADD", o.ADD), ("SUB", o.SUB), ("MUL", o.MUL)):
                g["arith_rand_%s_%d_%s" % (name, cplx, opname)] = o.ref_arith(op, a, b, bool(cplx))
            g["arith_rand_%s_%d_DIV" % (name, cplx)] = o.ref_arith(o.DIV, ad, bd, bool(cplx))

    # ---- 4. /comms/freq_demod = compiled std::complex multiply + compiled getAngle (docstring, 3.) -----------------
    rng4 = np.random.default_rng(20240404)
    for name, dt in ALL_TYPES.items():
        if name.startswith("int"):
            info = np.iinfo(dt)
            x = rng4.integers(info.min, info.max + 1, (3072, 2), dtype=dt)           # products wrap in complex<intN>
            x[:512] = rng4.integers(-90, 91, (512, 2))                                 # products that do not
            x[512:520] = [[info.min, info.min], [info.max, info.max], [info.min, info.max], [0, info.min], [info.min, 0], [0, 0], [1, 0], [0, 0]]
            amp = min(info.max, 20000)
            ph = np.cumsum(2 * np.pi * (0.02 + 0.01 * np.sin(2 * np.pi * np.arange(1024) / 100)))
            x[1024:2048] = np.trunc(np.stack([amp * np.cos(ph), amp * np.sin(ph)], 1)).astype(np.int64).astype(dt)   # an FM signal
        else:
            # SURVEY 8d's C4 signal (FM, small noise), then random samples over forty orders of magnitude, then zeros, signed zero and axis points
            n = np.arange(2048)
            ph = np.cumsum(2 * np.pi * (0.02 + 0.01 * np.sin(2 * np.pi * n / 1000)))
            fm = np.stack([np.cos(ph), np.sin(ph)], 1) + 1e-3 * rng4.uniform(-1, 1, (2048, 2))
            wide = rng4.normal(size=(1016, 2)) * np.exp(rng4.uniform(-20, 20, (1016, 1)))
            corner = np.array([[0, 0], [0, 0], [1, 0], [0, -1], [-1, 0], [-0.0, 0.0], [1e-30, -1e-30], [3, 4]], np.float64)
            x = np.concatenate([fm, wide, corner]).astype(dt)
        prev = np.concatenate([np.zeros((1, 2), dt), x[:-1]])                          # _prev = 0 at activate(), FreqDemod.cpp:44-47
        with np.errstate(over="ignore"):
            prev[:, 1] = (-prev[:, 1].astype(np.int64)).astype(dt) if name.startswith("int") else -prev[:, 1]   # std::conj; -MIN wraps to MIN
        diff = o.ref_arith(o.MUL, np.ascontiguousarray(x), np.ascontiguousarray(prev), True)
        g["freqdemod_in_" + name] = x
        g["freqdemod_out_" + name] = o.ref_angle(diff)

    # ---- 5. /comms/fir_filter (float types) = compiled std::complex multiply-accumulate in the loop's order (docstring, 4.) -----
    from pothoscomms_amd import taps as tp            # tap recipes only (numpy): SURVEY 8d's windowed-sinc sets
    rng5 = np.random.default_rng(20240505)
    # Lengths: every float32 fixture runs across TWO seams of the frequency-domain kernel's overlap-save blocks (payload S = 4096 minus the
    # overlap rounded up to 16: 4032 at 63 taps, 3840 at 255, 3968 at 127 real taps, 2048 at 2049 taps -- the smallest payload of the
    # 4096-sample plan), so that tests/test_golden_gpu.py compares that kernel with reference-operator outputs ACROSS block boundaries,
    # no oracle in between (Weak 1).
    for key, taps, ctaps, dt, nout in (("c0_63c_f32", tp.c0_taps(), True, np.float32, 8192), ("c1_255c_f32", tp.c1_taps(), True, np.float32, 8192),
                                       ("c4_127r_f32", tp.c4_taps(), False, np.float32, 8192),
                                       ("2049c_f32", tp.complex_bandpass(2049, 0.02, 0.03), True, np.float32, 6144),
                                       ("31c_f64", tp.c0_taps()[:31], True, np.float64, 1024)):
        K = len(taps)
        x = rng5.uniform(-1, 1, (K - 1 + nout, 2)).astype(dt)
        x[5] = [0.0, -0.0]
        t = np.asarray(taps)
        tq = np.stack([t.real, t.imag], 1).astype(dt) if ctaps else np.real(t).astype(dt)       # floatToQ<QTapsType>: one narrowing cast (:348)
        acc = np.zeros((nout, 2), dt)                                                           # QType y_n = 0
        for k in range(K):
            xs = np.ascontiguousarray(x[K - 1 - k:K - 1 - k + nout])                            # x[n - k], n = 0 .. nout-1 (x = in + K-1, :281)
            if ctaps:
                prod = o.ref_arith(o.MUL, np.ascontiguousarray(np.broadcast_to(tq[k], (nout, 2))), xs, True)
            else:                                                                               # T * complex<T>: both parts scaled
                prod = o.ref_arith(o.MUL, np.full(2 * nout, tq[k], dt), xs.reshape(-1), False).reshape(nout, 2)
            acc = o.ref_arith(o.ADD, acc, prod, True)
        g["fir_%s_in" % key], g["fir_%s_taps" % key], g["fir_%s_out" % key] = x, np.asarray(taps, np.complex128 if ctaps else np.float64), acc

    # ---- 6. /comms/rotate, /comms/scale (float types) = the same compiled operators (math/Rotate.cpp:15-23,71-75, math/Scale.cpp:15-23) ----
    # arrayRotate: out = phasor * QType(in), phasor = complex<T>(std::polar(1.0, phase)); arrayScale: out = factor * QType(in) (T * complex<T>
    # scales both parts).  std::polar in an optimised build is one glibc sincos() call: the phasor stored here comes from that call.
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    phases = np.array([0.3, 2.747554270528532, -1.1, np.pi / 2])
    factors = np.array([0.3337, -1.77, 100.25, 0.5])
    ph = []
    for p in phases:
        sn, cs = ctypes.c_double(), ctypes.c_double()
        libm.sincos(float(p), ctypes.byref(sn), ctypes.byref(cs))
        ph.append([cs.value, sn.value])
    g["rotscale_phases"], g["rotscale_phasors"], g["rotscale_factors"] = phases, np.array(ph), factors
    rng6 = np.random.default_rng(20240606)
    for name, dt in (("float32", np.float32), ("float64", np.float64)):
        x = (rng6.normal(size=(512, 2)) * np.exp(rng6.uniform(-10, 10, (512, 1)))).astype(dt)
        g["rotscale_in_" + name] = x
        for k in range(len(phases)):
            pz = np.ascontiguousarray(np.broadcast_to(np.array(ph[k]).astype(dt), x.shape))      # floatToQ<complex<T>>: narrowing cast
            g["rotate_out_%s_%d" % (name, k)] = o.ref_arith(o.MUL, pz, x, True)
            f = np.full(x.size, dt(factors[k]), dt)
            g["scale_out_%s_%d" % (name, k)] = o.ref_arith(o.MUL, f, x.reshape(-1), False).reshape(x.shape)

    np.savez_compressed(OUT, **g)
    print("wrote %s: %d arrays, %d bytes" % (OUT, len(g), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
