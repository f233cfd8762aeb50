/*
 * pcx_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the arithmetic and the work() host logic of the
 * PothosComms streaming-DSP hot path (FIRFilter, FFT/kiss_fft, FreqDemod,
 * Rotate, Scale, Abs, Conjugate).  Every function cites the reference
 * file:line it follows (paths relative to the reference checkout).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 * The product (pothoscomms_amd/, libpcx_hip.so) never links or calls it.
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   - FFT float/double/int16, fxpt_atan2, getAngle, getAbs: bit-exact against
 *     the reference sources compiled as oracle/_ref (kissfft.hh, kiss_fft.c,
 *     fxpt_atan2.cpp, FxptHelpers.hpp need no Pothos headers) and against the
 *     known-answer vectors in fft/TestFFT.cpp.
 *   - Rotate/Scale/Abs/Conjugate: the vectors of math/Test{Rotate,Scale,Abs,
 *     Conjugate}.cpp (tests/golden/).
 *   - FIR / FreqDemod: the reference tests hold no numeric vectors for them
 *     (RMS threshold only / no test); the restatement is pinned by an
 *     independent float64 convolution and the survey's observed anchors, and
 *     -- since their loops are nothing but std::complex operator*, operator+=
 *     and getAngle -- by fixtures composed of exactly those pieces as the
 *     COMPILED reference toolchain evaluates them (oracle/_ref ref_std_arith,
 *     ref_angle_*; tests/golden/make_golden.py sections 4 and 5): bit for bit,
 *     float FIR on the BASELINE tap sets and FreqDemod for all six types.
 *   - Integer Q-format (fromQ/floatToQ live in PothosCore, which is not in
 *     the reference tree): restated as "half of the Q word is fractional,
 *     floatToQ = T(ldexp(x, n)), fromQ = T(q >> n)".  PARITY UNPINNED: of the
 *     96 candidate semantics run through math/TestRotate.cpp and
 *     math/TestScale.cpp with the reference's own integer arithmetic, 12
 *     survive (tools/qformat_enumeration.py) -- the fractional
 *     bit count is half the Q word or half the element word, the same in both
 *     directions; rounding is not pinned at all.  This restatement is one of
 *     the 12 by default and takes ANY of the 12 as a parameter (orc_set_qformat,
 *     mirroring pcx_qformat of include/pcx.h): integer FIR/Rotate/Scale are
 *     bit-exact against it under every reading, not against the (absent) header.
 *   - The pinning tests run on BOTH boxes (tests/test_oracle_cpu.py under
 *     -m "not gpu" and under -m gpu): this library binds the glibc / libgcc_s
 *     of the machine it is loaded on.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math -shared -fPIC (see Makefile)
 * -ffp-contract=off keeps every product and sum separately rounded, as the
 * reference's baseline x86-64 build (no FMA) does.
 */
#define _GNU_SOURCE   /* sincos */
#include <complex.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

/* scalar type codes -- same numbering as include/pcx.h */
enum { ORC_F64 = 0, ORC_F32 = 1, ORC_I64 = 2, ORC_I32 = 3, ORC_I16 = 4, ORC_I8 = 5,
       ORC_U64 = 6, ORC_U32 = 7, ORC_U16 = 8, ORC_U8 = 9 };   /* unsigned: /comms/arithmetic only */

static int scalar_bytes(int st)
{
    switch (st) {
    case ORC_F64: case ORC_I64: case ORC_U64: return 8;
    case ORC_F32: case ORC_I32: case ORC_U32: return 4;
    case ORC_I16: case ORC_U16: return 2;
    case ORC_I8: case ORC_U8: return 1;
    }
    return 0;
}
static int is_float_type(int st) { return st == ORC_F64 || st == ORC_F32; }

/* bits of the Q (accumulator) scalar for an integer element type:
 * FIRFilter.cpp:377-382, Rotate.cpp:151-154, Scale.cpp:150-153
 * int64->int64, int32->int64, int16->int32, int8->int16 */
static int q_bits(int st)
{
    switch (st) {
    case ORC_I64: case ORC_I32: return 64;
    case ORC_I16: return 32;
    case ORC_I8: return 16;
    }
    return 0;
}

static int64_t load_int(const void *p, size_t idx, int st)
{
    switch (st) {
    case ORC_I64: return ((const int64_t *)p)[idx];
    case ORC_I32: return ((const int32_t *)p)[idx];
    case ORC_I16: return ((const int16_t *)p)[idx];
    default: return ((const int8_t *)p)[idx];
    }
}
static void store_int(void *p, size_t idx, int st, int64_t v)
{
    switch (st) {
    case ORC_I64: ((int64_t *)p)[idx] = v; break;
    case ORC_I32: ((int32_t *)p)[idx] = (int32_t)(uint32_t)(uint64_t)v; break;
    case ORC_I16: ((int16_t *)p)[idx] = (int16_t)(uint16_t)(uint64_t)v; break;
    default: ((int8_t *)p)[idx] = (int8_t)(uint8_t)(uint64_t)v; break;
    }
}
/* wrap an exact (mod 2^64) value to a signed `bits`-wide integer */
static int64_t wrap_bits(uint64_t v, int bits)
{
    if (bits >= 64) return (int64_t)v;
    const uint64_t m = (1ull << bits) - 1;
    v &= m;
    if (v >> (bits - 1)) v |= ~m;
    return (int64_t)v;
}
/* The Q-format READING in force (include/pcx.h, pcx_qformat): QFormat.hpp is not in the reference tree and the reference's
 * tests leave twelve readings standing, so the restatement takes the reading as a parameter exactly as the product does --
 * fractional bits n = half the Q word (0) or half the ELEMENT word (1); floatToQ truncating (0) or to nearest, ties away (1);
 * fromQ floor (0), toward zero (1) or to nearest, ties up (2).  All zero = the default.  Process-wide (this is a test library):
 * set it BEFORE creating a filter or setting its taps -- they are quantised when they are set. */
static int g_q_frac = 0, g_q_to = 0, g_q_from = 0;
ORC_EXPORT void orc_set_qformat(int frac, int float_to_q_mode, int from_q_mode)
{
    g_q_frac = frac; g_q_to = float_to_q_mode; g_q_from = from_q_mode;
}
static int elem_bits(int st) { return st == ORC_I64 ? 64 : st == ORC_I32 ? 32 : st == ORC_I16 ? 16 : 8; }
static int q_frac_bits(int st) { return g_q_frac ? elem_bits(st) / 2 : q_bits(st) / 2; }
/* Pothos::Util::floatToQ<T>(x), integer T of q_bits(st) bits: T(std::ldexp(x, n))
 * (PothosCore QFormat.hpp; call sites FIRFilter.cpp:348, Rotate.cpp:74, Scale.cpp:73) */
static int64_t float_to_q(double x, int st)
{
    const int qbits = q_bits(st);
    double v = ldexp(x, q_frac_bits(st));
    if (g_q_to) v = round(v);
    /* double -> integer conversion as x86-64 cvttsd2si does (out of range -> MIN) */
    if (qbits == 64) {
        if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return INT64_MIN;
        return (int64_t)v;
    }
    if (qbits == 32) {
        if (!(v >= -2147483648.0 && v < 2147483648.0)) return INT32_MIN;
        return (int32_t)v;
    }
    /* int16: conversion goes through int32 then truncates */
    if (!(v >= -2147483648.0 && v < 2147483648.0)) return 0;
    return (int16_t)(uint16_t)(uint32_t)(int32_t)v;
}
/* Pothos::Util::fromQ<T>(q), q already wrapped to the Q width: the quotient by 2^n under the reading's rounding, in 128-bit
 * arithmetic so that no intermediate can overflow (the device does the same with shifts and masks: an independent formulation) */
static int64_t from_q(int64_t q, int st)
{
    const int n = q_frac_bits(st);
    const __int128 d = (__int128)1 << n, v = q;
    if (g_q_from == 1) return (int64_t)(v / d);                             /* toward zero: C division */
    const __int128 num = g_q_from == 2 ? v + (d >> 1) : v;                  /* nearest: add half first */
    __int128 fl = num / d;
    if (num % d != 0 && num < 0) fl -= 1;                                   /* floor */
    return (int64_t)fl;
}

/* ===================================================================== *
 *  FIR  (filter/FIRFilter.cpp)
 * ===================================================================== */
typedef struct {
    char id;        /* 'S' = frame-start id match, 'E' = frame-end id match, other = no match */
    uint64_t index; /* Label::index */
    uint64_t width; /* Label::width */
    uint64_t length;/* label.data converted to size_t (frame start only) */
    int has_length; /* label.data.canConvert(size_t) */
} orc_label;

typedef struct orc_fir {
    int st;           /* scalar type */
    int cplx;         /* element type complex? */
    int ctaps;        /* taps complex? ("COMPLEX") */
    size_t ntaps;
    double *taps;     /* ntaps (real) or 2*ntaps (complex, interleaved) */
    size_t M, L, K, inputRequire;
    /* polyphase split, FIRFilter.cpp:340-350: row j holds taps[j + k*L] */
    size_t *rowLen;   /* L entries */
    double *rowTapsF; /* L*K*(ctaps?2:1) floating taps narrowed to Q precision (stored as double) */
    float *rowTapsF32;/* the same rows as float: what floatToQ<complex<float>> leaves in _interpTaps (FIRFilter.cpp:348),
                         so the float loop reads its taps at their own width like the reference does */
    int64_t *rowTapsQ;/* same for integer Q */
    int waitTapsMode, waitTapsArmed;
    int haveStartId, haveEndId;
    size_t eobSampsLeft;
} orc_fir;

/* FIRFilter::updateInternals, FIRFilter.cpp:327-354 */
static void fir_update_internals(orc_fir *f)
{
    const size_t L = f->L, n = f->ntaps;
    f->K = n / L + (((n % L) == 0) ? 0 : 1);
    const size_t K = f->K, w = f->ctaps ? 2 : 1;
    free(f->rowLen); free(f->rowTapsF); free(f->rowTapsF32); free(f->rowTapsQ);
    f->rowLen = (size_t *)calloc(L, sizeof(size_t));
    f->rowTapsF = (double *)calloc(L * K * w, sizeof(double));
    f->rowTapsF32 = (float *)calloc(L * K * w, sizeof(float));
    f->rowTapsQ = (int64_t *)calloc(L * K * w, sizeof(int64_t));
    for (size_t j = 0; j < L; j++) {
        size_t len = 0;
        for (size_t k = 0; k < K; k++) {
            const size_t i = j + k * L;
            if (i >= n) continue;
            for (size_t c = 0; c < w; c++) {
                const double t = f->taps[i * w + c];
                /* floatToQ<QTapsType>: float Q -> plain narrowing cast */
                f->rowTapsF[(j * K + len) * w + c] = (f->st == ORC_F32) ? (double)(float)t : t;
                f->rowTapsF32[(j * K + len) * w + c] = (float)t;
                if (!is_float_type(f->st)) f->rowTapsQ[(j * K + len) * w + c] = float_to_q(t, f->st);
            }
            len++;
        }
        f->rowLen[j] = len;
    }
    f->inputRequire = f->M + (K - 1);
}

ORC_EXPORT orc_fir *orc_fir_create(int scalar_type, int is_complex, int complex_taps)
{
    /* factory matrix FIRFilter.cpp:371-382: REAL taps on real or complex data,
     * COMPLEX taps only on complex data */
    if (scalar_type < 0 || scalar_type > ORC_I8) return NULL;
    if (complex_taps && !is_complex) return NULL;
    orc_fir *f = (orc_fir *)calloc(1, sizeof(orc_fir));
    f->st = scalar_type; f->cplx = is_complex; f->ctaps = complex_taps;
    f->M = f->L = 1;
    /* ctor: setTaps({1}) FIRFilter.cpp:125 */
    f->ntaps = 1;
    f->taps = (double *)calloc(2, sizeof(double));
    f->taps[0] = 1.0;
    fir_update_internals(f);
    return f;
}
ORC_EXPORT void orc_fir_destroy(orc_fir *f)
{
    if (!f) return;
    free(f->taps); free(f->rowLen); free(f->rowTapsF); free(f->rowTapsF32); free(f->rowTapsQ); free(f);
}
/* FIRFilter::setTaps FIRFilter.cpp:138-144 */
ORC_EXPORT int orc_fir_set_taps(orc_fir *f, const double *taps, size_t ntaps)
{
    if (ntaps == 0) return -1; /* InvalidArgumentException "taps cannot be empty" */
    const size_t w = f->ctaps ? 2 : 1;
    free(f->taps);
    f->taps = (double *)malloc(ntaps * w * sizeof(double));
    memcpy(f->taps, taps, ntaps * w * sizeof(double));
    f->ntaps = ntaps;
    f->waitTapsArmed = 0;
    fir_update_internals(f);
    return 0;
}
/* FIRFilter.cpp:151-168 */
ORC_EXPORT int orc_fir_set_decimation(orc_fir *f, size_t m) { if (!m) return -1; f->M = m; fir_update_internals(f); return 0; }
ORC_EXPORT int orc_fir_set_interpolation(orc_fir *f, size_t l) { if (!l) return -1; f->L = l; fir_update_internals(f); return 0; }
ORC_EXPORT void orc_fir_set_wait_taps(orc_fir *f, int w) { f->waitTapsMode = w; }
ORC_EXPORT void orc_fir_set_frame_ids(orc_fir *f, int have_start, int have_end) { f->haveStartId = have_start; f->haveEndId = have_end; }
/* FIRFilter::activate FIRFilter.cpp:201-205 */
ORC_EXPORT void orc_fir_activate(orc_fir *f) { f->waitTapsArmed = f->waitTapsMode; f->eobSampsLeft = 0; }
ORC_EXPORT size_t orc_fir_K(const orc_fir *f) { return f->K; }
ORC_EXPORT size_t orc_fir_input_require(const orc_fir *f) { return f->inputRequire; }

/* std::complex<float/double> operator* as GCC compiles it (FIRFilter.cpp:298, Rotate.cpp:20): the plain formula, every product
 * and sum rounded separately, and -- only when BOTH parts come out NaN -- libgcc's __mulsc3 / __muldc3, which recovers the
 * infinities of C99 Annex G.  The libgcc functions are called directly so that the slow path is the reference's own. */
extern float _Complex __mulsc3(float a, float b, float c, float d);
extern double _Complex __muldc3(double a, double b, double c, double d);
static inline void orc_cmul_f32(float a, float b, float c, float d, float *x, float *y)
{
    const float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    *x = ac - bd; *y = ad + bc;
    if (*x != *x && *y != *y) { const float _Complex z = __mulsc3(a, b, c, d); *x = __real__ z; *y = __imag__ z; }
}
static inline void orc_cmul_f64(double a, double b, double c, double d, double *x, double *y)
{
    const double ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    *x = ac - bd; *y = ad + bc;
    if (*x != *x && *y != *y) { const double _Complex z = __muldc3(a, b, c, d); *x = __real__ z; *y = __imag__ z; }
}

/* the convolution loop FIRFilter.cpp:286-302, floating point element types.
 * Accumulation is sequential in k in the Q type (== element type for floats);
 * complex*complex is orc_cmul_* above: (ac-bd, ad+bc), every product and sum rounded
 * separately, with libgcc's slow path when both parts are NaN. */
#define ORC_FIR_FLOAT(NAME, T, ROWS, CMUL)                                                             \
    static size_t NAME(const orc_fir *f, const T *x, T *y, size_t N)                            \
    {                                                                                           \
        const size_t L = f->L, M = f->M, K = f->K;                                              \
        size_t decim = M, nout = 0;                                                             \
        for (size_t n = 0; n < N; n++) {                                                        \
            for (size_t j = 0; j < L; j++) {                                                    \
                if (--decim != 0) continue;                                                     \
                decim = M;                                                                      \
                const T *tp = f->ROWS + j * K * (f->ctaps ? 2 : 1);                             \
                const size_t len = f->rowLen[j];                                                \
                if (!f->cplx) {                                                                 \
                    T acc = 0;                                                                  \
                    for (size_t k = 0; k < len; k++) {                                          \
                        const T p = tp[k] * x[(ptrdiff_t)n - (ptrdiff_t)k];                     \
                        acc = acc + p;                                                          \
                    }                                                                           \
                    y[nout++] = acc;                                                            \
                } else if (!f->ctaps) {                                                         \
                    T ar = 0, ai = 0;                                                           \
                    for (size_t k = 0; k < len; k++) {                                          \
                        const T *xp = x + 2 * ((ptrdiff_t)n - (ptrdiff_t)k);                    \
                        const T h = tp[k];                                                      \
                        const T pr = xp[0] * h, pi = xp[1] * h;                                 \
                        ar = ar + pr; ai = ai + pi;                                             \
                    }                                                                           \
                    y[2 * nout] = ar; y[2 * nout + 1] = ai; nout++;                             \
                } else {                                                                        \
                    T ar = 0, ai = 0;                                                           \
                    for (size_t k = 0; k < len; k++) {                                          \
                        const T *xp = x + 2 * ((ptrdiff_t)n - (ptrdiff_t)k);                    \
                        T pr, pi;                                                               \
                        CMUL(tp[2 * k], tp[2 * k + 1], xp[0], xp[1], &pr, &pi);                 \
                        ar = ar + pr; ai = ai + pi;                                             \
                    }                                                                           \
                    y[2 * nout] = ar; y[2 * nout + 1] = ai; nout++;                             \
                }                                                                               \
            }                                                                                   \
        }                                                                                       \
        return nout;                                                                            \
    }
ORC_FIR_FLOAT(fir_loop_f32, float, rowTapsF32, orc_cmul_f32)
ORC_FIR_FLOAT(fir_loop_f64, double, rowTapsF, orc_cmul_f64)

/* same loop for the integer element types: all operations are ring operations
 * modulo 2^qbits (std::complex<intN> products/sums wrap), so the result is the
 * exact integer convolution reduced modulo 2^qbits, then fromQ (>> qbits/2),
 * then truncation to the element width (FIRFilter.cpp:295-300). */
static size_t fir_loop_int(const orc_fir *f, const void *xbase, ptrdiff_t x0, void *y, size_t N)
{
    const size_t L = f->L, M = f->M, K = f->K;
    const int st = f->st, qb = q_bits(st);
    size_t decim = M, nout = 0;
    for (size_t n = 0; n < N; n++) {
        for (size_t j = 0; j < L; j++) {
            if (--decim != 0) continue;
            decim = M;
            const int64_t *tp = f->rowTapsQ + j * K * (f->ctaps ? 2 : 1);
            const size_t len = f->rowLen[j];
            uint64_t ar = 0, ai = 0;
            for (size_t k = 0; k < len; k++) {
                const ptrdiff_t i = x0 + (ptrdiff_t)n - (ptrdiff_t)k;
                if (!f->cplx) {
                    ar += (uint64_t)tp[k] * (uint64_t)load_int(xbase, (size_t)i, st);
                } else {
                    const uint64_t c = (uint64_t)load_int(xbase, 2 * (size_t)i, st);
                    const uint64_t d = (uint64_t)load_int(xbase, 2 * (size_t)i + 1, st);
                    if (!f->ctaps) {
                        ar += (uint64_t)tp[k] * c; ai += (uint64_t)tp[k] * d;
                    } else {
                        const uint64_t a = (uint64_t)tp[2 * k], b = (uint64_t)tp[2 * k + 1];
                        ar += a * c - b * d; ai += a * d + b * c;
                    }
                }
            }
            if (!f->cplx) store_int(y, nout, st, from_q(wrap_bits(ar, qb), st));
            else {
                store_int(y, 2 * nout, st, from_q(wrap_bits(ar, qb), st));
                store_int(y, 2 * nout + 1, st, from_q(wrap_bits(ai, qb), st));
            }
            nout++;
        }
    }
    return nout;
}

/*
 * FIRFilter::work, FIRFilter.cpp:207-309.  `in` is the input port buffer
 * (in_elems elements available, K-1 history at the front), `out` the output
 * port buffer with room for out_elems elements.  Returns consumed/produced and
 * the reserve requested through setReserve (SIZE_MAX if setReserve not called).
 */
ORC_EXPORT int orc_fir_work(orc_fir *f, const void *in, size_t in_elems, const orc_label *labels, size_t nlabels,
                            void *out, size_t out_elems, size_t *consumed, size_t *produced, size_t *reserve)
{
    *consumed = 0; *produced = 0; *reserve = SIZE_MAX;
    if (f->waitTapsArmed) return 0;                       /* :209 */
    size_t inputAvailable = in_elems;
    if (inputAvailable == 0) return 0;                    /* :213 */
    const size_t K = f->K, M = f->M, L = f->L;

    if (f->eobSampsLeft == 0) for (size_t i = 0; i < nlabels; i++) {  /* :218-231 */
        const orc_label *lb = &labels[i];
        if (f->haveStartId && lb->id == 'S' && lb->has_length) {
            f->eobSampsLeft = lb->index + lb->length * lb->width;
            break;
        } else if (f->haveEndId && lb->id == 'E') {
            f->eobSampsLeft = lb->index + lb->width;
            break;
        }
    }
    if (f->eobSampsLeft != 0) {                           /* :237-248 */
        if (f->eobSampsLeft <= inputAvailable) inputAvailable = f->eobSampsLeft;
        else { *reserve = f->eobSampsLeft; return 0; }
    } else if (inputAvailable < f->inputRequire) {        /* :251-255 */
        *reserve = f->inputRequire; return 0;
    }
    *reserve = 0;                                         /* :258 */

    const size_t esz = (size_t)scalar_bytes(f->st) * (f->cplx ? 2 : 1);
    const void *buf = in;
    size_t bufElems = inputAvailable;
    void *flush = NULL;
    if (f->eobSampsLeft != 0 && f->eobSampsLeft < f->inputRequire) {  /* :265-272 */
        bufElems = f->eobSampsLeft + K - 1;
        flush = calloc(bufElems, esz);
        memcpy(flush, in, f->eobSampsLeft * esz);
        buf = flush;
    }
    /* :278  N = min((elems-(K-1))/M, outElems/L)*M  (size_t arithmetic) */
    size_t a = (bufElems - (K - 1)) / M, b = out_elems / L;
    const size_t N = (a < b ? a : b) * M;
    size_t nout;
    if (f->st == ORC_F32) nout = fir_loop_f32(f, (const float *)buf + (K - 1) * (f->cplx ? 2 : 1), (float *)out, N);
    else if (f->st == ORC_F64) nout = fir_loop_f64(f, (const double *)buf + (K - 1) * (f->cplx ? 2 : 1), (double *)out, N);
    else nout = fir_loop_int(f, buf, (ptrdiff_t)(K - 1), out, N);
    free(flush);
    if (f->eobSampsLeft != 0) f->eobSampsLeft -= N;       /* :306 */
    *consumed = N;                                        /* :307 */
    *produced = (N / M) * L;                              /* :308 */
    (void)nout;
    return 0;
}

/* ===================================================================== *
 *  FFT, floating point  (fft/kissfft.hh)
 * ===================================================================== */
#define ORC_MAXFACTORS 64

#define ORC_KISSFFT(PFX, T, CT, CEXP)                                                                    \
    typedef struct { T r, i; } PFX##_cpx;                                                                \
    typedef struct {                                                                                     \
        int nfft, inverse, nstages;                                                                      \
        int radix[ORC_MAXFACTORS], remainder[ORC_MAXFACTORS];                                            \
        PFX##_cpx *tw;                                                                                   \
    } PFX##_plan;                                                                                        \
    /* kissfft_utils::traits::fill_twiddles + prepare, kissfft.hh:21-56 */                               \
    static PFX##_plan *PFX##_make(int nfft, int inverse)                                                 \
    {                                                                                                    \
        PFX##_plan *p = (PFX##_plan *)calloc(1, sizeof(*p));                                             \
        p->nfft = nfft; p->inverse = inverse;                                                            \
        p->tw = (PFX##_cpx *)malloc(sizeof(PFX##_cpx) * (size_t)nfft);                                   \
        /* unqualified acos((T)-1) binds to ::acos(double) in the reference TU, */                       \
        /* so phinc is formed in double and rounded once (pinned against _ref)  */                       \
        const T phinc = (T)((inverse ? 2 : -2) * acos((double)(T)-1) / nfft);                            \
        for (int i = 0; i < nfft; ++i) {                                                                 \
            CT z = CEXP(CMPLX_##PFX((T)0, i * phinc));                                                   \
            p->tw[i].r = __real__ z; p->tw[i].i = __imag__ z;                                            \
        }                                                                                                \
        int n = nfft, q = 4;                                                                             \
        do {                                                                                             \
            while (n % q) {                                                                              \
                switch (q) { case 4: q = 2; break; case 2: q = 3; break; default: q += 2; break; }       \
                if (q * q > n) q = n;                                                                    \
            }                                                                                            \
            n /= q;                                                                                      \
            p->radix[p->nstages] = q; p->remainder[p->nstages] = n; p->nstages++;                        \
        } while (n > 1);                                                                                 \
        return p;                                                                                        \
    }                                                                                                    \
    static void PFX##_free(PFX##_plan *p) { if (p) { free(p->tw); free(p); } }                           \
    /* std::complex<T> operator* (finite path of __mul?c3) */                                            \
    static inline PFX##_cpx PFX##_mul(PFX##_cpx a, PFX##_cpx b)                                          \
    {                                                                                                    \
        const T ac = a.r * b.r, bd = a.i * b.i, ad = a.r * b.i, bc = a.i * b.r;                          \
        PFX##_cpx m; m.r = ac - bd; m.i = ad + bc; return m;                                             \
    }                                                                                                    \
    static inline PFX##_cpx PFX##_add(PFX##_cpx a, PFX##_cpx b) { PFX##_cpx m; m.r = a.r + b.r; m.i = a.i + b.i; return m; } \
    static inline PFX##_cpx PFX##_sub(PFX##_cpx a, PFX##_cpx b) { PFX##_cpx m; m.r = a.r - b.r; m.i = a.i - b.i; return m; } \
    /* kf_bfly2 kissfft.hh:132-139 */                                                                    \
    static void PFX##_bfly2(const PFX##_plan *p, PFX##_cpx *Fout, size_t fstride, int m)                 \
    {                                                                                                    \
        for (int k = 0; k < m; ++k) {                                                                    \
            PFX##_cpx t = PFX##_mul(Fout[m + k], p->tw[k * fstride]);                                    \
            Fout[m + k] = PFX##_sub(Fout[k], t);                                                         \
            Fout[k] = PFX##_add(Fout[k], t);                                                             \
        }                                                                                                \
    }                                                                                                    \
    /* kf_bfly4 kissfft.hh:141-161 */                                                                    \
    static void PFX##_bfly4(const PFX##_plan *p, PFX##_cpx *Fout, size_t fstride, size_t m)              \
    {                                                                                                    \
        PFX##_cpx s[7];                                                                                  \
        const int neg = p->inverse * -2 + 1;                                                             \
        for (size_t k = 0; k < m; ++k) {                                                                 \
            s[0] = PFX##_mul(Fout[k + m], p->tw[k * fstride]);                                           \
            s[1] = PFX##_mul(Fout[k + 2 * m], p->tw[k * fstride * 2]);                                   \
            s[2] = PFX##_mul(Fout[k + 3 * m], p->tw[k * fstride * 3]);                                   \
            s[5] = PFX##_sub(Fout[k], s[1]);                                                             \
            Fout[k] = PFX##_add(Fout[k], s[1]);                                                          \
            s[3] = PFX##_add(s[0], s[2]);                                                                \
            s[4] = PFX##_sub(s[0], s[2]);                                                                \
            { PFX##_cpx r; r.r = s[4].i * neg; r.i = -s[4].r * neg; s[4] = r; }                          \
            Fout[k + 2 * m] = PFX##_sub(Fout[k], s[3]);                                                  \
            Fout[k] = PFX##_add(Fout[k], s[3]);                                                          \
            Fout[k + m] = PFX##_add(s[5], s[4]);                                                         \
            Fout[k + 3 * m] = PFX##_sub(s[5], s[4]);                                                     \
        }                                                                                                \
    }                                                                                                    \
    /* kf_bfly3 kissfft.hh:163-198 */                                                                    \
    static void PFX##_bfly3(const PFX##_plan *p, PFX##_cpx *Fout, size_t fstride, size_t m)              \
    {                                                                                                    \
        size_t k = m; const size_t m2 = 2 * m;                                                           \
        const PFX##_cpx *tw1 = p->tw, *tw2 = p->tw;                                                      \
        PFX##_cpx s[5];                                                                                  \
        const PFX##_cpx epi3 = p->tw[fstride * m];                                                       \
        do {                                                                                             \
            s[1] = PFX##_mul(Fout[m], *tw1);                                                             \
            s[2] = PFX##_mul(Fout[m2], *tw2);                                                            \
            s[3] = PFX##_add(s[1], s[2]);                                                                \
            s[0] = PFX##_sub(s[1], s[2]);                                                                \
            tw1 += fstride; tw2 += fstride * 2;                                                          \
            Fout[m].r = Fout->r - (T)(s[3].r * .5);                                                      \
            Fout[m].i = Fout->i - (T)(s[3].i * .5);                                                      \
            s[0].r = s[0].r * epi3.i; s[0].i = s[0].i * epi3.i;                                          \
            *Fout = PFX##_add(*Fout, s[3]);                                                              \
            Fout[m2].r = Fout[m].r + s[0].i;                                                             \
            Fout[m2].i = Fout[m].i - s[0].r;                                                             \
            { PFX##_cpx r; r.r = -s[0].i; r.i = s[0].r; Fout[m] = PFX##_add(Fout[m], r); }               \
            ++Fout;                                                                                      \
        } while (--k);                                                                                   \
    }                                                                                                    \
    /* kf_bfly5 kissfft.hh:200-262 */                                                                    \
    static void PFX##_bfly5(const PFX##_plan *p, PFX##_cpx *Fout, size_t fstride, size_t m)              \
    {                                                                                                    \
        PFX##_cpx *F0 = Fout, *F1 = Fout + m, *F2 = Fout + 2 * m, *F3 = Fout + 3 * m, *F4 = Fout + 4 * m;\
        PFX##_cpx s[13];                                                                                 \
        const PFX##_cpx *tw = p->tw;                                                                     \
        const PFX##_cpx ya = tw[fstride * m], yb = tw[fstride * 2 * m];                                  \
        for (size_t u = 0; u < m; ++u) {                                                                 \
            s[0] = *F0;                                                                                  \
            s[1] = PFX##_mul(*F1, tw[u * fstride]);                                                      \
            s[2] = PFX##_mul(*F2, tw[2 * u * fstride]);                                                  \
            s[3] = PFX##_mul(*F3, tw[3 * u * fstride]);                                                  \
            s[4] = PFX##_mul(*F4, tw[4 * u * fstride]);                                                  \
            s[7] = PFX##_add(s[1], s[4]); s[10] = PFX##_sub(s[1], s[4]);                                 \
            s[8] = PFX##_add(s[2], s[3]); s[9] = PFX##_sub(s[2], s[3]);                                  \
            *F0 = PFX##_add(*F0, s[7]); *F0 = PFX##_add(*F0, s[8]);                                      \
            { PFX##_cpx t; T p1, p2;                                                                     \
              p1 = s[7].r * ya.r; p2 = s[8].r * yb.r; t.r = p1 + p2;                                     \
              p1 = s[7].i * ya.r; p2 = s[8].i * yb.r; t.i = p1 + p2;                                     \
              s[5] = PFX##_add(s[0], t);                                                                 \
              p1 = s[10].i * ya.i; p2 = s[9].i * yb.i; s[6].r = p1 + p2;                                 \
              p1 = s[10].r * ya.i; p2 = s[9].r * yb.i; s[6].i = -p1 - p2; }                              \
            *F1 = PFX##_sub(s[5], s[6]); *F4 = PFX##_add(s[5], s[6]);                                    \
            { PFX##_cpx t; T p1, p2;                                                                     \
              p1 = s[7].r * yb.r; p2 = s[8].r * ya.r; t.r = p1 + p2;                                     \
              p1 = s[7].i * yb.r; p2 = s[8].i * ya.r; t.i = p1 + p2;                                     \
              s[11] = PFX##_add(s[0], t);                                                                \
              p1 = s[10].i * yb.i; p2 = s[9].i * ya.i; s[12].r = -p1 + p2;                               \
              p1 = s[10].r * yb.i; p2 = s[9].r * ya.i; s[12].i = p1 - p2; }                              \
            *F2 = PFX##_add(s[11], s[12]); *F3 = PFX##_sub(s[11], s[12]);                                \
            ++F0; ++F1; ++F2; ++F3; ++F4;                                                                \
        }                                                                                                \
    }                                                                                                    \
    /* kf_bfly_generic kissfft.hh:264-303 */                                                             \
    static void PFX##_bfly_generic(const PFX##_plan *p, PFX##_cpx *Fout, size_t fstride, int m, int q)   \
    {                                                                                                    \
        const int Norig = p->nfft;                                                                       \
        PFX##_cpx *sb = (PFX##_cpx *)malloc(sizeof(PFX##_cpx) * (size_t)q);                              \
        for (int u = 0; u < m; ++u) {                                                                    \
            int k = u;                                                                                   \
            for (int q1 = 0; q1 < q; ++q1) { sb[q1] = Fout[k]; k += m; }                                 \
            k = u;                                                                                       \
            for (int q1 = 0; q1 < q; ++q1) {                                                             \
                int twidx = 0;                                                                           \
                Fout[k] = sb[0];                                                                         \
                for (int qq = 1; qq < q; ++qq) {                                                         \
                    twidx += (int)fstride * k;                                                           \
                    if (twidx >= Norig) twidx -= Norig;                                                  \
                    Fout[k] = PFX##_add(Fout[k], PFX##_mul(sb[qq], p->tw[twidx]));                       \
                }                                                                                        \
                k += m;                                                                                  \
            }                                                                                            \
        }                                                                                                \
        free(sb);                                                                                        \
    }                                                                                                    \
    /* kf_work kissfft.hh:87-120 */                                                                      \
    static void PFX##_work(const PFX##_plan *p, int stage, PFX##_cpx *Fout, const PFX##_cpx *f,          \
                           size_t fstride, size_t in_stride)                                             \
    {                                                                                                    \
        const int q = p->radix[stage], m = p->remainder[stage];                                          \
        PFX##_cpx *Fout_beg = Fout, *Fout_end = Fout + q * m;                                            \
        if (m == 1) {                                                                                    \
            do { *Fout = *f; f += fstride * in_stride; } while (++Fout != Fout_end);                     \
        } else {                                                                                         \
            do { PFX##_work(p, stage + 1, Fout, f, fstride * q, in_stride); f += fstride * in_stride; }  \
            while ((Fout += m) != Fout_end);                                                             \
        }                                                                                                \
        Fout = Fout_beg;                                                                                 \
        switch (q) {                                                                                     \
        case 2: PFX##_bfly2(p, Fout, fstride, m); break;                                                 \
        case 3: PFX##_bfly3(p, Fout, fstride, (size_t)m); break;                                         \
        case 4: PFX##_bfly4(p, Fout, fstride, (size_t)m); break;                                         \
        case 5: PFX##_bfly5(p, Fout, fstride, (size_t)m); break;                                         \
        default: PFX##_bfly_generic(p, Fout, fstride, m, q); break;                                      \
        }                                                                                                \
    }

#define CMPLX_kf32(a, b) CMPLXF(a, b)
#define CMPLX_kf64(a, b) CMPLX(a, b)
ORC_KISSFFT(kf32, float, float complex, cexpf)
ORC_KISSFFT(kf64, double, double complex, cexp)

/* ===================================================================== *
 *  FFT, Q15 fixed point  (fft/kiss_fft.c + fft/_kiss_fft_guts.h, -DFIXED_POINT=16)
 * ===================================================================== */
typedef struct { int16_t r, i; } k16_cpx;
typedef struct {
    int nfft, inverse;
    int factors[2 * ORC_MAXFACTORS];
    k16_cpx *tw;
} k16_plan;

/* _kiss_fft_guts.h:64-65 */
#define K16_SMUL(a, b) ((int32_t)(a) * (b))
#define K16_SROUND(x) ((int16_t)(((x) + (1 << 14)) >> 15))
static inline k16_cpx k16_mul(k16_cpx a, k16_cpx b)   /* C_MUL _kiss_fft_guts.h:69-71 */
{
    k16_cpx m;
    m.r = K16_SROUND(K16_SMUL(a.r, b.r) - K16_SMUL(a.i, b.i));
    m.i = K16_SROUND(K16_SMUL(a.r, b.i) + K16_SMUL(a.i, b.r));
    return m;
}
static inline void k16_fixdiv(k16_cpx *c, int div)    /* C_FIXDIV/DIVSCALAR _kiss_fft_guts.h:73-78 */
{
    c->r = K16_SROUND(K16_SMUL(c->r, 32767 / div));
    c->i = K16_SROUND(K16_SMUL(c->i, 32767 / div));
}
static inline int16_t k16_smulr(int16_t a, int16_t b) { return K16_SROUND(K16_SMUL(a, b)); } /* S_MUL */
static inline k16_cpx k16_add(k16_cpx a, k16_cpx b) { k16_cpx m; m.r = (int16_t)(a.r + b.r); m.i = (int16_t)(a.i + b.i); return m; }
static inline k16_cpx k16_sub(k16_cpx a, k16_cpx b) { k16_cpx m; m.r = (int16_t)(a.r - b.r); m.i = (int16_t)(a.i - b.i); return m; }

/* kiss_fft_alloc kiss_fft.c:339-368 + kf_factor :309-328 */
static k16_plan *k16_make(int nfft, int inverse)
{
    k16_plan *st = (k16_plan *)calloc(1, sizeof(*st));
    st->nfft = nfft; st->inverse = inverse;
    st->tw = (k16_cpx *)malloc(sizeof(k16_cpx) * (size_t)nfft);
    for (int i = 0; i < nfft; ++i) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / nfft;
        if (inverse) phase *= -1;
        st->tw[i].r = (int16_t)floor(.5 + 32767 * cos(phase));
        st->tw[i].i = (int16_t)floor(.5 + 32767 * sin(phase));
    }
    int n = nfft, p = 4, *facbuf = st->factors;
    const double floor_sqrt = floor(sqrt((double)n));
    do {
        while (n % p) {
            switch (p) { case 4: p = 2; break; case 2: p = 3; break; default: p += 2; break; }
            if (p > floor_sqrt) p = n;
        }
        n /= p;
        *facbuf++ = p; *facbuf++ = n;
    } while (n > 1);
    return st;
}
static void k16_free(k16_plan *p) { if (p) { free(p->tw); free(p); } }

static void k16_bfly2(const k16_plan *st, k16_cpx *Fout, size_t fstride, int m) /* kiss_fft.c:21-42 */
{
    k16_cpx *Fout2 = Fout + m; const k16_cpx *tw1 = st->tw;
    do {
        k16_fixdiv(Fout, 2); k16_fixdiv(Fout2, 2);
        k16_cpx t = k16_mul(*Fout2, *tw1);
        tw1 += fstride;
        *Fout2 = k16_sub(*Fout, t);
        *Fout = k16_add(*Fout, t);
        ++Fout2; ++Fout;
    } while (--m);
}
static void k16_bfly4(const k16_plan *st, k16_cpx *Fout, size_t fstride, size_t m) /* kiss_fft.c:44-90 */
{
    const k16_cpx *tw1 = st->tw, *tw2 = st->tw, *tw3 = st->tw;
    k16_cpx s[6];
    size_t k = m; const size_t m2 = 2 * m, m3 = 3 * m;
    do {
        k16_fixdiv(Fout, 4); k16_fixdiv(&Fout[m], 4); k16_fixdiv(&Fout[m2], 4); k16_fixdiv(&Fout[m3], 4);
        s[0] = k16_mul(Fout[m], *tw1);
        s[1] = k16_mul(Fout[m2], *tw2);
        s[2] = k16_mul(Fout[m3], *tw3);
        s[5] = k16_sub(*Fout, s[1]);
        *Fout = k16_add(*Fout, s[1]);
        s[3] = k16_add(s[0], s[2]);
        s[4] = k16_sub(s[0], s[2]);
        Fout[m2] = k16_sub(*Fout, s[3]);
        tw1 += fstride; tw2 += fstride * 2; tw3 += fstride * 3;
        *Fout = k16_add(*Fout, s[3]);
        if (st->inverse) {
            Fout[m].r = (int16_t)(s[5].r - s[4].i); Fout[m].i = (int16_t)(s[5].i + s[4].r);
            Fout[m3].r = (int16_t)(s[5].r + s[4].i); Fout[m3].i = (int16_t)(s[5].i - s[4].r);
        } else {
            Fout[m].r = (int16_t)(s[5].r + s[4].i); Fout[m].i = (int16_t)(s[5].i - s[4].r);
            Fout[m3].r = (int16_t)(s[5].r - s[4].i); Fout[m3].i = (int16_t)(s[5].i + s[4].r);
        }
        ++Fout;
    } while (--k);
}
static void k16_bfly3(const k16_plan *st, k16_cpx *Fout, size_t fstride, size_t m) /* kiss_fft.c:92-135 */
{
    size_t k = m; const size_t m2 = 2 * m;
    const k16_cpx *tw1 = st->tw, *tw2 = st->tw;
    k16_cpx s[5];
    const k16_cpx epi3 = st->tw[fstride * m];
    do {
        k16_fixdiv(Fout, 3); k16_fixdiv(&Fout[m], 3); k16_fixdiv(&Fout[m2], 3);
        s[1] = k16_mul(Fout[m], *tw1);
        s[2] = k16_mul(Fout[m2], *tw2);
        s[3] = k16_add(s[1], s[2]);
        s[0] = k16_sub(s[1], s[2]);
        tw1 += fstride; tw2 += fstride * 2;
        Fout[m].r = (int16_t)(Fout->r - (s[3].r >> 1));
        Fout[m].i = (int16_t)(Fout->i - (s[3].i >> 1));
        s[0].r = k16_smulr(s[0].r, epi3.i); s[0].i = k16_smulr(s[0].i, epi3.i);
        *Fout = k16_add(*Fout, s[3]);
        Fout[m2].r = (int16_t)(Fout[m].r + s[0].i);
        Fout[m2].i = (int16_t)(Fout[m].i - s[0].r);
        Fout[m].r = (int16_t)(Fout[m].r - s[0].i);
        Fout[m].i = (int16_t)(Fout[m].i + s[0].r);
        ++Fout;
    } while (--k);
}
static void k16_bfly5(const k16_plan *st, k16_cpx *Fout, size_t fstride, int m) /* kiss_fft.c:137-199 */
{
    k16_cpx *F0 = Fout, *F1 = Fout + m, *F2 = Fout + 2 * m, *F3 = Fout + 3 * m, *F4 = Fout + 4 * m;
    k16_cpx s[13];
    const k16_cpx *tw = st->tw;
    const k16_cpx ya = tw[fstride * m], yb = tw[fstride * 2 * m];
    for (int u = 0; u < m; ++u) {
        k16_fixdiv(F0, 5); k16_fixdiv(F1, 5); k16_fixdiv(F2, 5); k16_fixdiv(F3, 5); k16_fixdiv(F4, 5);
        s[0] = *F0;
        s[1] = k16_mul(*F1, tw[u * fstride]);
        s[2] = k16_mul(*F2, tw[2 * u * fstride]);
        s[3] = k16_mul(*F3, tw[3 * u * fstride]);
        s[4] = k16_mul(*F4, tw[4 * u * fstride]);
        s[7] = k16_add(s[1], s[4]); s[10] = k16_sub(s[1], s[4]);
        s[8] = k16_add(s[2], s[3]); s[9] = k16_sub(s[2], s[3]);
        F0->r = (int16_t)(F0->r + (s[7].r + s[8].r));
        F0->i = (int16_t)(F0->i + (s[7].i + s[8].i));
        s[5].r = (int16_t)(s[0].r + k16_smulr(s[7].r, ya.r) + k16_smulr(s[8].r, yb.r));
        s[5].i = (int16_t)(s[0].i + k16_smulr(s[7].i, ya.r) + k16_smulr(s[8].i, yb.r));
        s[6].r = (int16_t)(k16_smulr(s[10].i, ya.i) + k16_smulr(s[9].i, yb.i));
        s[6].i = (int16_t)(-k16_smulr(s[10].r, ya.i) - k16_smulr(s[9].r, yb.i));
        *F1 = k16_sub(s[5], s[6]); *F4 = k16_add(s[5], s[6]);
        s[11].r = (int16_t)(s[0].r + k16_smulr(s[7].r, yb.r) + k16_smulr(s[8].r, ya.r));
        s[11].i = (int16_t)(s[0].i + k16_smulr(s[7].i, yb.r) + k16_smulr(s[8].i, ya.r));
        s[12].r = (int16_t)(-k16_smulr(s[10].i, yb.i) + k16_smulr(s[9].i, ya.i));
        s[12].i = (int16_t)(k16_smulr(s[10].r, yb.i) - k16_smulr(s[9].r, ya.i));
        *F2 = k16_add(s[11], s[12]); *F3 = k16_sub(s[11], s[12]);
        ++F0; ++F1; ++F2; ++F3; ++F4;
    }
}
static void k16_bfly_generic(const k16_plan *st, k16_cpx *Fout, size_t fstride, int m, int p) /* kiss_fft.c:202-235 */
{
    const int Norig = st->nfft;
    k16_cpx *sb = (k16_cpx *)malloc(sizeof(k16_cpx) * (size_t)p);
    for (int u = 0; u < m; ++u) {
        int k = u;
        for (int q1 = 0; q1 < p; ++q1) { sb[q1] = Fout[k]; k16_fixdiv(&sb[q1], p); k += m; }
        k = u;
        for (int q1 = 0; q1 < p; ++q1) {
            int twidx = 0;
            Fout[k] = sb[0];
            for (int q = 1; q < p; ++q) {
                twidx += (int)fstride * k;
                if (twidx >= Norig) twidx -= Norig;
                Fout[k] = k16_add(Fout[k], k16_mul(sb[q], st->tw[twidx]));
            }
            k += m;
        }
    }
    free(sb);
}
static void k16_work(const k16_plan *st, k16_cpx *Fout, const k16_cpx *f, size_t fstride, int in_stride, const int *factors) /* kiss_fft.c:237-302 */
{
    k16_cpx *Fout_beg = Fout;
    const int p = *factors++, m = *factors++;
    const k16_cpx *Fout_end = Fout + p * m;
    if (m == 1) {
        do { *Fout = *f; f += fstride * (size_t)in_stride; } while (++Fout != Fout_end);
    } else {
        do { k16_work(st, Fout, f, fstride * (size_t)p, in_stride, factors); f += fstride * (size_t)in_stride; }
        while ((Fout += m) != Fout_end);
    }
    Fout = Fout_beg;
    switch (p) {
    case 2: k16_bfly2(st, Fout, fstride, m); break;
    case 3: k16_bfly3(st, Fout, fstride, (size_t)m); break;
    case 4: k16_bfly4(st, Fout, fstride, (size_t)m); break;
    case 5: k16_bfly5(st, Fout, fstride, m); break;
    default: k16_bfly_generic(st, Fout, fstride, m, p); break;
    }
}

/* ---- the /comms/fft block: FFT.cpp:43-72 + FFTAux.h:16-48 ---- */
typedef struct orc_fft {
    int st, nbins, inverse;
    kf32_plan *p32; kf64_plan *p64; k16_plan *p16;
} orc_fft;

/* FFTFactory FFT.cpp:83-93: complex<double>, complex<float>, complex<int16> only */
ORC_EXPORT orc_fft *orc_fft_create(int scalar_type, size_t nbins, int inverse)
{
    if (nbins == 0) return NULL;
    if (scalar_type != ORC_F64 && scalar_type != ORC_F32 && scalar_type != ORC_I16) return NULL;
    orc_fft *h = (orc_fft *)calloc(1, sizeof(*h));
    h->st = scalar_type; h->nbins = (int)nbins; h->inverse = inverse;
    if (scalar_type == ORC_F32) h->p32 = kf32_make((int)nbins, inverse);
    else if (scalar_type == ORC_F64) h->p64 = kf64_make((int)nbins, inverse);
    else h->p16 = k16_make((int)nbins, inverse);
    return h;
}
ORC_EXPORT void orc_fft_destroy(orc_fft *h)
{
    if (!h) return;
    kf32_free(h->p32); kf64_free(h->p64); k16_free(h->p16); free(h);
}
/* one transform (what FFT::work does per call, FFT.cpp:66-71) on `nframes`
 * consecutive frames -- the block itself does one per work() */
ORC_EXPORT int orc_fft_transform(const orc_fft *h, const void *in, void *out, size_t nframes)
{
    const size_t n = (size_t)h->nbins;
    for (size_t fr = 0; fr < nframes; fr++) {
        if (h->st == ORC_F32) kf32_work(h->p32, 0, (kf32_cpx *)out + fr * n, (const kf32_cpx *)in + fr * n, 1, 1);
        else if (h->st == ORC_F64) kf64_work(h->p64, 0, (kf64_cpx *)out + fr * n, (const kf64_cpx *)in + fr * n, 1, 1);
        else k16_work(h->p16, (k16_cpx *)out + fr * n, (const k16_cpx *)in + fr * n, 1, 1, h->p16->factors);
    }
    return 0;
}
/* FFT::work FFT.cpp:61-72: exactly one frame per call, no elements() check */
ORC_EXPORT int orc_fft_work(const orc_fft *h, const void *in, void *out, size_t *consumed, size_t *produced)
{
    orc_fft_transform(h, in, out, 1);
    *consumed = (size_t)h->nbins; *produced = (size_t)h->nbins;
    return 0;
}

/* ===================================================================== *
 *  fxpt_atan2 / getAngle / getAbs  (functions/fxpt_atan2.cpp, FxptHelpers.hpp)
 * ===================================================================== */
static inline int16_t q15_from_double(double d) { return (int16_t)lround(d * 32768); }  /* fxpt_atan2.cpp:36-38 */
static inline int16_t s16_nabs(int16_t j)                                                 /* :48-58 */
{
    const int16_t negSign = (int16_t)~(j >> 15);
    return (int16_t)((j ^ negSign) - negSign);
}
static inline int16_t q15_mul(int16_t j, int16_t k)                                       /* :68-77, unbiased rounding */
{
    const int32_t im = j * (int32_t)k;
    return (int16_t)((im + ((im & 0x7FFF) == 0x4000 ? 0 : 0x4000)) >> 15);
}
static inline int16_t q15_div(int16_t numer, int16_t denom)                               /* :89-91 */
{
    return (int16_t)(((int32_t)((uint32_t)(int32_t)numer << 15)) / denom);
}
ORC_EXPORT uint16_t orc_fxpt_atan2(int16_t y, int16_t x)                                  /* :108-138 */
{
    if (x == y) {
        if (y > 0) return 8192;
        else if (y < 0) return 40960;
        else return 0;
    }
    const int16_t nabs_y = s16_nabs(y), nabs_x = s16_nabs(x);
    if (nabs_x < nabs_y) {
        const int16_t y_over_x = q15_div(y, x);
        const int16_t correction = q15_mul(q15_from_double(0.273 * M_1_PI), s16_nabs(y_over_x));
        const int16_t unrotated = q15_mul((int16_t)(q15_from_double(0.25 + 0.273 * M_1_PI) + correction), y_over_x);
        if (x > 0) return (uint16_t)unrotated;
        else return (uint16_t)(32768 + unrotated);
    } else {
        const int16_t x_over_y = q15_div(x, y);
        const int16_t correction = q15_mul(q15_from_double(0.273 * M_1_PI), s16_nabs(x_over_y));
        const int16_t unrotated = q15_mul((int16_t)(q15_from_double(0.25 + 0.273 * M_1_PI) + correction), x_over_y);
        if (y > 0) return (uint16_t)(16384 - unrotated);
        else return (uint16_t)(49152 - unrotated);
    }
}

/* getAngle on complex integers FxptHelpers.hpp:21-29: truncate both parts to
 * int16, fxpt_atan2, then Type(uint16) (wraps for int8/int16) */
static int64_t get_angle_int(int64_t re, int64_t im)
{
    return (int64_t)orc_fxpt_atan2((int16_t)(uint16_t)(uint64_t)im, (int16_t)(uint16_t)(uint64_t)re);
}

/* ===================================================================== *
 *  FreqDemod  (demod/FreqDemod.cpp:44-71)
 * ===================================================================== */
typedef struct orc_freqdemod { int st; double pr, pi; int64_t ipr, ipi; } orc_freqdemod;
ORC_EXPORT orc_freqdemod *orc_freqdemod_create(int scalar_type)
{
    if (scalar_type < 0 || scalar_type > ORC_I8) return NULL;
    orc_freqdemod *h = (orc_freqdemod *)calloc(1, sizeof(*h));
    h->st = scalar_type;
    return h;
}
ORC_EXPORT void orc_freqdemod_destroy(orc_freqdemod *h) { free(h); }
ORC_EXPORT void orc_freqdemod_activate(orc_freqdemod *h) { h->pr = h->pi = 0; h->ipr = h->ipi = 0; } /* :44-47 */
/* work(): in = complex elements, out = real elements of the same scalar type */
ORC_EXPORT int orc_freqdemod_work(orc_freqdemod *h, const void *in, void *out, size_t N)
{
    if (h->st == ORC_F32) {
        const float *x = (const float *)in; float *y = (float *)out;
        float pr = (float)h->pr, pi = (float)h->pi;
        for (size_t i = 0; i < N; i++) {
            const float a = x[2 * i], b = x[2 * i + 1];
            float dr, di;
            orc_cmul_f32(a, b, pr, pi, &dr, &di);                             /* in_i * _prev  :63 */
            y[i] = atan2f(di, dr);                                            /* std::arg FxptHelpers.hpp:18 */
            pr = a; pi = -b;                                                  /* _prev = conj(in_i) :65 */
        }
        h->pr = pr; h->pi = pi;
    } else if (h->st == ORC_F64) {
        const double *x = (const double *)in; double *y = (double *)out;
        double pr = h->pr, pi = h->pi;
        for (size_t i = 0; i < N; i++) {
            const double a = x[2 * i], b = x[2 * i + 1];
            double dr, di;
            orc_cmul_f64(a, b, pr, pi, &dr, &di);
            y[i] = atan2(di, dr);
            pr = a; pi = -b;
        }
        h->pr = pr; h->pi = pi;
    } else {
        const int bits = scalar_bytes(h->st) * 8;
        int64_t pr = h->ipr, pi = h->ipi;
        for (size_t i = 0; i < N; i++) {
            const uint64_t a = (uint64_t)load_int(in, 2 * i, h->st), b = (uint64_t)load_int(in, 2 * i + 1, h->st);
            const int64_t dr = wrap_bits(a * (uint64_t)pr - b * (uint64_t)pi, bits);  /* complex<intN> product wraps */
            const int64_t di = wrap_bits(a * (uint64_t)pi + b * (uint64_t)pr, bits);
            store_int(out, i, h->st, get_angle_int(dr, di));
            pr = wrap_bits(a, bits); pi = wrap_bits((uint64_t)0 - b, bits);
        }
        h->ipr = pr; h->ipi = pi;
    }
    return 0;
}

/* ===================================================================== *
 *  Rotate (math/Rotate.cpp), Scale (math/Scale.cpp)
 * ===================================================================== */
/* arrayRotate Rotate.cpp:15-23 with phasor = floatToQ<QType>(std::polar(1.0, phase)) Rotate.cpp:71-75.
 * `n` counts complex elements (already multiplied by dtype.dimension(), :126). */
ORC_EXPORT int orc_rotate(int st, double phase, const void *in, void *out, size_t n)
{
    /* std::polar(1.0, phase) = (1.0*cos(phase), 1.0*sin(phase)) (libstdc++ <complex>).  An optimised GCC
     * build turns the pair into ONE glibc sincos() call, and sincos()'s sine is not always the bit
     * pattern sin() returns (phase 2.747554270528532: 0.3839204418969204 vs 0.38392044189692043); the
     * call is spelled out here so that the oracle does not depend on its own optimisation level. */
    double c, s;
    sincos(phase, &s, &c);
    if (st == ORC_F32) {
        const float pr = (float)c, pi = (float)s;
        const float *x = (const float *)in; float *y = (float *)out;
        for (size_t i = 0; i < n; i++) {
            orc_cmul_f32(pr, pi, x[2 * i], x[2 * i + 1], &y[2 * i], &y[2 * i + 1]);
        }
    } else if (st == ORC_F64) {
        const double *x = (const double *)in; double *y = (double *)out;
        for (size_t i = 0; i < n; i++) {
            orc_cmul_f64(c, s, x[2 * i], x[2 * i + 1], &y[2 * i], &y[2 * i + 1]);
        }
    } else {
        const int qb = q_bits(st);
        const uint64_t a = (uint64_t)float_to_q(c, st), b = (uint64_t)float_to_q(s, st);
        for (size_t i = 0; i < n; i++) {
            const uint64_t cc = (uint64_t)load_int(in, 2 * i, st), d = (uint64_t)load_int(in, 2 * i + 1, st);
            store_int(out, 2 * i, st, from_q(wrap_bits(a * cc - b * d, qb), st));
            store_int(out, 2 * i + 1, st, from_q(wrap_bits(a * d + b * cc, qb), st));
        }
    }
    return 0;
}
/* Rotate before any setPhase: _phasor value-initialises to 0 (Rotate.cpp:60-62
 * leaves it unset; std::complex default ctor zeroes) -> all-zero output */
ORC_EXPORT int orc_rotate_unset(int st, const void *in, void *out, size_t n)
{
    (void)in;
    memset(out, 0, n * 2 * (size_t)scalar_bytes(st));
    return 0;
}
/* arrayScale Scale.cpp:15-23, factorScaled = floatToQ<ScaleType>(factor) Scale.cpp:70-74;
 * `n` counts scalars for real types and complex elements for complex types */
ORC_EXPORT int orc_scale(int st, int is_complex, double factor, const void *in, void *out, size_t n)
{
    const size_t ns = n * (is_complex ? 2 : 1);  /* real factor: componentwise either way */
    if (st == ORC_F32) {
        const float f = (float)factor; const float *x = (const float *)in; float *y = (float *)out;
        for (size_t i = 0; i < ns; i++) y[i] = x[i] * f;
    } else if (st == ORC_F64) {
        const double *x = (const double *)in; double *y = (double *)out;
        for (size_t i = 0; i < ns; i++) y[i] = x[i] * factor;
    } else {
        const int qb = q_bits(st);
        const uint64_t f = (uint64_t)float_to_q(factor, st);
        for (size_t i = 0; i < ns; i++)
            store_int(out, i, st, from_q(wrap_bits(f * (uint64_t)load_int(in, i, st), qb), st));
    }
    return 0;
}
/* label handling shared by Rotate::work / Scale::work (Rotate.cpp:105-123,
 * Scale.cpp:104-122).  labels are (index, matches-id) pairs in port order.
 * Returns the number of elements to process this call; *apply_idx = index into
 * `labels` of a label whose data must be applied (setPhase/setFactor) before
 * processing, or -1. */
ORC_EXPORT size_t orc_coeff_label_scan(size_t elems, const uint64_t *label_index, const int *label_match,
                                       size_t nlabels, int have_label_id, long *apply_idx)
{
    *apply_idx = -1;
    if (!have_label_id) return elems;
    for (size_t i = 0; i < nlabels; i++) {
        if (label_index[i] >= elems) break;
        if (label_match[i]) {
            if (label_index[i] == 0) *apply_idx = (long)i;
            else { elems = label_index[i]; break; }
        }
    }
    return elems;
}

/* ===================================================================== *
 *  Abs (math/Abs.cpp:40-43 + FxptHelpers.hpp:36-49), Conjugate (math/Conjugate.cpp:36-39)
 * ===================================================================== */
ORC_EXPORT int orc_abs(int st, int is_complex, const void *in, void *out, size_t n)
{
    if (st == ORC_F32) {
        const float *x = (const float *)in; float *y = (float *)out;
        if (is_complex) for (size_t i = 0; i < n; i++) y[i] = hypotf(x[2 * i], x[2 * i + 1]); /* std::abs(complex) = cabsf */
        else for (size_t i = 0; i < n; i++) y[i] = fabsf(x[i]);
    } else if (st == ORC_F64) {
        const double *x = (const double *)in; double *y = (double *)out;
        if (is_complex) for (size_t i = 0; i < n; i++) y[i] = hypot(x[2 * i], x[2 * i + 1]);
        else for (size_t i = 0; i < n; i++) y[i] = fabs(x[i]);
    } else if (!is_complex) {
        /* OutType(std::abs(in)): int8/int16 promote to int; MIN wraps back to MIN */
        for (size_t i = 0; i < n; i++) {
            const int64_t v = load_int(in, i, st);
            store_int(out, i, st, v < 0 ? (int64_t)((uint64_t)0 - (uint64_t)v) : v);
        }
    } else {
        /* mag2 = re*re + im*im in the promoted type (int for int8/16, the type
         * itself for int32/int64: wraps), OutType(std::sqrt(float(mag2))) */
        const int pbits = (st == ORC_I64) ? 64 : 32;
        for (size_t i = 0; i < n; i++) {
            const uint64_t re = (uint64_t)load_int(in, 2 * i, st), im = (uint64_t)load_int(in, 2 * i + 1, st);
            const int64_t mag2 = wrap_bits(re * re + im * im, pbits);
            const float r = sqrtf((float)mag2);
            /* float -> integer conversion (truncation toward zero); NaN for negative
             * wrapped mag2 converts as x86 cvttss2si does (MIN) */
            int64_t o;
            if (r != r) o = (st == ORC_I64) ? INT64_MIN : INT32_MIN;
            else o = (int64_t)r;
            store_int(out, i, st, o);
        }
    }
    return 0;
}
ORC_EXPORT int orc_conj(int st, const void *in, void *out, size_t n)
{
    if (st == ORC_F32) {
        const float *x = (const float *)in; float *y = (float *)out;
        for (size_t i = 0; i < n; i++) { y[2 * i] = x[2 * i]; y[2 * i + 1] = -x[2 * i + 1]; }
    } else if (st == ORC_F64) {
        const double *x = (const double *)in; double *y = (double *)out;
        for (size_t i = 0; i < n; i++) { y[2 * i] = x[2 * i]; y[2 * i + 1] = -x[2 * i + 1]; }
    } else {
        for (size_t i = 0; i < n; i++) {
            store_int(out, 2 * i, st, load_int(in, 2 * i, st));
            store_int(out, 2 * i + 1, st, (int64_t)((uint64_t)0 - (uint64_t)load_int(in, 2 * i + 1, st)));
        }
    }
    return 0;
}

/* ===================================================================== *
 *  /comms/arithmetic, /comms/split_complex, /comms/combine_complex  
 *
 *  Arithmetic::work (math/Arithmetic.cpp:205-231) left-folds the block's input ports through
 *  one of four array functions (:70-110):  out[i] = in0[i] OP in1[i]  with the C++ operator of the
 *  element type.  Types: float, double, (u)int8..64 and std::complex of each (:284-296).
 *  What those operators are, on the reference's toolchain (libstdc++ <complex>, libgcc):
 *    - integers narrower than int promote to int and narrow back on the store (wrap mod 2^bits);
 *      32/64-bit signed overflow is undefined in C++ and wraps on every build of the reference;
 *    - x / 0 traps (SIGFPE) in the reference and INT_MIN / -1 is undefined: this restatement
 *      returns 0 resp. the wrapped quotient there, and the tests keep those inputs out of
 *      parity claims;
 *    - complex<float|double> * and / are the C99 operations (std::complex<T>::operator*= and /=
 *      act on the __complex__ representation: libgcc __mulsc3/__divsc3/__muldc3/__divdc3), which
 *      this file gets from the same libgcc_s.so.1 by using C99 _Complex arithmetic (Makefile:
 *      -shared-libgcc).  Measured on this image (libgcc_s from GCC 12): complex<float> division is
 *      evaluated in double -- den = c*c + d*d, ((a*c + b*d)/den, (b*c - a*d)/den), rounded once to
 *      float; complex<double> division is Smith's ratio form; both multiplications are
 *      (a*c - b*d, a*d + b*c) with separately rounded products for finite operands;
 *    - complex<integer> uses libstdc++'s generic members: operator*= is
 *        r = re*z.re - im*z.im;  im = re*z.im + im*z.re;  re = r          (each narrowed to T)
 *      and operator/= is
 *        r = re*z.re + im*z.im;  n = norm(z) = z.re*z.re + z.im*z.im      (r, n narrowed to T)
 *        im = (im*z.re - re*z.im) / n;  re = r / n
 *      -- note the asymmetry: the new imaginary part divides the un-narrowed promoted numerator.
 *  Pinned by math/TestArithmeticBlocks.cpp:47-245 (closed-form vectors, all 20 types x 4 ops) through
 *  tests/golden and, for the complex operators, against std::complex itself (oracle/ref_driver.cpp).
 * ===================================================================== */
enum { ORC_ADD = 0, ORC_SUB = 1, ORC_MUL = 2, ORC_DIV = 3 };

/* T: element type; P: the type the operands are promoted to (int for the narrow types, the
 * unsigned twin for 32/64-bit so that + - * wrap without undefined behaviour); D: the type the
 * division is carried out in (T's own signedness after promotion) */
#define ORC_ARITH_INT(NAME, T, P, D, DMIN)                                                          \
    static inline D NAME##_div(D a, D b)                                                            \
    {                                                                                               \
        if (b == 0) return 0;               /* reference: SIGFPE */                                 \
        if ((DMIN) != 0 && a == (D)(DMIN) && b == (D)-1) return a; /* reference: undefined */       \
        return (D)(a / b);                                                                          \
    }                                                                                               \
    static void NAME##_real(int op, const T *a, const T *b, T *o, size_t n)                         \
    {                                                                                               \
        for (size_t i = 0; i < n; i++) {                                                            \
            const P x = (P)a[i], y = (P)b[i];                                                       \
            switch (op) {                                                                           \
            case ORC_ADD: o[i] = (T)(P)(x + y); break;                                              \
            case ORC_SUB: o[i] = (T)(P)(x - y); break;                                              \
            case ORC_MUL: o[i] = (T)(P)(x * y); break;                                              \
            default: o[i] = (T)NAME##_div((D)a[i], (D)b[i]); break;                                 \
            }                                                                                       \
        }                                                                                           \
    }                                                                                               \
    static void NAME##_cplx(int op, const T *a, const T *b, T *o, size_t n)                         \
    {                                                                                               \
        for (size_t i = 0; i < n; i++) {                                                            \
            const P ar = (P)a[2 * i], ai = (P)a[2 * i + 1], br = (P)b[2 * i], bi = (P)b[2 * i + 1]; \
            T re, im;                                                                               \
            switch (op) {                                                                           \
            case ORC_ADD: re = (T)(P)(ar + br); im = (T)(P)(ai + bi); break;                        \
            case ORC_SUB: re = (T)(P)(ar - br); im = (T)(P)(ai - bi); break;                        \
            case ORC_MUL: re = (T)(P)(ar * br - ai * bi); im = (T)(P)(ar * bi + ai * br); break;    \
            default: {                                                                              \
                const T r = (T)(P)(ar * br + ai * bi);                                              \
                const T nn = (T)(P)(br * br + bi * bi);                                             \
                /* numerator of the new imaginary part stays in the promoted type */                \
                const P num = (P)(ai * br - ar * bi);                                               \
                im = (T)NAME##_div((D)num, (D)nn);                                                  \
                re = (T)NAME##_div((D)r, (D)nn);                                                    \
            } break;                                                                                \
            }                                                                                       \
            o[2 * i] = re; o[2 * i + 1] = im;                                                       \
        }                                                                                           \
    }
/* narrow types: operands promote to int, so P = D = int */
ORC_ARITH_INT(ar_i8, int8_t, int, int, 0)
ORC_ARITH_INT(ar_u8, uint8_t, int, int, 0)
ORC_ARITH_INT(ar_i16, int16_t, int, int, 0)
ORC_ARITH_INT(ar_u16, uint16_t, int, int, 0)
/* 32/64-bit: + - * in the unsigned twin (wraps), division in the type itself */
ORC_ARITH_INT(ar_i32, int32_t, uint32_t, int32_t, INT32_MIN)
ORC_ARITH_INT(ar_u32, uint32_t, uint32_t, uint32_t, 0)
ORC_ARITH_INT(ar_i64, int64_t, uint64_t, int64_t, INT64_MIN)
ORC_ARITH_INT(ar_u64, uint64_t, uint64_t, uint64_t, 0)

#define ORC_ARITH_FLT(NAME, T, CT)                                                                  \
    static void NAME##_real(int op, const T *a, const T *b, T *o, size_t n)                         \
    {                                                                                               \
        for (size_t i = 0; i < n; i++) {                                                            \
            switch (op) {                                                                           \
            case ORC_ADD: o[i] = a[i] + b[i]; break;                                                \
            case ORC_SUB: o[i] = a[i] - b[i]; break;                                                \
            case ORC_MUL: o[i] = a[i] * b[i]; break;                                                \
            default: o[i] = a[i] / b[i]; break;                                                     \
            }                                                                                       \
        }                                                                                           \
    }                                                                                               \
    static void NAME##_cplx(int op, const T *a, const T *b, T *o, size_t n)                         \
    {                                                                                               \
        for (size_t i = 0; i < n; i++) {                                                            \
            CT x, y, z;                                                                             \
            __real__ x = a[2 * i]; __imag__ x = a[2 * i + 1];                                       \
            __real__ y = b[2 * i]; __imag__ y = b[2 * i + 1];                                       \
            switch (op) {                                                                           \
            case ORC_ADD: z = x + y; break;                                                         \
            case ORC_SUB: z = x - y; break;                                                         \
            case ORC_MUL: z = x * y; break;   /* __mulsc3 / __muldc3 */                             \
            default: z = x / y; break;        /* __divsc3 / __divdc3 */                             \
            }                                                                                       \
            o[2 * i] = __real__ z; o[2 * i + 1] = __imag__ z;                                       \
        }                                                                                           \
    }
ORC_ARITH_FLT(ar_f32, float, float _Complex)
ORC_ARITH_FLT(ar_f64, double, double _Complex)

/* out[i] = in0[i] OP in1[i]; n counts elements (complex elements when is_complex) */
ORC_EXPORT int orc_arith(int st, int is_complex, int op, const void *in0, const void *in1, void *out, size_t n)
{
    if (op < ORC_ADD || op > ORC_DIV) return -1;
#define ORC_ARITH_CASE(CODE, NAME, T)                                                          \
    case CODE:                                                                                 \
        if (is_complex) NAME##_cplx(op, (const T *)in0, (const T *)in1, (T *)out, n);          \
        else NAME##_real(op, (const T *)in0, (const T *)in1, (T *)out, n);                     \
        return 0;
    switch (st) {
        ORC_ARITH_CASE(ORC_F64, ar_f64, double)
        ORC_ARITH_CASE(ORC_F32, ar_f32, float)
        ORC_ARITH_CASE(ORC_I64, ar_i64, int64_t)
        ORC_ARITH_CASE(ORC_I32, ar_i32, int32_t)
        ORC_ARITH_CASE(ORC_I16, ar_i16, int16_t)
        ORC_ARITH_CASE(ORC_I8, ar_i8, int8_t)
        ORC_ARITH_CASE(ORC_U64, ar_u64, uint64_t)
        ORC_ARITH_CASE(ORC_U32, ar_u32, uint32_t)
        ORC_ARITH_CASE(ORC_U16, ar_u16, uint16_t)
        ORC_ARITH_CASE(ORC_U8, ar_u8, uint8_t)
    }
#undef ORC_ARITH_CASE
    return -1;
}

/* arraySplitComplex / arrayCombineComplex (utility/SplitComplex.cpp:10-18,
 * utility/CombineComplex.cpp:10-17): pure data movement, any scalar width */
ORC_EXPORT int orc_split_complex(int st, const void *in, void *re, void *im, size_t n)
{
    const size_t w = (size_t)scalar_bytes(st);
    if (w == 0) return -1;
    for (size_t i = 0; i < n; i++) {
        memcpy((char *)re + i * w, (const char *)in + 2 * i * w, w);
        memcpy((char *)im + i * w, (const char *)in + (2 * i + 1) * w, w);
    }
    return 0;
}
ORC_EXPORT int orc_combine_complex(int st, const void *re, const void *im, void *out, size_t n)
{
    const size_t w = (size_t)scalar_bytes(st);
    if (w == 0) return -1;
    for (size_t i = 0; i < n; i++) {
        memcpy((char *)out + 2 * i * w, (const char *)re + i * w, w);
        memcpy((char *)out + (2 * i + 1) * w, (const char *)im + i * w, w);
    }
    return 0;
}

/* getAngle for arrays (shared with /comms/angle; pins FreqDemod's angle stage
 * against math/TestAngle.cpp vectors) */
ORC_EXPORT int orc_angle(int st, const void *in, void *out, size_t n)
{
    if (st == ORC_F32) {
        const float *x = (const float *)in; float *y = (float *)out;
        for (size_t i = 0; i < n; i++) y[i] = atan2f(x[2 * i + 1], x[2 * i]);
    } else if (st == ORC_F64) {
        const double *x = (const double *)in; double *y = (double *)out;
        for (size_t i = 0; i < n; i++) y[i] = atan2(x[2 * i + 1], x[2 * i]);
    } else {
        for (size_t i = 0; i < n; i++)
            store_int(out, i, st, get_angle_int(load_int(in, 2 * i, st), load_int(in, 2 * i + 1, st)));
    }
    return 0;
}

/* ===================================================================== *
 *  Deterministic synthetic streams: splitmix64 counter hash ->
 *  24-bit mantissa -> uniform [-1,1).  Not from the reference; shared by the
 *  tests, bench.py's CPU leg and (restated in HIP) the device-side generator.
 * ===================================================================== */
static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
ORC_EXPORT void orc_fill_uniform_f32(float *dst, size_t n_scalars, uint64_t seed, uint64_t offset)
{
    for (size_t i = 0; i < n_scalars; i++) {
        const uint64_t h = splitmix64(seed * 0x100000001B3ull + offset + i);
        dst[i] = (float)((int32_t)(h >> 40) - (1 << 23)) * (1.0f / (float)(1 << 23));
    }
}

/* multi-threaded helper for bench.py's CPU baseline: plain static chunking of
 * the M=L=1 float FIR with K-1 overlap ("parallelised restatement, not
 * reference behaviour"); single call = one thread's chunk. */
ORC_EXPORT int orc_fir_cf32_chunk(const orc_fir *f, const float *in_with_history, float *out, size_t n_out)
{
    if (f->st != ORC_F32 || !f->cplx || f->L != 1 || f->M != 1) return -1;
    fir_loop_f32(f, in_with_history + 2 * (f->K - 1), out, n_out);
    return 0;
}
