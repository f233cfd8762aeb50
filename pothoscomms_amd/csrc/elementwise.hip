// elementwise.hip -- HBM-bound maps for /comms/{rotate,scale,abs,conjugate,freq_demod}
//
// One grid-stride kernel template: each lane moves 16-byte vectors (the coalescing
// sweet spot on gfx950: 1 KiB per wave-instruction), UNROLL vectors in flight per
// lane, <= 2048 blocks of 256 threads.  No LDS, no MFMA: these ops have no reuse
// and ~1 flop/byte, so the only roof is HBM.
//
// This TU is compiled with -ffp-contract=off: the reference computes
// (a*c - b*d, a*d + b*c) with every product and sum rounded separately
// (std::complex operator* on baseline x86-64), and keeping the same unfused
// sequence makes Rotate/Scale/Conjugate/Abs(float) BIT-IDENTICAL to it.
#include "pcx_cplx.hpp"
#include "pcx_internal.hpp"
#include "vec_io.hpp"

#include <type_traits>

namespace pcx {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

// IN_PER / OUT_PER scalars per stream item; ITEMS items per lane-vector.
// No __restrict__: pcx.h documents out == in for the same-size maps (rotate, scale, conj), and Arithmetic's buffer
// inlining relies on it.  Every lane loads the vectors it is about to overwrite before it stores them, and no lane
// touches another lane's elements, so the in-place call is well defined as written.
template <typename Op, typename = void>
struct HasFix : std::false_type {};
template <typename Op>
struct HasFix<Op, std::enable_if_t<Op::kHasFix>> : std::true_type {};
// the rare second look at a lane's vector of results (an Op with kHasFix: Rotate's complex multiply)
template <int IN_PER, int OUT_PER, int ITEMS, typename Op, typename In, typename Out>
__device__ __forceinline__ void map_fix(const Op &op, const In *a, Out *b)
{
    if constexpr (HasFix<Op>::value) {
        bool bad = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) bad |= op.suspect(&b[k * OUT_PER]);
        if (__builtin_expect(bad, 0)) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) op.fix(&a[k * IN_PER], &b[k * OUT_PER]);
        }
    }
}

template <typename In, typename Out, int IN_PER, int OUT_PER, int ITEMS, typename Op>
__global__ __launch_bounds__(kBlock) void map_kernel(const In *in, Out *out, size_t nitems, Op op)
{
    using VIn = Vec<In, IN_PER * ITEMS>;
    using VOut = Vec<Out, OUT_PER * ITEMS>;
    const size_t nvec = nitems / ITEMS;
    const size_t chunk = (size_t)kBlock * kUnroll;
    const size_t nchunks = nvec / chunk;
    const VIn *vin = reinterpret_cast<const VIn *>(in);
    VOut *vout = reinterpret_cast<VOut *>(out);
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        VIn a[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) a[u] = nt_load(&vin[base + (size_t)u * kBlock]);
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            VOut b;
#pragma unroll
            for (int k = 0; k < ITEMS; k++) op(&a[u].v[k * IN_PER], &b.v[k * OUT_PER]);
            map_fix<IN_PER, OUT_PER, ITEMS>(op, a[u].v, b.v);
            nt_store(&vout[base + (size_t)u * kBlock], b);
        }
    }
    // remaining whole vectors, then the scalar tail
    const size_t gtid = (size_t)blockIdx.x * kBlock + threadIdx.x, gstride = (size_t)gridDim.x * kBlock;
    for (size_t i = nchunks * chunk + gtid; i < nvec; i += gstride) {
        VIn a = vin[i];
        VOut b;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) op(&a.v[k * IN_PER], &b.v[k * OUT_PER]);
        map_fix<IN_PER, OUT_PER, ITEMS>(op, a.v, b.v);
        vout[i] = b;
    }
    for (size_t i = nvec * ITEMS + gtid; i < nitems; i += gstride) {
        Out y[OUT_PER];
        In x[IN_PER];
#pragma unroll
        for (int k = 0; k < IN_PER; k++) x[k] = in[i * IN_PER + k];     // (out may be in: read the element before writing it)
        op(x, y);
        map_fix<IN_PER, OUT_PER, 1>(op, x, y);
#pragma unroll
        for (int k = 0; k < OUT_PER; k++) out[i * OUT_PER + k] = y[k];
    }
}

template <typename In, typename Out, int IN_PER, int OUT_PER, typename Op>
static int launch_map(const void *in, void *out, size_t nitems, Op op, hipStream_t st)
{
    if (nitems == 0) return PCX_OK;
    constexpr int item_bytes = (int)sizeof(In) * IN_PER;
    constexpr int ITEMS = item_bytes >= 16 ? 1 : 16 / item_bytes;
    const In *pin = static_cast<const In *>(in);
    Out *pout = static_cast<Out *>(out);
    const bool aligned = (reinterpret_cast<uintptr_t>(in) % (sizeof(In) * IN_PER * ITEMS) == 0) &&
                         (reinterpret_cast<uintptr_t>(out) % (sizeof(Out) * OUT_PER * ITEMS) == 0);
    if (aligned && ITEMS > 1) {
        const unsigned grid = stream_grid(nitems / ITEMS / kUnroll + 1, kBlock);
        hipLaunchKernelGGL((map_kernel<In, Out, IN_PER, OUT_PER, ITEMS, Op>), dim3(grid), dim3(kBlock), 0, st, pin, pout, nitems, op);
    } else {
        const unsigned grid = stream_grid(nitems / kUnroll + 1, kBlock);
        hipLaunchKernelGGL((map_kernel<In, Out, IN_PER, OUT_PER, 1, Op>), dim3(grid), dim3(kBlock), 0, st, pin, pout, nitems, op);
    }
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// ---- Q-format integer helpers (device): exact ring arithmetic modulo 2^bits(Q) ----
template <typename Q>
struct QCompute {
    using type = typename std::conditional<(sizeof(Q) < 4), uint32_t, typename std::make_unsigned<Q>::type>::type;
};
template <typename S, typename Q>
__device__ inline S from_q_dev(typename QCompute<Q>::type v, QShift qs)
{
    const Q q = (Q)v;                       // wrap to the Q width
    return (S)from_q_bits<Q>(q, qs);        // fromQ under the reading in force (pcx_qformat.hpp), then truncate to the element
}

// ---- Rotate  (math/Rotate.cpp:15-23) ----
template <typename T>
struct RotateF {
    T pr, pi;
    __device__ void operator()(const T *x, T *y) const
    {
        const T ac = pr * x[0], bd = pi * x[1], ad = pr * x[1], bc = pi * x[0];
        y[0] = ac - bd;
        y[1] = ad + bc;
    }
    // std::complex operator*'s slow path (pcx_cplx.hpp): map_kernel asks `suspect` for every result of a lane's vector and
    // runs `fix` on the vector only when one of them says yes
    static constexpr bool kHasFix = true;
    __device__ bool suspect(const T *y) const { return both_nan(y[0], y[1]); }
    __device__ void fix(const T *x, T *y) const
    {
        if (both_nan(y[0], y[1])) cmul_annex_g(pr, pi, x[0], x[1], y[0], y[1]);
    }
};
template <typename S, typename Q>
struct RotateI {
    using C = typename QCompute<Q>::type;
    C pr, pi;
    QShift qs;
    __device__ void operator()(const S *x, S *y) const
    {
        const C c = (C)(Q)x[0], d = (C)(Q)x[1];
        y[0] = from_q_dev<S, Q>(pr * c - pi * d, qs);
        y[1] = from_q_dev<S, Q>(pr * d + pi * c, qs);
    }
};
int launch_rotate(int scalar, double pr, double pi, const QFormat &qf, const void *in, void *out, size_t n, hipStream_t st)
{
    // phasor = floatToQ<QType>(std::polar(1.0, phase)), Rotate.cpp:74; out = fromQ<Type>(phasor * QType(in)), :21
    const QShift qs = q_shift(qf, scalar);
    switch (scalar) {
    case PCX_F32: return launch_map<float, float, 2, 2>(in, out, n, RotateF<float>{(float)pr, (float)pi}, st);
    case PCX_F64: return launch_map<double, double, 2, 2>(in, out, n, RotateF<double>{pr, pi}, st);
    case PCX_I64: return launch_map<int64_t, int64_t, 2, 2>(in, out, n, RotateI<int64_t, int64_t>{(uint64_t)float_to_q(pr, scalar, qf), (uint64_t)float_to_q(pi, scalar, qf), qs}, st);
    case PCX_I32: return launch_map<int32_t, int32_t, 2, 2>(in, out, n, RotateI<int32_t, int64_t>{(uint64_t)float_to_q(pr, scalar, qf), (uint64_t)float_to_q(pi, scalar, qf), qs}, st);
    case PCX_I16: return launch_map<int16_t, int16_t, 2, 2>(in, out, n, RotateI<int16_t, int32_t>{(uint32_t)float_to_q(pr, scalar, qf), (uint32_t)float_to_q(pi, scalar, qf), qs}, st);
    case PCX_I8: return launch_map<int8_t, int8_t, 2, 2>(in, out, n, RotateI<int8_t, int16_t>{(uint32_t)float_to_q(pr, scalar, qf), (uint32_t)float_to_q(pi, scalar, qf), qs}, st);
    }
    set_error("rotate: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

// ---- Scale  (math/Scale.cpp:15-23); real factor => componentwise on scalars ----
template <typename T>
struct ScaleF {
    T f;
    __device__ void operator()(const T *x, T *y) const { y[0] = x[0] * f; }
};
template <typename S, typename Q>
struct ScaleI {
    using C = typename QCompute<Q>::type;
    C f;
    QShift qs;
    __device__ void operator()(const S *x, S *y) const { y[0] = from_q_dev<S, Q>(f * (C)(Q)x[0], qs); }
};
int launch_scale(int scalar, int is_complex, double factor, const QFormat &qf, const void *in, void *out, size_t n, hipStream_t st)
{
    // factorScaled = floatToQ<ScaleType>(factor), Scale.cpp:73; out = fromQ<Type>(factorScaled * QType(in)), :21
    const size_t ns = n * (is_complex ? 2 : 1);
    const QShift qs = q_shift(qf, scalar);
    switch (scalar) {
    case PCX_F32: return launch_map<float, float, 1, 1>(in, out, ns, ScaleF<float>{(float)factor}, st);
    case PCX_F64: return launch_map<double, double, 1, 1>(in, out, ns, ScaleF<double>{factor}, st);
    case PCX_I64: return launch_map<int64_t, int64_t, 1, 1>(in, out, ns, ScaleI<int64_t, int64_t>{(uint64_t)float_to_q(factor, scalar, qf), qs}, st);
    case PCX_I32: return launch_map<int32_t, int32_t, 1, 1>(in, out, ns, ScaleI<int32_t, int64_t>{(uint64_t)float_to_q(factor, scalar, qf), qs}, st);
    case PCX_I16: return launch_map<int16_t, int16_t, 1, 1>(in, out, ns, ScaleI<int16_t, int32_t>{(uint32_t)float_to_q(factor, scalar, qf), qs}, st);
    case PCX_I8: return launch_map<int8_t, int8_t, 1, 1>(in, out, ns, ScaleI<int8_t, int16_t>{(uint32_t)float_to_q(factor, scalar, qf), qs}, st);
    }
    set_error("scale: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

// ---- Conjugate  (math/Conjugate.cpp:36-39) ----
template <typename T>
struct ConjOp {
    __device__ void operator()(const T *x, T *y) const
    {
        y[0] = x[0];
        if constexpr (std::is_floating_point<T>::value) y[1] = -x[1];
        else y[1] = (T)((typename std::make_unsigned<T>::type)0 - (typename std::make_unsigned<T>::type)x[1]);
    }
};
int launch_conj(int scalar, const void *in, void *out, size_t n, hipStream_t st)
{
    switch (scalar) {
    case PCX_F32: return launch_map<float, float, 2, 2>(in, out, n, ConjOp<float>{}, st);
    case PCX_F64: return launch_map<double, double, 2, 2>(in, out, n, ConjOp<double>{}, st);
    case PCX_I64: return launch_map<int64_t, int64_t, 2, 2>(in, out, n, ConjOp<int64_t>{}, st);
    case PCX_I32: return launch_map<int32_t, int32_t, 2, 2>(in, out, n, ConjOp<int32_t>{}, st);
    case PCX_I16: return launch_map<int16_t, int16_t, 2, 2>(in, out, n, ConjOp<int16_t>{}, st);
    case PCX_I8: return launch_map<int8_t, int8_t, 2, 2>(in, out, n, ConjOp<int8_t>{}, st);
    }
    set_error("conjugate: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

// ---- Abs  (math/Abs.cpp:40-43, FxptHelpers.hpp:36-49) ----
struct AbsRealF32 { __device__ void operator()(const float *x, float *y) const { y[0] = fabsf(x[0]); } };
struct AbsRealF64 { __device__ void operator()(const double *x, double *y) const { y[0] = fabs(x[0]); } };
// std::abs(std::complex<float>) = hypotf.  glibc 2.35's hypotf is
// (float)sqrt((double)x*x + (double)y*y) with inf/nan special cases; fp64 sqrt on
// the device is correctly rounded, so this reproduces it bit for bit.
struct AbsCplxF32 {
    __device__ void operator()(const float *x, float *y) const
    {
        const float a = x[0], b = x[1];
        if (isinf(a) || isinf(b)) { y[0] = INFINITY; return; }
        const double da = (double)a, db = (double)b;
        y[0] = (float)sqrt(da * da + db * db);
    }
};
struct AbsCplxF64 { __device__ void operator()(const double *x, double *y) const { y[0] = hypot(x[0], x[1]); } };
template <typename S>
struct AbsRealI {
    __device__ void operator()(const S *x, S *y) const
    {
        using U = typename std::make_unsigned<S>::type;
        const S v = x[0];
        y[0] = v < 0 ? (S)((U)0 - (U)v) : v;
    }
};
// complex integer: mag2 in the promoted type (int for int8/int16/int32 products,
// int64 for int64), OutType(std::sqrt(float(mag2))); float->int as cvttss2si
template <typename S>
struct AbsCplxI {
    __device__ void operator()(const S *x, S *y) const
    {
        using P = typename std::conditional<(sizeof(S) == 8), int64_t, int32_t>::type;
        using UP = typename std::make_unsigned<P>::type;
        const UP re = (UP)(P)x[0], im = (UP)(P)x[1];
        const P mag2 = (P)(re * re + im * im);
        const float r = sqrtf((float)mag2);
        P o;
        if (r != r) o = (sizeof(S) == 8) ? (P)INT64_MIN : (P)INT32_MIN;
        else o = (P)r;
        y[0] = (S)o;
    }
};
int launch_abs(int scalar, int is_complex, const void *in, void *out, size_t n, hipStream_t st)
{
    if (!is_complex) {
        switch (scalar) {
        case PCX_F32: return launch_map<float, float, 1, 1>(in, out, n, AbsRealF32{}, st);
        case PCX_F64: return launch_map<double, double, 1, 1>(in, out, n, AbsRealF64{}, st);
        case PCX_I64: return launch_map<int64_t, int64_t, 1, 1>(in, out, n, AbsRealI<int64_t>{}, st);
        case PCX_I32: return launch_map<int32_t, int32_t, 1, 1>(in, out, n, AbsRealI<int32_t>{}, st);
        case PCX_I16: return launch_map<int16_t, int16_t, 1, 1>(in, out, n, AbsRealI<int16_t>{}, st);
        case PCX_I8: return launch_map<int8_t, int8_t, 1, 1>(in, out, n, AbsRealI<int8_t>{}, st);
        }
    } else {
        switch (scalar) {
        case PCX_F32: return launch_map<float, float, 2, 1>(in, out, n, AbsCplxF32{}, st);
        case PCX_F64: return launch_map<double, double, 2, 1>(in, out, n, AbsCplxF64{}, st);
        case PCX_I64: return launch_map<int64_t, int64_t, 2, 1>(in, out, n, AbsCplxI<int64_t>{}, st);
        case PCX_I32: return launch_map<int32_t, int32_t, 2, 1>(in, out, n, AbsCplxI<int32_t>{}, st);
        case PCX_I16: return launch_map<int16_t, int16_t, 2, 1>(in, out, n, AbsCplxI<int16_t>{}, st);
        case PCX_I8: return launch_map<int8_t, int8_t, 2, 1>(in, out, n, AbsCplxI<int8_t>{}, st);
        }
    }
    set_error("abs: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

// ---- fxpt_atan2 on the device (functions/fxpt_atan2.cpp:36-138), integer-exact ----
__device__ inline int16_t d_s16_nabs(int16_t j)
{
    const int16_t negSign = (int16_t)~(j >> 15);
    return (int16_t)((j ^ negSign) - negSign);
}
__device__ inline int16_t d_q15_mul(int16_t j, int16_t k)
{
    const int32_t im = j * (int32_t)k;
    return (int16_t)((im + ((im & 0x7FFF) == 0x4000 ? 0 : 0x4000)) >> 15);
}
__device__ inline int16_t d_q15_div(int16_t numer, int16_t denom)
{
    return (int16_t)(((int32_t)((uint32_t)(int32_t)numer << 15)) / denom);
}
__device__ inline uint16_t d_fxpt_atan2(int16_t y, int16_t x)
{
    // q15_from_double(0.273/pi) = 2847, q15_from_double(0.25 + 0.273/pi) = 11039
    if (x == y) return y > 0 ? 8192 : (y < 0 ? 40960 : 0);
    const int16_t nabs_y = d_s16_nabs(y), nabs_x = d_s16_nabs(x);
    if (nabs_x < nabs_y) {
        const int16_t q = d_q15_div(y, x);
        const int16_t corr = d_q15_mul(2847, d_s16_nabs(q));
        const int16_t un = d_q15_mul((int16_t)(11039 + corr), q);
        return x > 0 ? (uint16_t)un : (uint16_t)(32768 + un);
    } else {
        const int16_t q = d_q15_div(x, y);
        const int16_t corr = d_q15_mul(2847, d_s16_nabs(q));
        const int16_t un = d_q15_mul((int16_t)(11039 + corr), q);
        return y > 0 ? (uint16_t)(16384 - un) : (uint16_t)(49152 - un);
    }
}

// ---- Angle  (math/Angle.cpp:23-26 via getAngle, FxptHelpers.hpp:14-29) ----
// the arithmetic FreqDemod shares; /comms/angle is the first "next" sibling block
template <typename T>
struct AngleOp {
    __device__ void operator()(const T *x, T *y) const
    {
        if constexpr (std::is_same<T, float>::value) y[0] = atan2f(x[1], x[0]);
        else if constexpr (std::is_same<T, double>::value) y[0] = atan2(x[1], x[0]);
        else y[0] = (T)d_fxpt_atan2((int16_t)x[1], (int16_t)x[0]);
    }
};
int launch_angle(int scalar, const void *in, void *out, size_t n, hipStream_t st)
{
    switch (scalar) {
    case PCX_F32: return launch_map<float, float, 2, 1>(in, out, n, AngleOp<float>{}, st);
    case PCX_F64: return launch_map<double, double, 2, 1>(in, out, n, AngleOp<double>{}, st);
    case PCX_I64: return launch_map<int64_t, int64_t, 2, 1>(in, out, n, AngleOp<int64_t>{}, st);
    case PCX_I32: return launch_map<int32_t, int32_t, 2, 1>(in, out, n, AngleOp<int32_t>{}, st);
    case PCX_I16: return launch_map<int16_t, int16_t, 2, 1>(in, out, n, AngleOp<int16_t>{}, st);
    case PCX_I8: return launch_map<int8_t, int8_t, 2, 1>(in, out, n, AngleOp<int8_t>{}, st);
    }
    set_error("angle: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

// ---- FreqDemod  (demod/FreqDemod.cpp:60-67) ----
// out[i] = angle(in[i] * _prev), _prev = conj(in[i-1]); for i = 0 _prev comes from
// *prev_in (carried from the previous call; zero after activate()); *prev_out gets the
// new _prev.  prev_in/prev_out are distinct device slots (ping-pong), so no lane of this
// launch can observe the update.  Each lane handles ITEMS consecutive
// samples from one 16-byte load plus one extra neighbour load (an L1/L2 hit).
// (a,b) = in[i], (c,d) = _prev = conj(in[i-1]) (value-initialised zero after activate)
template <typename T>
__device__ inline T demod_one(T a, T b, T c, T d);
template <>
__device__ inline float demod_one<float>(float a, float b, float c, float d)
{
    // in_i * _prev, unfused: re = a*c - b*d, im = a*d + b*c
    const float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    float re = ac - bd, im = ad + bc;
    if (__builtin_expect(both_nan(re, im), 0)) cmul_annex_g(a, b, c, d, re, im);   // operator*'s slow path (pcx_cplx.hpp)
    return atan2f(im, re);
}
template <>
__device__ inline double demod_one<double>(double a, double b, double c, double d)
{
    const double ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    double re = ac - bd, im = ad + bc;
    if (__builtin_expect(both_nan(re, im), 0)) cmul_annex_g(a, b, c, d, re, im);
    return atan2(im, re);
}
template <typename T>
__device__ inline T conj_im(T d)
{
    if constexpr (std::is_floating_point<T>::value) return -d;
    else return (T)((typename std::make_unsigned<T>::type)0 - (typename std::make_unsigned<T>::type)d);
}
template <typename S>
__device__ inline S demod_one_int(S a, S b, S c, S d)
{
    // complex<intN> product wraps modulo 2^N; getAngle truncates both parts to int16
    using U = typename std::conditional<(sizeof(S) == 8), uint64_t, uint32_t>::type;
    const U ua = (U)a, ub = (U)b, uc = (U)c, ud = (U)d;
    const S re = (S)(ua * uc - ub * ud), im = (S)(ua * ud + ub * uc);
    return (S)d_fxpt_atan2((int16_t)im, (int16_t)re);
}
template <>
__device__ inline int64_t demod_one<int64_t>(int64_t a, int64_t b, int64_t c, int64_t d) { return demod_one_int(a, b, c, d); }
template <>
__device__ inline int32_t demod_one<int32_t>(int32_t a, int32_t b, int32_t c, int32_t d) { return demod_one_int(a, b, c, d); }
template <>
__device__ inline int16_t demod_one<int16_t>(int16_t a, int16_t b, int16_t c, int16_t d) { return demod_one_int(a, b, c, d); }
template <>
__device__ inline int8_t demod_one<int8_t>(int8_t a, int8_t b, int8_t c, int8_t d) { return demod_one_int(a, b, c, d); }

template <typename T, int ITEMS>
__global__ __launch_bounds__(kBlock) void freqdemod_kernel(const T *__restrict__ in, T *__restrict__ out, size_t n,
                                                           const T *__restrict__ prev_in, T *__restrict__ prev_out)
{
    using VIn = Vec<T, 2 * ITEMS>;
    using VOut = Vec<T, ITEMS>;
    const size_t nvec = n / ITEMS;
    const size_t gtid = (size_t)blockIdx.x * kBlock + threadIdx.x, gstride = (size_t)gridDim.x * kBlock;
    const VIn *vin = reinterpret_cast<const VIn *>(in);
    VOut *vout = reinterpret_cast<VOut *>(out);
    // one vector: ITEMS outputs from ITEMS inputs and the sample before them (the previous lane's
    // last element: a second, scalar-sized read of a line the wave fetches anyway)
    auto one = [&](size_t i, const VIn &a, T c, T d) {
        VOut o;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            o.v[k] = demod_one<T>(a.v[2 * k], a.v[2 * k + 1], c, d);
            c = a.v[2 * k]; d = conj_im<T>(a.v[2 * k + 1]);
        }
        nt_store(&vout[i], o);
    };
    // kUnroll vectors per lane in flight, as in map_kernel
    const size_t chunk = (size_t)kBlock * kUnroll;
    const size_t nchunks = nvec / chunk;
    for (size_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const size_t base = ch * chunk + threadIdx.x;
        VIn a[kUnroll];
        T c[kUnroll], d[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const size_t i = base + (size_t)u * kBlock;
            a[u] = vin[i];
            if (i == 0) { c[u] = prev_in[0]; d[u] = prev_in[1]; }
            else { c[u] = in[2 * (i * ITEMS - 1)]; d[u] = conj_im<T>(in[2 * (i * ITEMS - 1) + 1]); }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) one(base + (size_t)u * kBlock, a[u], c[u], d[u]);
    }
    for (size_t i = nchunks * chunk + gtid; i < nvec; i += gstride) {
        const VIn a = vin[i];
        T c, d;
        if (i == 0) { c = prev_in[0]; d = prev_in[1]; }
        else { c = in[2 * (i * ITEMS - 1)]; d = conj_im<T>(in[2 * (i * ITEMS - 1) + 1]); }
        one(i, a, c, d);
    }
    for (size_t i = nvec * ITEMS + gtid; i < n; i += gstride) {
        T c, d;
        if (i == 0) { c = prev_in[0]; d = prev_in[1]; }
        else { c = in[2 * (i - 1)]; d = conj_im<T>(in[2 * (i - 1) + 1]); }
        out[i] = demod_one<T>(in[2 * i], in[2 * i + 1], c, d);
    }
    if (gtid == 0 && n > 0) { prev_out[0] = in[2 * (n - 1)]; prev_out[1] = conj_im<T>(in[2 * (n - 1) + 1]); }
}
template <typename T>
static int launch_freqdemod_t(const void *in, void *out, size_t n, const void *prev_in, void *prev_out, hipStream_t st)
{
    if (n == 0) return PCX_OK;
    constexpr int ITEMS = (2 * sizeof(T) >= 16) ? 1 : (int)(16 / (2 * sizeof(T)));
    const bool aligned = (reinterpret_cast<uintptr_t>(in) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % (sizeof(T) * ITEMS) == 0);
    if (aligned && ITEMS > 1) {
        const unsigned grid = stream_grid(n / ITEMS / kUnroll + 1, kBlock);
        hipLaunchKernelGGL((freqdemod_kernel<T, ITEMS>), dim3(grid), dim3(kBlock), 0, st, (const T *)in, (T *)out, n, (const T *)prev_in, (T *)prev_out);
    } else {
        const unsigned grid = stream_grid(n, kBlock);
        hipLaunchKernelGGL((freqdemod_kernel<T, 1>), dim3(grid), dim3(kBlock), 0, st, (const T *)in, (T *)out, n, (const T *)prev_in, (T *)prev_out);
    }
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}
// Zeroing a handle's carried state as a KERNEL, not as hipMemsetAsync: a reset captured into a hipGraph then replays correctly.  (With
// the memset node the first replay was right and later ones started from non-zero state once other work had run in between --
// torch's allocator and kernels in the same process; tools/graph_probe*.py.  A kernel node carries its own copy of its arguments.)
__global__ void zero_words_kernel(unsigned *p, unsigned nwords)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwords) p[i] = 0u;
}
// the halo gate of a sharded stream (pcx_sched.hpp Gate): queued on the stream that carried the halo, behind the transfer
__global__ void gate_signal_kernel(unsigned *word, unsigned value)
{
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int launch_gate_signal(void *gate_word, unsigned value, hipStream_t st)
{
    hipLaunchKernelGGL(gate_signal_kernel, dim3(1), dim3(1), 0, st, (unsigned *)gate_word, value);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

int launch_zero_words(void *p, size_t nwords, hipStream_t st)
{
    if (nwords == 0) return PCX_OK;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((nwords + 63) / 64)), dim3(64), 0, st, static_cast<unsigned *>(p), (unsigned)nwords);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

int launch_freqdemod(int scalar, const void *in, void *out, size_t n, const void *prev_in, void *prev_out, hipStream_t st)
{
    switch (scalar) {
    case PCX_F32: return launch_freqdemod_t<float>(in, out, n, prev_in, prev_out, st);
    case PCX_F64: return launch_freqdemod_t<double>(in, out, n, prev_in, prev_out, st);
    case PCX_I64: return launch_freqdemod_t<int64_t>(in, out, n, prev_in, prev_out, st);
    case PCX_I32: return launch_freqdemod_t<int32_t>(in, out, n, prev_in, prev_out, st);
    case PCX_I16: return launch_freqdemod_t<int16_t>(in, out, n, prev_in, prev_out, st);
    case PCX_I8: return launch_freqdemod_t<int8_t>(in, out, n, prev_in, prev_out, st);
    }
    set_error("freq_demod: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

// ---- synthetic stream generator (same hash as oracle orc_fill_uniform_f32) ----
__device__ inline uint64_t d_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void fill_uniform_kernel(float *dst, size_t n, uint64_t seed, uint64_t offset)
{
    const size_t gstride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += gstride) {
        const uint64_t h = d_splitmix64(seed * 0x100000001B3ull + offset + i);
        dst[i] = (float)((int32_t)(h >> 40) - (1 << 23)) * (1.0f / (float)(1 << 23));
    }
}
int launch_fill_uniform_f32(float *dst, size_t n, uint64_t seed, uint64_t offset, hipStream_t st)
{
    if (n == 0) return PCX_OK;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, st, dst, n, seed, offset);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// ---- shader clock probe ----
// One wave spins for `spin_us` of wall time and reports how many shader cycles went by: s_memtime counts shader-clock cycles on
// gfx950 (tools/clk_lab.hip: 2,397-2,424 per microsecond on an idle device), s_memrealtime the constant 100 MHz reference.  Queued
// on a stream of its own BESIDE a running workload it reads the clock that workload runs at -- the package power cap holds it far
// below the 2.4 GHz boost on real data -- which is what turns an instruction count into a share of SIMD time.
__global__ __launch_bounds__(64) void clock_probe_kernel(float *mhz, unsigned spin_ticks)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) *mhz = (float)(100.0 * (double)(t1 - t0) / (double)(r1 - r0));
}
int launch_clock_probe(float *mhz_dev, unsigned spin_us, hipStream_t st)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, st, mhz_dev, spin_us * 100u);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace pcx
