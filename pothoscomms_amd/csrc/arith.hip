// arith.hip -- /comms/arithmetic, /comms/split_complex, /comms/combine_complex:
// two-stream element-wise kernels on the same 16-byte-vector, non-temporal, grid-stride scheme as
// elementwise.hip.  HBM-bound: 3 streams of sizeof(element) each (24 B per cf32 element).
//
// Element operators = the C++ operators math/Arithmetic.cpp:70-110 applies (`out[i] = in0[i] OP in1[i]`),
// restated for the device type by type -- see oracle/pcx_oracle.c (orc_arith) for the derivation and
// the cross-check against std::complex:
//   * integers narrower than int promote to int, 32/64-bit + - * wrap; results narrow on the store;
//   * x / 0 (a trap in the reference) yields 0, INT_MIN / -1 (undefined there) yields INT_MIN;
//   * complex<integer>: libstdc++'s generic operator*= / operator/= including the narrowing of r and
//     norm(z) to T and the un-narrowed numerator of the new imaginary part;
//   * complex<float> *: (ac - bd, ad + bc), every product and sum rounded (this TU is built with
//     -ffp-contract=off); complex<float> /: evaluated in double, rounded once (libgcc_s __divsc3);
//     complex<double> /: Smith's ratio form (__divdc3's path for operands in the normal range).
// Float results are bit-identical to the reference for finite, normal-range operands; the parity bar
// is 1e-5 relative.
#include "pcx_cplx.hpp"
#include "pcx_internal.hpp"
#include "vec_io.hpp"

#include <type_traits>

namespace pcx {

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 2;   // two input streams: 2 x 2 x 16 B in flight per lane

// ---- promoted / division types per element type (see the header comment) ----
template <typename T> struct Prom {
    typedef typename std::conditional<(sizeof(T) < 4), int, typename std::make_unsigned<T>::type>::type P;   // + - *
    typedef typename std::conditional<(sizeof(T) < 4), int, T>::type D;                                      // /
};
template <typename D>
__device__ __forceinline__ D int_div(D a, D b)
{
    if (b == 0) return 0;
    if (std::is_signed<D>::value && sizeof(D) >= 4 && b == (D)-1) return (D)(0 - (typename std::make_unsigned<D>::type)a);   // wraps at MIN
    return (D)(a / b);
}

template <typename T, int OP, bool FLT = std::is_floating_point<T>::value>
struct RealOp;
template <typename T, int OP>
struct RealOp<T, OP, true> {
    __device__ void operator()(const T *a, const T *b, T *o) const
    {
        o[0] = OP == PCX_ARITH_ADD ? a[0] + b[0] : OP == PCX_ARITH_SUB ? a[0] - b[0] : OP == PCX_ARITH_MUL ? a[0] * b[0] : a[0] / b[0];
    }
};
template <typename T, int OP>
struct RealOp<T, OP, false> {
    __device__ void operator()(const T *a, const T *b, T *o) const
    {
        typedef typename Prom<T>::P P;
        typedef typename Prom<T>::D D;
        const P x = (P)a[0], y = (P)b[0];
        if (OP == PCX_ARITH_ADD) o[0] = (T)(P)(x + y);
        else if (OP == PCX_ARITH_SUB) o[0] = (T)(P)(x - y);
        else if (OP == PCX_ARITH_MUL) o[0] = (T)(P)(x * y);
        else o[0] = (T)int_div<D>((D)a[0], (D)b[0]);
    }
};

template <typename T, int OP, bool FLT = std::is_floating_point<T>::value>
struct CplxOp;
// complex<integer>: libstdc++ generic members
template <typename T, int OP>
struct CplxOp<T, OP, false> {
    __device__ void operator()(const T *a, const T *b, T *o) const
    {
        typedef typename Prom<T>::P P;
        typedef typename Prom<T>::D D;
        const P ar = (P)a[0], ai = (P)a[1], br = (P)b[0], bi = (P)b[1];
        if (OP == PCX_ARITH_ADD) { o[0] = (T)(P)(ar + br); o[1] = (T)(P)(ai + bi); }
        else if (OP == PCX_ARITH_SUB) { o[0] = (T)(P)(ar - br); o[1] = (T)(P)(ai - bi); }
        else if (OP == PCX_ARITH_MUL) { o[0] = (T)(P)(ar * br - ai * bi); o[1] = (T)(P)(ar * bi + ai * br); }
        else {
            const T r = (T)(P)(ar * br + ai * bi);
            const T nn = (T)(P)(br * br + bi * bi);
            const P num = (P)(ai * br - ar * bi);     // stays in the promoted type
            o[1] = (T)int_div<D>((D)num, (D)nn);
            o[0] = (T)int_div<D>((D)r, (D)nn);
        }
    }
};
template <int OP>
struct CplxOp<float, OP, true> {
    __device__ void operator()(const float *a, const float *b, float *o) const
    {
        if (OP == PCX_ARITH_ADD) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; }
        else if (OP == PCX_ARITH_SUB) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; }
        else if (OP == PCX_ARITH_MUL) {
            const float ac = a[0] * b[0], bd = a[1] * b[1], ad = a[0] * b[1], bc = a[1] * b[0];
            float x = ac - bd, y = ad + bc;
            if (both_nan(x, y)) cmul_annex_g(a[0], a[1], b[0], b[1], x, y);   // __mulsc3's slow path
            o[0] = x; o[1] = y;
        } else {
            const double aa = a[0], bb = a[1], cc = b[0], dd = b[1];
            const double den = (cc * cc) + (dd * dd);
            float x = (float)(((aa * cc) + (bb * dd)) / den);
            float y = (float)(((bb * cc) - (aa * dd)) / den);
            if (both_nan(x, y)) cdiv_annex_g(a[0], a[1], b[0], b[1], x, y);   // __divsc3's
            o[0] = x; o[1] = y;
        }
    }
};
template <int OP>
struct CplxOp<double, OP, true> {
    __device__ void operator()(const double *x, const double *y, double *o) const
    {
        double a = x[0], b = x[1], c = y[0], d = y[1];
        if (OP == PCX_ARITH_ADD) { o[0] = a + c; o[1] = b + d; }
        else if (OP == PCX_ARITH_SUB) { o[0] = a - c; o[1] = b - d; }
        else if (OP == PCX_ARITH_MUL) {
            const double ac = a * c, bd = b * d, ad = a * d, bc = b * c;
            double x = ac - bd, y = ad + bc;
            if (both_nan(x, y)) cmul_annex_g(a, b, c, d, x, y);   // __muldc3's slow path
            o[0] = x; o[1] = y;
        } else {
            // libgcc's __divdc3 as shipped since GCC 12 (what the reference's operator/ calls; restated here and checked
            // bit for bit against this box's libgcc on 4 M wide-range and special operands): Smith's division with the
            // operands rescaled when the denominator is huge or tiny or a numerator part would underflow, and the other
            // order of operations when the ratio itself is subnormal
            constexpr double kBig = 1.7976931348623157e308 / 2, kMin = 2.2250738585072014e-308, kMin2 = 2.220446049250313e-16,
                             kScale = 1.0 / 2.220446049250313e-16, kMax2 = kBig * kMin2;
            double aa = a, bb = b, cc = c, dd = d;
            const bool first = fabs(cc) < fabs(dd);
            const double m = first ? fabs(dd) : fabs(cc);
            double f = 1.0;
            if (m >= kBig) f = 0.5;
            else if (m < kMin2) f = kScale;
            else if ((fabs(aa) < kMin && fabs(bb) < kMax2 && m < kMax2) || (fabs(bb) < kMin && fabs(aa) < kMax2 && m < kMax2)) f = kScale;
            aa *= f; bb *= f; cc *= f; dd *= f;
            double x, y;
            if (first) {
                const double ratio = cc / dd, den = (cc * ratio) + dd;
                if (fabs(ratio) > kMin) { x = ((aa * ratio) + bb) / den; y = ((bb * ratio) - aa) / den; }
                else { x = ((cc * (aa / dd)) + bb) / den; y = ((cc * (bb / dd)) - aa) / den; }
            } else {
                const double ratio = dd / cc, den = (dd * ratio) + cc;
                if (fabs(ratio) > kMin) { x = ((bb * ratio) + aa) / den; y = (bb - (aa * ratio)) / den; }
                else { x = ((dd * (bb / cc)) + aa) / den; y = (bb - (dd * (aa / cc))) / den; }
            }
            a = aa; b = bb; c = cc; d = dd;     // the slow path below sees the rescaled operands, as libgcc's does
            if (both_nan(x, y)) cdiv_annex_g(a, b, c, d, x, y);   // __divdc3's
            o[0] = x; o[1] = y;
        }
    }
};

// PER scalars per element, ITEMS elements per 16-byte lane vector
template <typename T, int PER, int ITEMS, typename Op>
__global__ __launch_bounds__(kBlock) void map2_kernel(const T *in0, const T *in1, T *out, size_t nitems, Op op)
{
    // `out` may alias in0 or in1 exactly (the reference forwards input 0's buffer to the output
    // and folds further ports in place, Arithmetic.cpp:157-158,217-224): every lane reads its own
    // elements before it writes them, no lane touches another's
    using V = Vec<T, PER * ITEMS>;
    const size_t nvec = nitems / ITEMS;
    const size_t chunk = (size_t)kBlock * kUnroll;
    const size_t nchunks = nvec / chunk;
    const V *va = reinterpret_cast<const V *>(in0), *vb = reinterpret_cast<const V *>(in1);
    V *vo = reinterpret_cast<V *>(out);
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        V a[kUnroll], b[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            a[u] = nt_load(&va[base + (size_t)u * kBlock]);
            b[u] = nt_load(&vb[base + (size_t)u * kBlock]);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            V o;
#pragma unroll
            for (int k = 0; k < ITEMS; k++) op(&a[u].v[k * PER], &b[u].v[k * PER], &o.v[k * PER]);
            nt_store(&vo[base + (size_t)u * kBlock], o);
        }
    }
    const size_t gtid = (size_t)blockIdx.x * kBlock + threadIdx.x, gstride = (size_t)gridDim.x * kBlock;
    for (size_t i = nchunks * chunk + gtid; i < nvec; i += gstride) {
        const V a = va[i], b = vb[i];
        V o;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) op(&a.v[k * PER], &b.v[k * PER], &o.v[k * PER]);
        vo[i] = o;
    }
    for (size_t i = nvec * ITEMS + gtid; i < nitems; i += gstride) {
        T t[PER];
        op(in0 + i * PER, in1 + i * PER, t);
#pragma unroll
        for (int k = 0; k < PER; k++) out[i * PER + k] = t[k];
    }
}

template <typename T, int PER, typename Op>
int launch_map2(const void *in0, const void *in1, void *out, size_t nitems, Op op, hipStream_t st)
{
    constexpr int item_bytes = (int)sizeof(T) * PER;
    constexpr int ITEMS = item_bytes >= 16 ? 1 : 16 / item_bytes;
    const uintptr_t al = reinterpret_cast<uintptr_t>(in0) | reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(out);
    const T *a = static_cast<const T *>(in0), *b = static_cast<const T *>(in1);
    T *o = static_cast<T *>(out);
    if (ITEMS > 1 && al % 16 == 0) {
        const unsigned grid = stream_grid(nitems / ITEMS / kUnroll + 1, kBlock);
        hipLaunchKernelGGL((map2_kernel<T, PER, ITEMS, Op>), dim3(grid), dim3(kBlock), 0, st, a, b, o, nitems, op);
    } else {
        const unsigned grid = stream_grid(nitems / kUnroll + 1, kBlock);
        hipLaunchKernelGGL((map2_kernel<T, PER, 1, Op>), dim3(grid), dim3(kBlock), 0, st, a, b, o, nitems, op);
    }
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

template <typename T>
int launch_arith_t(int is_complex, int op, const void *a, const void *b, void *o, size_t n, hipStream_t st)
{
    if (is_complex) {
        switch (op) {
        case PCX_ARITH_ADD: return launch_map2<T, 2>(a, b, o, n, CplxOp<T, PCX_ARITH_ADD>{}, st);
        case PCX_ARITH_SUB: return launch_map2<T, 2>(a, b, o, n, CplxOp<T, PCX_ARITH_SUB>{}, st);
        case PCX_ARITH_MUL: return launch_map2<T, 2>(a, b, o, n, CplxOp<T, PCX_ARITH_MUL>{}, st);
        default: return launch_map2<T, 2>(a, b, o, n, CplxOp<T, PCX_ARITH_DIV>{}, st);
        }
    }
    switch (op) {
    case PCX_ARITH_ADD: return launch_map2<T, 1>(a, b, o, n, RealOp<T, PCX_ARITH_ADD>{}, st);
    case PCX_ARITH_SUB: return launch_map2<T, 1>(a, b, o, n, RealOp<T, PCX_ARITH_SUB>{}, st);
    case PCX_ARITH_MUL: return launch_map2<T, 1>(a, b, o, n, RealOp<T, PCX_ARITH_MUL>{}, st);
    default: return launch_map2<T, 1>(a, b, o, n, RealOp<T, PCX_ARITH_DIV>{}, st);
    }
}

// ---- split / combine: W-byte scalars, ITEMS = 16 / W complex elements per lane: two 16-byte
// vectors of interleaved input <-> one 16-byte vector per plane ----
template <typename W, bool SPLIT>
__global__ __launch_bounds__(kBlock) void planes_kernel(const W *inter_in, W *inter_out, const W *re_in, const W *im_in, W *re_out,
                                                       W *im_out, size_t n, bool vec)
{
    constexpr int ITEMS = 16 / (int)sizeof(W);
    using VP = Vec<W, ITEMS>;         // one plane vector
    const size_t gtid = (size_t)blockIdx.x * kBlock + threadIdx.x, gstride = (size_t)gridDim.x * kBlock;
    const size_t nvec = vec ? n / ITEMS : 0;
    constexpr int U = 2;    // plane vectors per lane and iteration: 4 x 16 B in flight
    const size_t chunk = (size_t)kBlock * U, nchunks = nvec / chunk;
    auto split1 = [&](const VP &lo, const VP &hi, size_t i) {
        VP re, im;
#pragma unroll
        for (int k = 0; k < ITEMS / 2; k++) {
            re.v[k] = lo.v[2 * k]; im.v[k] = lo.v[2 * k + 1];
            re.v[ITEMS / 2 + k] = hi.v[2 * k]; im.v[ITEMS / 2 + k] = hi.v[2 * k + 1];
        }
        nt_store(reinterpret_cast<VP *>(re_out) + i, re);
        nt_store(reinterpret_cast<VP *>(im_out) + i, im);
    };
    auto combine1 = [&](const VP &re, const VP &im, size_t i) {
        VP lo, hi;
#pragma unroll
        for (int k = 0; k < ITEMS / 2; k++) {
            lo.v[2 * k] = re.v[k]; lo.v[2 * k + 1] = im.v[k];
            hi.v[2 * k] = re.v[ITEMS / 2 + k]; hi.v[2 * k + 1] = im.v[ITEMS / 2 + k];
        }
        nt_store(reinterpret_cast<VP *>(inter_out) + 2 * i, lo);
        nt_store(reinterpret_cast<VP *>(inter_out) + 2 * i + 1, hi);
    };
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        VP a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * kBlock;
            if (SPLIT) { a[u] = nt_load(reinterpret_cast<const VP *>(inter_in) + 2 * i); b[u] = nt_load(reinterpret_cast<const VP *>(inter_in) + 2 * i + 1); }
            else { a[u] = nt_load(reinterpret_cast<const VP *>(re_in) + i); b[u] = nt_load(reinterpret_cast<const VP *>(im_in) + i); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * kBlock;
            if (SPLIT) split1(a[u], b[u], i); else combine1(a[u], b[u], i);
        }
    }
    for (size_t i = nchunks * chunk + gtid; i < nvec; i += gstride) {
        if (SPLIT) split1(nt_load(reinterpret_cast<const VP *>(inter_in) + 2 * i), nt_load(reinterpret_cast<const VP *>(inter_in) + 2 * i + 1), i);
        else combine1(nt_load(reinterpret_cast<const VP *>(re_in) + i), nt_load(reinterpret_cast<const VP *>(im_in) + i), i);
    }
    for (size_t i = nvec * ITEMS + gtid; i < n; i += gstride) {
        if (SPLIT) { re_out[i] = inter_in[2 * i]; im_out[i] = inter_in[2 * i + 1]; }
        else { inter_out[2 * i] = re_in[i]; inter_out[2 * i + 1] = im_in[i]; }
    }
}
template <typename W>
int launch_planes(bool split, const void *inter_in, void *inter_out, const void *re_in, const void *im_in, void *re_out, void *im_out,
                  size_t n, hipStream_t st)
{
    const uintptr_t al = reinterpret_cast<uintptr_t>(inter_in) | reinterpret_cast<uintptr_t>(inter_out) | reinterpret_cast<uintptr_t>(re_in) |
                         reinterpret_cast<uintptr_t>(im_in) | reinterpret_cast<uintptr_t>(re_out) | reinterpret_cast<uintptr_t>(im_out);
    const bool vec = al % 16 == 0;
    const unsigned grid = stream_grid(vec ? n / (16 / sizeof(W)) / 2 + 1 : n, kBlock);
    if (split)
        hipLaunchKernelGGL((planes_kernel<W, true>), dim3(grid), dim3(kBlock), 0, st, (const W *)inter_in, (W *)nullptr, (const W *)nullptr,
                           (const W *)nullptr, (W *)re_out, (W *)im_out, n, vec);
    else
        hipLaunchKernelGGL((planes_kernel<W, false>), dim3(grid), dim3(kBlock), 0, st, (const W *)nullptr, (W *)inter_out, (const W *)re_in,
                           (const W *)im_in, (W *)nullptr, (W *)nullptr, n, vec);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}
int launch_planes_w(int width, bool split, const void *ii, void *io, const void *ri, const void *mi, void *ro, void *mo, size_t n, hipStream_t st)
{
    switch (width) {
    case 8: return launch_planes<uint64_t>(split, ii, io, ri, mi, ro, mo, n, st);
    case 4: return launch_planes<uint32_t>(split, ii, io, ri, mi, ro, mo, n, st);
    case 2: return launch_planes<uint16_t>(split, ii, io, ri, mi, ro, mo, n, st);
    case 1: return launch_planes<uint8_t>(split, ii, io, ri, mi, ro, mo, n, st);
    }
    set_error("split/combine complex: unsupported scalar width %d", width);
    return PCX_ERR_ARG;
}

}  // namespace

int launch_arith(int scalar, int is_complex, int op, const void *in0, const void *in1, void *out, size_t n, hipStream_t st)
{
    if (n == 0) return PCX_OK;
    switch (scalar) {
    case PCX_F64: return launch_arith_t<double>(is_complex, op, in0, in1, out, n, st);
    case PCX_F32: return launch_arith_t<float>(is_complex, op, in0, in1, out, n, st);
    case PCX_I64: return launch_arith_t<int64_t>(is_complex, op, in0, in1, out, n, st);
    case PCX_I32: return launch_arith_t<int32_t>(is_complex, op, in0, in1, out, n, st);
    case PCX_I16: return launch_arith_t<int16_t>(is_complex, op, in0, in1, out, n, st);
    case PCX_I8: return launch_arith_t<int8_t>(is_complex, op, in0, in1, out, n, st);
    case PCX_U64: return launch_arith_t<uint64_t>(is_complex, op, in0, in1, out, n, st);
    case PCX_U32: return launch_arith_t<uint32_t>(is_complex, op, in0, in1, out, n, st);
    case PCX_U16: return launch_arith_t<uint16_t>(is_complex, op, in0, in1, out, n, st);
    case PCX_U8: return launch_arith_t<uint8_t>(is_complex, op, in0, in1, out, n, st);
    }
    set_error("arithmetic: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}
int launch_split_complex(int scalar, const void *in, void *re, void *im, size_t n, hipStream_t st)
{
    if (n == 0) return PCX_OK;
    return launch_planes_w(scalar_bytes(scalar), true, in, nullptr, nullptr, nullptr, re, im, n, st);
}
int launch_combine_complex(int scalar, const void *re, const void *im, void *out, size_t n, hipStream_t st)
{
    if (n == 0) return PCX_OK;
    return launch_planes_w(scalar_bytes(scalar), false, nullptr, out, re, im, nullptr, nullptr, n, st);
}

}  // namespace pcx
