// pcx_cplx.hpp -- the slow paths of complex multiply and divide as the reference's compiler emits them.
//
// std::complex<float/double> operator* and operator/ compile to the plain formulas PLUS, when both parts of the result
// are NaN, a call into libgcc (__mulsc3 / __muldc3 / __divsc3 / __divdc3), which recovers the infinities C99 Annex G asks
// for: (inf + i nan) * (2 + 3i) is (inf + i inf), not (nan + i nan).  math/Rotate.cpp:20, math/Arithmetic.cpp:70-110 and
// the complex-taps convolution of filter/FIRFilter.cpp:298 all go through them.  The plain formulas stay in the kernels (same
// operations, same roundings; -ffp-contract=off); these functions are what a kernel runs once it sees NaN + i NaN -- inlined
// into a cold branch: a real call would put the whole kernel under the function-call ABI (measured on /comms/rotate: -8 %).
// The algorithm is the one printed in C99 G.5.1 (and followed by libgcc2.c); it only ever produces 0, inf or NaN parts.
#pragma once
#include <hip/hip_runtime.h>

namespace pcx {

template <typename T>
__device__ __forceinline__ bool both_nan(T x, T y) { return x != x && y != y; }

template <typename T>
__device__ __forceinline__ void cmul_annex_g(T a, T b, T c, T d, T &x, T &y)
{
    const T ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    const T inf = (T)__builtin_huge_valf();
    bool recalc = false;
    if (__builtin_isinf(a) || __builtin_isinf(b)) {
        a = __builtin_copysign(__builtin_isinf(a) ? (T)1 : (T)0, a);
        b = __builtin_copysign(__builtin_isinf(b) ? (T)1 : (T)0, b);
        if (c != c) c = __builtin_copysign((T)0, c);
        if (d != d) d = __builtin_copysign((T)0, d);
        recalc = true;
    }
    if (__builtin_isinf(c) || __builtin_isinf(d)) {
        c = __builtin_copysign(__builtin_isinf(c) ? (T)1 : (T)0, c);
        d = __builtin_copysign(__builtin_isinf(d) ? (T)1 : (T)0, d);
        if (a != a) a = __builtin_copysign((T)0, a);
        if (b != b) b = __builtin_copysign((T)0, b);
        recalc = true;
    }
    if (!recalc && (__builtin_isinf(ac) || __builtin_isinf(bd) || __builtin_isinf(ad) || __builtin_isinf(bc))) {
        if (a != a) a = __builtin_copysign((T)0, a);
        if (b != b) b = __builtin_copysign((T)0, b);
        if (c != c) c = __builtin_copysign((T)0, c);
        if (d != d) d = __builtin_copysign((T)0, d);
        recalc = true;
    }
    if (recalc) {
        x = inf * (a * c - b * d);
        y = inf * (a * d + b * c);
    }
}

// (a + ib) / (c + id) came out NaN + i NaN
template <typename T>
__device__ __forceinline__ void cdiv_annex_g(T a, T b, T c, T d, T &x, T &y)
{
    const T inf = (T)__builtin_huge_valf();
    if (c == (T)0 && d == (T)0 && (a == a || b == b)) {
        x = __builtin_copysign(inf, c) * a;
        y = __builtin_copysign(inf, c) * b;
    } else if ((__builtin_isinf(a) || __builtin_isinf(b)) && __builtin_isfinite(c) && __builtin_isfinite(d)) {
        a = __builtin_copysign(__builtin_isinf(a) ? (T)1 : (T)0, a);
        b = __builtin_copysign(__builtin_isinf(b) ? (T)1 : (T)0, b);
        x = inf * (a * c + b * d);
        y = inf * (b * c - a * d);
    } else if ((__builtin_isinf(c) || __builtin_isinf(d)) && __builtin_isfinite(a) && __builtin_isfinite(b)) {
        c = __builtin_copysign(__builtin_isinf(c) ? (T)1 : (T)0, c);
        d = __builtin_copysign(__builtin_isinf(d) ? (T)1 : (T)0, d);
        x = (T)0 * (a * c + b * d);
        y = (T)0 * (b * c - a * d);
    }
}

}  // namespace pcx
