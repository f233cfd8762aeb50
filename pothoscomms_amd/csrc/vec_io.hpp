// vec_io.hpp -- 16-byte lane vectors and non-temporal access for the HBM-bound element-wise kernels
// (elementwise.hip, arith.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace pcx {

template <typename T, int N>
struct alignas(sizeof(T) * N) Vec {
    T v[N];
};

// The maps touch every byte exactly once: loads and stores carry the non-temporal hint (measured on
// MI355X, tools/ubench.hip: copy 5.4 -> 5.8 TB/s with nt).  The builtins take scalar / ext-vector
// types, so a Vec goes through a same-sized integer vector.
template <int BYTES> struct RawVec;
template <> struct RawVec<1> { typedef unsigned char type; };
template <> struct RawVec<2> { typedef unsigned short type; };
template <> struct RawVec<4> { typedef unsigned int type; };
template <> struct RawVec<8> { typedef unsigned int type __attribute__((ext_vector_type(2))); };
template <> struct RawVec<16> { typedef unsigned int type __attribute__((ext_vector_type(4))); };
template <typename V>
__device__ __forceinline__ V nt_load(const V *p)
{
    static_assert(sizeof(V) <= 16, "vector wider than one 16-byte access");
    typedef typename RawVec<sizeof(V)>::type R;
    const R r = __builtin_nontemporal_load(reinterpret_cast<const R *>(p));
    V v;
    __builtin_memcpy(&v, &r, sizeof(V));
    return v;
}
template <typename V>
__device__ __forceinline__ void nt_store(V *p, const V &v)
{
    static_assert(sizeof(V) <= 16, "vector wider than one 16-byte access");
    typedef typename RawVec<sizeof(V)>::type R;
    R r;
    __builtin_memcpy(&r, &v, sizeof(V));
    __builtin_nontemporal_store(r, reinterpret_cast<R *>(p));
}

}  // namespace pcx
