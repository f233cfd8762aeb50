// pcx_qformat.hpp -- Pothos::Util::floatToQ / fromQ for integer Q types, as a PARAMETER (include/pcx.h, pcx_qformat).
//
// The header that defines the two functions (PothosCore, include/Pothos/Util/QFormat.hpp) is not under /root/reference; the call
// sites are filter/FIRFilter.cpp:300,348, math/Rotate.cpp:21,74, math/Scale.cpp:21,73, and the reference's own tests leave
// twelve readings standing (profiles/r02/qformat_enumeration.txt).  kDefaultQFormat is the one the library uses when nobody says
// otherwise: THE line to change the day the header is read.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "pcx.h"

namespace pcx {

struct QFormat {
    int frac, to, from;   // pcx_q_frac, pcx_q_to, pcx_q_from
};
constexpr QFormat kDefaultQFormat{PCX_Q_FRAC_HALF_Q, PCX_Q_TRUNCATE, PCX_Q_FLOOR};

// what a kernel needs of it: fromQ = an arithmetic shift by `shift` bits under rounding `mode` (pcx_q_from)
struct QShift {
    int shift, mode;
};

// fromQ on a value already wrapped to the signed Q type.  FLOOR: q >> n.  TOWARD_ZERO: the C++ quotient q / 2^n.  ROUND:
// floor((q + 2^(n-1)) / 2^n) evaluated without the intermediate overflow.  1 <= n < bits(Q).
template <typename Q>
__host__ __device__ __forceinline__ Q from_q_bits(Q q, QShift s)
{
    using U = typename std::make_unsigned<Q>::type;
    const Q fl = (Q)(q >> s.shift);
    if (s.mode == PCX_Q_FLOOR) return fl;
    const U rem = (U)((U)q & (U)(((U)1 << s.shift) - 1));
    if (s.mode == PCX_Q_TOWARD_ZERO) return (Q)(fl + (Q)((q < 0) && rem != 0));
    return (Q)(fl + (Q)(rem >= (U)((U)1 << (s.shift - 1))));
}

}  // namespace pcx
