// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" exports over the reference sources that compile WITHOUT any
// PothosCore header: fft/kissfft.hh, fft/kiss_fft.c (-DFIXED_POINT=16, as
// fft/CMakeLists.txt:19-22 builds it), functions/fxpt_atan2.cpp and
// functions/FxptHelpers.hpp.  The reference files are compiled where they lie
// under /root/reference (see oracle/Makefile, target _ref); nothing of them is
// copied into this repository.  The resulting oracle/_ref/libpcx_ref.so pins
// oracle/pcx_oracle.c bit-for-bit and may serve as bench.py's "reference" CPU
// baseline for the FFT row.
//
// The block files themselves (filter/FIRFilter.cpp, fft/FFT.cpp,
// demod/FreqDemod.cpp, math/{Rotate,Scale,Abs,Conjugate}.cpp) include
// <Pothos/Framework.hpp>, which this image does not have: they are
// unbuildable here and are restated in pcx_oracle.c instead.
#include <complex>
#include <cstddef>
#include <cstdint>

#include "kissfft.hh"      // -I/root/reference/fft
#include "kiss_fft.h"      // -I/root/reference/fft   (FIXED_POINT=16 -> int16)
#include "FxptHelpers.hpp" // -I/root/reference/functions

#define REF_EXPORT extern "C" __attribute__((visibility("default")))

// FFTAux<std::complex<float>>::transform  (fft/FFTAux.h:16-27)
REF_EXPORT int ref_kissfft_f32(int nfft, int inverse, const float *in, float *out, size_t nframes)
{
    kissfft<float> fft(nfft, inverse != 0);
    for (size_t f = 0; f < nframes; f++)
        fft.transform(reinterpret_cast<const std::complex<float> *>(in) + f * nfft,
                      reinterpret_cast<std::complex<float> *>(out) + f * nfft);
    return 0;
}
REF_EXPORT int ref_kissfft_f64(int nfft, int inverse, const double *in, double *out, size_t nframes)
{
    kissfft<double> fft(nfft, inverse != 0);
    for (size_t f = 0; f < nframes; f++)
        fft.transform(reinterpret_cast<const std::complex<double> *>(in) + f * nfft,
                      reinterpret_cast<std::complex<double> *>(out) + f * nfft);
    return 0;
}
// FFTAux<std::complex<kiss_fft_scalar>>::transform  (fft/FFTAux.h:29-48)
REF_EXPORT int ref_kiss_fft_i16(int nfft, int inverse, const int16_t *in, int16_t *out, size_t nframes)
{
    static_assert(sizeof(kiss_fft_scalar) == 2, "reference builds kiss_fft with FIXED_POINT=16");
    kiss_fft_cfg cfg = kiss_fft_alloc(nfft, inverse, nullptr, nullptr);
    if (!cfg) return -1;
    for (size_t f = 0; f < nframes; f++)
        kiss_fft(cfg, reinterpret_cast<const kiss_fft_cpx *>(in) + f * nfft,
                 reinterpret_cast<kiss_fft_cpx *>(out) + f * nfft);
    kiss_fft_free(cfg);
    return 0;
}

REF_EXPORT uint16_t ref_fxpt_atan2(int16_t y, int16_t x) { return fxpt_atan2(y, x); }

// getAngle / getAbs templates, one export per instantiation the factories use
// (demod/FreqDemod.cpp:80-93, math/Abs.cpp:107-122)
#define REF_ANGLE(NAME, T)                                                                  \
    REF_EXPORT void NAME(const T *in, T *out, size_t n)                                     \
    {                                                                                       \
        for (size_t i = 0; i < n; i++) out[i] = T(getAngle(std::complex<T>(in[2 * i], in[2 * i + 1]))); \
    }
REF_ANGLE(ref_angle_f64, double)
REF_ANGLE(ref_angle_f32, float)
REF_ANGLE(ref_angle_i64, int64_t)
REF_ANGLE(ref_angle_i32, int32_t)
REF_ANGLE(ref_angle_i16, int16_t)
REF_ANGLE(ref_angle_i8, int8_t)

#define REF_ABS(NAME, T)                                                                    \
    REF_EXPORT void NAME##_real(const T *in, T *out, size_t n)                              \
    {                                                                                       \
        for (size_t i = 0; i < n; i++) out[i] = getAbs<T>(in[i]);                           \
    }                                                                                       \
    REF_EXPORT void NAME##_cplx(const T *in, T *out, size_t n)                              \
    {                                                                                       \
        for (size_t i = 0; i < n; i++) out[i] = getAbs<T>(std::complex<T>(in[2 * i], in[2 * i + 1])); \
    }
REF_ABS(ref_abs_f64, double)
REF_ABS(ref_abs_f32, float)
REF_ABS(ref_abs_i64, int64_t)
REF_ABS(ref_abs_i32, int32_t)
REF_ABS(ref_abs_i16, int16_t)
REF_ABS(ref_abs_i8, int8_t)

// ---- /comms/arithmetic: the element operators themselves ----
// math/Arithmetic.cpp:70-110 is `out[i] = in0[i] OP in1[i]` on `Type` = T or std::complex<T>; the file
// needs <Pothos/Framework.hpp> and cannot be compiled here, but the operators it applies are the
// toolchain's (C++ arithmetic conversions, libstdc++ <complex>, libgcc), and math/TestArithmeticBlocks.cpp
// :146-151,201-206 computes its own expected complex outputs with the very same expressions.  These
// exports evaluate those expressions so that oracle/pcx_oracle.c's C restatement of the operators can
// be checked bit for bit and the golden vectors of the reference test generated.
template <typename Type>
static void std_arith(int op, const void *a, const void *b, void *o, size_t n)
{
    const Type *in0 = static_cast<const Type *>(a), *in1 = static_cast<const Type *>(b);
    Type *out = static_cast<Type *>(o);
    for (size_t i = 0; i < n; i++) {
        switch (op) {
        case 0: out[i] = in0[i] + in1[i]; break;
        case 1: out[i] = in0[i] - in1[i]; break;
        case 2: out[i] = in0[i] * in1[i]; break;
        default: out[i] = in0[i] / in1[i]; break;   // the caller keeps zero divisors out
        }
    }
}
// st: include/pcx.h scalar codes (0 f64, 1 f32, 2..5 i64..i8, 6..9 u64..u8)
REF_EXPORT int ref_std_arith(int st, int is_complex, int op, const void *in0, const void *in1, void *out, size_t n)
{
#define REF_ARITH_CASE(CODE, T)                                                       \
    case CODE:                                                                        \
        if (is_complex) std_arith<std::complex<T>>(op, in0, in1, out, n);             \
        else std_arith<T>(op, in0, in1, out, n);                                      \
        return 0;
    switch (st) {
        REF_ARITH_CASE(0, double)
        REF_ARITH_CASE(1, float)
        REF_ARITH_CASE(2, int64_t)
        REF_ARITH_CASE(3, int32_t)
        REF_ARITH_CASE(4, int16_t)
        REF_ARITH_CASE(5, int8_t)
        REF_ARITH_CASE(6, uint64_t)
        REF_ARITH_CASE(7, uint32_t)
        REF_ARITH_CASE(8, uint16_t)
        REF_ARITH_CASE(9, uint8_t)
    }
#undef REF_ARITH_CASE
    return -1;
}
