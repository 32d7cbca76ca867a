// Scalar helpers shared by the kernels: float64 and a complex128 type whose abs / max /
// sqrt follow NumPy's rules, because the reference's default Jacobian-vector product is the
// complex step Im R(Q + i eps v)/eps (solvers/matvec.py:56-61) and inherits NumPy semantics:
//   abs(z)      -> modulus, a REAL number (no tangent flows through |u|)
//   maximum(a,b)-> lexicographic compare (real part first), carries the winner's imag part
// (the reference's C++ kernels hard-code the same choices, pde/definitions.hpp:45-69).
#pragma once
#include <hip/hip_runtime.h>

namespace wx {

// reference common/definitions.py:5-12
constexpr double kGravity = 9.80616;
constexpr double kP0 = 100000.0;
constexpr double kRd = 287.05;
constexpr double kCpd = 1005.46;
constexpr double kCvd = kCpd - kRd;
constexpr double kGamma = kCpd / kCvd;  // heat_capacity_ratio
constexpr double kRdOverP0 = kRd / kP0;
constexpr double kLogP0 = 11.512925464970228420;  // log(1e5)
constexpr double kLogRdOverP0 = -5.853269048356583;     // log(287.05/1e5)

struct cplx {
    double re, im;
    __host__ __device__ cplx() = default;
    __host__ __device__ constexpr cplx(double r) : re(r), im(0.0) {}
    __host__ __device__ constexpr cplx(double r, double i) : re(r), im(i) {}
};

__device__ __forceinline__ cplx operator+(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx operator-(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx operator-(cplx a) { return {-a.re, -a.im}; }
__device__ __forceinline__ cplx operator*(cplx a, cplx b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cplx operator*(double a, cplx b) { return {a * b.re, a * b.im}; }
__device__ __forceinline__ cplx operator*(cplx a, double b) { return {a.re * b, a.im * b}; }
__device__ __forceinline__ cplx operator+(cplx a, double b) { return {a.re + b, a.im}; }
__device__ __forceinline__ cplx operator+(double a, cplx b) { return {a + b.re, b.im}; }
__device__ __forceinline__ cplx operator-(double a, cplx b) { return {a - b.re, -b.im}; }
__device__ __forceinline__ cplx operator-(cplx a, double b) { return {a.re - b, a.im}; }
__device__ __forceinline__ cplx operator/(cplx a, cplx b) {
    // Smith's algorithm (what NumPy uses for complex division)
    if (fabs(b.re) >= fabs(b.im)) {
        const double r = b.im / b.re, d = 1.0 / (b.re + b.im * r);
        return {(a.re + a.im * r) * d, (a.im - a.re * r) * d};
    }
    const double r = b.re / b.im, d = 1.0 / (b.re * r + b.im);
    return {(a.re * r + a.im) * d, (a.im * r - a.re) * d};
}
__device__ __forceinline__ cplx operator/(cplx a, double b) { return {a.re / b, a.im / b}; }
__device__ __forceinline__ cplx operator/(double a, cplx b) { return cplx(a, 0.0) / b; }
__device__ __forceinline__ cplx& operator+=(cplx& a, cplx b) { a.re += b.re; a.im += b.im; return a; }
__device__ __forceinline__ cplx& operator-=(cplx& a, cplx b) { a.re -= b.re; a.im -= b.im; return a; }

__device__ __forceinline__ double w_exp(double x) { return exp(x); }
__device__ __forceinline__ double w_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double w_abs(double x) { return fabs(x); }
__device__ __forceinline__ double w_max(double a, double b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ double w_min(double a, double b) { return (a < b || a != a) ? a : b; }
__device__ __forceinline__ double w_real(double x) { return x; }

// Lean float64 logarithm and exponential for the one-kernel form of the low orders (csrc/euler3d_brick.h), whose face stage is
// bound by the instruction count of its transcendentals: the library's log is 98 vector instructions (double-double
// arithmetic, special cases), its exp 42.  Arguments here are densities, rho theta and their extrapolated logarithms: positive,
// finite, normal; |x| < 700 for the exponential - no special cases.  Error < 1 ulp (the classical fdlibm forms).
//   lean_log: x = 2^k m, m in [sqrt(1/2), sqrt(2)); f = m - 1, s = f / (2 + f); log m = f - f^2/2 + s (f^2/2 + R(s^2))   ~40 instructions
//   lean_exp: x = k ln2 + r, |r| <= ln2 / 2; exp r by its Taylor polynomial of degree 13 (truncation 4e-18)          ~20 instructions
__device__ __forceinline__ double lean_log(double x) {
    double f = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = f < 0.70710678118654752440;
    f = lo ? f + f : f;
    e = lo ? e - 1 : e;
    f -= 1.0;                                              // [-0.2929, 0.4142)
    const double s = f / (2.0 + f);
    const double dk = (double)e;
    const double z = s * s, w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double r = dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
    // a state that has blown up must show: NaN for a negative argument (and NaN), -inf at zero, as the library's log - ADDED to the
    // (finite) value the arithmetic above leaves there, not selected: a select of the result becomes a branch around the whole
    // computation, which cuts every kernel's schedule at every logarithm
    const double bad = x > 0.0 ? 0.0 : (x == 0.0 ? -__builtin_huge_val() : __builtin_nan(""));
    return r + bad;
}
// The float64 logarithm of EVERY form goes through lean_log (round 6): the two-kernel form takes three per point at n = 8 - the
// pressure's and two per face point (the interface buffer holds rho theta, as the reference's arrays do) - and the extrapolation
// kernel two; with the library's 84-instruction log those were a quarter of the fused kernel's vector instructions.  Measured
// (profiles/r06_lean_log_everywhere.txt): extrapolation kernel -6 %, fused kernel -2 % at n = 8, whole R(Q) -1...-5 % at n = 3...6.
// -DWX_LEAN_LOG_K2=0 builds the library's log back in (A/B).
#ifndef WX_LEAN_LOG_K2
#define WX_LEAN_LOG_K2 1
#endif
__device__ __forceinline__ double w_log(double x) { return WX_LEAN_LOG_K2 ? lean_log(x) : log(x); }
__device__ __forceinline__ double lean_exp(double x) {
    const double k = __builtin_rint(x * 1.44269504088896338700e+00);
    const double r = __builtin_fma(-k, 1.90821492927058770002e-10, __builtin_fma(-k, 6.93147180369123816490e-01, x));
    double p = 1.0 / 6227020800.0;                          // 1/13!
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)k);
}
// value-type dispatch: LEAN selects the forms above for float64 and for the value part of a dual number
template <bool LEAN> __device__ __forceinline__ double w_logx(double x) { return LEAN ? lean_log(x) : log(x); }
template <bool LEAN> __device__ __forceinline__ double w_expx(double x) { return LEAN ? lean_exp(x) : exp(x); }

__device__ __forceinline__ cplx w_log(cplx z) { return {log(hypot(z.re, z.im)), atan2(z.im, z.re)}; }
__device__ __forceinline__ cplx w_exp(cplx z) {
    const double e = exp(z.re);
    double s, c;
    sincos(z.im, &s, &c);
    return {e * c, e * s};
}
__device__ __forceinline__ cplx w_sqrt(cplx z) {
    // principal branch, as csqrt
    const double m = hypot(z.re, z.im);
    if (m == 0.0) return {0.0, z.im};
    if (z.re >= 0.0) {
        const double t = sqrt(0.5 * (m + z.re));
        return {t, z.im / (2.0 * t)};
    }
    const double t = sqrt(0.5 * (m - z.re));
    return {fabs(z.im) / (2.0 * t), copysign(t, z.im)};
}
__device__ __forceinline__ double w_abs(cplx z) { return hypot(z.re, z.im); }
__device__ __forceinline__ cplx w_max(cplx a, cplx b) {
    // numpy.maximum on complex: a if a >= b lexicographically (or a is nan) else b
    const bool ge = (a.re > b.re) || (a.re == b.re && a.im >= b.im) || (a.re != a.re) || (a.im != a.im);
    return {ge ? a.re : b.re, ge ? a.im : b.im};
}
__device__ __forceinline__ cplx w_min(cplx a, cplx b) {
    const bool le = (a.re < b.re) || (a.re == b.re && a.im <= b.im) || (a.re != a.re) || (a.im != a.im);
    return {le ? a.re : b.re, le ? a.im : b.im};
}
__device__ __forceinline__ double w_real(cplx z) { return z.re; }

// ------------------------------------------------------------------------------------------------
// dual: first-order (dual-number) arithmetic in complex128 storage - re = value, im = tangent.
// The complex step Im R(Q + i eps v)/eps equals the directional derivative up to O(eps^2) (2e-16 here),
// because every product drops only the eps^2 term im*im.  With the reference's conventions kept
// (|z| has no tangent, maximum/minimum are lexicographic and carry the winner's tangent:
// pde/definitions.hpp:45-69 and NumPy) a dual evaluation returns the same real and "imaginary" parts as
// the complex one to rounding, without hypot / atan2 / sincos and with 3-flop instead of 6-flop products.
// ------------------------------------------------------------------------------------------------
struct dual {
    double re, im;
    __host__ __device__ dual() = default;
    __host__ __device__ constexpr dual(double r) : re(r), im(0.0) {}
    __host__ __device__ constexpr dual(double r, double i) : re(r), im(i) {}
};
__device__ __forceinline__ dual operator+(dual a, dual b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ dual operator-(dual a, dual b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ dual operator-(dual a) { return {-a.re, -a.im}; }
__device__ __forceinline__ dual operator*(dual a, dual b) { return {a.re * b.re, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ dual operator*(double a, dual b) { return {a * b.re, a * b.im}; }
__device__ __forceinline__ dual operator*(dual a, double b) { return {a.re * b, a.im * b}; }
__device__ __forceinline__ dual operator+(dual a, double b) { return {a.re + b, a.im}; }
__device__ __forceinline__ dual operator+(double a, dual b) { return {a + b.re, b.im}; }
__device__ __forceinline__ dual operator-(double a, dual b) { return {a - b.re, -b.im}; }
__device__ __forceinline__ dual operator-(dual a, double b) { return {a.re - b, a.im}; }
__device__ __forceinline__ dual operator/(dual a, dual b) {
    const double r = 1.0 / b.re, v = a.re * r;
    return {v, (a.im - v * b.im) * r};
}
__device__ __forceinline__ dual operator/(dual a, double b) { return {a.re / b, a.im / b}; }
__device__ __forceinline__ dual operator/(double a, dual b) {
    const double r = 1.0 / b.re, v = a * r;
    return {v, -v * b.im * r};
}
__device__ __forceinline__ dual& operator+=(dual& a, dual b) { a.re += b.re; a.im += b.im; return a; }
__device__ __forceinline__ dual& operator-=(dual& a, dual b) { a.re -= b.re; a.im -= b.im; return a; }
__device__ __forceinline__ dual w_log(dual z) { return {w_log(z.re), z.im / z.re}; }
__device__ __forceinline__ dual w_exp(dual z) { const double e = exp(z.re); return {e, e * z.im}; }
__device__ __forceinline__ dual w_sqrt(dual z) { const double s = sqrt(z.re); return {s, z.im / (2.0 * s)}; }
__device__ __forceinline__ double w_abs(dual z) { return fabs(z.re); }  // modulus to first order; no tangent
__device__ __forceinline__ dual w_max(dual a, dual b) {
    const bool ge = (a.re > b.re) || (a.re == b.re && a.im >= b.im) || (a.re != a.re) || (a.im != a.im);
    return {ge ? a.re : b.re, ge ? a.im : b.im};
}
__device__ __forceinline__ dual w_min(dual a, dual b) {
    const bool le = (a.re < b.re) || (a.re == b.re && a.im <= b.im) || (a.re != a.re) || (a.im != a.im);
    return {le ? a.re : b.re, le ? a.im : b.im};
}
__device__ __forceinline__ double w_real(dual z) { return z.re; }

__device__ __forceinline__ dual lean_log(dual z) { return {lean_log(z.re), z.im / z.re}; }   // (w_log(dual) with the lean value part)

// select between values: component-wise for the two-double types (a ternary on the aggregate makes
// the compiler route register arrays of them through scratch memory)
__device__ __forceinline__ double w_sel(bool c, double a, double b) { return c ? a : b; }
__device__ __forceinline__ cplx w_sel(bool c, cplx a, cplx b) { return {c ? a.re : b.re, c ? a.im : b.im}; }
__device__ __forceinline__ dual w_sel(bool c, dual a, dual b) { return {c ? a.re : b.re, c ? a.im : b.im}; }

// 16-byte scalar types (LDS sizing, launch bounds)
template <typename T> struct is_complex { static constexpr bool value = false; };
template <> struct is_complex<cplx> { static constexpr bool value = true; };
template <> struct is_complex<dual> { static constexpr bool value = true; };

}  // namespace wx
