// Shared device math for the fused RBF kernels (Gram build, MeanITE pass, D/Delta build, RHS).
#pragma once
#include <hip/hip_runtime.h>

// exp(x) for x <= 0 (the only domain the RBF path needs; x is clamped at -800 where exp underflows
// to 0).  Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor polynomial in Horner form
// (truncation 4e-18 relative), v_ldexp_f64 scaling: < 1 ulp from the polynomial's rounding, about
// 2.5x fewer instructions than ocml's exp (no overflow / subnormal-input branches).
// exp(-0.0) and exp(+0.0) return exactly 1.0 (the doT == T exact-zero identities rely on it).
__device__ __forceinline__ double gp_exp_neg(double x) {
    x = fmax(x, -800.0);
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;            // 1/13!
    p = fma(p, r, 2.08767569878681e-09);          // 1/12!
    p = fma(p, r, 2.505210838544172e-08);         // 1/11!
    p = fma(p, r, 2.755731922398589e-07);         // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);        // 1/9!
    p = fma(p, r, 2.48015873015873e-05);          // 1/8!
    p = fma(p, r, 1.984126984126984e-04);         // 1/7!
    p = fma(p, r, 1.388888888888889e-03);         // 1/6!
    p = fma(p, r, 8.333333333333333e-03);         // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);        // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// The two arithmetic flavours of the fused RBF evaluation: fp64 (default) and fp32 (mixed-precision mode).
template <typename RT> struct RbfMath;
template <> struct RbfMath<double> {
    static __device__ __forceinline__ double exp_neg(double x) { return gp_exp_neg(x); }
};
template <> struct RbfMath<float> {
    static __device__ __forceinline__ float exp_neg(float x) { return expf(x); }   // exact 1 at +-0
};
