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


// Table-driven variant for the two fp64-VALU-bound kernels of unit A (Gram build, MeanITE pass): x = (32 e + j) ln2/32 + r,
// |r| <= ln2/64, exp(x) = 2^e * T[j] * (1 + q(r)) with T[j] = 2^(j/32) (32 correctly rounded doubles staged in LDS: a 256-byte
// table holds one entry per pair of banks, so ANY pattern of lane indices is conflict-free) and q = exp(r) - 1 from a
// degree-6 polynomial (truncation 3.5e-18).  14 fp64-rate instructions + 3 integer + one ds_read_b64 against 20 + 1 for
// gp_exp_neg (the pair loop of both kernels is two of these per element).  exp(+-0) == 1 exactly
// (r = +-0, q = +-0, T[0] = 1); every kernel that mixes values of the two routines does so at rounding level only
// (both are < 1 ulp from the polynomial's rounding + 0.5 ulp of the table entry).
static __device__ const double gp_exp2_tab[32] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};
#define GP_EXP_TAB_DOUBLES 32

// the first 32 threads of a workgroup copy the table into LDS (the caller's next barrier publishes it)
__device__ __forceinline__ void gp_exp_tab_stage(double* lds_tab, int tid) {
    if (tid < GP_EXP_TAB_DOUBLES) lds_tab[tid] = gp_exp2_tab[tid];
}

__device__ __forceinline__ double gp_exp_neg_tab(double x, const double* __restrict__ lds_tab) {
    x = fmax(x, -800.0);
    const double k = rint(x * 0x1.71547652b82fep+5);               // 32 / ln2
    double r = fma(k, -0x1.62e42fee00000p-6, x);                   // ln2 / 32, high 32 bits (k * hi is exact)
    r = fma(k, -0x1.a39ef35793c76p-38, r);
    const int ki = (int)k;
    const double T = lds_tab[ki & 31];
    double p = 1.0 / 720.0;
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    const double q = p * r;                                        // exp(r) - 1
    return ldexp(fma(T, q, T), ki >> 5);
}

// The two arithmetic flavours of the fused RBF evaluation: fp64 (default) and fp32 (mixed-precision mode).
template <typename RT> struct RbfMath;
template <> struct RbfMath<double> {
    static __device__ __forceinline__ double exp_neg(double x) { return gp_exp_neg(x); }
    static __device__ __forceinline__ double exp_neg_t(double x, const double* tab) { return gp_exp_neg_tab(x, tab); }
};
template <> struct RbfMath<float> {
    static __device__ __forceinline__ float exp_neg(float x) { return expf(x); }   // exact 1 at +-0
    static __device__ __forceinline__ float exp_neg_t(float x, const double*) { return expf(x); }
};
