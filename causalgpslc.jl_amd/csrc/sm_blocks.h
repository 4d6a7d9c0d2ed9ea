// 16 x 16 fp64 block primitives shared by the single-workgroup node-score kernels (k_small.hip) and the robust
// (substitution-based) tile factorisation (k_robust.hip): in-register Cholesky column operations with the pivot row
// travelling by v_readlane, and the f64-MFMA rank-16 block update.
#pragma once
#include <hip/hip_runtime.h>

typedef double d4 __attribute__((ext_vector_type(4)));

#define SB 16
#define SM_THREADS 512
#define SM_WAVES (SM_THREADS / 64)
#define SBLK(i, j) (P + ((((i) * ((i) + 1)) / 2 + (j)) << 8))

__device__ __forceinline__ double sm_readlane(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
// fragment of a packed 16 x 16 block (column-major, ld 16): element (row = lane&15, k = 4kk + lane>>4)
__device__ __forceinline__ double sm_frag(const double* blk, int kk, int lane) {
    return blk[(4 * kk + (lane >> 4)) * SB + (lane & 15)];
}

// The column operations of the 16 x 16 Cholesky applied to this lane's row r[0..15] of block column p.
// Lanes 0..15 of the wave hold rows 0..15 of the diagonal block itself (row i in lane i): their lower triangle ends up
// holding L_pp — except the diagonal entry, which is left as pivot * pivot^-1/2 (uncorrected) inside the loop and is
// returned, properly rounded, through `lcc` (lane c: L_cc), and the upper triangle, which holds rounding residue nobody
// reads.  Every other lane holds a row below the block (or the right-hand side) and ends up holding that row of
// A_ip L_pp^-T.  Nothing but the pivot chain sits in the loop: readlane -> rsq -> 4 dependent flops -> scale -> first
// update -> next readlane.  bad = 1-based first non-positive pivot (0 = ok), wave-uniform.
// y = d^-1/2: hardware estimate (~2^-23) + one third-order step (error^3 ~ 2^-69)
__device__ __forceinline__ double sm_rsqrt(double d) {
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

__device__ __forceinline__ void sm_factor_rows(double (&r)[SB], int li, int base, int& bad, double& lcc) {
    double dsave = 1.0, ysave = 1.0;          // pivot and its reciprocal square root of THIS lane's column (lane c: column c)
    double d = sm_readlane(r[0], 0);          // pivot: row c of the diagonal block lives in lane c
    double y = sm_rsqrt(d);
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        if (!(d > 0.0) && bad == 0) bad = base + c + 1;
        if (li == c) { dsave = d; ysave = y; }
        r[c] *= y;
        // software-pipelined by hand: update column c+1 first and START the next pivot's reciprocal square root, so that
        // its dependent chain runs under the remaining 14 - c column updates instead of after them
        double dn = 1.0, yn = 1.0;
        if (c + 1 < SB) {
            const double l1 = sm_readlane(r[c], c + 1);        // L[c+1][c]: row c+1 of the diagonal block = lane c+1
            r[c + 1] = fma(-r[c], l1, r[c + 1]);
            dn = sm_readlane(r[c + 1], c + 1);
            yn = sm_rsqrt(dn);
        }
#pragma unroll
        for (int j = c + 2; j < SB; ++j) {
            const double ljc = sm_readlane(r[c], j);           // L[j][c], row j of the diagonal block = lane j
            r[j] = fma(-r[c], ljc, r[j]);
        }
        d = dn;
        y = yn;
    }
    // L_cc = sqrt(pivot_c): d*y with one Newton correction, once per lane, off the chain
    double sq = dsave * ysave;
    sq = fma(fma(-sq, sq, dsave), 0.5 * ysave, sq);
    lcc = sq;
}

// A_ij -= X_i X_j^T for one 16 x 16 block (X_i = block (i, p), X_j = block (j, p))
__device__ __forceinline__ void sm_update(double* Aij, const double* Xi, const double* Xj, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    d4 acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = Aij[(lg + 4 * v) * SB + li];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm_frag(Xj, kk, lane), sm_frag(Xi, kk, lane), acc, 0, 0, 1);
#pragma unroll
    for (int v = 0; v < 4; ++v) Aij[(lg + 4 * v) * SB + li] = acc[v];
}

// the same for two blocks at once (independent accumulators)
__device__ __forceinline__ void sm_update2(double* A0, const double* Xi0, const double* Xj0, double* A1, const double* Xi1,
                                           const double* Xj1, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    d4 acc0, acc1;
#pragma unroll
    for (int v = 0; v < 4; ++v) { acc0[v] = A0[(lg + 4 * v) * SB + li]; acc1[v] = A1[(lg + 4 * v) * SB + li]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sm_frag(Xj0, kk, lane), sm_frag(Xi0, kk, lane), acc0, 0, 0, 1);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sm_frag(Xj1, kk, lane), sm_frag(Xi1, kk, lane), acc1, 0, 0, 1);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) { A0[(lg + 4 * v) * SB + li] = acc0[v]; A1[(lg + 4 * v) * SB + li] = acc1[v]; }
}

