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

// y = d^-1/2: hardware estimate (~2^-23) + one third-order step (error^3 ~ 2^-69)
__device__ __forceinline__ double sm_rsqrt(double d) {
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

// The column operations of the 16 x 16 Cholesky applied to this lane's row r[0..15] of block column p.
// Lanes 0..15 of the wave hold rows 0..15 of the diagonal block itself (row i in lane i): their lower triangle ends up
// holding L_pp — except the diagonal entry, which is left as pivot * pivot^-1/2 (uncorrected) inside the loop and is
// returned, properly rounded, through `lcc` (lane c: L_cc), and the upper triangle, which holds rounding residue nobody
// reads.  Every other lane holds a row below the block (or the right-hand side) and ends up holding that row of
// A_ip L_pp^-T.  Nothing but the pivot chain sits on the critical path: readlane -> rsq -> 4 dependent flops -> scale ->
// first update -> next readlane.  bad = 1-based first non-positive pivot (0 = ok), wave-uniform.
//
// Most multipliers travel through LDS instead of v_readlane: with v_readlane alone a pivot c issues 2 (15 - c)
// v_readlane_b32 (+ their SGPR hazard nops) to turn the multipliers L[j][c] = (lane j, column c) into scalar operands —
// 240 per block, and the pass is bound by instruction issue, not by the pivot chain (that form: 5.0 k clocks per block
// column undisturbed, 6.9 k next to an updater wave; this form: 5.9 k next to an updater wave).  Here lanes 0-15 write
// their scaled column entry to a 16-double per-wave LDS line `bc` (one ds_write_b64), and every lane reads the
// multipliers of the FAR columns j >= c + 3 back with broadcast ds_read_b128 (two per instruction, wave-uniform address);
// only the two columns the next pivots need at once (c + 1, c + 2) keep the v_readlane path.  The far updates of pivot
// c are applied one iteration later — after pivot c + 1's critical readlane / rsqrt work has been issued — so the LDS
// round trip (~100+ cycles) never sits on the chain; per column j the updates are still applied in ascending pivot
// order, i.e. the arithmetic is that of the plain column sweep.  LDS operations of one wave execute in order, so the
// single line is safe: the reads of pivot c are issued before the write of pivot c + 1 (a wavefront-scope fence keeps
// the compiler from reordering or forwarding across the store — without it the compiler, reasoning per thread, reuses a
// lane's earlier loads for the lanes that did not store).
__device__ __forceinline__ void sm_factor_rows_lds(double (&r)[SB], int li, int lane, int base, int& bad, double& lcc,
                                                   double* bc /* this wave's 16-double LDS line, 16-byte aligned */) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    double dsave = 1.0, ysave = 1.0;
    double d = sm_readlane(r[0], 0);
    double y = sm_rsqrt(d);
    double mp[SB];                 // multipliers of the previous pivot's far columns
#pragma unroll
    for (int j = 0; j < SB; ++j) mp[j] = 0.0;
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        if (!(d > 0.0) && bad == 0) bad = base + c + 1;
        if (li == c) { dsave = d; ysave = y; }
        r[c] *= y;
        if (c + 3 < SB) {                                        // L[., c] for the far columns of this pivot
            if (lane < SB) bc[lane] = r[c];
            // lanes talk to each other through this line: without the fence the compiler may (and does) forward a lane's
            // earlier loads past the other lanes' stores.  Wavefront scope: no instruction, LDS executes a wave's
            // operations in order.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        double dn = 1.0, yn = 1.0;
        if (c + 1 < SB) {                                        // critical: column c + 1 and the next pivot
            const double l1 = sm_readlane(r[c], c + 1);
            r[c + 1] = fma(-r[c], l1, r[c + 1]);
            dn = sm_readlane(r[c + 1], c + 1);
            yn = sm_rsqrt(dn);
        }
        if (c >= 1) {                                            // far columns of pivot c - 1: j >= c + 2
#pragma unroll
            for (int j = c + 2; j < SB; ++j) r[j] = fma(-r[c - 1], mp[j], r[j]);
        }
        if (c + 2 < SB) {                                        // column c + 2 of this pivot (after pivot c - 1's update of it)
            const double l2 = sm_readlane(r[c], c + 2);
            r[c + 2] = fma(-r[c], l2, r[c + 2]);
        }
        if (c + 3 < SB) {                                        // fetch this pivot's far multipliers: j >= c + 3
#pragma unroll
            for (int j0 = (c + 3) & ~1; j0 < SB; j0 += 2) {
                const d2v m = *reinterpret_cast<const d2v*>(bc + j0);
                mp[j0] = m[0];
                mp[j0 + 1] = m[1];
            }
        }
        d = dn;
        y = yn;
    }
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

