// Substitution-based ("robust") diagonal-tile factorisation and panel solve of the tiled Cholesky, for matrices
// that are positive definite only by a hair: CovITE + 1e-10 I (src/estimation.jl:82, :105; cond ~ 1e10) and the
// U-prior covariance SigmaU * uNoise with its 1e-13 jitter (src/utils.jl:17-33).
//
// Why: the fast path (k_diag.hip / k_tilegemm.hip) solves every panel by MULTIPLYING with an explicitly inverted
// triangular block — 16 x 16 sub-blocks inside the diagonal tile, the whole 128 x 128 inverse across tiles — so that
// the panel is one more f64-MFMA product.  A product with a computed inverse carries an error of eps * cond(L_kk); for
// A = K + yNoise I (cond(L) ~ 1e2..1e3) that is rounding noise, for a near-singular leading block (cond(L) ~ 1e5) it
// reaches the 1e-10 jitter and a pivot of the trailing block goes negative although LAPACK's potrf succeeds on the very
// same matrix — measured on the reference's documented NEEC example (tools/check_neec_pd.py: pivot 135 of a 150 x 150
// CovITE + 1e-10 I, numpy.linalg.cholesky fine).  Substitution is backward stable whatever the conditioning.
//
// How: the column operations of k_small.hip's register-resident Cholesky (sm_blocks.h).  The Cholesky of a 16 x 16
// block is a sequence of column operations; applied to the rows below the block they ARE the triangular solve
// x <- x L_pp^-T.  No inverse is formed anywhere.
//   diag_potrf_robust_kernel   tile (k, k) resident in LDS as 36 packed blocks; per sub-block column: factor + panel in
//                              one register-resident pass (every wave carries its own copy of the diagonal block's 16
//                              rows in lanes 0-15 and 48 rows below in lanes 16-63), MFMA trailing update.
//   tile_trsm_robust_kernel    tile (i, k), i > k: X <- A(i,k) L_kk^-T.  Every wave owns 32 rows for the whole solve
//                              (rows are independent: no barrier after the load): per sub-block column p the MFMA
//                              update  a_p -= sum_{q<p} x_q L_pq^T  (L_pq fragments straight from the factored tile in
//                              L2) and the 16 x 16 solve with L_pp by column operations (L_pp's rows in lanes 0-15).
#include "gpslc_internal.h"
#include "sm_blocks.h"

#define RB_NSB (GP_TS / SB)       // 8 sub-block rows / columns per tile

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diag_potrf_robust_kernel(TRef M, int k, int* info, int info_base, int info_div) {
    extern __shared__ __attribute__((aligned(16))) double P[];        // 36 packed blocks + [4 waves][16] broadcast lines
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: branches on it are scalar branches
    const int li = lane & 15;
    const long long b = blockIdx.x;
    double* tile = tref_tile(M, b, k, k);
    const int er = tid & 15, ec = tid >> 4;

    for (int bi = 0; bi < RB_NSB; ++bi)
        for (int bj = 0; bj <= bi; ++bj) SBLK(bi, bj)[tid] = tile[(SB * bj + ec) * GP_TS + SB * bi + er];
    int bad = 0;
    __syncthreads();

    for (int p = 0; p < RB_NSB; ++p) {
        const int rows = GP_TS - SB * (p + 1);               // rows of the tile below the diagonal sub-block
        if (wave * 48 < rows || wave == 0) {                 // wave-uniform: this wave carries rows (wave 0 always: the factor)
            const bool is_diag = lane < SB;
            const int q = wave * 48 + (lane - SB);
            double r[SB];
            double* dst = nullptr;
            if (is_diag) dst = SBLK(p, p) + li;
            else if (q < rows) { const int gr = SB * (p + 1) + q; dst = SBLK(gr >> 4, p) + (gr & 15); }
            if (dst) {
#pragma unroll
                for (int c = 0; c < SB; ++c) r[c] = dst[c * SB];
            } else {
#pragma unroll
                for (int c = 0; c < SB; ++c) r[c] = 0.0;
            }
            double lcc;
            sm_factor_rows_lds(r, li, lane, SB * p, bad, lcc, P + 36 * 256 + wave * SB);
            if (is_diag) {
                if (wave == 0) {                             // L_pp: lower triangle, exact diagonal, zeros above
#pragma unroll
                    for (int c = 0; c < SB; ++c) dst[c * SB] = (li > c) ? r[c] : (li == c ? lcc : 0.0);
                }
            } else if (dst) {
#pragma unroll
                for (int c = 0; c < SB; ++c) dst[c * SB] = r[c];
            }
        }
        __syncthreads();
        {
            const int m = RB_NSB - p - 1;
            const int nt_ = m * (m + 1) / 2;
            for (int t = wave; t < nt_; t += 4) {
                int ii = 0, rem = t;
                while (rem > ii) { rem -= ii + 1; ++ii; }
                sm_update(SBLK(p + 1 + ii, p + 1 + rem), SBLK(p + 1 + ii, p), SBLK(p + 1 + rem, p), lane);
            }
        }
        __syncthreads();
    }
    if (tid == 0 && bad != 0) atomicCAS(&info[b / info_div], 0, info_base + GP_TS * k + bad);   // info per posterior sample
    // factor back to the tile: lower blocks, zeros in the strictly-upper sub-blocks
    for (int bi = 0; bi < RB_NSB; ++bi)
        for (int bj = 0; bj < RB_NSB; ++bj)
            tile[(SB * bj + ec) * GP_TS + SB * bi + er] = (bj <= bi) ? SBLK(bi, bj)[tid] : 0.0;
}

// column operations with an already factored 16 x 16 block: lanes 0..15 hold the rows of L_pp (untouched), every other
// lane a row x that becomes x L_pp^-T
__device__ __forceinline__ void rb_solve_rows(double (&r)[SB], bool is_diag) {
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        const double lcc = sm_readlane(r[c], c);               // L_cc (lane c, column c)
        const double y = 1.0 / lcc;
        if (!is_diag) r[c] *= y;
#pragma unroll
        for (int j = c + 1; j < SB; ++j) {
            const double ljc = sm_readlane(r[c], j);           // L[j][c]: row j of L_pp = lane j
            if (!is_diag) r[j] = fma(-r[c], ljc, r[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_trsm_robust_kernel(TRef X, TRef L, int k, int i0) {
    extern __shared__ __attribute__((aligned(16))) double XB[];      // [rb 8][cb 8] blocks of 256 doubles: the tile (i, k)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: branches on it are scalar branches
    const int li = lane & 15, lg = lane >> 4;
    const long long b = blockIdx.y;
    const int i = i0 + blockIdx.x;
    double* tile = tref_tile(X, b, i, k);
    const double* __restrict__ Lkk = tref_tile(L, b, k, k);          // factored: lower triangle valid, zeros above
    // load: thread (er, ec) of block (rb, cb)
    {
        const int er = tid & 15, ec = tid >> 4;
        for (int rb = 0; rb < RB_NSB; ++rb)
            for (int cb = 0; cb < RB_NSB; ++cb) XB[((rb * RB_NSB + cb) << 8) + tid] = tile[(SB * cb + ec) * GP_TS + SB * rb + er];
    }
    __syncthreads();
    // from here on wave w works alone on row blocks w and w + 4
    for (int p = 0; p < RB_NSB; ++p) {
        // update: block (rb, p) -= sum_{q<p} X(rb, q) L_pq^T; acc[v] <-> (row li of the X block, col lg + 4v of block p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rb = wave + 4 * h;
            double* Bp = XB + ((rb * RB_NSB + p) << 8);
            d4 acc;
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = Bp[(lg + 4 * v) * SB + li];
            for (int q = 0; q < p; ++q) {
                const double* Xq = XB + ((rb * RB_NSB + q) << 8);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    // L_pq fragment: element (row = li of block p, k = 4kk + lg of block q) of the factored tile
                    const double lf = Lkk[(SB * q + 4 * kk + lg) * GP_TS + SB * p + li];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lf, sm_frag(Xq, kk, lane), acc, 0, 0, 1);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) Bp[(lg + 4 * v) * SB + li] = acc[v];
        }
        // the wave's lanes hand data to one another through its private part of XB without a barrier: LDS executes a
        // wave's operations in order, the wavefront-scope fence (no instruction) makes the compiler honour that order
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // solve with L_pp: lanes 0-15 = rows of L_pp, lanes 16-31 = rows of block (wave, p), lanes 32-47 = rows of
        // block (wave + 4, p)
        {
            double r[SB];
            const bool is_diag = lane < SB;
            double* src = nullptr;
            if (lane >= SB && lane < 3 * SB) {
                const int rb = wave + 4 * ((lane >> 4) - 1);
                src = XB + ((rb * RB_NSB + p) << 8) + li;
            }
            if (is_diag) {
#pragma unroll
                for (int c = 0; c < SB; ++c) r[c] = Lkk[(SB * p + c) * GP_TS + SB * p + li];
            } else if (src) {
#pragma unroll
                for (int c = 0; c < SB; ++c) r[c] = src[c * SB];
            } else {
#pragma unroll
                for (int c = 0; c < SB; ++c) r[c] = 0.0;
            }
            rb_solve_rows(r, is_diag);
            if (src) {
#pragma unroll
                for (int c = 0; c < SB; ++c) src[c * SB] = r[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    {
        const int er = tid & 15, ec = tid >> 4;
        for (int rb = 0; rb < RB_NSB; ++rb)
            for (int cb = 0; cb < RB_NSB; ++cb) tile[(SB * cb + ec) * GP_TS + SB * rb + er] = XB[((rb * RB_NSB + cb) << 8) + tid];
    }
}

void launch_diag_robust(const TRef& M, int k, int* info, int info_base, int nbatch, hipStream_t st, int info_div) {
    const int bytes = (36 * 256 + 4 * SB) * 8;
    static DeviceOnce once;
    lds_opt_in(once, (const void*)diag_potrf_robust_kernel, bytes);
    hipLaunchKernelGGL(diag_potrf_robust_kernel, dim3(nbatch), dim3(256), bytes, st, M, k, info, info_base, info_div < 1 ? 1 : info_div);
}

// X(i, k) <- X(i, k) L_kk^-T for the tiles i = i0 .. i0 + count - 1 of tile column k of X (X may be the factor's own
// tile matrix — the panel below the diagonal tile — or a separate rectangular one: right-hand sides as rows)
void launch_trsm_robust(const TRef& X, const TRef& L, int k, int i0, int count, int nbatch, hipStream_t st) {
    if (count <= 0) return;
    const int bytes = RB_NSB * RB_NSB * 256 * 8;       // 128 KiB: one workgroup per CU
    static DeviceOnce once;
    lds_opt_in(once, (const void*)tile_trsm_robust_kernel, bytes);
    hipLaunchKernelGGL(tile_trsm_robust_kernel, dim3(count, nbatch), dim3(256), bytes, st, X, L, k, i0);
}
