// 128 x 128 diagonal-tile Cholesky + inverse on a PACKED LDS image (the body of diag_potrf_inv_v2_kernel, k_diag.hip),
// as a device function: the diagonal-block kernel runs it on a tile it loads itself; the measurement build's chained
// in-panel kernel (tile_fused_chain_kernel, k_tilegemm.hip, GPSLC_CHAIN=2) runs it on the image its own diagonal-tile
// update has just left in LDS.
// Tried on it in round 4 and not kept (profiles/r04_ab_experiments.md §4): barriers that wait for LDS traffic only
// (__syncthreads() also waits for the acknowledgement of the row-p stores to HBM) and two trailing blocks per pass —
// 638.7 vs 614.7 us per 4,096 matrices: the kernel is bound by its pivot chains, not by its barriers.
#pragma once
#include "gpslc_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define SB 16     // sub-block
#define NSB (GP_TS / SB)
#define DIAG2_LDS_BYTES ((36 * 256 + 256) * 8)

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
// acc[v] <-> (row = lane&15 of the `rowside` operand's row index, col = (lane>>4)+4v of `colside`'s)
__device__ __forceinline__ d4 mma(double colside, double rowside, d4 acc) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(colside, rowside, acc, 0, 0, 0);
}
__device__ __forceinline__ d4 mma_neg(double colside, double rowside, d4 acc) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(colside, rowside, acc, 0, 0, 1);
}
// ---------------------------------------------------------------------------------------
// Version 2: the same arithmetic on a PACKED image — only the 36 lower 16 x 16 sub-blocks live in LDS
// (72 KiB + one 2 KiB slot for the current inv(L_pp)), so TWO workgroups fit a CU.  The kernel is latency
// bound (one 8-step dependency chain per matrix), so residency is throughput: 352 -> ~190 us per 1024
// matrices.  What makes the packing possible:
//   * block row p of L is final after step p: it is written to the output tile right away and its LDS slots
//     are then overwritten, in place, by block row p of inv(L):
//         W_pq = -W_pp * sum_{m=q}^{p-1} L_pm W_mq        (q < p; needs rows < p of W only),
//     i.e. the inversion runs row-wise inside the factorisation loop instead of column-wise after it;
//   * W blocks are stored transposed (element (r, c) at r*16 + c) so that they are read as MFMA operands with
//     the conflict-free fragment pattern; slot (p, p) receives W_pp^T once L_pp has been written out.
// Summation orders are those of version 1 (same MFMA chains).
// ---------------------------------------------------------------------------------------
#define BLK(i, j) (P + ((((i) * ((i) + 1)) / 2 + (j)) << 8))
// fragment of a packed 16 x 16 block (column-major, ld 16): element (row = lane&15, k = 4kk + lane>>4)
__device__ __forceinline__ double bfrag(const double* blk, int kk, int lane) {
    return blk[(4 * kk + (lane >> 4)) * SB + (lane & 15)];
}

// P: 36 packed blocks + 256 doubles (DIAG2_LDS_BYTES of LDS); tile / invt: the output tiles (L_kk, inv(L_kk)) in HBM;
// image_ready: the lower blocks are already in P (the caller has NOT yet synchronised: the first barrier is in here);
// a non-positive pivot c (0-based) is reported as info_code0 + c + 1 through an atomicCAS on *info_word.
// Called by all 256 threads of the workgroup (it contains barriers).
__device__ __forceinline__ void diag_potrf_inv_v2_body(double* P, double* tile, double* invt, int* info_word,
                                                       int info_code0, int tid, bool image_ready) {
    double* Wcur = P + 36 * 256;     // Wcur[c'*16 + c] = inv(L_pp)[c][c'] of the current step
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int er = tid & 15, ec = tid >> 4;       // this thread's element of a 16 x 16 block

    if (!image_ready) {
        for (int bi = 0; bi < NSB; ++bi)
            for (int bj = 0; bj <= bi; ++bj)
                BLK(bi, bj)[tid] = tile[(SB * bj + ec) * GP_TS + SB * bi + er];
    }
    int bad = 0;
    __syncthreads();

    for (int p = 0; p < NSB; ++p) {
        double* Dpp = BLK(p, p);
        if (wave == 0) {
            // ---- (a) 16 x 16 Cholesky in registers: every group of 16 lanes mirrors rows 0..15
            double r[SB], isd[SB];
#pragma unroll
            for (int c = 0; c < SB; ++c) r[c] = Dpp[c * SB + li];
#pragma unroll
            for (int c = 0; c < SB; ++c) {
                const double d = readlane_f64(r[c], c);
                if (!(d > 0.0) && bad == 0) bad = SB * p + c + 1;
                double y = __builtin_amdgcn_rsq(d);
                y = y * (1.5 - 0.5 * d * y * y);
                y = y * (1.5 - 0.5 * d * y * y);
                double s = d * y;
                s = fma(fma(-s, s, d), 0.5 * y, s);       // sqrt(d), Newton-corrected
                y = fma(fma(-s, y, 1.0), y, y);           // 1/s
                isd[c] = y;
                r[c] = (li > c) ? r[c] * y : (li == c ? s : 0.0);
#pragma unroll
                for (int j = c + 1; j < SB; ++j) {
                    const double ljc = readlane_f64(r[c], j);
                    r[j] = fma(-r[c], ljc, r[j]);
                }
            }
            // ---- inverse of the 16 x 16 factor: lane j owns column j of W
            double w[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < i; ++m) acc = fma(readlane_f64(r[m], i), w[m], acc);
                w[i] = (i == li) ? isd[i] : ((i > li) ? -acc * isd[i] : 0.0);
            }
            if (lane < SB) {
#pragma unroll
                for (int c = 0; c < SB; ++c) Dpp[c * SB + li] = r[c];
#pragma unroll
                for (int i = 0; i < SB; ++i) Wcur[li * SB + i] = w[i];
            }
        }
        __syncthreads();
        // ---- phase 1: reads of block row p (final L) and of the rows < p of W; writes to column p and to HBM
        // (b) panel: X_i = A_i * W_pp^T for the sub-blocks below the diagonal block
        for (int i = p + 1 + wave; i < NSB; i += 4) {
            double* Aip = BLK(i, p);
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mma(bfrag(Wcur, kk, lane), bfrag(Aip, kk, lane), acc);
#pragma unroll
            for (int v = 0; v < 4; ++v) Aip[(lg + 4 * v) * SB + li] = acc[v];
        }
        // (b') block row p of inv(L): W_pq = -W_pp sum_{m=q}^{p-1} L_pm W_mq, kept in registers until phase 2
        d4 wq[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = wave + 4 * u;
            wq[u] = (d4){0.0, 0.0, 0.0, 0.0};
            if (q < p) {
                d4 accT = (d4){0.0, 0.0, 0.0, 0.0};   // accT[v] = T[(lane>>4)+4v][lane&15]
                for (int m = q; m < p; ++m) {
                    const double* Lpm = BLK(p, m);
                    const double* Wmq = BLK(m, q);    // transposed image: W_mq[r][c] at r*16 + c
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) accT = mma(bfrag(Lpm, kk, lane), bfrag(Wmq, kk, lane), accT);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wq[u] = mma_neg(accT[kk], bfrag(Wcur, kk, lane), wq[u]);
                // W_pq[r = lane&15][c = (lane>>4)+4v] -> output tile of the inverse
#pragma unroll
                for (int v = 0; v < 4; ++v) invt[(SB * q + lg + 4 * v) * GP_TS + SB * p + li] = wq[u][v];
            }
        }
        // block row p of the outputs: L (slots (p, 0..p)), the diagonal block of inv(L), zeros right of them
        for (int j = 0; j < NSB; ++j) {
            double lv = 0.0;
            if (j < p) lv = BLK(p, j)[tid];
            else if (j == p) lv = (er >= ec) ? Dpp[tid] : 0.0;
            tile[(SB * j + ec) * GP_TS + SB * p + er] = lv;
            if (j == p) invt[(SB * j + ec) * GP_TS + SB * p + er] = (er >= ec) ? Wcur[tid] : 0.0;
            else if (j > p) invt[(SB * j + ec) * GP_TS + SB * p + er] = 0.0;
        }
        __syncthreads();
        // ---- phase 2: block row p of W into its slots (transposed), trailing update of the blocks (i, j > p)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = wave + 4 * u;
            if (q < p) {
                double* Wpq = BLK(p, q);
#pragma unroll
                for (int v = 0; v < 4; ++v) Wpq[li * SB + lg + 4 * v] = wq[u][v];
            }
        }
        Dpp[er * SB + ec] = Wcur[tid];          // W_pp^T: element (r = c, c = c') of W at r*16 + c
        {
            const int m = NSB - p - 1;
            const int nt_ = m * (m + 1) / 2;
            for (int t = wave; t < nt_; t += 4) {
                int ii = 0, rem = t;
                while (rem > ii) { rem -= ii + 1; ++ii; }
                const int i = p + 1 + ii, j = p + 1 + rem;
                double* Aij = BLK(i, j);
                const double* Xi = BLK(i, p);
                const double* Xj = BLK(j, p);
                d4 acc;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = Aij[(lg + 4 * v) * SB + li];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = mma_neg(bfrag(Xj, kk, lane), bfrag(Xi, kk, lane), acc);
#pragma unroll
                for (int v = 0; v < 4; ++v) Aij[(lg + 4 * v) * SB + li] = acc[v];
            }
        }
        __syncthreads();
    }
    if (wave == 0 && lane == 0 && bad != 0) atomicCAS(info_word, 0, info_code0 + bad);
}

