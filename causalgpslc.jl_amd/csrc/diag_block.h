// 128 x 128 diagonal-tile Cholesky + inverse on a PACKED LDS image, as a device function: the diagonal-block kernel
// (diag_potrf_inv_la_kernel, k_diag.hip) runs it on a tile it loads itself, diag_update_potrf_kernel (k_tilegemm.hip) on the
// image its own diagonal-tile update has just left in LDS.
// Tried on it in round 4 and not kept (profiles/r04_ab_experiments.md §4): barriers that wait for LDS traffic only
// (__syncthreads() also waits for the acknowledgement of the row-p stores to HBM) and two trailing blocks per pass.
// Round 5: one-block lookahead (below) — 1,220 -> 1,133 us per 8,192 matrices; the loop without it is
// profiles/r05_potrf_no_lookahead.patch.
#pragma once
#include "gpslc_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define SB 16     // sub-block
#define NSB (GP_TS / SB)

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
// The arithmetic runs on a PACKED image — only the 36 lower 16 x 16 sub-blocks live in LDS
// (72 KiB + 2 KiB slots for inv(L_pp)), so TWO workgroups fit a CU.  The kernel is latency
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

// ---------------------------------------------------------------------------------------
// The factorisation loop, with a ONE-BLOCK LOOKAHEAD (round 5).  Step p = (a) Cholesky + inverse of the 16 x 16 diagonal
// sub-block in registers (lane i holds row i, pivot rows by v_readlane: a chain of dependent fp64 operations on ONE wave),
// (b) panel X_i = A_i W_pp^T, (b') block row p of inv(L), the output rows, (c) trailing update A_ij -= X_i X_j^T.  Until round 4
// (a) ran while the other three waves waited at a barrier and (b), (c) while that wave had nothing to do.  Now wave 0 is taken
// off the block updates: in step p it computes the panel block
// (p+1, p), applies it to the diagonal block (p+1, p+1) — the only update of step p that block needs — and factorises it
// (Cholesky before the mid-step barrier, inverse after it) into the SECOND W slot, while waves 1..3 do everything else of
// step p: the remaining panel blocks, block row p of inv(L), the output rows, the trailing blocks.  Step p + 1 then starts
// with its diagonal block already factorised.  Arithmetic: every block receives the same MFMA chains in the same order as
// the loop without lookahead — results are bit-identical; only who computes what, and when, changes.  LDS: one more 2 KiB W slot.
// ---------------------------------------------------------------------------------------
#define DIAG3_LDS_BYTES ((36 * 256 + 512) * 8)

// wave-level: Cholesky of the 16 x 16 block at Dpp (column-major, ld 16) in registers; r = the factor's columns (lane & 15 =
// row), isd = 1 / diagonal.  Returns the 1-based failing pivot inside the block (0 = ok).
__device__ __forceinline__ int sb_chol16(const double* Dpp, int li, double (&r)[SB], double (&isd)[SB]) {
    int bad = 0;
#pragma unroll
    for (int c = 0; c < SB; ++c) r[c] = Dpp[c * SB + li];
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        const double d = readlane_f64(r[c], c);
        if (!(d > 0.0) && bad == 0) bad = c + 1;
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
    return bad;
}
// wave-level: inverse of that factor (lane j owns column j of W) -> Wdst[c' * 16 + c] = inv(L_pp)[c][c'], and the factor itself
// back to its slot
__device__ __forceinline__ void sb_inv16_store(const double (&r)[SB], const double (&isd)[SB], double* Dpp, double* Wdst,
                                               int lane, int li) {
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
        for (int i = 0; i < SB; ++i) Wdst[li * SB + i] = w[i];
    }
}

// P: 36 packed blocks + TWO 256-double W slots (DIAG3_LDS_BYTES of LDS); tile / invt: the output tiles (L_kk, inv(L_kk)) in HBM;
// image_ready: the lower blocks are already in P (the caller has NOT yet synchronised: the first barrier is in here);
// a non-positive pivot c (0-based) is reported as info_code0 + c + 1 through an atomicCAS on *info_word.
// Called by all 256 threads of the workgroup (it contains barriers).
__device__ __forceinline__ void diag_potrf_inv_la_body(double* P, double* tile, double* invt, int* info_word,
                                                       int info_code0, int tid, bool image_ready) {
    double* Wslot = P + 36 * 256;    // [2][256]
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int er = tid & 15, ec = tid >> 4;       // this thread's element of a 16 x 16 block (image load)
    const int t3 = tid - 64;                      // thread index among waves 1..3 (0..191)

    if (!image_ready) {
        for (int bi = 0; bi < NSB; ++bi)
            for (int bj = 0; bj <= bi; ++bj)
                BLK(bi, bj)[tid] = tile[(SB * bj + ec) * GP_TS + SB * bi + er];
    }
    int bad = 0;
    __syncthreads();
    if (wave == 0) {                 // block (0, 0): nothing to overlap it with
        double r[SB], isd[SB];
        const int b0 = sb_chol16(BLK(0, 0), li, r, isd);
        if (b0) bad = b0;
        sb_inv16_store(r, isd, BLK(0, 0), Wslot, lane, li);
    }
    __syncthreads();

    for (int p = 0; p < NSB; ++p) {
        double* Dpp = BLK(p, p);
        const double* Wcur = Wslot + (p & 1) * 256;          // inv(L_pp), transposed image
        double* Wnext = Wslot + ((p + 1) & 1) * 256;
        // ---- phase 1: reads of block row p (final L) and of the rows < p of W; writes to column p and to HBM
        double r[SB], isd[SB];                   // wave 0: the factor of block (p+1, p+1) between the two phases
        d4 wq[3];                                // waves 1..3: their blocks of row p of inv(L) between the two phases
        if (wave == 0) {
            if (p + 1 < NSB) {
                // panel block (p+1, p), then the one update the next diagonal block needs, then its Cholesky
                double* Aip = BLK(p + 1, p);
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = mma(bfrag(Wcur, kk, lane), bfrag(Aip, kk, lane), acc);
#pragma unroll
                for (int v = 0; v < 4; ++v) Aip[(lg + 4 * v) * SB + li] = acc[v];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                double* Ann = BLK(p + 1, p + 1);
                d4 ad;
#pragma unroll
                for (int v = 0; v < 4; ++v) ad[v] = Ann[(lg + 4 * v) * SB + li];
                // the wave reads back its own LDS stores: same wave, in-order LDS queue
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) ad = mma_neg(bfrag(Aip, kk, lane), bfrag(Aip, kk, lane), ad);
#pragma unroll
                for (int v = 0; v < 4; ++v) Ann[(lg + 4 * v) * SB + li] = ad[v];
                // the lanes of this wave now read each other's stores of the block: the LDS queue keeps a wave's operations in
                // order, the wavefront-scope fence (no instruction) makes the compiler honour that order
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int bn = sb_chol16(Ann, li, r, isd);
                if (bn && bad == 0) bad = SB * (p + 1) + bn;
            }
        } else {
            // (b) panel: X_i = A_i * W_pp^T for the sub-blocks below (p+1, p)
            for (int i = p + 2 + (wave - 1); i < NSB; i += 3) {
                double* Aip = BLK(i, p);
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = mma(bfrag(Wcur, kk, lane), bfrag(Aip, kk, lane), acc);
#pragma unroll
                for (int v = 0; v < 4; ++v) Aip[(lg + 4 * v) * SB + li] = acc[v];
            }
            // (b') block row p of inv(L): W_pq = -W_pp sum_{m=q}^{p-1} L_pm W_mq, kept in registers until phase 2
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = (wave - 1) + 3 * u;
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
#pragma unroll
                    for (int v = 0; v < 4; ++v) invt[(SB * q + lg + 4 * v) * GP_TS + SB * p + li] = wq[u][v];
                }
            }
            // block row p of the outputs: L (slots (p, 0..p)), the diagonal block of inv(L), zeros right of them
            for (int j = 0; j < NSB; ++j)
                for (int e = t3; e < 256; e += 192) {
                    const int rr = e & 15, cc = e >> 4;
                    double lv = 0.0;
                    if (j < p) lv = BLK(p, j)[e];
                    else if (j == p) lv = (rr >= cc) ? Dpp[e] : 0.0;
                    tile[(SB * j + cc) * GP_TS + SB * p + rr] = lv;
                    if (j == p) invt[(SB * j + cc) * GP_TS + SB * p + rr] = (rr >= cc) ? Wcur[e] : 0.0;
                    else if (j > p) invt[(SB * j + cc) * GP_TS + SB * p + rr] = 0.0;
                }
        }
        __syncthreads();
        // ---- phase 2: wave 0 inverts the next diagonal block; waves 1..3: block row p of W into its slots (transposed) and
        // the trailing update of the blocks (i, j > p) other than (p+1, p+1)
        if (wave == 0) {
            if (p + 1 < NSB) sb_inv16_store(r, isd, BLK(p + 1, p + 1), Wnext, lane, li);
        } else {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = (wave - 1) + 3 * u;
                if (q < p) {
                    double* Wpq = BLK(p, q);
#pragma unroll
                    for (int v = 0; v < 4; ++v) Wpq[li * SB + lg + 4 * v] = wq[u][v];
                }
            }
            for (int e = t3; e < 256; e += 192) Dpp[(e & 15) * SB + (e >> 4)] = Wcur[e];     // W_pp^T into slot (p, p)
            const int m = NSB - p - 1;
            const int nt_ = m * (m + 1) / 2;
            for (int t = 1 + (wave - 1); t < nt_; t += 3) {       // t = 0 is block (p+1, p+1): wave 0 did it in phase 1
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

