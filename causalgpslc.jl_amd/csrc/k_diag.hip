// Diagonal-block kernel of the blocked Cholesky: factor the 128 x 128 tile (k, k) and produce
// inv(L_kk), so that the panel solve L(i,k) = A(i,k) * inv(L_kk)^T becomes one more f64-MFMA tile
// product (k_tilegemm.hip).  One workgroup (4 wave64) per matrix of the batch; the tile lives in LDS.
//
// Blocked with 16 x 16 sub-blocks (8 block columns):
//   (a) wave 0 factors the diagonal sub-block in registers: lane i holds row i, the pivot row is
//       broadcast with v_readlane (no LDS, no barrier inside the 16 columns), and inverts it;
//   (b) panel   X_i = A_i * inv(L_pp)^T            4 MFMAs per sub-tile, sub-tiles dealt to the 4 waves
//   (c) update  A_ij -= X_i X_j^T  (p < j <= i)    4 MFMAs per sub-tile, accumulators round-trip LDS
//   after the 8 steps: inv(L) by block forward substitution, one block column per wave (column q and
//   7-q to balance), products on the MFMA; the transposed blocks of inv(L) are parked in the unused
//   strictly-upper part of the LDS image.  The f64 16x16x4 accumulator layout (row = (lane>>4)+4v)
//   is exactly the k-layout of the next MFMA's operand, so T = L_pm W_mq feeds -inv(L_pp) T without
//   leaving registers.
//
// Failure semantics mirror LAPACK potrf / PDMats: the first non-positive (or NaN) pivot is reported
// as info = info_base + 128*k + c + 1 (1-based) through an atomicCAS on the batch element's info
// word; the kernel always terminates (no data-dependent loops).
#include "gpslc_internal.h"
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

#define DLD 130   // LDS leading dimension (doubles)
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
// fragment of a column-major LDS matrix: element (row = rbase + lane&15, k = kbase + 4kk + lane>>4)
__device__ __forceinline__ double frag(const double* X, int ld, int rbase, int kbase, int kk, int lane) {
    return X[(kbase + 4 * kk + (lane >> 4)) * ld + rbase + (lane & 15)];
}

#ifdef GPSLC_DIAG   // version 1 (133 KiB square image, one workgroup per CU): measurement build only, for the A/B of
                    // profiles/r02_ab_experiments.md; the production library compiles and launches version 2 alone
__global__ __launch_bounds__(256) void diag_potrf_inv_kernel(TRef M, int k, double* inv,
                                                             long long inv_bstride, int* info,
                                                             int info_base) {
    extern __shared__ __attribute__((aligned(16))) double S[];   // [128][DLD] + Wl[8][16*16]
    double* Wl = S + GP_TS * DLD;   // Wl[p][c'*16 + c] = inv(L_pp)[c][c']
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15;
    const long long b = blockIdx.x;
    double* tile = tref_tile(M, b, k, k);

    for (int idx = tid; idx < GP_TSQ; idx += 256) S[(idx >> 7) * DLD + (idx & 127)] = tile[idx];
    int bad = 0;
    __syncthreads();

    for (int p = 0; p < NSB; ++p) {
        const int o = SB * p;
        if (wave == 0) {
            // ---- (a) 16 x 16 Cholesky in registers: every group of 16 lanes mirrors rows 0..15
            double r[SB], isd[SB];
#pragma unroll
            for (int c = 0; c < SB; ++c) r[c] = S[(o + c) * DLD + o + li];
#pragma unroll
            for (int c = 0; c < SB; ++c) {
                const double d = readlane_f64(r[c], c);
                if (!(d > 0.0) && bad == 0) bad = o + c + 1;
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
                for (int c = 0; c < SB; ++c) S[(o + c) * DLD + o + li] = r[c];
#pragma unroll
                for (int i = 0; i < SB; ++i) Wl[p * SB * SB + li * SB + i] = w[i];
            }
        }
        __syncthreads();
        // ---- (b) panel: X_i = A_i * W^T for the sub-tiles below the diagonal block
        for (int i = p + 1 + wave; i < NSB; i += 4) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc = mma(frag(Wl + p * SB * SB, SB, 0, 0, kk, lane), frag(S, DLD, SB * i, o, kk, lane), acc);
#pragma unroll
            for (int v = 0; v < 4; ++v) S[(o + (lane >> 4) + 4 * v) * DLD + SB * i + li] = acc[v];
        }
        __syncthreads();
        // ---- (c) trailing update of the lower sub-tiles (i >= j > p)
        {
            const int m = NSB - p - 1;
            const int nt_ = m * (m + 1) / 2;
            for (int t = wave; t < nt_; t += 4) {
                int ii = 0, rem = t;
                while (rem > ii) { rem -= ii + 1; ++ii; }
                const int i = p + 1 + ii, j = p + 1 + rem;
                d4 acc;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = S[(SB * j + (lane >> 4) + 4 * v) * DLD + SB * i + li];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc = mma_neg(frag(S, DLD, SB * j, o, kk, lane), frag(S, DLD, SB * i, o, kk, lane), acc);
#pragma unroll
                for (int v = 0; v < 4; ++v) S[(SB * j + (lane >> 4) + 4 * v) * DLD + SB * i + li] = acc[v];
            }
        }
        __syncthreads();
    }
    if (wave == 0 && lane == 0 && bad != 0) atomicCAS(&info[b], 0, info_base + GP_TS * k + bad);

    // ---- inverse of the 128 x 128 factor, block column q (and 7 - q) per wave.
    // W_pq^T is parked at sub-block (q, p) of S (strictly upper part): S[(16p + r)*DLD + 16q + c] = W_pq[r][c]
    for (int qq = 0; qq < 2; ++qq) {
        const int q = qq == 0 ? wave : NSB - 1 - wave;
        for (int p = q + 1; p < NSB; ++p) {
            d4 accT = (d4){0.0, 0.0, 0.0, 0.0};   // accT[v] = T[k = (lane>>4)+4v][c = lane&15], T = sum_m L_pm W_mq
            for (int m = q; m < p; ++m) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double lf = frag(S, DLD, SB * p, SB * m, kk, lane);                 // L_pm[r][k]
                    const double wfr = (m == q)
                        ? Wl[q * SB * SB + li * SB + 4 * kk + (lane >> 4)]                   // W_qq[k][c]
                        : frag(S, DLD, SB * q, SB * m, kk, lane);                            // W_mq[k][c]
                    accT = mma(lf, wfr, accT);
                }
            }
            d4 c2 = (d4){0.0, 0.0, 0.0, 0.0};     // W_pq[r = lane&15][c = (lane>>4)+4v] = -inv(L_pp) T
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                c2 = mma_neg(accT[kk], frag(Wl + p * SB * SB, SB, 0, 0, kk, lane), c2);
#pragma unroll
            for (int v = 0; v < 4; ++v) S[(SB * p + li) * DLD + SB * q + (lane >> 4) + 4 * v] = c2[v];
        }
    }
    __syncthreads();

    // ---- write back: factor (lower, zero strictly-upper) and inverse (lower, zero strictly-upper)
    double* invt = inv + b * inv_bstride + (long long)k * GP_TSQ;
    for (int idx = tid; idx < GP_TSQ; idx += 256) {
        const int c = idx >> 7, rr = idx & 127;
        double lv = 0.0, wv = 0.0;
        if (rr >= c) {
            lv = S[c * DLD + rr];
            if ((rr >> 4) == (c >> 4)) wv = Wl[(c >> 4) * SB * SB + (c & 15) * SB + (rr & 15)];
            else wv = S[rr * DLD + c];
        }
        tile[idx] = lv;
        invt[idx] = wv;
    }
}

#endif  // GPSLC_DIAG
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

__global__ __launch_bounds__(256, 2) void diag_potrf_inv_v2_kernel(TRef M, int k, double* inv,
                                                                    long long inv_bstride, int* info,
                                                                    int info_base) {
    extern __shared__ __attribute__((aligned(16))) double P[];   // 36 blocks x 256 + Wcur[256]
    double* Wcur = P + 36 * 256;     // Wcur[c'*16 + c] = inv(L_pp)[c][c'] of the current step
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const long long b = blockIdx.x;
    double* tile = tref_tile(M, b, k, k);
    double* invt = inv + b * inv_bstride + (long long)k * GP_TSQ;
    const int er = tid & 15, ec = tid >> 4;       // this thread's element of a 16 x 16 block

    for (int bi = 0; bi < NSB; ++bi)
        for (int bj = 0; bj <= bi; ++bj)
            BLK(bi, bj)[tid] = tile[(SB * bj + ec) * GP_TS + SB * bi + er];
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
    if (wave == 0 && lane == 0 && bad != 0) atomicCAS(&info[b], 0, info_base + GP_TS * k + bad);
}

#define DIAG2_LDS_BYTES ((36 * 256 + 256) * 8)

void launch_diag(const TRef& M, int k, double* inv, long long inv_bstride, int* info,
                 int info_base, int nbatch, hipStream_t st) {
#ifdef GPSLC_DIAG
#define DIAG_LDS_BYTES ((GP_TS * DLD + NSB * SB * SB) * 8)
    if (diag_env("GPSLC_DIAG_V1", 0) == 1) {
        static DeviceOnce attr_set;
        lds_opt_in(attr_set, (const void*)diag_potrf_inv_kernel, DIAG_LDS_BYTES);
        hipLaunchKernelGGL(diag_potrf_inv_kernel, dim3(nbatch), dim3(256), DIAG_LDS_BYTES, st, M, k, inv,
                           inv_bstride, info, info_base);
        return;
    }
#endif
    static DeviceOnce attr2;
    lds_opt_in(attr2, (const void*)diag_potrf_inv_v2_kernel, DIAG2_LDS_BYTES);
    hipLaunchKernelGGL(diag_potrf_inv_v2_kernel, dim3(nbatch), dim3(256), DIAG2_LDS_BYTES, st, M, k, inv,
                       inv_bstride, info, info_base);
}
