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

#include "diag_block.h"

#define DLD 130   // LDS leading dimension (doubles), version 1

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
__global__ __launch_bounds__(256, 2) void diag_potrf_inv_v2_kernel(TRef M, int k, double* inv,
                                                                    long long inv_bstride, int* info,
                                                                    int info_base) {
    extern __shared__ __attribute__((aligned(16))) double P[];   // 36 blocks x 256 + Wcur[256]
    const long long b = blockIdx.x;
    diag_potrf_inv_v2_body(P, tref_tile(M, b, k, k), inv + b * inv_bstride + (long long)k * GP_TSQ, info + b,
                           info_base + GP_TS * k, (int)threadIdx.x, false);
}

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
