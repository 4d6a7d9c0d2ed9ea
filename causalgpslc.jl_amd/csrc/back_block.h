// Back-substitution L^T alpha = z on 128 x 128 tiles: the tile-times-vector block of backsolve_alpha_kernel /
// backsolve_update_kernel (k_solve.hip), and the whole back-substitution of ONE matrix as a task of the persistent
// factorisation launch (potrf_tasks_kernel, k_tilegemm.hip).  Same block routine, same order of the updates of every z_k
// (tile rows i = nt-1 .. 0): alpha is bit-identical whichever way it is computed.
#pragma once
#include "gpslc_internal.h"

// (t^T v)_c for the 32 columns c = wave*32 .. +31 of a column-major 128 x 128 tile: every lane loads its
// two rows of all 32 columns first (64 independent 8-byte loads in flight), then the wave reduces.
__device__ __forceinline__ void tile_tv32(const double* __restrict__ t, const double* v /*LDS[128]*/,
                                          int wave, int lane, double out[32]) {
    const double v0 = v[lane], v1 = v[lane + 64];
    double p[32];
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) {
        const double* col = t + (wave * 32 + cc) * GP_TS;
        p[cc] = col[lane] * v0 + col[lane + 64] * v1;
    }
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) {
        double x = p[cc];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
        out[cc] = x;
    }
}

// tile_tv32 in pieces for a software pipeline over HALF tiles (16 of a wave's 32 columns): two register sets, the loads of the
// next tile's half fly while the current half is reduced (the loads never depend on the solve, only the products do).
__device__ __forceinline__ void tile_tv16_issue(const double* __restrict__ t, int wave, int half, int lane, double (&raw)[32]) {
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const double* col = t + (wave * 32 + half * 16 + cc) * GP_TS;
        raw[cc] = __builtin_nontemporal_load(col + lane);
        raw[16 + cc] = __builtin_nontemporal_load(col + lane + 64);
    }
}
// The butterfly sums of tile_tv32 (x += shfl_xor(x, o), o = 32, 16, .., 1) for 16 columns without their redundant copies: at
// the steps o = 32 .. 4 a lane keeps the half of its values selected by its own lane bit and receives the partner's partial
// sums for exactly those — the same additions between the same lanes in the same order as the butterfly (an fp add commutes).
// Products as tile_tv32 forms them.  Returns, in lane l, element (l >> 2) of the 16 results.
__device__ __forceinline__ double tile_tv16_finish(const double (&raw)[32], const double* v /*LDS[128]*/, int lane) {
    const double v0 = v[lane], v1 = v[lane + 64];
    double p[16], q8[8], q4[4], q2[2];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) p[cc] = raw[cc] * v0 + raw[16 + cc] * v1;
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double send = b5 ? p[j] : p[8 + j], keep = b5 ? p[8 + j] : p[j];
        q8[j] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double send = b4 ? q8[j] : q8[4 + j], keep = b4 ? q8[4 + j] : q8[j];
        q4[j] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double send = b3 ? q4[j] : q4[2 + j], keep = b3 ? q4[2 + j] : q4[j];
        q2[j] = keep + __shfl_xor(send, 8, 64);
    }
    const double send = b2 ? q2[0] : q2[1], keep = b2 ? q2[1] : q2[0];
    double x = keep + __shfl_xor(send, 4, 64);
    x += __shfl_xor(x, 2, 64);
    x += __shfl_xor(x, 1, 64);
    return x;
}

// The back-substitution of batch element b by ONE workgroup (256 threads): z = row 0 of the augmented tile row, then for
// i = nt-1 .. 0:  alpha_i = inv(L_ii)^T z_i,  z_k -= L(i, k)^T alpha_i (k = i-1 .. 0).  lds: 2 * nt * 128 doubles.
// Tile sequence inv(nt-1), (nt-1, nt-2), ..., (nt-1, 0), inv(nt-2), (nt-2, nt-3), ...
__device__ __forceinline__ void back_task_body(const TRef& M, const TRef& F, const int b, const int nt, double* __restrict__ alpha_out,
                                               double* lds, const int tid) {
    double* z = lds;
    double* al = lds + nt * GP_TS;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = wave * 32 + (lane >> 2);          // this lane's column of a half tile's results (+ 16 for the second half)
    const bool writer = (lane & 3) == 0;
    double ra[32], rb[32];                            // the two halves of the tile in flight
    tile_tv16_issue(tref_tile(F, b, 0, nt - 1), wave, 0, lane, ra);
    tile_tv16_issue(tref_tile(F, b, 0, nt - 1), wave, 1, lane, rb);
    for (int e = tid; e < nt * GP_TS; e += 256) z[e] = tref_tile(M, b, nt, e >> 7)[(e & 127) * GP_TS];
    __syncthreads();
    for (int i = nt - 1; i >= 0; --i) {
        {   // alpha_i = inv(L_ii)^T z_i; next tile: (i, i - 1)
            const double* nxt = i > 0 ? tref_tile(M, b, i, i - 1) : nullptr;
            const double m0 = tile_tv16_finish(ra, z + i * GP_TS, lane);
            if (nxt) tile_tv16_issue(nxt, wave, 0, lane, ra);
            const double m1 = tile_tv16_finish(rb, z + i * GP_TS, lane);
            if (nxt) tile_tv16_issue(nxt, wave, 1, lane, rb);
            if (writer) {
                al[i * GP_TS + col] = m0;
                al[i * GP_TS + col + 16] = m1;
                alpha_out[i * GP_TS + col] = m0;
                alpha_out[i * GP_TS + col + 16] = m1;
            }
        }
        __syncthreads();
        for (int k = i - 1; k >= 0; --k) {
            // z_k -= L(i, k)^T alpha_i; next tile: (i, k - 1), or inv(L_{i-1,i-1}) when the row is done
            const double* nxt = k > 0 ? tref_tile(M, b, i, k - 1) : tref_tile(F, b, 0, i - 1);
            const double m0 = tile_tv16_finish(ra, al + i * GP_TS, lane);
            tile_tv16_issue(nxt, wave, 0, lane, ra);
            const double m1 = tile_tv16_finish(rb, al + i * GP_TS, lane);
            tile_tv16_issue(nxt, wave, 1, lane, rb);
            if (writer) {
                z[k * GP_TS + col] -= m0;
                z[k * GP_TS + col + 16] -= m1;
            }
        }
        __syncthreads();
    }
}
