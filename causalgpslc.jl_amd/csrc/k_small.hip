// Single-launch Gaussian-process node score for the reference's own problem sizes (n <= ~190 instances:
// NEEC 150, the test CSVs 20-100; SURVEY.md §8f next-1, src/model_likelihood.jl:4-120, src/inference.jl:21-56).
//
//     log N(target; 0, scale * exp.(rbfKernelLog(F, F, ls)) + noise * I)
//
// One workgroup = one node (one Gen address: :X => k => :X, :T / :logitT, :Y), any number of nodes per launch,
// every node with its own feature block.  The whole chain — Gram build, blocked Cholesky, forward solve of the
// target, log-determinant, quadratic form — runs inside the workgroup on an LDS-resident image of the matrix:
// the general path spends ~15 dependent launches (20-60 us each) on the same score, and an `mh` / slice step of
// the Markov chain waits for exactly that.
//
// CDNA4 mapping (8 wave64 per workgroup, one workgroup per CU)
//   * the lower triangle lives in LDS as packed 16 x 16 fp64 blocks (2 KiB each, column-major): n = 150 -> 55 blocks
//     = 110 KiB of the CU's 160 KiB; the scaled features (n x nF) and the right-hand side sit beside it;
//   * right-looking blocked Cholesky, block size 16 = one f64 MFMA tile.  Per block column p:
//       factor + panel in ONE register-resident pass: the Cholesky of the 16 x 16 diagonal block is a sequence of
//               column operations (scale column c by 1/sqrt(pivot), subtract multiples of it from the columns right
//               of it); applied to the rows BELOW the block they turn A_ip into X_i = A_ip L_pp^-T, and applied to
//               the right-hand side row they turn y_p into z_p.  Every wave keeps its own copy of the diagonal
//               block's 16 rows in lanes 0-15 (row i in lane i, the pivot row travels by v_readlane) and 48 of the
//               rows below in lanes 16-63: the panel solve costs no instruction and no barrier of its own, and no
//               inverse of L_pp is ever formed;
//       update  A_ij -= X_i X_j^T (4 MFMAs per block) dealt to the 8 waves, y_j -= z_p X_j^T on the VALU;
//     two barriers per block column.  The critical path is the chain of 16 dependent pivots per block: reciprocal
//     square root by v_rsq_f64 + one third-order correction, the column scaling uses it directly (the square root
//     itself is only needed for the diagonal entry and is finished off the chain);
//   * inputs are read straight from the pinned host staging buffer (a few KB, coalesced, once) and the three result
//     words are written back to it: a score costs one kernel launch and one stream synchronisation.
#include "gpslc_internal.h"
#include "gp_math.h"

#include "sm_blocks.h"

// one Gram entry: scale * exp(-sum_f (x_f/l_f - x'_f/l_f)^2) (+ noise on the diagonal); identity on the padding
__device__ __forceinline__ double sm_gram_entry(const double* fs, int NP, int nF, int n, int i, int j, double scale,
                                                double noise) {
    if (i >= n || j >= n) return (i == j) ? 1.0 : 0.0;
    double lux = 0.0;
    for (int f = 0; f < nF; ++f) {
        const double d = fs[f * NP + i] - fs[f * NP + j];
        lux = fma(d, d, lux);
    }
    const double v = scale * gp_exp_neg(-lux);
    return (i == j) ? v + noise : v;
}

__global__ __launch_bounds__(SM_THREADS) void small_gp_logpdf_kernel(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) double P[];
    const unsigned long long tstart = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
    const SmallNode nd = blockIdx.x < SMALL_INLINE_NODES ? a.inl[blockIdx.x] : a.nodes[blockIdx.x];
    const int n = a.n, NB = a.NB, NP = NB * SB;
    const int NBLK = NB * (NB + 1) / 2;
    double* yv = P + NBLK * 256;      // right-hand side, updated in place block by block
    double* zv = yv + NP;             // z = L^-1 target
    double* ldv = zv + NP;            // diag(L), for the log-determinant
    double* bcl = ldv + NP;           // [8 waves][16]: multiplier broadcast lines of the pivot chains (sm_factor_rows_lds)
    double* fs = bcl + SM_WAVES * SB; // scaled features, fs[f*NP + i] = F[i, f] * (1 / ls[f]) (scaled by the host)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: branches on it are scalar branches
    const int li = lane & 15;

    for (int idx = tid; idx < nd.nF * NP; idx += SM_THREADS) {
        const int f = idx / NP, i = idx - f * NP;
        fs[idx] = (i < n) ? nd.Fs[(long long)f * n + i] : 0.0;
    }
    for (int i = tid; i < NP; i += SM_THREADS) yv[i] = (i < n) ? nd.target[i] : 0.0;
    __syncthreads();
    const unsigned long long t0 = a.stamps ? __builtin_amdgcn_s_memtime() : 0;

    // ---- Gram build into the packed blocks (src/kernel.jl:13-32, 53-59): two blocks per wave and iteration = eight
    // independent entries per lane, the feature loop outermost, so that eight distance accumulations and then eight exp
    // chains interleave (a wave's dependent chains would otherwise run back to back: one or two waves per SIMD here)
    {
        int bi = 0, bj = 0;                              // block `wave` of the row-major lower-triangle enumeration
        for (int t = 0; t < wave; ++t) { if (++bj > bi) { ++bi; bj = 0; } }
        for (int blk = wave; blk < NBLK; blk += 2 * SM_WAVES) {
            int bi2 = bi, bj2 = bj;                      // the second block of this iteration: blk + SM_WAVES
            for (int t = 0; t < SM_WAVES; ++t) { if (++bj2 > bi2) { ++bi2; bj2 = 0; } }
            const bool two = blk + SM_WAVES < NBLK;
            int gi[8], gj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int w = lane + 64 * (u & 3);
                gi[u] = SB * (u < 4 ? bi : bi2) + (w & 15);
                gj[u] = SB * (u < 4 ? bj : bj2) + (w >> 4);
            }
            double v[8];
            if (nd.cov) {          // dense covariance node (uCov = SigmaU * uNoise, src/model_likelihood.jl:4-10)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = (gi[u] < n && gj[u] < n && (u < 4 || two)) ? nd.covscale * nd.cov[(long long)gj[u] * n + gi[u]]
                                                                       : (gi[u] == gj[u] ? 1.0 : 0.0);
            } else {
                double lux[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) lux[u] = 0.0;
                for (int f = 0; f < nd.nF; ++f) {
                    const double* ff = fs + f * NP;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double d = ff[gi[u]] - ff[gj[u]];      // padded instances carry zeros: harmless, masked below
                        lux[u] = fma(d, d, lux[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double e = nd.scale * gp_exp_neg(-lux[u]);
                    v[u] = (gi[u] < n && gj[u] < n) ? (gi[u] == gj[u] ? e + nd.noise : e) : (gi[u] == gj[u] ? 1.0 : 0.0);
                }
            }
            double* B = P + (blk << 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) B[lane + 64 * u] = v[u];
            if (two) {
                double* B2 = P + ((blk + SM_WAVES) << 8);
#pragma unroll
                for (int u = 0; u < 4; ++u) B2[lane + 64 * u] = v[4 + u];
            }
            for (int t = 0; t < 2 * SM_WAVES; ++t) { if (++bj > bi) { ++bi; bj = 0; } }
        }
    }
    int bad = 0;
    unsigned long long t_fac = 0, t_upd = 0, tq = 0, t_chain = 0;    // measurement build only
    __syncthreads();
    if (a.stamps) tq = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = tq;

    // ---- factor + panel of block column p: the column operations of chol(A_pp) on the diagonal block's rows (lanes 0-15
    // of every wave that carries rows) and on the rows below / the right-hand side (lanes 16-63: 48 rows per wave)
    auto factor_panel = [&](int p) {
        const int rows = NP - SB * (p + 1);              // matrix rows below the block; row index `rows` = y
        if (wave * 48 <= rows) {                         // scalar: this wave has rows to carry
            const bool is_diag = lane < SB;
            const int q = wave * 48 + (lane - SB);       // this lane's row below the block (lanes >= 16)
            double r[SB];
            double* dst = nullptr;                       // where this lane's row lives (null: nothing to carry)
            if (is_diag) dst = SBLK(p, p) + li;
            else if (q < rows) { const int gr = SB * (p + 1) + q; dst = SBLK(gr >> 4, p) + (gr & 15); }
            const bool is_y = !is_diag && q == rows;
            if (dst) {
#pragma unroll
                for (int c = 0; c < SB; ++c) r[c] = dst[c * SB];
            } else if (is_y) {
#pragma unroll
                for (int c = 0; c < SB; ++c) r[c] = yv[SB * p + c];
            } else {
#pragma unroll
                for (int c = 0; c < SB; ++c) r[c] = 0.0;
            }
            double lcc;
            sm_factor_rows_lds(r, li, lane, SB * p, bad, lcc, bcl + wave * SB);
            if (is_diag) {
                if (wave == 0) {
                    ldv[SB * p + li] = lcc;              // the score never reads the factor's diagonal block again
                    if (a.draw) {
                        // a draw does.  The other carrying waves may still be loading their copy of the block's rows, of
                        // which only the lower triangle matters: L_pp goes, transposed, into the strictly UPPER triangle
                        // (element (c, li) <- L[li][c], c < li); its diagonal is ldv.
                        double* up = SBLK(p, p) + li * SB;
#pragma unroll
                        for (int c = 0; c < SB; ++c)
                            if (c < li) up[c] = r[c];
                    }
                }
            } else if (dst) {
#pragma unroll
                for (int c = 0; c < SB; ++c) dst[c * SB] = r[c];
            } else if (is_y) {
#pragma unroll
                for (int c = 0; c < SB; ++c) zv[SB * p + c] = r[c];
            }
        }
    };
    // y_j[c] -= sum_k z_p[k] X_jp[c][k] for the entries q = first, first + step, ... of the blocks j = p + 1 + (q >> 4)
    auto update_y = [&](int p, int first, int step, int q_begin, int q_end) {
        for (int q = q_begin + first; q < q_end; q += step) {
            const int j = p + 1 + (q >> 4), c = q & 15;
            const double* X = SBLK(j, p);
            double acc = yv[SB * j + c];
#pragma unroll
            for (int k = 0; k < SB; ++k) acc = fma(-zv[SB * p + k], X[k * SB + c], acc);
            yv[SB * j + c] = acc;
        }
    };

    // Right-looking with a one-column lookahead.  After column p is factored, only the blocks of column p + 1 (and the
    // right-hand side entries of block p + 1) are brought up to date by everybody; then the waves that carry rows
    // factor column p + 1 while the OTHER waves apply column p to the rest of the trailing matrix — the serial pivot
    // chain of the next column runs under the bulk of this column's updates.
    factor_panel(0);
    __syncthreads();
    if (a.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_fac += t - tq; tq = t; }
    for (int p = 0; p + 1 < NB; ++p) {
        const int m = NB - p - 1;
        // ---- blocks (p + 1 + ii, p + 1), ii < m: two independent blocks per wave and iteration (their MFMA chains interleave)
        for (int t = wave; t < m; t += 2 * SM_WAVES) {
            const int t2 = t + SM_WAVES;
            if (t2 < m)
                sm_update2(SBLK(p + 1 + t, p + 1), SBLK(p + 1 + t, p), SBLK(p + 1, p),
                           SBLK(p + 1 + t2, p + 1), SBLK(p + 1 + t2, p), SBLK(p + 1, p), lane);
            else
                sm_update(SBLK(p + 1 + t, p + 1), SBLK(p + 1 + t, p), SBLK(p + 1, p), lane);
        }
        if (wave == SM_WAVES - 1) update_y(p, lane, 64, 0, SB);
        __syncthreads();
        if (a.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_upd += t - tq; tq = t; }
        // ---- factor + panel of column p + 1  ||  the rest of column p's trailing update: blocks (i, j), p + 2 <= j <= i
        {
            const int rows1 = NP - SB * (p + 2);
            const int nc = min(SM_WAVES, rows1 / 48 + 1);               // waves that carry rows of column p + 1 (scalar)
            const unsigned long long tf0 = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
            if (wave < nc) factor_panel(p + 1);
            if (a.stamps) t_chain += __builtin_amdgcn_s_memtime() - tf0;
            // the updaters: every wave that carries nothing (all of them, after their chains, should every wave carry)
            const int u0 = nc == SM_WAVES ? 0 : nc;
            if (wave >= u0) {
                const int w = wave - u0, nw = SM_WAVES - u0;
                const int m1 = m - 1, cnt = m1 * (m1 + 1) / 2;          // triangle over (ii, rem), 1 <= rem <= ii <= m - 1
                for (int t = w; t < cnt; t += 2 * nw) {
                    int ii = 0, rem = t;
                    while (rem > ii) { rem -= ii + 1; ++ii; }
                    const int bi = p + 2 + ii, bj = p + 2 + rem;
                    const int t2 = t + nw;
                    if (t2 < cnt) {
                        int i2 = 0, rem2 = t2;
                        while (rem2 > i2) { rem2 -= i2 + 1; ++i2; }
                        sm_update2(SBLK(bi, bj), SBLK(bi, p), SBLK(bj, p),
                                   SBLK(p + 2 + i2, p + 2 + rem2), SBLK(p + 2 + i2, p), SBLK(p + 2 + rem2, p), lane);
                    } else {
                        sm_update(SBLK(bi, bj), SBLK(bi, p), SBLK(bj, p), lane);
                    }
                }
                update_y(p, w * 64 + lane, nw * 64, SB, m * SB);
            }
        }
        __syncthreads();
        if (a.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_fac += t - tq; tq = t; }
    }

    if (a.draw) {          // draw = L * target: row i of the factor from the LDS blocks, target from the (pinned) input
        for (int i = tid; i < NP; i += SM_THREADS) yv[i] = (i < n) ? nd.target[i] : 0.0;
        __syncthreads();
        for (int i = tid; i < n; i += SM_THREADS) {
            const int bi = i >> 4, ri = i & 15;
            double acc = 0.0;
            for (int bk = 0; bk < bi; ++bk) {
                const double* B = SBLK(bi, bk);
                for (int j = 0; j < SB; ++j) acc = fma(B[j * SB + ri], yv[SB * bk + j], acc);
            }
            const double* D = SBLK(bi, bi) + ri * SB;            // row ri of L_pp, kept transposed above the diagonal
            for (int j = 0; j < ri; ++j) acc = fma(D[j], yv[SB * bi + j], acc);
            acc = fma(ldv[i], yv[i], acc);
            a.draw[(long long)blockIdx.x * n + i] = acc;
        }
    }
    if (wave == 0) {
        double q = 0.0, ld = 0.0;
        for (int i = lane; i < n; i += 64) {
            q = fma(zv[i], zv[i], q);
            ld += log(ldv[i]);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            q += __shfl_xor(q, o, 64);
            ld += __shfl_xor(ld, o, 64);
        }
        if (lane == 0) {
            double* o = a.out + 4 * (long long)blockIdx.x;
            o[0] = 2.0 * ld;
            o[1] = q;
            o[2] = (double)bad;
            o[3] = 0.0;
            if (a.stamps) {    // shader-clock ticks: inputs, Gram, factor+panel phases, update phases, -, total
                double* sp = a.stamps + 8 * (long long)blockIdx.x;
                sp[0] = (double)(t0 - tstart); sp[1] = (double)(t1 - t0); sp[2] = (double)t_fac; sp[3] = (double)t_upd;
                sp[4] = (double)t_chain; sp[5] = (double)(__builtin_amdgcn_s_memtime() - tstart);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same score for the sizes between the LDS-resident kernel and the batched tiled path (176 < n <= 640: IHDP
// n = 272): one workgroup per node, LEFT-looking over block columns.  Only the current block column lives in LDS; the
// finished columns X(i, k) = L blocks sit in a per-node global scratch that stays in L2 (n = 272: 306 KiB).
// Every wave owns up to MID_MAXI block rows of a column and keeps their 16 x 16 blocks in MFMA accumulators from the
// Gram evaluation to the hand-over to the factorisation.  Step p of the main loop, two barriers:
//   phase B   waves that carry rows: factor + panel of column p — exactly the register-resident column operations of the
//             small kernel (sm_factor_rows_lds) on the LDS image of the column, 48 rows per wave; the finished rows go
//             from the registers straight to the scratch.
//             EVERY wave, for its rows of column p + 1: Gram blocks straight into the accumulator layout, then
//             -= sum_{k<p} X(i,k) X(p+1,k)^T  (operand fragments: 512-byte coalesced reads of the scratch through a
//             ring of register sets; X(p+1,k) is shared by the wave's rows).  None of this needs column p, so it runs
//             while column p is being factored; block rows are dealt from the LAST wave downwards, so that the waves
//             that carry the factorisation (the first ones) own the fewest rows.
//   phase A   the one k step that needs column p, then the accumulators become the LDS image of column p + 1.
// The right-hand side rides along as block row NB (its row 0 = target, the other 15 rows zero): z = L^-1 target
// falls out of the same block operations, no special case.
// ---------------------------------------------------------------------------------------------------------------------
#define MID_MAXI 6         // block rows per wave at most: (NB + 1) <= 8 * MID_MAXI  ->  NB <= 47, limited to n <= 640 by the host
#define MBLK(i, j) (X + ((((long long)(i) * ((i) + 1)) / 2 + (j)) << 8))     // scratch: packed lower blocks, rows 0..NB

// operand-fragment register sets of the k loop for a wave that owns R block rows: the fragments of steps
// k+1 .. k+NSET-1 are in flight (L2 latency ~ 1000 cycles) while the MFMAs of step k (256 cycles per block row) run;
// a set is 8 (1 + R) VGPRs
__host__ __device__ constexpr int mid_nset(int R) { return R <= 1 ? 6 : R == 2 ? 5 : R == 3 ? 4 : R == 4 ? 3 : 2; }

struct MidCtx {
    double* X;              // scratch: finished block columns
    double* Cp;             // LDS image of the block column being factored: slot (i - p) = block (i, p)
    double* Cn;             // LDS image of the next block column under construction (== Cp without the second buffer)
    double* ldv;            // LDS: diag(L)
    double* bc;             // LDS: this wave's 16-double multiplier broadcast line (sm_factor_rows_lds)
    double* Xrow;           // LDS (XR): the finished blocks X(q, 0 .. q-2) of the block row the next k loop shares
    const double* fl;       // LDS features (FL: [f][NP]; else the block column's [f][16])
    const double* featg;    // !FL: features in the scratch [f][NP]
    const double* tgt;      // right-hand side [NP]
    int n, NB, NP, wave, lane, tid;
    int own;                // block rows of column q owned by this wave: i = q + own + 8u (own = 7 - wave)
    int draw;               // the factor's diagonal blocks are kept too (SmallArgs::draw)
#ifdef GPSLC_DIAG
    long long* tt;          // per-wave phase clocks (measurement build)
#endif
};
#ifdef GPSLC_DIAG
#define MID_STAMP(tt, j) do { const long long now_ = __builtin_amdgcn_s_memtime(); (tt)[j] += now_ - (tt)[7]; (tt)[7] = now_; } while (0)
#else
#define MID_STAMP(tt, j) do { } while (0)
#endif

// Gram blocks (i, q) of the wave's R rows together: element (row 16i + li, col 16q + lg + 4v); the feature loop is the outer
// one: 4 R independent distance / exp chains in flight.  A right-hand side row (i == NB) computes a throw-away block
// on the last real block row's features and is then overwritten with the target.
template <int R, bool FL>
__device__ __forceinline__ void mid_gram(const SmallNode& nd, const MidCtx& c, int q, d4 (&acc)[R]) {
    const int lane = c.lane, li = lane & 15, lg = lane >> 4, NP = c.NP, NB = c.NB, n = c.n;
    const int i0 = q + c.own;
    const bool last_rhs = (i0 + SM_WAVES * (R - 1) == NB);       // scalar
    double lux[R][4];
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) lux[u][v] = 0.0;
    if (!nd.cov) {
        for (int f = 0; f < nd.nF; ++f) {
            double cf[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) cf[v] = FL ? c.fl[f * NP + SB * q + lg + 4 * v] : c.fl[f * SB + lg + 4 * v];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = min(i0 + SM_WAVES * u, NB - 1);
                const double xr = FL ? c.fl[f * NP + SB * i + li] : c.featg[f * NP + SB * i + li];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double d = xr - cf[v];
                    lux[u][v] = fma(d, d, lux[u][v]);
                }
            }
        }
    }
    if (nd.cov) {        // dense covariance handed over by the caller (scalar branch: one code path per node)
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int gi = SB * min(i0 + SM_WAVES * u, NB - 1) + li;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int gj = SB * q + lg + 4 * v;
                const bool in = gi < n && gj < n;
                const double cv = nd.cov[in ? (long long)gj * n + gi : 0];
                acc[u][v] = in ? nd.covscale * cv : (gi == gj ? 1.0 : 0.0);
            }
        }
    } else {             // branch-free: the exp chains of all 4 R entries interleave
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int gi = SB * min(i0 + SM_WAVES * u, NB - 1) + li;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int gj = SB * q + lg + 4 * v;
                const double e = nd.scale * gp_exp_neg(-lux[u][v]) + (gi == gj ? nd.noise : 0.0);
                acc[u][v] = (gi < n && gj < n) ? e : (gi == gj ? 1.0 : 0.0);
            }
        }
    }
    if (last_rhs) {      // right-hand side block: row 0 = target
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[R - 1][v] = (li == 0) ? c.tgt[SB * q + lg + 4 * v] : 0.0;
    }
}

// acc(i, q) -= sum_{k0 <= k < k1} X(i,k) X(q,k)^T for the wave's R rows: a ring of NSET operand sets, unrolled by NSET so
// that the set indices are static.  R is a template parameter and the wave index a scalar, so the body is straight-line
// code: the compiler counts the outstanding loads exactly (s_waitcnt vmcnt(N)) and the prefetch ring really overlaps.
// XR: the shared operand X(q, k) comes from its LDS copy (staged once per column for all eight waves) instead of the scratch
template <int R, bool XR>
__device__ __forceinline__ void mid_kloop(const MidCtx& c, int q, int k0, int k1, d4 (&acc)[R]) {
    constexpr int NSET = mid_nset(R);
    const double* X = c.X;
    const int lane = c.lane;
    const int i0 = q + c.own;
    double fj[NSET][4], fi[NSET][R][4];
    auto fetch = [&](int k, double (&fjs)[4], double (&fis)[R][4]) {
        const double* Xqk = XR ? c.Xrow + (k << 8) : MBLK(q, k);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fjs[kk] = sm_frag(Xqk, kk, lane);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const double* Xik = MBLK(i0 + SM_WAVES * u, k);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fis[u][kk] = sm_frag(Xik, kk, lane);
        }
    };
    auto apply = [&](const double (&fjs)[4], const double (&fis)[R][4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int u = 0; u < R; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fjs[kk], fis[u][kk], acc[u], 0, 0, 1);
    };
#pragma unroll
    for (int s = 0; s < NSET - 1; ++s)
        if (k0 + s < k1) fetch(k0 + s, fj[s], fi[s]);
    for (int kb = k0; kb < k1; kb += NSET) {
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            const int k = kb + s;
            if (k < k1) {
                if (k + NSET - 1 < k1) fetch(k + NSET - 1, fj[(s + NSET - 1) % NSET], fi[(s + NSET - 1) % NSET]);
                apply(fj[s], fi[s]);
            }
        }
    }
}

// factor + panel of column p on its LDS image: rows below the diagonal block = 16 (NBa - p - 1), 48 per wave and pass.
// DB: the finished rows also replace their LDS image, which phase A reads its operands from.
template <bool DB>
__device__ __forceinline__ void mid_factor(const MidCtx& c, int p, int& bad) {
    double* X = c.X;
    const int lane = c.lane, li = lane & 15, wave = c.wave;
    const int rows = SB * (c.NB + 1 - p - 1);
    for (int q0 = 0; q0 < rows; q0 += 48 * SM_WAVES) {
        if (q0 + wave * 48 < rows) {                                 // scalar: this wave carries rows
            const bool is_diag = lane < SB;
            const int q = q0 + wave * 48 + (lane - SB);
            double r[SB];
            double* src = nullptr;
            if (is_diag) src = c.Cp + li;
            else if (q < rows) src = c.Cp + (((q >> 4) + 1) << 8) + (q & 15);
            if (src) {
#pragma unroll
                for (int cc = 0; cc < SB; ++cc) r[cc] = src[cc * SB];
            } else {
#pragma unroll
                for (int cc = 0; cc < SB; ++cc) r[cc] = 0.0;
            }
            double lcc;
            sm_factor_rows_lds(r, li, lane, SB * p, bad, lcc, c.bc);
            if (is_diag) {
                if (wave == 0 && q0 == 0) {
                    c.ldv[SB * p + li] = lcc;
                    if (c.draw) {                        // L_pp: lower triangle, exact diagonal, zeros above
                        double* dd = MBLK(p, p) + li;
#pragma unroll
                        for (int cc = 0; cc < SB; ++cc) dd[cc * SB] = (li > cc) ? r[cc] : (li == cc ? lcc : 0.0);
                    }
                }
            } else if (src) {
                double* dst = MBLK(p + 1 + (q >> 4), p) + (q & 15);
#pragma unroll
                for (int cc = 0; cc < SB; ++cc) dst[cc * SB] = r[cc];
                if (DB) {
#pragma unroll
                    for (int cc = 0; cc < SB; ++cc) src[cc * SB] = r[cc];
                }
            }
        }
    }
}

// the k step that needs column p, operands from the LDS image of column p: X(q, p) = slot 1, X(i, p) = slot i - p
template <int R>
__device__ __forceinline__ void mid_kstep_lds(const MidCtx& c, d4 (&acc)[R]) {
    const int lane = c.lane;
    double fj[4], fi[R][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fj[kk] = sm_frag(c.Cp + 256, kk, lane);
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fi[u][kk] = sm_frag(c.Cp + ((c.own + SM_WAVES * u + 1) << 8), kk, lane);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int u = 0; u < R; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[kk], fi[u][kk], acc[u], 0, 0, 1);
}

// one step of the main loop for a wave that owns EXACTLY R block rows of column q = p + 1 (p = -1: the first column)
template <int R, bool FL, bool DB, bool XR>
__device__ __forceinline__ void mid_step(const SmallNode& nd, const MidCtx& c, int p, int& bad) {
    const int q = p + 1;
    constexpr int RA = R > 0 ? R : 1;
    d4 acc[RA];
    if (p >= 0) mid_factor<DB>(c, p, bad);
    MID_STAMP(c.tt, 4);
    if (R > 0) {
        mid_gram<RA, FL>(nd, c, q, acc);
        MID_STAMP(c.tt, 0);
        if (p > 0) mid_kloop<RA, XR>(c, q, 0, p, acc);
        MID_STAMP(c.tt, 1);
    }
    __threadfence_block();
    __syncthreads();                 // column p is finished: in the scratch (and, DB, in its LDS image)
    MID_STAMP(c.tt, 3);
    if (XR && p >= 0 && q + 1 < c.NB) {      // the next column's shared operand row: blocks (q + 1, 0 .. p), contiguous in the scratch
        const double* X = c.X;
        const double* src = MBLK(q + 1, 0);
        for (int idx = c.tid; idx < (p + 1) * 256; idx += SM_THREADS) c.Xrow[idx] = src[idx];
    }
    if (R > 0) {
        if (p >= 0) {
            if (DB) mid_kstep_lds<RA>(c, acc);
            else mid_kloop<RA, false>(c, q, p, p + 1, acc);
        }
        const int li = c.lane & 15, lg = c.lane >> 4;
#pragma unroll
        for (int u = 0; u < RA; ++u) {
            double* B = c.Cn + ((c.own + SM_WAVES * u) << 8);
#pragma unroll
            for (int v = 0; v < 4; ++v) B[(lg + 4 * v) * SB + li] = acc[u][v];
        }
    }
    MID_STAMP(c.tt, 2);
    __syncthreads();                 // the LDS image of column q is complete
    MID_STAMP(c.tt, 5);
}

// FL    scaled features + target resident in LDS for the whole kernel (they fit for every size the reference uses);
//       otherwise they sit in the scratch and the 16 instances of the current block column are staged per column
// DB    two LDS column images: the factored column p stays readable (operands of the one k step that needs it) while
//       column p + 1 is written
// XR    an LDS copy of the block row X(q, .) every wave's k loop shares
template <bool FL, bool DB, bool XR>
__global__ __launch_bounds__(SM_THREADS) void mid_gp_logpdf_kernel(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) double P[];
    const SmallNode nd = blockIdx.x < SMALL_INLINE_NODES ? a.inl[blockIdx.x] : a.nodes[blockIdx.x];
    const int n = a.n, NB = a.NB, NP = NB * SB, NBa = NB + 1;
    double* __restrict__ X = a.scratch + (long long)blockIdx.x * a.scratch_stride;     // finished columns
    double* C0 = P;                       // block column image(s): slot (i - p) = block (i, p), i = p..NB
    double* ldv = C0 + (DB ? 2 : 1) * NBa * 256;   // diag(L)
    double* bcl = ldv + NP;               // [8 waves][16] multiplier broadcast lines
    double* xrow = bcl + SM_WAVES * SB;   // XR: [NB][256]
    double* fl = xrow + (XR ? NB * 256 : 0);                // FL: features [f][NP] then target [NP];  else: the block column's features [f][16]
    double* featg = X + ((long long)NBa * (NBa + 1) / 2) * 256;                        // !FL: features [f][NP], target [NP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: every branch on it is a scalar branch

    {
        double* dstf = FL ? fl : featg;
        for (int idx = tid; idx < nd.nF * NP; idx += SM_THREADS) {
            const int f = idx / NP, i = idx - f * NP;
            dstf[idx] = (i < n) ? nd.Fs[(long long)f * n + i] : 0.0;
        }
        // the right-hand side too: it is read once per block column, and nd.target lives in pinned HOST memory
        double* dstt = dstf + (long long)nd.nF * NP;
        for (int i = tid; i < NP; i += SM_THREADS) dstt[i] = (i < n) ? nd.target[i] : 0.0;
    }
    MidCtx mc;
    mc.X = X; mc.Cp = C0; mc.Cn = C0; mc.ldv = ldv; mc.Xrow = xrow; mc.tid = tid; mc.bc = bcl + wave * SB; mc.draw = a.draw != nullptr; mc.fl = fl; mc.featg = featg; mc.tgt = (FL ? fl : featg) + (long long)nd.nF * NP;
    mc.n = n; mc.NB = NB; mc.NP = NP; mc.wave = wave; mc.lane = lane; mc.own = SM_WAVES - 1 - wave;
#ifdef GPSLC_DIAG
    long long tt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mc.tt = tt;
#endif
    __threadfence_block();
    __syncthreads();
    int bad = 0;
#ifdef GPSLC_DIAG
    tt[7] = __builtin_amdgcn_s_memtime();
#endif

    for (int p = -1; p < NB; ++p) {
        const int q = p + 1;
        // image of column p (factored in this step) / of column q (built in this step)
        mc.Cp = C0 + ((DB && (p & 1)) ? NBa * 256 : 0);
        mc.Cn = C0 + ((DB && (q & 1)) ? NBa * 256 : 0);
        if (!FL && q < NB) {       // stage the features of block column q (read by mid_gram only, which nobody runs right now)
            for (int idx = tid; idx < nd.nF * SB; idx += SM_THREADS) fl[idx] = featg[(idx >> 4) * NP + SB * q + (idx & 15)];
            __syncthreads();
        }
        // block rows of column q owned by this wave: i = q + own + 8u <= NB (none once q == NB)
        const int span = NB - q - mc.own;
        const int nrows = (q < NB && span >= 0) ? span / SM_WAVES + 1 : 0;               // scalar
        switch (nrows) {
            case 0: mid_step<0, FL, DB, XR>(nd, mc, p, bad); break;
            case 1: mid_step<1, FL, DB, XR>(nd, mc, p, bad); break;
            case 2: mid_step<2, FL, DB, XR>(nd, mc, p, bad); break;
            case 3: mid_step<3, FL, DB, XR>(nd, mc, p, bad); break;
            case 4: mid_step<4, FL, DB, XR>(nd, mc, p, bad); break;
            case 5: mid_step<5, FL, DB, XR>(nd, mc, p, bad); break;
            default: mid_step<6, FL, DB, XR>(nd, mc, p, bad); break;
        }
    }
#ifdef GPSLC_DIAG
    if (a.stamps && blockIdx.x == 0 && lane == 0)
        for (int j = 0; j < 6; ++j) a.stamps[8 * (long long)gridDim.x + 8 * wave + j] = (double)tt[j];
#endif

    if (a.draw) {          // draw = L * target: row i of the factor from the scratch (L2), target from its staged copy
        for (int i = tid; i < n; i += SM_THREADS) {
            const int bi = i >> 4, ri = i & 15;
            double acc = 0.0;
            for (int bk = 0; bk <= bi; ++bk) {
                const double* B = MBLK(bi, bk);
                const int jmax = bk == bi ? ri : SB - 1;
                for (int j = 0; j <= jmax; ++j) acc = fma(B[j * SB + ri], mc.tgt[SB * bk + j], acc);
            }
            a.draw[(long long)blockIdx.x * n + i] = acc;
        }
    }
    if (wave == 0) {      // z = row 0 of the right-hand side blocks (NB, k): z[16k + c] = X(NB, k)[c*16 + 0]
        double q = 0.0, ld = 0.0;
        for (int i = lane; i < n; i += 64) {
            const double z = MBLK(NB, i >> 4)[(i & 15) * SB];
            q = fma(z, z, q);
            ld += log(ldv[i]);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            q += __shfl_xor(q, o, 64);
            ld += __shfl_xor(ld, o, 64);
        }
        if (lane == 0) {
            double* o = a.out + 4 * (long long)blockIdx.x;
            o[0] = 2.0 * ld;
            o[1] = q;
            o[2] = (double)bad;
            o[3] = 0.0;
        }
    }
}

size_t mid_gp_scratch_doubles(int n, int nF) {
    const long long NB = (n + SB - 1) / SB, NBa = NB + 1;
    return (size_t)(NBa * (NBa + 1) / 2 * 256 + (long long)(nF + 1) * NB * SB);
}
// LDS: column image(s) + diag(L) [+ shared operand row] + the features resident (preferred) or staged per block column
static size_t mid_lds(int n, int nF, bool resident, bool two_images, bool xrow) {
    const size_t NB = (n + SB - 1) / SB;
    return ((two_images ? 2 : 1) * (NB + 1) * 256 + NB * SB + SM_WAVES * SB + (xrow ? NB * 256 : 0) +
            (resident ? (size_t)(nF + 1) * NB * SB : (size_t)nF * SB)) * 8;
}
static void mid_choose(int n, int nF, bool& resident, bool& two_images, bool& xrow) {
    const size_t cap = 160 * 1024;
    xrow = mid_lds(n, nF, true, true, true) <= cap;
    two_images = xrow || mid_lds(n, nF, true, true, false) <= cap;
    resident = two_images || mid_lds(n, nF, true, false, false) <= cap;
}
size_t mid_gp_lds_bytes(int n, int nF) {
    bool r, d, x;
    mid_choose(n, nF, r, d, x);
    return mid_lds(n, nF, r, d, x);
}
bool mid_gp_fits(int n) { return (n + SB - 1) / SB + 1 <= SM_WAVES * MID_MAXI && n <= 640; }

template <bool FL, bool DB, bool XR>
static void launch_mid_inst(const SmallArgs& a, int count, size_t bytes, hipStream_t st) {
    static DeviceOnce once;
    lds_opt_in(once, (const void*)mid_gp_logpdf_kernel<FL, DB, XR>, 160 * 1024);
    hipLaunchKernelGGL((mid_gp_logpdf_kernel<FL, DB, XR>), dim3(count), dim3(SM_THREADS), bytes, st, a);
}

void launch_mid_gp(const SmallArgs& a, int count, int nF_max, hipStream_t st) {
    bool resident, two, xrow;
    mid_choose(a.n, nF_max, resident, two, xrow);
    const size_t bytes = mid_lds(a.n, nF_max, resident, two, xrow);
    if (xrow) launch_mid_inst<true, true, true>(a, count, bytes, st);
    else if (two) launch_mid_inst<true, true, false>(a, count, bytes, st);
    else if (resident) launch_mid_inst<true, false, false>(a, count, bytes, st);
    else launch_mid_inst<false, false, false>(a, count, bytes, st);
}

size_t small_gp_lds_bytes(int n, int nF) {
    const int NB = (n + SB - 1) / SB, NP = NB * SB;
    return ((size_t)(NB * (NB + 1) / 2) * 256 + 3 * (size_t)NP + SM_WAVES * SB + (size_t)nF * NP) * 8;
}

void launch_small_gp(const SmallArgs& a, int count, int nF_max, hipStream_t st) {
    const size_t bytes = small_gp_lds_bytes(a.n, nF_max);
    static DeviceOnce once;
    lds_opt_in(once, (const void*)small_gp_logpdf_kernel, 160 * 1024);
    hipLaunchKernelGGL(small_gp_logpdf_kernel, dim3(count), dim3(SM_THREADS), bytes, st, a);
}
