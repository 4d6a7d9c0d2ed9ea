// Gram-matrix build and the O(N^2)/O(N) pieces around the factorisation.
//
//   gram_kernel        A = yScale*exp(Lu+Lx) .* exp(Lt) + yNoise*I   (src/kernel.jl:13-32, 53-59;
//                      src/likelihood.jl:24-32; src/model_likelihood.jl:83-120), lower tiles only,
//                      fused with the per-tile partial column sums of B = yScale*exp(Lu+Lx) and
//                      K = B.*E that the SATE path needs.
//   rhs_prepare/tiles  column sums -> augmented right-hand sides [Y, c(1..L)] and sum(Delta).
//   epilogue           Schur complement of the augmented block -> MeanSATE, VarSATE, logdet, quad.
//   rbf_log / process_cov  the two src/kernel.jl entry points as stand-alone dense kernels.
#include "gpslc_internal.h"
#include "gp_math.h"
#include <type_traits>

#define MAXF 32   // max nU + nX handled by the fused Gram kernel

// block-wide sum with a fixed reduction tree (deterministic); result valid in every thread
__device__ __forceinline__ double block_sum_256(double v, double* red /* >= 4 doubles */) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------
// Gram build: one workgroup per lower tile (ti >= tj) per posterior sample.
// Thread (tx = tid>>4, ty = tid&15) owns rows ty + 16p and columns 8 tx + q, p, q = 0..7.
// ---------------------------------------------------------------------------------------
template <typename RT, int BIN, int FT>   // FT: exact feature count (0 = runtime): the distance loop unrolls fully.  BIN: binary treatments (e_ij is 1 or exp(-1/tyLS^2)); a template parameter so that the
                                  // 16 exp chains of a column are branch-free and interleave
__global__ __launch_bounds__(256) void gram_kernel(GramArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int F = FT > 0 ? FT : g.nU + g.nX;
    double* red = sm;                    // [4][128][2] cross-wave row-sum staging
    double* etab = red + 4 * GP_TS * 2;  // [32] 2^(j/32) for the table-driven exp (gp_math.h)
    RT* fr = reinterpret_cast<RT*>(etab + GP_EXP_TAB_DOUBLES);   // [F][128] row-block features / LS
    RT* fc = fr + F * GP_TS;             // [F][128] column-block features / LS
    RT* tr = fc + F * GP_TS;             // [128] T of row block
    RT* tc = tr + GP_TS;                 // [128]

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const long long s = g.s0 + b;
#ifdef GPSLC_DIAG
    const unsigned long long dt0 = g.dbg ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long dr0 = g.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
    int ti, tj;
    {
        const int t = blockIdx.x;
        int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)(r + 1) * (r + 2) / 2 <= t) ++r;
        while ((long long)r * (r + 1) / 2 > t) --r;
        ti = r; tj = t - r * (r + 1) / 2;
    }
    const int n = g.n;
    const int gi0 = ti * GP_TS, gj0 = tj * GP_TS;

    // stage features scaled by 1/LS (zeros on the padding): (x - x')^2 / LS^2 = (x/LS - x'/LS)^2
    if (FT > 0) {
        // exact feature count: ALL loads of the staging (lengthscale, row value, column value per (feature, instance)
        // pair of this thread) are issued before the first is used — branch-free, clamped addresses.  As a loop with
        // guarded loads every iteration waited for its own round trips: in-kernel stamps showed 17.7 k clocks of staging
        // per workgroup, a quarter of its life.
        constexpr int NIT = (FT * GP_TS + 255) / 256;
        double lv[NIT > 0 ? NIT : 1], av[NIT > 0 ? NIT : 1], cv[NIT > 0 ? NIT : 1];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int idx = tid + 256 * k;
            const int f = min(idx >> 7, FT - 1), r = idx & 127;
            const bool isu = f < g.nU;
            const double* src = isu ? g.p.U + s * g.p.u_sstride + (long long)f * n : g.X + (long long)(f - g.nU) * n;
            const double* lp = isu ? g.p.uyLS + s * g.nU + f : g.p.xyLS + s * g.nX + (f - g.nU);
            lv[k] = *lp;
            av[k] = src[min(gi0 + r, n - 1)];
            cv[k] = src[min(gj0 + r, n - 1)];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int idx = tid + 256 * k;
            if (idx < FT * GP_TS) {
                const int f = idx >> 7, r = idx & 127;
                const double il = 1.0 / lv[k];
                fr[f * GP_TS + r] = (RT)((gi0 + r < n) ? av[k] * il : 0.0);
                fc[f * GP_TS + r] = (RT)((gj0 + r < n) ? cv[k] * il : 0.0);
            }
        }
    } else {
        for (int idx = tid; idx < F * GP_TS; idx += 256) {
            const int f = idx >> 7, r = idx & 127;
            const double* src;
            double l;
            if (f < g.nU) { src = g.p.U + s * g.p.u_sstride + (long long)f * n; l = g.p.uyLS[s * g.nU + f]; }
            else { src = g.X + (long long)(f - g.nU) * n; l = g.p.xyLS[s * g.nX + (f - g.nU)]; }
            const double il = 1.0 / l;
            fr[f * GP_TS + r] = (RT)((gi0 + r < n) ? src[gi0 + r] * il : 0.0);
            fc[f * GP_TS + r] = (RT)((gj0 + r < n) ? src[gj0 + r] * il : 0.0);
        }
    }
    if (tid < GP_TS) {
        tr[tid] = (RT)((gi0 + tid < n) ? g.T[gi0 + tid] : 0.0);
        tc[tid] = (RT)((gj0 + tid < n) ? g.T[gj0 + tid] : 0.0);
    }
    gp_exp_tab_stage(etab, tid);
    __syncthreads();
#ifdef GPSLC_DIAG
    const unsigned long long dt1 = g.dbg ? __builtin_amdgcn_s_memtime() : 0;
#endif

    const double ys = g.p.yScale[s];
    const double yn = g.p.yNoise[s];
    const double tl = g.p.tyLS[s];
    const RT wt = (RT)(1.0 / (tl * tl));
    const RT ew = RbfMath<RT>::exp_neg_t(-wt, etab);     // e_ij for |T_i - T_j| = 1 (binary treatments)
    const int ty = tid & 15, tx = tid >> 4;

    RT tra[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) tra[p] = tr[ty + 16 * p];

    double* tile = tref_tile(g.M, b, ti, tj);
    const int Np = g.nt * GP_TS;
    double* partB = g.part + ((long long)b * 2 + 0) * g.nt * Np;
    double* partK = g.part + ((long long)b * 2 + 1) * g.nt * Np;
    double rsB[8], rsK[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) { rsB[p] = 0.0; rsK[p] = 0.0; }

    // runtime loop over this thread's 8 columns keeps the (inlined) exp code small enough for the
    // instruction cache; the row features are re-read from LDS per column (cheap next to 16 exps)
    // interior off-diagonal tiles (all but nt + a few of the nt(nt+1)/2) need no padding / diagonal selects:
    // v_cndmask issues at a quarter of the v_fma_f64 rate (tools/valu_rate_bench), 8.5 of them per element
    // cost as much as a third of the arithmetic
    auto columns = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll 1
        for (int q = 0; q < 8; ++q) {
            const int cq = 8 * tx + q;
            const int gj = gj0 + cq;
            RT lux[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) lux[p] = (RT)0;
#pragma unroll 2
            for (int f = 0; f < (FT > 0 ? FT : F); ++f) {
                const RT c = fc[f * GP_TS + cq];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const RT d = fr[f * GP_TS + ty + 16 * p] - c;
                    lux[p] = fma(d, d, lux[p]);
                }
            }
            const RT tcq = tc[cq];
            double csB = 0.0, csK = 0.0;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int rp = ty + 16 * p;
                const int gi = gi0 + rp;
                const RT dt = tra[p] - tcq;
                const RT Bq = (RT)ys * RbfMath<RT>::exp_neg_t(-lux[p], etab);
                const RT Eq = BIN ? (dt == (RT)0 ? (RT)1 : ew) : RbfMath<RT>::exp_neg_t(-((dt * dt) * wt), etab);
                double Bv, Kv, Av;
                // K = B .* E with the product taken in fp64 (exact for two fp32 factors): every consumer of K agrees on it to
                // the last bit whichever precision evaluated the kernel
                if (FAST) {
                    Bv = (double)Bq; Kv = (double)Bq * (double)Eq; Av = Kv;
                } else {
                    const bool inside = (gi < n) && (gj < n);
                    Bv = inside ? (double)Bq : 0.0;
                    Kv = inside ? (double)Bq * (double)Eq : 0.0;
                    // diagonal: + yNoise inside, identity on the padding
                    Av = (gi == gj) ? (inside ? Kv + yn : 1.0) : Kv;
                }
                __builtin_nontemporal_store(Av, &tile[cq * GP_TS + rp]);
                rsB[p] += Bv; rsK[p] += Kv;
                csB += Bv; csK += Kv;
            }
            if (g.with_sums) {
                // column sums: reduce over the 16 ty lanes (lane bits 0..3)
#pragma unroll
                for (int o = 1; o <= 8; o <<= 1) { csB += __shfl_xor(csB, o, 64); csK += __shfl_xor(csK, o, 64); }
                if (ty == 0) {
                    partB[(long long)ti * Np + gj] = csB;
                    partK[(long long)ti * Np + gj] = csK;
                }
            }
        }
    };
    if (ti != tj && gi0 + GP_TS <= n && gj0 + GP_TS <= n) columns(std::true_type{});
    else columns(std::false_type{});
#ifdef GPSLC_DIAG
    if (g.dbg) {     // [entry, staged, columns done, tile stores drained]
        const unsigned long long dt2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            unsigned long long* d = g.dbg + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            d[0] = dt0; d[1] = dt1; d[2] = dt2; d[3] = __builtin_amdgcn_s_memtime();
            d[4] = dr0; d[5] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
    // diagonal tile: the full square was computed, column sums are complete
    if (!g.with_sums || ti == tj) return;

    // row sums: reduce over tx = (lane>>4) + 4*wave : lane bits 4..5, then the 4 waves through LDS
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();   // all feature reads done before `red` (aliasing nothing, but keep phases clean)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        double vb = rsB[p], vk = rsK[p];
        vb += __shfl_xor(vb, 16, 64); vk += __shfl_xor(vk, 16, 64);
        vb += __shfl_xor(vb, 32, 64); vk += __shfl_xor(vk, 32, 64);
        if (lane < 16) {
            red[(wave * GP_TS + ty + 16 * p) * 2 + 0] = vb;
            red[(wave * GP_TS + ty + 16 * p) * 2 + 1] = vk;
        }
    }
    __syncthreads();
    if (tid < GP_TS) {
        const double vb = (red[(0 * GP_TS + tid) * 2] + red[(1 * GP_TS + tid) * 2]) +
                          (red[(2 * GP_TS + tid) * 2] + red[(3 * GP_TS + tid) * 2]);
        const double vk = (red[(0 * GP_TS + tid) * 2 + 1] + red[(1 * GP_TS + tid) * 2 + 1]) +
                          (red[(2 * GP_TS + tid) * 2 + 1] + red[(3 * GP_TS + tid) * 2 + 1]);
        partB[(long long)tj * Np + gi0 + tid] = vb;
        partK[(long long)tj * Np + gi0 + tid] = vk;
    }
}

#define GRAM_LDS_BYTES(F, RTS) (4 * GP_TS * 2 * 8 + GP_EXP_TAB_DOUBLES * 8 + (2 * (F) * GP_TS + 2 * GP_TS) * (RTS))

template <typename RT, int BIN, int FT>
static void launch_gram_t(const GramArgs& g, int nbatch, hipStream_t st) {
    const int F = g.nU + g.nX;
    static DeviceOnce attr_set;
    lds_opt_in(attr_set, (const void*)gram_kernel<RT, BIN, FT>, GRAM_LDS_BYTES(MAXF, (int)sizeof(RT)));
    // one workgroup per (tile, sample): a persistent variant measured 0.7 % slower (same-box A/B)
    const int nlow = g.nt * (g.nt + 1) / 2;
    hipLaunchKernelGGL((gram_kernel<RT, BIN, FT>), dim3(nlow, nbatch), dim3(256), GRAM_LDS_BYTES(F, (int)sizeof(RT)), st, g);
}
template <typename RT, int BIN>
static void launch_gram_b(const GramArgs& g, int nbatch, hipStream_t st) {
    switch (g.nU + g.nX) {   // exact instantiations for the common feature counts
        case 4: launch_gram_t<RT, BIN, 4>(g, nbatch, st); break;
        case 5: launch_gram_t<RT, BIN, 5>(g, nbatch, st); break;
        case 6: launch_gram_t<RT, BIN, 6>(g, nbatch, st); break;
        case 8: launch_gram_t<RT, BIN, 8>(g, nbatch, st); break;
        case 10: launch_gram_t<RT, BIN, 10>(g, nbatch, st); break;
        case 12: launch_gram_t<RT, BIN, 12>(g, nbatch, st); break;
        case 20: launch_gram_t<RT, BIN, 20>(g, nbatch, st); break;
        default: launch_gram_t<RT, BIN, 0>(g, nbatch, st); break;
    }
}
void launch_gram(const GramArgs& g, int nbatch, hipStream_t st) {
    if (g.f32) { if (g.binary_t) launch_gram_b<float, 1>(g, nbatch, st); else launch_gram_b<float, 0>(g, nbatch, st); }
    else { if (g.binary_t) launch_gram_b<double, 1>(g, nbatch, st); else launch_gram_b<double, 0>(g, nbatch, st); }
}

// ---------------------------------------------------------------------------------------
// rhs_prepare: per sample, reduce the per-tile partial sums in a fixed order, form the totals and
// sum(Delta_l) = sum K - 2 r.bsum + sum B for every level.  One workgroup per sample.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rhs_prepare_kernel(RhsArgs a) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const long long s = a.s0 + b;
    const int Np = a.nt * GP_TS;
    const double* partB = a.part + ((long long)b * 2 + 0) * a.nt * Np;
    const double* partK = a.part + ((long long)b * 2 + 1) * a.nt * Np;
    double* bs = a.bsum + (long long)b * Np;
    double* ks = a.ksum + (long long)b * Np;
    double tb = 0.0, tk = 0.0;
    for (int j = tid; j < Np; j += 256) {
        double vb = 0.0, vk = 0.0;
        for (int slot = 0; slot < a.nt; ++slot) {
            vb += partB[(long long)slot * Np + j];
            vk += partK[(long long)slot * Np + j];
        }
        bs[j] = vb; ks[j] = vk;
        tb += vb; tk += vk;
    }
    const double btot = block_sum_256(tb, red);
    const double ktot = block_sum_256(tk, red);
    const double tl = a.tyLS[s];
    const double wt = 1.0 / (tl * tl);
    for (int l = 0; l < a.L; ++l) {
        const double dot = a.doT[l];
        double acc = 0.0;
        for (int j = tid; j < Np; j += 256) {
            // same thread -> j assignment and the same tree as btot, so r == 1 reproduces btot bit for bit
            const double dt = (j < a.n ? a.T[j] : 0.0) - dot;
            const double r = gp_exp_neg(-((dt * dt) * wt));
            acc += r * bs[j];
        }
        const double rb = block_sum_256(acc, red);
        if (tid == 0) a.sumdelta[(long long)b * a.L + l] = (ktot - 2.0 * rb) + btot;
    }
}

// rhs_tiles: write the augmented row tiles: row q of the augmented block is right-hand side q
// (q = 0: Y, q = 1 + l: c_l = r_l .* bsum - ksum), zero elsewhere; zero the aug x aug tiles.
// grid (nt + naug, naug, batch): tile (nt + a, j) with j = blockIdx.x, a = blockIdx.y (j <= nt + a).
__global__ __launch_bounds__(256) void rhs_tiles_kernel(RhsArgs a) {
    const int tid = threadIdx.x;
    const int j = blockIdx.x, au = blockIdx.y, b = blockIdx.z;
    if (j > a.nt + au) return;
    const long long s = a.s0 + b;
    double* tile = tref_tile(a.M, b, a.nt + au, j);
    if (j >= a.nt) {
        if (a.live_rows > 0) return;      // the augmented diagonal tile has no reader (EpiArgs::from_rows, skip_aug_diag)
        for (int idx = tid; idx < GP_TSQ; idx += 256) tile[idx] = 0.0;
        return;
    }
    const int Np = a.nt * GP_TS;
    const double* bs = a.bsum + (long long)b * Np;
    const double* ks = a.ksum + (long long)b * Np;
    const double tl = a.tyLS[s];
    const double wt = 1.0 / (tl * tl);
    // element (row q, col c) at c*128 + q; thread -> consecutive q for coalescing
    if (a.live_rows > 0) {
        // 32 live rows at most: 32 x 128 elements, 256 contiguous bytes per column.  ALL 32 rows are written (zeros beyond
        // the last right-hand side): the strip and trailing kernels handle these tiles in 32-row units, so wave 0 loads,
        // multiplies and stores rows 16..31 even when only 16 are live — they must not hold what a reused arena left there
        // (NaN / Inf).  The other 96 rows keep whatever the workspace held: no kernel touches them, and a row of an MFMA
        // product depends on its own row only.
        for (int idx = tid; idx < 32 * GP_TS; idx += 256) {
            const int c = idx >> 5, q = idx & 31;
            const int gj = j * GP_TS + c;
            double v = 0.0;
            if (gj < a.n) {
                if (q == 0) v = a.Y[s * a.y_sstride + gj];
                else if (q <= a.L) {
                    const double dt = a.T[gj] - a.doT[q - 1];
                    const double r = gp_exp_neg(-((dt * dt) * wt));
                    v = r * bs[gj] - ks[gj];
                }
            }
            tile[c * GP_TS + q] = v;
        }
        return;
    }
    for (int idx = tid; idx < GP_TSQ; idx += 256) {
        const int c = idx >> 7, q = idx & 127;
        const int gq = au * GP_TS + q;        // right-hand side index
        const int gj = j * GP_TS + c;         // instance index
        double v = 0.0;
        if (gj < a.n) {
            if (gq == 0) v = a.Y[s * a.y_sstride + gj];
            else if (gq <= a.L) {
                const double dt = a.T[gj] - a.doT[gq - 1];
                const double r = gp_exp_neg(-((dt * dt) * wt));
                v = r * bs[gj] - ks[gj];
            }
        }
        tile[idx] = v;
    }
}

void launch_rhs(const RhsArgs& r, int nbatch, hipStream_t st) {
    if (r.with_sums) hipLaunchKernelGGL(rhs_prepare_kernel, dim3(nbatch), dim3(256), 0, st, r);
    hipLaunchKernelGGL(rhs_tiles_kernel, dim3(r.nt + r.naug, r.naug, nbatch), dim3(256), 0, st, r);
}

// ---------------------------------------------------------------------------------------
// epilogue: after the augmented factorisation tile (nt+a, nt+a') holds G = -R R^T with
// R = [z, w_1 .. w_L] (z = L^-1 Y, w_l = L^-1 c_l).  One workgroup per sample.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void epilogue_kernel(EpiArgs e) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const long long s = e.s0 + b;
    // log-determinant: 2 * sum log L_ii over the real rows
    double acc = 0.0;
    for (int i = tid; i < e.n; i += 256) {
        const int k = i >> 7, c = i & 127;
        const double* dt = tref_tile(e.M, b, k, k);
        acc += log(dt[c * GP_TS + c]);
    }
    const double ld = 2.0 * block_sum_256(acc, red);
    const double nn = (double)e.n;
    if (e.from_rows) {
        // The Schur block -R R^T used to come from one more tile update (an item streaming two full operand panels for
        // (L + 1)^2 live entries: 1.9 % of the GPU time at N = 1024) of which only z.z, z.w_l and w_l.w_l are read.  Here
        // they are summed from R directly: right-hand side q of column i sits at tile (nt, i / 128)[(i % 128) * 128 + q].
        // Wave w takes q = w, w + 4, ... (q = 0: z itself), lanes stride the columns; fixed butterfly: deterministic.
        // w_l == 0 (doT == T everywhere) gives exact zeros, as before.
        const int lane = tid & 63, wave = tid >> 6;
        const int ncol = e.nt * GP_TS;
        const int nq = (e.meanSATE || e.varSATE) ? e.L : 0;
        if (tid == 0 && e.logdet) e.logdet[s] = ld;
        auto put = [&](int q, double zw, double ww) {
            if (q == 0) { if (e.quad) e.quad[s] = ww; return; }
            const int l = q - 1;
            const double sd = e.sumdelta[(long long)b * e.L + l];
            if (e.meanSATE) e.meanSATE[s + e.S * l] = zw / nn;
            if (e.varSATE) e.varSATE[s + e.S * l] = ((sd - ww) + nn * e.pred_noise) / (nn * nn);
        };
        if (nq >= 16) {
            // Level sweeps: a lane reads 16 CONSECUTIVE right-hand sides of its column with each visit (one full 128-byte line)
            // instead of one 8-byte element per pass over the columns (a 1 KiB lane stride, once per q: up to 127 uncoalesced
            // passes per matrix).  Same lane -> column assignment, same fma order along the columns and the same butterfly per
            // q as the loop below: bit-identical sums.  Wave w takes the blocks q = 16 (w + 4 j) ...
            for (int qb = 16 * wave; qb <= nq; qb += 64) {
                double zw[16], ww[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { zw[j] = 0.0; ww[j] = 0.0; }
                for (int i = lane; i < ncol; i += 64) {
                    const double* col = tref_tile(e.M, b, e.nt, i >> 7) + (long long)(i & 127) * GP_TS;
                    const double z = col[0];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const double w = col[qb + j];
                        zw[j] = fma(z, w, zw[j]);
                        ww[j] = fma(w, w, ww[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    double a = zw[j], c = ww[j];
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
                    if (lane == 0 && qb + j <= nq) put(qb + j, a, c);
                }
            }
            return;
        }
        for (int q = wave; q <= nq; q += 4) {
            double zw = 0.0, ww = 0.0;
            for (int i = lane; i < ncol; i += 64) {
                const double* col = tref_tile(e.M, b, e.nt, i >> 7) + (long long)(i & 127) * GP_TS;
                const double z = col[0], w = col[q];
                zw = fma(z, w, zw);
                ww = fma(w, w, ww);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { zw += __shfl_xor(zw, o, 64); ww += __shfl_xor(ww, o, 64); }
            if (lane == 0) put(q, zw, ww);
        }
        return;
    }
    const double* g00 = tref_tile(e.M, b, e.nt, e.nt);
    if (tid == 0) {
        if (e.logdet) e.logdet[s] = ld;
        if (e.quad) e.quad[s] = -g00[0];
    }
    if (e.meanSATE == nullptr && e.varSATE == nullptr) return;
    for (int l = tid; l < e.L; l += 256) {
        const int q = 1 + l, au = q >> 7, qq = q & 127;
        const double wz = -tref_tile(e.M, b, e.nt + au, e.nt)[0 * GP_TS + qq];
        const double ww = -tref_tile(e.M, b, e.nt + au, e.nt + au)[qq * GP_TS + qq];
        const double sd = e.sumdelta[(long long)b * e.L + l];
        if (e.meanSATE) e.meanSATE[s + e.S * l] = wz / nn;
        if (e.varSATE) e.varSATE[s + e.S * l] = ((sd - ww) + nn * e.pred_noise) / (nn * nn);
    }
}

void launch_epilogue(const EpiArgs& e, int nbatch, hipStream_t st) {
    hipLaunchKernelGGL(epilogue_kernel, dim3(nbatch), dim3(256), 0, st, e);
}

// ---------------------------------------------------------------------------------------
// src/kernel.jl:24-32 rbfKernelLog and :53-59 processCov as plain dense kernels (column-major
// n x n, both triangles) for the Julia-visible entry points.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rbf_log_kernel(const double* X1, const double* X2, long long n,
                                                      int d, const double* ls, int ls_len, double* out) {
    // 16 x 16 thread tile, i fastest (column-major output: out[i + n*ip])
    const long long i = (long long)blockIdx.x * 16 + (threadIdx.x & 15);
    const long long ip = (long long)blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= n || ip >= n) return;
    double acc = 0.0;
    for (int k = 0; k < d; ++k) {
        const double l = ls[ls_len == 1 ? 0 : k];
        const double df = X1[i + n * k] - X2[ip + n * k];
        acc += (df * df) / (l * l);
    }
    out[i + n * ip] = -acc;
}

void launch_rbf_log(const double* X1, const double* X2, long long n, int d, const double* ls,
                    int ls_len, double* out, hipStream_t st) {
    dim3 grid((unsigned)((n + 15) / 16), (unsigned)((n + 15) / 16));
    hipLaunchKernelGGL(rbf_log_kernel, grid, dim3(256), 0, st, X1, X2, n, d, ls, ls_len, out);
}

__global__ __launch_bounds__(256) void process_cov_kernel(const double* in, long long n, double scale,
                                                          double noise, double* out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const long long i = idx % n, j = idx / n;
    double v = exp(in[idx]) * scale;
    if (i == j) v += noise;
    out[idx] = v;
}

void launch_process_cov(const double* in, long long n, double scale, double noise, double* out,
                        hipStream_t st) {
    const long long tot = n * n;
    hipLaunchKernelGGL(process_cov_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, in, n,
                       scale, noise, out);
}

// ---------------------------------------------------------------------------------------
// Generic MvNormal pieces: a dense column-major covariance into lower tiles (identity on the
// padding), S right-hand-side vectors as augmented rows, and the per-row quadratic forms.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_load_kernel(DenseLoadArgs a) {
    int ti, tj;
    {
        const int t = blockIdx.x;
        int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)(r + 1) * (r + 2) / 2 <= t) ++r;
        while ((long long)r * (r + 1) / 2 > t) --r;
        ti = r; tj = t - r * (r + 1) / 2;
    }
    double* tile = tref_tile(a.M, 0, ti, tj);
    for (int idx = threadIdx.x; idx < GP_TSQ; idx += 256) {
        const int c = idx >> 7, r = idx & 127;
        const long long gi = (long long)ti * GP_TS + r, gj = (long long)tj * GP_TS + c;
        double v;
        if (gi < a.n && gj < a.n) v = a.cov[gi + (long long)a.n * gj];
        else v = (gi == gj) ? 1.0 : 0.0;
        tile[idx] = v;
    }
}
void launch_dense_load(const DenseLoadArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(dense_load_kernel, dim3(a.nt * (a.nt + 1) / 2), dim3(256), 0, st, a);
}

// tile (nt + au, j): row q of the augmented block = x[:, 128 au + q] (zero beyond S and on the padding);
// aug x aug tiles zero.  grid (nt + naug, naug)
__global__ __launch_bounds__(256) void rows_rhs_kernel(RowsRhsArgs a) {
    const int j = blockIdx.x, au = blockIdx.y;
    if (a.rect ? (j >= a.nt) : (j > a.nt + au)) return;
    double* tile = tref_tile(a.M, 0, a.row0 + au, j);
    for (int idx = threadIdx.x; idx < GP_TSQ; idx += 256) {
        const int c = idx >> 7, q = idx & 127;
        const long long gq = (long long)au * GP_TS + q, gj = (long long)j * GP_TS + c;
        double v = 0.0;
        if (j < a.nt && gq < a.S && gj < a.n) v = a.x[gj + (long long)a.n * gq];
        tile[idx] = v;
    }
}
void launch_rows_rhs(const RowsRhsArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(rows_rhs_kernel, dim3(a.nt + a.naug, a.naug), dim3(256), 0, st, a);
}

__global__ __launch_bounds__(256) void quad_rows_kernel(QuadRowsArgs a) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int i = tid; i < a.n; i += 256) {
        const double* dt = tref_tile(a.M, 0, i >> 7, i >> 7);
        acc += log(dt[(i & 127) * GP_TS + (i & 127)]);
    }
    const double ld = 2.0 * block_sum_256(acc, red);
    if (tid == 0) a.logdet[0] = ld;
    for (long long q = tid; q < a.S; q += 256) {
        const int au = (int)(q >> 7), qq = (int)(q & 127);
        a.quad[q] = -tref_tile(a.M, 0, a.nt + au, a.nt + au)[qq * GP_TS + qq];
    }
}
// quad[q] = sum over the tile row's entries of W(au, j)[q][c]^2; lanes run along q (contiguous)
__global__ __launch_bounds__(128) void row_norms_kernel(RowNormArgs a) {
    const int au = blockIdx.x, q = threadIdx.x;
    double acc = 0.0;
    for (int j = 0; j < a.nt; ++j) {
        const double* t = tref_tile(a.W, 0, au, j);
        for (int c = 0; c < GP_TS; ++c) {
            const double v = t[c * GP_TS + q];
            acc = fma(v, v, acc);
        }
    }
    const long long gq = (long long)au * GP_TS + q;
    if (gq < a.S) a.quad[gq] = acc;
}
void launch_row_norms(const RowNormArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(row_norms_kernel, dim3(a.naug), dim3(128), 0, st, a);
}

void launch_quad_rows(const QuadRowsArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(quad_rows_kernel, dim3(1), dim3(256), 0, st, a);
}
