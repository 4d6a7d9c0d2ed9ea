// O(N^2) kernels around the factor: back-substitution for alpha = A^-1 Y, the MeanITE pass,
// construction of D / Delta for the full ITE covariance, CovITE gather and the predictive draws.
#include "gpslc_internal.h"
#include "gp_math.h"
#include "back_block.h"

#define MAXF 32

// ---------------------------------------------------------------------------------------
// Back-substitution L^T alpha = z, right-looking over tile rows i = nt-1 .. 0, two launches per row:
// alpha_i = inv(L_ii)^T z_i (one workgroup per matrix), then z_k -= L(i,k)^T alpha_i for every k < i
// (one workgroup per tile; reads L exactly once: HBM-bound).
// ---------------------------------------------------------------------------------------
// alpha_i = inv(L_ii)^T z_i, one workgroup per matrix
__global__ __launch_bounds__(256) void backsolve_alpha_kernel(BackArgs a, int i, double* alpha) {
    __shared__ double zi[GP_TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long b = blockIdx.x;
    const int Np = a.nt * GP_TS;
    if (tid < GP_TS) zi[tid] = a.zwork[b * Np + i * GP_TS + tid];
    __syncthreads();
    const double* invt = a.inv + b * a.inv_bstride + (long long)i * GP_TSQ;
    double o32[32];
    tile_tv32(invt, zi, wave, lane, o32);
    if (lane == 0) {
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) alpha[b * Np + i * GP_TS + wave * 32 + cc] = o32[cc];
    }
}
// z_k -= L(i,k)^T alpha_i for k < i, one workgroup per (k, matrix)
__global__ __launch_bounds__(256) void backsolve_update_kernel(BackArgs a, int i, const double* alpha) {
    __shared__ double ai[GP_TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x;
    const long long b = blockIdx.y;
    const int Np = a.nt * GP_TS;
    if (tid < GP_TS) ai[tid] = alpha[b * Np + i * GP_TS + tid];
    __syncthreads();
    double o32[32];
    tile_tv32(tref_tile(a.M, b, i, k), ai, wave, lane, o32);
    double* z = a.zwork + b * Np;
    if (lane == 0) {
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) z[k * GP_TS + wave * 32 + cc] -= o32[cc];
    }
}

// copy z = row 0 of the augmented tile row (nt, j) into zwork
__global__ __launch_bounds__(128) void extract_z_kernel(BackArgs a) {
    const int j = blockIdx.x;
    const long long b = blockIdx.y;
    const double* t = tref_tile(a.M, b, a.nt, j);
    a.zwork[b * a.nt * GP_TS + j * GP_TS + threadIdx.x] = t[threadIdx.x * GP_TS + 0];
}

void launch_backsolve(const BackArgs& a, int nbatch, hipStream_t st) {
    // alpha is stored right behind zwork (caller allocates 2 * nbatch * Np doubles)
    double* alpha = a.zwork + (long long)nbatch * a.nt * GP_TS;
    hipLaunchKernelGGL(extract_z_kernel, dim3(a.nt, nbatch), dim3(128), 0, st, a);
    for (int i = a.nt - 1; i >= 0; --i) {
        hipLaunchKernelGGL(backsolve_alpha_kernel, dim3(nbatch), dim3(256), 0, st, a, i, alpha);
        if (i > 0) hipLaunchKernelGGL(backsolve_update_kernel, dim3(i, nbatch), dim3(256), 0, st, a, i, alpha);
    }
}

// ---------------------------------------------------------------------------------------
// MeanITE (src/estimation.jl:46):  MeanITE = (Ks' - K) alpha,  alpha = A^-1 Y,  Ks'_ij = B_ij r_j(l),  K = B .* E.
// Round 3: K alpha needs no second pass over the pairs — A alpha = Y, so  K alpha = Y - yNoise alpha  comes from the solve
// itself, and
//     MeanITE_i(l) = sum_j B_ij (r_j(l) alpha_j)  -  (Y_i - yNoise alpha_i)
// costs ONE exp per pair (B_ij) instead of two (the e_ij of round 1/2: 56 -> 36 fp64-rate instructions per pair).
// Row i of D = Ks' - K is identically zero in the reference's own arithmetic whenever T_i == doT ((T_j - doT)^2 and
// (T_i - T_j)^2 are then the same square, so Ks'_ij == K_ij bit for bit): the reference returns an exact 0.0 for such an
// instance (test/estimation.jl:6-66 is the n = 1 case) and so does this kernel.
// One workgroup per (row block, sample); levels are processed LCT at a time.
// ---------------------------------------------------------------------------------------
template <int FREG, int LCT, typename RT, int RB>
__global__ __launch_bounds__(256) void ite_mean_kernel(IteMeanArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int F = a.nU + a.nX;
    double* etab = sm;                     // [32] 2^(j/32): table-driven exp (gp_math.h)
    double* rl = etab + GP_EXP_TAB_DOUBLES;   // [LCT][128]  r_j(l) * alpha_j of the staged column block
    double* red = rl + LCT * GP_TS;        // [RB][128][LCT]
    RT* fc = reinterpret_cast<RT*>(red + RB * GP_TS * LCT);   // [FREG][128] column features / LS (zero rows beyond F)
    const int tid = threadIdx.x, r = tid & 127, h = tid >> 7;
    const int ib = blockIdx.x;
    const long long b = blockIdx.y, s = a.s0 + b;
    const int n = a.n, Np = a.nt * GP_TS;

    auto feat_src = [&](int f) -> const double* {
        return (f < a.nU) ? a.p.U + s * a.p.u_sstride + (long long)f * n : a.X + (long long)(f - a.nU) * n;
    };
    auto feat_il = [&](int f) -> double {
        return 1.0 / ((f < a.nU) ? a.p.uyLS[s * a.nU + f] : a.p.xyLS[s * a.nX + (f - a.nU)]);
    };
    int gi[RB];
    RT af[RB][FREG];   // this thread's rows' features / LS
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        gi[q] = (ib * RB + q) * GP_TS + r;
#pragma unroll
        for (int f = 0; f < FREG; ++f) af[q][f] = (RT)((f < F && gi[q] < n) ? feat_src(f)[gi[q]] * feat_il(f) : 0.0);
    }
    const double ys = a.p.yScale[s];
    const double tl = a.p.tyLS[s];
    const double wt = 1.0 / (tl * tl);
    const RT wtq = (RT)wt;
    gp_exp_tab_stage(etab, tid);
    const double* alpha = a.alpha + b * Np;

    for (int l0 = 0; l0 < a.L; l0 += LCT) {
        const int nl = min(LCT, a.L - l0);
        double acc[RB][LCT];
#pragma unroll
        for (int q = 0; q < RB; ++q)
#pragma unroll
            for (int ll = 0; ll < LCT; ++ll) acc[q][ll] = 0.0;
        for (int jt = 0; jt < a.nt; ++jt) {
            __syncthreads();
            for (int idx = tid; idx < FREG * GP_TS; idx += 256) {
                const int f = idx >> 7, cc = idx & 127;
                const int g = jt * GP_TS + cc;
                fc[idx] = (RT)((f < F && g < n) ? feat_src(f)[g] * feat_il(f) : 0.0);
            }
            for (int idx = tid; idx < LCT * GP_TS; idx += 256) {
                const int ll = idx >> 7, cc = idx & 127;
                const int g = jt * GP_TS + cc;
                double v = 0.0;
                if (ll < nl && g < n) {
                    const RT dt = (RT)a.T[g] - (RT)a.doT[l0 + ll];
                    v = (double)RbfMath<RT>::exp_neg_t(-((dt * dt) * wtq), etab) * alpha[g];
                }
                rl[idx] = v;
            }
            __syncthreads();
#pragma unroll 2
            for (int cq = 0; cq < 64; ++cq) {
                const int c = h * 64 + cq;
                RT cf[FREG];
#pragma unroll
                for (int f = 0; f < FREG; ++f) cf[f] = fc[f * GP_TS + c];
                double rlc[LCT];
#pragma unroll
                for (int ll = 0; ll < LCT; ++ll) rlc[ll] = rl[ll * GP_TS + c];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    RT lux = (RT)0;
#pragma unroll
                    for (int f = 0; f < FREG; ++f) {
                        const RT d = af[q][f] - cf[f];
                        lux = fma(d, d, lux);
                    }
                    const double Bv = (double)((RT)ys * RbfMath<RT>::exp_neg_t(-lux, etab));
#pragma unroll
                    for (int ll = 0; ll < LCT; ++ll) acc[q][ll] = fma(Bv, rlc[ll], acc[q][ll]);
                }
            }
        }
        __syncthreads();
        if (h == 1) {
#pragma unroll
            for (int q = 0; q < RB; ++q)
#pragma unroll
                for (int ll = 0; ll < LCT; ++ll) red[(q * GP_TS + r) * LCT + ll] = acc[q][ll];
        }
        __syncthreads();
        if (h == 0) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                if (gi[q] >= n) continue;
                // (K alpha)_i = Y_i - yNoise alpha_i: alpha solves (K + yNoise I) alpha = Y
                const double ka = a.Y[s * a.y_sstride + gi[q]] - a.yNoise[s] * alpha[gi[q]];
                const double ti = a.T[gi[q]];
#pragma unroll
                for (int ll = 0; ll < LCT; ++ll)
                    if (ll < nl) {
                        const double v = (acc[q][ll] + red[(q * GP_TS + r) * LCT + ll]) - ka;
                        a.meanITE[(long long)gi[q] * a.si + s * a.ss + (long long)(l0 + ll) * a.sl] =
                            (ti == a.doT[l0 + ll]) ? 0.0 : v;
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// MeanITE for many intervention levels on the MFMA (L > 4, fp64):
//   MeanITE[i, l] = sum_j B_ij (r_j(l) alpha_j)  -  (Y_i - yNoise alpha_i)
// = (B R)[i, l] - (K alpha)[i],  R[j, l] = r_j(l) alpha_j  — a (128 x N)(N x 64) product per row block; K alpha from
// the solve itself (A alpha = Y; see ite_mean_kernel), so the pair loop evaluates one exp (B_ij) and no e_ij.
// No operand tile goes through LDS: the f64 16x16x4 MFMA wants one element per lane, and each lane computes exactly its
// own B_rc (row = 16m + lane&15, column = 4kk + lane>>4) from the staged features; R is staged per 64-column chunk.
// Instances with T_i == doT_l get the reference's exact 0.0 (row i of Ks' - K is identically zero there).
// One workgroup = 128 rows x up to 64 levels; wave w owns rows 32w..32w+31 (2 row sub-tiles x 4 level
// sub-tiles = 8 accumulators).
// ---------------------------------------------------------------------------------------
typedef double d4s __attribute__((ext_vector_type(4)));
typedef double d2s __attribute__((ext_vector_type(2)));
#define IM_CC 64          // columns per staged chunk
#define IM_RLD 80         // padded row of the R chunk (doubles): conflict-free ds_read_b64 across k rows
#define IM_NL 64          // levels per pass

template <int FREG>   // FREG > 0: this lane's two rows' features live in registers (F <= FREG); 0: read from LDS
__global__ __launch_bounds__(256, 2) void ite_mean_mfma_kernel(IteMeanArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int F = a.nU + a.nX;
    double* etab = sm;                       // [32] 2^(j/32): table-driven exp (gp_math.h)
    double* fr = etab + GP_EXP_TAB_DOUBLES;  // [F][128] row features / LS
    const int FSL = FREG > F ? FREG : F;
    double* fc = fr + F * GP_TS;             // [max(F, FREG)][IM_CC] column features / LS
    double* R = fc + FSL * IM_CC;            // [IM_CC][IM_RLD]  r_j(l) * alpha_j
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int ib = blockIdx.x;
    const long long b = blockIdx.y, s = a.s0 + b;
    const int n = a.n, Np = a.nt * GP_TS;

    auto feat_src = [&](int f) -> const double* {
        return (f < a.nU) ? a.p.U + s * a.p.u_sstride + (long long)f * n : a.X + (long long)(f - a.nU) * n;
    };
    auto feat_il = [&](int f) -> double {
        return 1.0 / ((f < a.nU) ? a.p.uyLS[s * a.nU + f] : a.p.xyLS[s * a.nX + (f - a.nU)]);
    };
    for (int idx = tid; idx < F * GP_TS; idx += 256) {
        const int f = idx >> 7, rr = idx & 127;
        const int g = ib * GP_TS + rr;
        fr[idx] = (g < n) ? feat_src(f)[g] * feat_il(f) : 0.0;
    }
    const double ys = a.p.yScale[s];
    const double tl = a.p.tyLS[s];
    const double wt = 1.0 / (tl * tl);
    const double* alpha = a.alpha + b * Np;
    gp_exp_tab_stage(etab, tid);
    const int r0 = 32 * wave + li;           // this lane's rows: r0 and r0 + 16
    __syncthreads();
    double af0[FREG > 0 ? FREG : 1], af1[FREG > 0 ? FREG : 1];
    if (FREG > 0) {
#pragma unroll
        for (int f = 0; f < FREG; ++f) {
            af0[f] = (f < F) ? fr[f * GP_TS + r0] : 0.0;
            af1[f] = (f < F) ? fr[f * GP_TS + r0 + 16] : 0.0;
        }
    }

    for (int l0 = 0; l0 < a.L; l0 += IM_NL) {
        const int nl = min(IM_NL, a.L - l0);
        const int nq = (nl + 15) >> 4;          // live 16-level sub-tiles of this pass (wave-uniform)
        d4s acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[m][q] = (d4s){0.0, 0.0, 0.0, 0.0};
        for (int c0 = 0; c0 < Np; c0 += IM_CC) {
            __syncthreads();
            const int FS = FREG > F ? FREG : F;      // staged feature rows (zero beyond F)
            for (int idx = tid; idx < FS * IM_CC; idx += 256) {
                const int f = idx / IM_CC, cc = idx - f * IM_CC;
                const int g = c0 + cc;
                fc[idx] = (f < F && g < n) ? feat_src(f)[g] * feat_il(f) : 0.0;
            }
            for (int idx = tid; idx < IM_CC * IM_NL; idx += 256) {
                const int cc = idx >> 6, ll = idx & 63;      // consecutive threads -> consecutive levels
                const int g = c0 + cc;
                double v = 0.0;
                if (ll < nl && g < n) {
                    const double dt = a.T[g] - a.doT[l0 + ll];
                    v = gp_exp_neg_tab(-((dt * dt) * wt), etab) * alpha[g];
                }
                R[cc * IM_RLD + ll] = v;
            }
            __syncthreads();
#pragma unroll 2
            for (int kk = 0; kk < IM_CC / 4; ++kk) {
                const int cc = 4 * kk + lq;            // this lane's column inside the chunk
                double lux0 = 0.0, lux1 = 0.0;
                if (FREG > 0) {
#pragma unroll
                    for (int f = 0; f < FREG; ++f) {       // fc rows beyond F are zero-filled
                        const double cf = fc[f * IM_CC + cc];
                        const double d0 = af0[f] - cf, d1 = af1[f] - cf;
                        lux0 = fma(d0, d0, lux0);
                        lux1 = fma(d1, d1, lux1);
                    }
                } else {
                    for (int f = 0; f < F; ++f) {
                        const double cf = fc[f * IM_CC + cc];
                        const double d0 = fr[f * GP_TS + r0] - cf;
                        const double d1 = fr[f * GP_TS + r0 + 16] - cf;
                        lux0 = fma(d0, d0, lux0);
                        lux1 = fma(d1, d1, lux1);
                    }
                }
                const double B0 = ys * gp_exp_neg_tab(-lux0, etab), B1 = ys * gp_exp_neg_tab(-lux1, etab);
                const double* Rrow = R + cc * IM_RLD + li;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < nq) {
                        const double rf = Rrow[16 * q];
                        acc[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(rf, B0, acc[0][q], 0, 0, 0);
                        acc[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(rf, B1, acc[1][q], 0, 0, 0);
                    }
                }
            }
        }
        // acc[m][q][v] = (B R)[row 32w + 16m + li][level 16q + lq + 4v]
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int gi = ib * GP_TS + 32 * wave + 16 * m + li;
            if (gi < n) {
                const double ka = a.Y[s * a.y_sstride + gi] - a.yNoise[s] * alpha[gi];     // (K alpha)_i from A alpha = Y
                const double ti = a.T[gi];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int ll = 16 * q + lq + 4 * v;
                        if (ll < nl)
                            a.meanITE[(long long)gi * a.si + s * a.ss + (long long)(l0 + ll) * a.sl] =
                                (ti == a.doT[l0 + ll]) ? 0.0 : acc[m][q][v] - ka;
                    }
            }
        }
    }
}

template <int FREG>
static void launch_ite_mean_mfma_t(const IteMeanArgs& a, int nbatch, hipStream_t st) {
    const int F = a.nU + a.nX;
    const int FS = FREG > F ? FREG : F;
    const int bytes = (GP_EXP_TAB_DOUBLES + F * GP_TS + FS * IM_CC + IM_CC * IM_RLD) * 8;
    static DeviceOnce attr_set;
    lds_opt_in(attr_set, (const void*)ite_mean_mfma_kernel<FREG>, (GP_EXP_TAB_DOUBLES + MAXF * GP_TS + MAXF * IM_CC + IM_CC * IM_RLD) * 8);
    hipLaunchKernelGGL((ite_mean_mfma_kernel<FREG>), dim3(a.nt, nbatch), dim3(256), bytes, st, a);
}
static void launch_ite_mean_mfma(const IteMeanArgs& a, int nbatch, hipStream_t st) {
    const int F = a.nU + a.nX;
    if (F <= 4) launch_ite_mean_mfma_t<4>(a, nbatch, st);
    else if (F <= 6) launch_ite_mean_mfma_t<6>(a, nbatch, st);
    else if (F <= 8) launch_ite_mean_mfma_t<8>(a, nbatch, st);
    else if (F <= 10) launch_ite_mean_mfma_t<10>(a, nbatch, st);
    else if (F <= 12) launch_ite_mean_mfma_t<12>(a, nbatch, st);
    else launch_ite_mean_mfma_t<0>(a, nbatch, st);
}

template <int FREG, int LCT, typename RT>
static void launch_ite_mean_t(const IteMeanArgs& a, int nbatch, hipStream_t st) {
    constexpr int RB = 1;     // row blocks per workgroup (2 measured slower: occupancy)
    const int bytes = (GP_EXP_TAB_DOUBLES + (1 + RB) * LCT * GP_TS) * 8 + (FREG * GP_TS) * (int)sizeof(RT);
    static DeviceOnce attr_set;
    lds_opt_in(attr_set, (const void*)ite_mean_kernel<FREG, LCT, RT, RB>, bytes);
    hipLaunchKernelGGL((ite_mean_kernel<FREG, LCT, RT, RB>), dim3((a.nt + RB - 1) / RB, nbatch), dim3(256), bytes, st, a);
}
template <int FREG, typename RT>
static void launch_ite_mean_f(const IteMeanArgs& a, int nbatch, hipStream_t st) {
    if (a.L <= 1) launch_ite_mean_t<FREG, 1, RT>(a, nbatch, st);
    else if (a.L <= 4) launch_ite_mean_t<FREG, 4, RT>(a, nbatch, st);
    else launch_ite_mean_t<FREG, 16, RT>(a, nbatch, st);
}
template <typename RT>
static void launch_ite_mean_r(const IteMeanArgs& a, int nbatch, hipStream_t st) {
    const int F = a.nU + a.nX;
    // exact register counts for the common feature widths: the pass is fp64-VALU bound (2 instructions per
    // feature per element), a padded feature is paid in full
    if (F <= 4) launch_ite_mean_f<4, RT>(a, nbatch, st);
    else if (F <= 5) launch_ite_mean_f<5, RT>(a, nbatch, st);      // BASELINE config 2: nU + nX = 1 + 4
    else if (F <= 6) launch_ite_mean_f<6, RT>(a, nbatch, st);
    else if (F <= 8) launch_ite_mean_f<8, RT>(a, nbatch, st);
    else if (F <= 10) launch_ite_mean_f<10, RT>(a, nbatch, st);
    else if (F <= 12) launch_ite_mean_f<12, RT>(a, nbatch, st);
    else if (F <= 16) launch_ite_mean_f<16, RT>(a, nbatch, st);
    else if (F <= 20) launch_ite_mean_f<20, RT>(a, nbatch, st);
    else launch_ite_mean_f<32, RT>(a, nbatch, st);
}
void launch_ite_mean(const IteMeanArgs& a, int nbatch, hipStream_t st) {
    // many levels: the (B R) product belongs on the matrix cores (fp64 path; the fp32 mode keeps the VALU kernel)
    if (!a.f32 && a.L > 4) { launch_ite_mean_mfma(a, nbatch, st); return; }
    if (a.f32) launch_ite_mean_r<float>(a, nbatch, st);
    else launch_ite_mean_r<double>(a, nbatch, st);
}

// ---------------------------------------------------------------------------------------
// dt_build: tiles of D (D_ij = B_ij (r_j - e_ij), src/estimation.jl:46 "CovWWs' - CovWW") into the
// rectangular matrix W and Delta + pred_noise*I (Delta_ij = B_ij (e_ij - r_i - r_j + 1) =
// CovWW - CovWWs - CovWWs' + CovWsWs, src/likelihood.jl:46-49 / estimation.jl:47, :82) into Cm.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dt_build_kernel(DtArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int F = a.nU + a.nX;
    double* fr = sm;
    double* fc = fr + F * GP_TS;
    double* tr = fc + F * GP_TS;
    double* tc = tr + GP_TS;
    double* rr_ = tc + GP_TS;   // r_i of the row block
    double* rc_ = rr_ + GP_TS;  // r_j of the column block
    const int tid = threadIdx.x;
    const int ti = blockIdx.x / a.nt, tj = blockIdx.x % a.nt;
    const long long b = blockIdx.y, s = a.s0 + b / a.lc;       // batch element = (sample, level) pair
    const double doT = a.doT[a.l0 + (int)(b % a.lc)];
    const int n = a.n;
    const int gi0 = ti * GP_TS, gj0 = tj * GP_TS;
    const double tl = a.p.tyLS[s];
    const double wt = 1.0 / (tl * tl);
    for (int idx = tid; idx < F * GP_TS; idx += 256) {
        const int f = idx >> 7, r = idx & 127;
        const double* src = (f < a.nU) ? a.p.U + s * a.p.u_sstride + (long long)f * n
                                       : a.X + (long long)(f - a.nU) * n;
        const double il = 1.0 / ((f < a.nU) ? a.p.uyLS[s * a.nU + f] : a.p.xyLS[s * a.nX + (f - a.nU)]);
        fr[idx] = (gi0 + r < n) ? src[gi0 + r] * il : 0.0;
        fc[idx] = (gj0 + r < n) ? src[gj0 + r] * il : 0.0;
    }
    if (tid < GP_TS) {
        const double t1 = (gi0 + tid < n) ? a.T[gi0 + tid] : 0.0;
        const double t2 = (gj0 + tid < n) ? a.T[gj0 + tid] : 0.0;
        tr[tid] = t1; tc[tid] = t2;
        const double d1 = t1 - doT, d2 = t2 - doT;
        rr_[tid] = gp_exp_neg(-((d1 * d1) * wt));
        rc_[tid] = gp_exp_neg(-((d2 * d2) * wt));
    }
    __syncthreads();
    const double ys = a.p.yScale[s];
    const int ty = tid & 15, tx = tid >> 4;
    double* wt_tile = tref_tile(a.W, b, ti, tj);
    double* c_tile = (ti >= tj) ? tref_tile(a.Cm, b, ti, tj) : nullptr;
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        const int cq = 8 * tx + q;
        const int gj = gj0 + cq;
        double lux[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) lux[p] = 0.0;
        for (int f = 0; f < F; ++f) {
            const double c = fc[f * GP_TS + cq];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const double d = fr[f * GP_TS + ty + 16 * p] - c;
                lux[p] = fma(d, d, lux[p]);
            }
        }
        const double tcq = tc[cq], rj = rc_[cq];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rp = ty + 16 * p;
            const int gi = gi0 + rp;
            const double dt = tr[rp] - tcq;
            const double Bv = ys * gp_exp_neg(-lux[p]);
            const double Ev = gp_exp_neg(-((dt * dt) * wt));
            const double ri = rr_[rp];
            double Dv = Bv * (rj - Ev);
            double Cv = Bv * (((Ev - ri) - rj) + 1.0);
            const bool inside = (gi < n) && (gj < n);
            if (!inside) { Dv = 0.0; Cv = (gi == gj) ? 1.0 : 0.0; }
            else if (gi == gj) Cv += a.pred_noise;
            wt_tile[cq * GP_TS + rp] = Dv;
            if (c_tile) c_tile[cq * GP_TS + rp] = Cv;
        }
    }
}
#define DT_LDS_BYTES(F) ((2 * (F) * GP_TS + 4 * GP_TS) * 8)

void launch_dt_build(const DtArgs& a, int nbatch, hipStream_t st) {
    static DeviceOnce attr_set;
    lds_opt_in(attr_set, (const void*)dt_build_kernel, DT_LDS_BYTES(MAXF));
    hipLaunchKernelGGL(dt_build_kernel, dim3(a.nt * a.nt, nbatch), dim3(256), DT_LDS_BYTES(a.nU + a.nX), st, a);
}

// CovITEs[s + S*(i + n*j)] (src/estimation.jl:75, :82 layout: sample index fastest), both triangles
__global__ __launch_bounds__(256) void gather_cov_kernel(GatherCovArgs a) {
    int ti, tj;
    {
        const int t = blockIdx.x;
        int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)(r + 1) * (r + 2) / 2 <= t) ++r;
        while ((long long)r * (r + 1) / 2 > t) --r;
        ti = r; tj = t - r * (r + 1) / 2;
    }
    const long long b = blockIdx.y, s = a.s0 + b;
    const double* t = tref_tile(a.Cm, b, ti, tj);
    for (int idx = threadIdx.x; idx < GP_TSQ; idx += 256) {
        const int c = idx >> 7, r = idx & 127;
        const long long gi = (long long)ti * GP_TS + r, gj = (long long)tj * GP_TS + c;
        if (gi >= a.n || gj >= a.n || gi < gj) continue;
        const double v = t[idx];
        a.out[s + a.S * (gi + a.n * gj)] = v;
        a.out[s + a.S * (gj + a.n * gi)] = v;
    }
}
void launch_gather_cov(const GatherCovArgs& a, int nbatch, hipStream_t st) {
    hipLaunchKernelGGL(gather_cov_kernel, dim3(a.nt * (a.nt + 1) / 2, nbatch), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------
// Predictive draws: ite[l, i, s*spp + d] = MeanITE_i + (L_c z)_i  (src/estimation.jl:95-109 with the
// factor computed once per (sample, level) instead of once per draw).
// Philox4x32-10 + Box-Muller, restated in oracle/gpslc_oracle.py:philox_normals.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                              unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int rd = 0; rd < 10; ++rd) {
        const unsigned long long p0 = 0xD2511F53ull * c0;
        const unsigned long long p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double philox_normal(unsigned long long seed, unsigned long long stream,
                                                unsigned long long e) {
    unsigned w[4];
    const unsigned long long pair = e >> 1;
    philox4x32_10((unsigned)pair, (unsigned)(pair >> 32), (unsigned)stream, (unsigned)(stream >> 32),
                  (unsigned)seed, (unsigned)(seed >> 32), w);
    const unsigned long long A = ((unsigned long long)w[0] << 21) ^ ((unsigned long long)w[1] >> 11);
    const unsigned long long Bq = ((unsigned long long)w[2] << 21) ^ ((unsigned long long)w[3] >> 11);
    const double u1 = ((double)A + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)Bq + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    return (e & 1ull) ? rad * sin(ang) : rad * cos(ang);
}

// One standard normal per (instance, draw) of every unit of the sub-batch, generated ONCE per unit into a
// workspace laid out like the caller-supplied form (z[g + n*d] per unit): thread = one Philox counter = the
// pair of elements (2p, 2p + 1) -> (cos, sin) branch of one Box-Muller transform, exactly the values
// philox_normal() returns for those two elements.
__global__ __launch_bounds__(256) void normals_kernel(unsigned long long seed, long long s0, long long S, int l, int lc,
                                                      long long n, int spp, double* out) {
    const long long b = blockIdx.y;             // (sample s0 + b / lc, level l + b % lc)
    const unsigned long long stream = (unsigned long long)(s0 + b / lc + S * (long long)(l + b % lc));
    const long long total = n * spp;                       // elements of this unit
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (2 * p >= total) return;
    double* o = out + b * total;
    o[2 * p] = philox_normal(seed, stream, (unsigned long long)(2 * p));
    if (2 * p + 1 < total) o[2 * p + 1] = philox_normal(seed, stream, (unsigned long long)(2 * p + 1));
}

// The draws of one unit as a triangular matrix product on the f64 MFMA:  out[:, d] = mu + L_c z[:, d] for ALL
// draws d of the unit in one pass over L_c (16 NQ draws per pass; spp <= 128 -> the factor is read exactly once).
// One workgroup = one tile row of L_c (128 instances); wave w owns rows 32w..32w+31 as two row sets
// {32w + 2j} and {32w + 2j + 1}, j = lane & 15: a lane fetches its two rows of a column with ONE 16-byte load
// straight from HBM (every element of L_c is used by exactly one wave, so it never goes through LDS) and stores
// its two results with one 16-byte store (16 lanes -> 256 contiguous bytes).  MFMA operands: A = z (draw index
// = lane & 15), B = L_c (row = lane & 15), k = lane >> 4, so D[draw = 4v + (lane >> 4)][row = lane & 15].
// z (64 columns x 16 NQ draws) is staged in LDS per half tile.  HBM-bound while 16 NQ <= 32 (one 128 KiB tile
// per 0.5 NQ MFMA-microseconds), MFMA-bound beyond.
#define DR_KC 64
template <int NQ>
__global__ __launch_bounds__(256) void draws_mfma_kernel(DrawArgs a) {
    extern __shared__ __attribute__((aligned(16))) double zs[];      // [DR_KC][ZLD]
    constexpr int ND = 16 * NQ;
    constexpr int ZLD = ND + 1;          // odd row stride: conflict-free column-wise staging writes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int ib = a.nt - 1 - (int)blockIdx.x;          // longest tile rows first
    const long long b = blockIdx.y, sb = b / a.lc, lb = b % a.lc;      // batch element = (sample, level) pair
    const long long s = a.s0 + sb, lev = a.l + lb;
    const long long n = a.n;
    const double* __restrict__ zu = a.z ? a.z + n * a.spp * (s + a.S * lev) : a.zgen + n * a.spp * b;
    const int r0 = 32 * wave + 2 * li;                  // this lane's two rows inside the tile: r0, r0 + 1
    const long long gi = (long long)ib * GP_TS + r0;

    for (int d0 = 0; d0 < a.spp; d0 += ND) {
        const int nd = min(ND, a.spp - d0);
        d4s acc[2][NQ];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[m][q] = (d4s){0.0, 0.0, 0.0, 0.0};
        // (rounds 2-4 ran 1..128 draws through this kernel, with a software-pipelined form for <= 16: since round 5 it serves units of
        // more than 128 draws only — several passes over L_c; profiles/r05_draws_lds_kernel.patch has the removed branches)
        for (int jt = 0; jt <= ib; ++jt) {
            const double* __restrict__ t = tref_tile(a.Lc, b, ib, jt) + r0;
#pragma unroll 1
            for (int kc = 0; kc < GP_TS / DR_KC; ++kc) {
                // this lane's rows of the 16 column groups of the chunk: 16 independent 16-byte loads in flight
                d2s lv[DR_KC / 4];
#pragma unroll
                for (int kk = 0; kk < DR_KC / 4; ++kk)
                    lv[kk] = *reinterpret_cast<const d2s*>(t + (kc * DR_KC + 4 * kk + lq) * GP_TS);
                __syncthreads();
                for (int idx = tid; idx < DR_KC * ND; idx += 256) {
                    const int k = idx & (DR_KC - 1), dd = idx / DR_KC;
                    const long long g = (long long)jt * GP_TS + kc * DR_KC + k;
                    zs[k * ZLD + dd] = (dd < nd && g < n) ? zu[g + n * (d0 + dd)] : 0.0;
                }
                __syncthreads();
                if (jt == ib) {     // diagonal tile: only the lower triangle belongs to L_c
#pragma unroll
                    for (int kk = 0; kk < DR_KC / 4; ++kk) {
                        const int c = kc * DR_KC + 4 * kk + lq;
                        if (c > r0) lv[kk].x = 0.0;
                        if (c > r0 + 1) lv[kk].y = 0.0;
                    }
                }
#pragma unroll
                for (int kk = 0; kk < DR_KC / 4; ++kk) {
                    const double* zr = zs + (4 * kk + lq) * ZLD + li;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const double zf = zr[16 * q];
                        acc[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zf, lv[kk].x, acc[0][q], 0, 0, 0);
                        acc[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zf, lv[kk].y, acc[1][q], 0, 0, 0);
                    }
                }
            }
        }
        // acc[m][q][v]: row r0 + m, draw d0 + 16 q + 4 v + lq
        if (gi < n) {
            const double mu0 = a.mean[gi + n * (s + a.S * lev)];
            const double mu1 = (gi + 1 < n) ? a.mean[gi + 1 + n * (s + a.S * lev)] : 0.0;
            double* __restrict__ ob = a.out + a.obase + sb * a.osb + lb * a.osl + gi * a.osi;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int dd = 16 * q + 4 * v + lq;
                    if (dd < nd) {
                        double* o = ob + (long long)(d0 + dd) * a.osd;
                        const double x0 = mu0 + acc[0][q][v], x1 = mu1 + acc[1][q][v];
                        if (a.osi == 1 && gi + 1 < n && ((reinterpret_cast<unsigned long long>(o) & 15ull) == 0)) {
                            *reinterpret_cast<d2s*>(o) = (d2s){x0, x1};
                        } else {
                            o[0] = x0;
                            if (gi + 1 < n) o[a.osi] = x1;
                        }
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Round 5: the draws of a unit with spp <= 128 (the reference's default is 10; NQ = 1, 2, 4 or 8 blocks of 16 draws per pass;
// more than 128 draws keep the LDS-staged multi-pass kernel above) as a PURE STREAM of L_c.
// What the LDS kernel above left on the table (4.3 of the ~6 TB/s a read-only sweep reaches on this chip): its 4 waves
// meet at two barriers per 64 columns (one wave's late line stalls four), its work items span 1..nt tiles (the last long
// row runs alone at the end of the launch), and the diagonal tile is read whole.  Here
//   * the unit's normals are laid out ONCE, by draws_zstage_kernel, as the MFMA A-operand image zt: the 16-byte word of
//     lane (lq, li) for the column groups (2m, 2m + 1) of a 32-column chunk sits at lane-contiguous addresses, so a wave
//     fetches its z operands with four fully coalesced 1 KiB loads per chunk — no LDS, no barrier, no dependence
//     between the waves of a workgroup (draws 10..15 and columns >= n are zeros in the image);
//   * a workgroup owns the tile-row PAIR (nt-1-p, p): every item streams nt + 1 tiles, whatever p;
//   * wave w reads only the columns of the diagonal tile at or left of its own 32 rows (8 (w + 1) of the 32 groups);
//   * three register sets of 4 factor loads + 2 z loads each (16-column chunks) in rotation, the next two chunks in flight
//     under the current chunk's MFMAs, 104 VGPRs -> four workgroups per CU; factor loads carry the non-temporal hint (each
//     line is used exactly once).  Measured (same box, 64 units x 10 draws at N = 4096): 4.30 -> 6.04 TB/s of factor stream;
//     32-column chunks with two sets 5.6, and the number of workgroups per CU (2 / 3 / 4) does not matter;
// MFMA operands, accumulators and the order of the k groups along a row are those of draws_mfma_kernel<1>: the chains
// are the same, so the draws are bit-identical (skipped groups of the diagonal tile only ever added +-0).
// ---------------------------------------------------------------------------------------
// blocks of 16 draws the stream kernel works on for spp draws per unit: its template parameter NQ
__host__ __device__ inline int draws_nq(int spp) { return spp <= 16 ? 1 : spp <= 32 ? 2 : spp <= 64 ? 4 : 8; }
// index of z[column g][draw d] in a unit's operand image (16 * Np doubles)
__host__ __device__ inline long long draws_zt_index(long long g, int d) {
    return ((g >> 3) << 7) + ((g & 3) << 5) + ((long long)d << 1) + ((g >> 2) & 1);
}

// one thread = one 16-byte word of the image: columns g0 = 8 blk + lq and g0 + 4, draw d
__global__ __launch_bounds__(256) void draws_zstage_kernel(DrawArgs a) {
    const long long b = blockIdx.y, sb = b / a.lc, lb = b % a.lc;
    const long long s = a.s0 + sb, lev = a.l + lb;
    const long long n = a.n, Np = (long long)a.nt * GP_TS;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;      // word t of the image; block q of 16 draws = words [q 8 Np, ...)
    const int nq = draws_nq(a.spp);              // the stream kernel's NQ: every one of its blocks is written (zeros beyond spp)
    if (t >= Np * 8 * nq) return;
    const long long tq = t % (Np * 8);
    const int d = (int)(tq & 15) + 16 * (int)(t / (Np * 8)), lq = (int)(tq >> 4) & 3;
    const long long g0 = ((tq >> 6) << 3) + lq, g1 = g0 + 4;
    double v0 = 0.0, v1 = 0.0;
    if (d < a.spp) {
        if (a.z) {
            const double* __restrict__ zu = a.z + n * a.spp * (s + a.S * lev);
            if (g0 < n) v0 = zu[g0 + n * d];
            if (g1 < n) v1 = zu[g1 + n * d];
        } else if ((n & 1) == 0) {
            // n even: the elements e = g + n d and e ^ 1 of a unit's stream share one Philox counter and one Box-Muller
            // transform (even -> cos, odd -> sin), and the lane that holds column g ^ 1 of the same draw is lane ^ 16 (lq ^ 1).
            // The even-lq lane evaluates the pair of g0, the odd-lq lane the pair of g1 = g0 + 4, and they swap the halves:
            // one counter, one log, one sqrt, one sincos per lane instead of two of each — the same values philox_normal()
            // returns element by element.  (Whole waves reach this point together: Np * 8 * nq is a multiple of 64.)
            const unsigned long long stream = (unsigned long long)((a.rs0 + sb) + a.rS * lev);
            const bool odd = (lq & 1) != 0;
            const long long ge = odd ? (g1 & ~1ll) : g0;            // the even column of the pair this lane evaluates
            double c = 0.0, sn = 0.0;
            if (d < a.spp && ge < n) {
                unsigned w[4];
                const unsigned long long pair = (unsigned long long)(ge + n * d) >> 1;
                philox4x32_10((unsigned)pair, (unsigned)(pair >> 32), (unsigned)stream, (unsigned)(stream >> 32),
                              (unsigned)a.seed, (unsigned)(a.seed >> 32), w);
                const unsigned long long A = ((unsigned long long)w[0] << 21) ^ ((unsigned long long)w[1] >> 11);
                const unsigned long long Bq = ((unsigned long long)w[2] << 21) ^ ((unsigned long long)w[3] >> 11);
                const double u1 = ((double)A + 0.5) * (1.0 / 9007199254740992.0);
                const double u2 = ((double)Bq + 0.5) * (1.0 / 9007199254740992.0);
                const double rad = sqrt(-2.0 * log(u1));
                const double ang = 6.283185307179586476925286766559 * u2;
                c = rad * cos(ang);
                sn = rad * sin(ang);
            }
            // even-lq lane: keeps cos as its v0 (column g0), sends sin to the partner's v0 (column g0 + 1);
            // odd-lq lane: keeps sin as its v1 (column g1), sends cos to the partner's v1 (column g1 - 1)
            const double give = odd ? c : sn;
            const double got = __shfl_xor(give, 16, 64);
            if (odd) { v0 = got; v1 = sn; } else { v0 = c; v1 = got; }
            if (g0 >= n) v0 = 0.0;
            if (g1 >= n) v1 = 0.0;
        } else {
            const unsigned long long stream = (unsigned long long)((a.rs0 + sb) + a.rS * lev);
            if (g0 < n) v0 = philox_normal(a.seed, stream, (unsigned long long)(g0 + n * d));
            if (g1 < n) v1 = philox_normal(a.seed, stream, (unsigned long long)(g1 + n * d));
        }
    }
    *reinterpret_cast<d2s*>(a.zt + b * Np * 16 * nq + 2 * t) = (d2s){v0, v1};
}

// CC columns per chunk (CC / 4 k groups of the 16x16x4 MFMA), NS register sets in rotation, WPE waves per SIMD the
// register allocation is held to
template <int CC, int NS, int WPE, int NQ = 1>
__global__ __launch_bounds__(256, WPE) void draws_stream_kernel(DrawArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const long long b = blockIdx.y, sb = b / a.lc, lb = b % a.lc;      // batch element = (sample, level) pair
    const long long s = a.s0 + sb, lev = a.l + lb;
    const long long n = a.n;
    const long long Np = (long long)a.nt * GP_TS;
    const double* __restrict__ zt = a.zt + b * Np * 16 * NQ + 2 * lane;      // block q of 16 draws: + q * 16 Np
    const int r0 = 32 * wave + 2 * li;                  // this lane's two rows inside the tile: r0, r0 + 1
    const int p = blockIdx.x;
    constexpr int NL = CC / 4, NZ = CC / 8;             // 16-byte factor / z loads per chunk and lane

    for (int half = 0; half < 2; ++half) {
        const int ib = half == 0 ? a.nt - 1 - p : p;
        if (half == 1 && 2 * p == a.nt - 1) break;      // odd nt: the middle row has no partner
        // the tiles (ib, 0..ib) of a row are contiguous in both tile layouts
        const double* __restrict__ Lrow = tref_tile(a.Lc, b, ib, 0) + r0 + lq * GP_TS;
        const int nch = (GP_TS / CC) * ib + (32 / CC) * wave + 32 / CC;   // chunks up to and including the wave's diagonal block
        d4s acc0[NQ], acc1[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { acc0[q] = (d4s){0.0, 0.0, 0.0, 0.0}; acc1[q] = acc0[q]; }
        d2s lv[NS][NL], zv[NS][NQ * NZ];
        auto load = [&](int c, d2s (&l)[NL], d2s (&z)[NQ * NZ]) {
            c = min(c, nch - 1);                         // past the end: the last chunk again (keeps the loop body branch-free)
            const double* __restrict__ zp = zt + (long long)c * (CC * 16);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int m = 0; m < NZ; ++m) z[q * NZ + m] = *reinterpret_cast<const d2s*>(zp + q * (16 * Np) + m * 128);
            const double* __restrict__ lp = Lrow + (long long)c * (CC * GP_TS);
#pragma unroll
            for (int kk = 0; kk < NL; ++kk)
                l[kk] = __builtin_nontemporal_load(reinterpret_cast<const d2s*>(lp + kk * 4 * GP_TS));
        };
        auto compute = [&](const d2s (&l)[NL], const d2s (&z)[NQ * NZ]) {
#pragma unroll
            for (int kk = 0; kk < NL; ++kk)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const double zf = (kk & 1) ? z[q * NZ + (kk >> 1)].y : z[q * NZ + (kk >> 1)].x;
                    acc0[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zf, l[kk].x, acc0[q], 0, 0, 0);
                    acc1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zf, l[kk].y, acc1[q], 0, 0, 0);
                }
        };
        // the chunks of the wave's own 32 x 32 diagonal block: only the lower triangle belongs to L_c
        auto mask_diag = [&](d2s (&l)[NL], int c) {
            const int cb = (c * CC) & 31;                // first column of the chunk relative to the wave's first row
#pragma unroll
            for (int kk = 0; kk < NL; ++kk) {
                const int cc = cb + 4 * kk + lq;
                if (cc > 2 * li) l[kk].x = 0.0;
                if (cc > 2 * li + 1) l[kk].y = 0.0;
            }
        };
        constexpr int ND = 32 / CC;                      // chunks of the diagonal block (1 or 2)
#pragma unroll
        for (int i = 0; i < NS; ++i) load(i, lv[i], zv[i]);
        int c = 0;
        for (; c + NS + ND - 1 < nch; c += NS) {         // chunks c .. c + NS - 1 are all left of the diagonal block
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                compute(lv[i], zv[i]);
                load(c + NS + i, lv[i], zv[i]);
            }
        }
        const int rem = nch - c;                         // ND .. NS + ND - 1 chunks left; the first min(rem, NS) are loaded
#pragma unroll
        for (int i = 0; i < NS + ND - 1; ++i) {
            if (i < rem) {
                if (i >= NS) load(c + i, lv[i % NS], zv[i % NS]);
                if (i >= rem - ND) mask_diag(lv[i % NS], c + i);
                compute(lv[i % NS], zv[i % NS]);
            }
        }
        // acc{0,1}[q][v]: row r0 + {0,1}, draw 16 q + 4 v + lq
        const long long gi = (long long)ib * GP_TS + r0;
        if (gi < n) {
            const double mu0 = a.mean[gi + n * (s + a.S * lev)];
            const double mu1 = (gi + 1 < n) ? a.mean[gi + 1 + n * (s + a.S * lev)] : 0.0;
            double* __restrict__ ob = a.out + a.obase + sb * a.osb + lb * a.osl + gi * a.osi;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int dd = 16 * q + 4 * v + lq;
                if (dd < a.spp) {
                    double* o = ob + (long long)dd * a.osd;
                    const double x0 = mu0 + acc0[q][v], x1 = mu1 + acc1[q][v];
                    if (a.osi == 1 && gi + 1 < n && ((reinterpret_cast<unsigned long long>(o) & 15ull) == 0)) {
                        *reinterpret_cast<d2s*>(o) = (d2s){x0, x1};
                    } else {
                        o[0] = x0;
                        if (gi + 1 < n) o[a.osi] = x1;
                    }
                }
            }
        }
    }
}

// Level sweep (L > 1): the draws of the sub-batch are produced level by level into tmp[b][l][d][i] (instance
// fastest: coalesced stores) and rearranged ONCE into the reference's level-fastest tensor
// ite[l + L*(i + n*(s*spp + d))] (src/prediction.jl:30-33) through LDS, so that both the reads (1 KiB runs along i)
// and the writes (runs along l) are contiguous.
#define SC_LC 32
__global__ __launch_bounds__(256) void draws_scatter_kernel(const double* __restrict__ tmp, double* __restrict__ out,
                                                            long long n, int L, int spp, long long s0) {
    __shared__ double tl[SC_LC][GP_TS + 1];
    const int tid = threadIdx.x;
    const long long i0 = (long long)blockIdx.x * GP_TS;
    const int d = blockIdx.y;
    const long long b = blockIdx.z;
    const double* src = tmp + ((b * L) * spp + d) * n;                                // + l*spp*n + i
    double* dst = out + (long long)L * n * ((s0 + b) * spp + d);                       // + l + L*i
    for (int l0 = 0; l0 < L; l0 += SC_LC) {
        const int nl = min(SC_LC, L - l0);
        __syncthreads();
        for (int idx = tid; idx < SC_LC * GP_TS; idx += 256) {
            const int ll = idx >> 7, ii = idx & 127;
            if (ll < nl && i0 + ii < n) tl[ll][ii] = src[(long long)(l0 + ll) * spp * n + i0 + ii];
        }
        __syncthreads();
        for (int idx = tid; idx < SC_LC * GP_TS; idx += 256) {
            const int ll = idx & (SC_LC - 1), ii = idx / SC_LC;
            if (ll < nl && i0 + ii < n) dst[(l0 + ll) + (long long)L * (i0 + ii)] = tl[ll][ii];
        }
    }
}

template <int NQ>
static void launch_draws_t(const DrawArgs& a, int nbatch, hipStream_t st) {
    const int bytes = DR_KC * (16 * NQ + 1) * 8;
    static DeviceOnce once;
    lds_opt_in(once, (const void*)draws_mfma_kernel<NQ>, bytes);
    hipLaunchKernelGGL((draws_mfma_kernel<NQ>), dim3(a.nt, nbatch), dim3(256), bytes, st, a);
}
void launch_draws(const DrawArgs& a, int nbatch, hipStream_t st) {
    if (a.spp <= 128 && a.zt) {    // up to 128 draws = one pass over L_c: the barrier-free stream of L_c (round 5)
        const long long words = (long long)a.nt * GP_TS * 8 * draws_nq(a.spp);
        hipLaunchKernelGGL(draws_zstage_kernel, dim3((unsigned)((words + 255) / 256), nbatch), dim3(256), 0, st, a);
        const dim3 grid((a.nt + 1) / 2, nbatch);
        // more than 16 draws: NQ z blocks and accumulator pairs per wave.  HBM-bound up to 32 draws, MFMA-bound beyond (the
        // factor is still read exactly once): fewer, fatter waves
        if (a.spp > 64) { hipLaunchKernelGGL((draws_stream_kernel<16, 2, 1, 8>), grid, dim3(256), 0, st, a); return; }
        if (a.spp > 32) { hipLaunchKernelGGL((draws_stream_kernel<16, 3, 2, 4>), grid, dim3(256), 0, st, a); return; }
        if (a.spp > 16) { hipLaunchKernelGGL((draws_stream_kernel<16, 3, 3, 2>), grid, dim3(256), 0, st, a); return; }
        // register-set arrangements tried (profiles/r05_ab_experiments.md §1): <32,2,2> 5.62-5.65 TB/s, <32,2,3> 5.59, <32,3,2> 5.55,
        // <16,4,3> 5.60, <16,4,4> 6.02, <16,3,4> 6.04 (this one) on one box; 5.5-5.6 for every one of them on another
        hipLaunchKernelGGL((draws_stream_kernel<16, 3, 4>), grid, dim3(256), 0, st, a);
        return;
    }
    if (!a.z) {     // the library's own stream: every normal of the unit is generated exactly once
        const long long pairs = ((long long)a.n * a.spp + 1) / 2;
        hipLaunchKernelGGL(normals_kernel, dim3((unsigned)((pairs + 255) / 256), nbatch), dim3(256), 0, st, a.seed, a.rs0,
                           a.rS, a.l, a.lc, (long long)a.n, a.spp, a.zgen);
    }
    launch_draws_t<8>(a, nbatch, st);      // more than 128 draws per unit: passes of 128
}
void launch_draws_scatter(const double* tmp, double* out, long long n, int L, int spp, long long s0, int nbatch,
                          hipStream_t st) {
    hipLaunchKernelGGL(draws_scatter_kernel, dim3((unsigned)((n + GP_TS - 1) / GP_TS), spp, nbatch), dim3(256), 0, st,
                       tmp, out, n, L, spp, s0);
}

// ---------------------------------------------------------------------------------------
// likelihoodDistribution blocks (src/likelihood.jl:24-39): K = B.*E, Ks = diag(r) B, Ks' = B diag(r), Kss = B
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ld_build_kernel(LdBuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int F = a.nU + a.nX;
    double* fr = sm;
    double* fc = fr + F * GP_TS;
    double* tr = fc + F * GP_TS;
    double* tc = tr + GP_TS;
    double* rr_ = tc + GP_TS;
    double* rc_ = rr_ + GP_TS;
    const int tid = threadIdx.x;
    const int ti = blockIdx.x / a.nt, tj = blockIdx.x % a.nt;
    const long long s = a.s0;
    const int n = a.n;
    const int gi0 = ti * GP_TS, gj0 = tj * GP_TS;
    const double tl = a.p.tyLS[s];
    const double wt = 1.0 / (tl * tl);
    for (int idx = tid; idx < F * GP_TS; idx += 256) {
        const int f = idx >> 7, r = idx & 127;
        const double* src = (f < a.nU) ? a.p.U + s * a.p.u_sstride + (long long)f * n
                                       : a.X + (long long)(f - a.nU) * n;
        const double il = 1.0 / ((f < a.nU) ? a.p.uyLS[s * a.nU + f] : a.p.xyLS[s * a.nX + (f - a.nU)]);
        fr[idx] = (gi0 + r < n) ? src[gi0 + r] * il : 0.0;
        fc[idx] = (gj0 + r < n) ? src[gj0 + r] * il : 0.0;
    }
    if (tid < GP_TS) {
        const double t1 = (gi0 + tid < n) ? a.T[gi0 + tid] : 0.0;
        const double t2 = (gj0 + tid < n) ? a.T[gj0 + tid] : 0.0;
        tr[tid] = t1; tc[tid] = t2;
        const double d1 = t1 - a.doT, d2 = t2 - a.doT;
        rr_[tid] = gp_exp_neg(-((d1 * d1) * wt));
        rc_[tid] = gp_exp_neg(-((d2 * d2) * wt));
    }
    __syncthreads();
    const double ys = a.p.yScale[s];
    double* tK = tref_tile(a.K, 0, ti, tj);
    double* tKs = tref_tile(a.Ks, 0, ti, tj);
    double* tKsT = tref_tile(a.KsT, 0, ti, tj);
    double* tKss = tref_tile(a.Kss, 0, ti, tj);
    for (int idx = tid; idx < GP_TSQ; idx += 256) {
        const int c = idx >> 7, r = idx & 127;
        double lux = 0.0;
        for (int f = 0; f < F; ++f) {
            const double d = fr[f * GP_TS + r] - fc[f * GP_TS + c];
            lux = fma(d, d, lux);
        }
        const double dt = tr[r] - tc[c];
        double Bv = ys * gp_exp_neg(-lux);
        double Ev = gp_exp_neg(-((dt * dt) * wt));
        if (gi0 + r >= n || gj0 + c >= n) { Bv = 0.0; Ev = 0.0; }
        tK[idx] = Bv * Ev;
        tKs[idx] = rr_[r] * Bv;
        tKsT[idx] = Bv * rc_[c];
        tKss[idx] = Bv;
    }
}
void launch_ld_build(const LdBuildArgs& a, hipStream_t st) {
    const int F = a.nU + a.nX;
    const int bytes = (2 * F * GP_TS + 4 * GP_TS) * 8;
    static DeviceOnce attr_set;
    lds_opt_in(attr_set, (const void*)ld_build_kernel, (2 * MAXF * GP_TS + 4 * GP_TS) * 8);
    hipLaunchKernelGGL(ld_build_kernel, dim3(a.nt * a.nt), dim3(256), bytes, st, a);
}

__global__ __launch_bounds__(256) void rect_gather_kernel(RectGatherArgs a) {
    const int ti = blockIdx.x / a.nt, tj = blockIdx.x % a.nt;
    const double* t = tref_tile(a.R, 0, ti, tj);
    for (int idx = threadIdx.x; idx < GP_TSQ; idx += 256) {
        const int c = idx >> 7, r = idx & 127;
        const long long gi = (long long)ti * GP_TS + r, gj = (long long)tj * GP_TS + c;
        if (gi < a.n && gj < a.n) a.out[gi + (long long)a.n * gj] = t[idx] + (gi == gj ? a.diag_add : 0.0);
    }
}
void launch_rect_gather(const RectGatherArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(rect_gather_kernel, dim3(a.nt * a.nt), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------
// summarizeEstimates: one workgroup per individual.  The row is gathered into LDS (padded with +inf to a
// power of two), sorted with a bitonic network, and the two order statistics are interpolated exactly as
// Julia's Statistics.quantile does (type 7: aleph = m p + (1 - p), j = trunc(aleph), a + gamma (b - a)).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double julia_quantile_sorted(const double* v, int m, double p) {
#pragma clang fp contract(off)   // Julia evaluates these expressions without fused multiply-adds
    if (m == 1) return v[0];
    const double aleph = (double)m * p + (1.0 - p);
    int j = (int)aleph;                       // trunc
    j = min(max(j, 1), m - 1);
    double gam = aleph - (double)j;
    gam = fmin(fmax(gam, 0.0), 1.0);
    const double a = v[j - 1], b = v[j];
    return a + gam * (b - a);
}

__global__ __launch_bounds__(256) void summarize_kernel(SummArgs a) {
    extern __shared__ __attribute__((aligned(16))) double v[];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const long long i = blockIdx.x;
    double acc = 0.0;
    for (int j = tid; j < a.mpad; j += 256) {
        double x = INFINITY;
        if (j < a.m) { x = a.x[i * a.rs + (long long)j * a.cs]; acc += x; }
        v[j] = x;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    const double total = (red[0] + red[1]) + (red[2] + red[3]);
    for (int k = 2; k <= a.mpad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < a.mpad; t += 256) {
                const int u = t ^ j;
                if (u > t) {
                    const bool up = (t & k) == 0;
                    const double p = v[t], q = v[u];
                    if ((p > q) == up) { v[t] = q; v[u] = p; }
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        a.mean[i] = total / (double)a.m;
        a.lower[i] = julia_quantile_sorted(v, a.m, a.lowerQ);
        a.upper[i] = julia_quantile_sorted(v, a.m, a.upperQ);
    }
}
// Rows longer than one LDS image (m > 16384: S * spp of a BASELINE-size posterior): exact order statistics by radix
// select on order-preserving 64-bit keys instead of a sort.  One workgroup = 16 consecutive individuals (for the
// column-major n x m sample matrix a wave then reads whole 128-byte lines: 16 rows x 4 samples), thread (r, jj) walks
// samples jj, jj + 16, ... of row r.  Eight passes of 8-bit digits from the top; both quantiles' selections run in the
// same pass (two histograms per row in LDS, ds_add_u32); a ninth pass finds the successor of each selected element
// (count of keys <= it, minimum key above it).  Julia's type-7 interpolation as in julia_quantile_sorted.
#define SUMM_RW 16
__device__ __forceinline__ unsigned long long summ_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double summ_unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__global__ __launch_bounds__(256) void summarize_select_kernel(SummArgs a) {
    __shared__ unsigned int hist[SUMM_RW][2][256];
    __shared__ unsigned long long prefix[SUMM_RW][2], mingt[SUMM_RW][2];
    __shared__ unsigned int rank[SUMM_RW][2], cntle[SUMM_RW][2];
    __shared__ double rsum[4][SUMM_RW];
    const int tid = threadIdx.x, r = tid & (SUMM_RW - 1), jj = tid >> 4;
    const long long i = (long long)blockIdx.x * SUMM_RW + r;
    const bool live = i < a.n;
    const double* row = a.x + (live ? i : 0) * a.rs;
    const int m = a.m;
    // ranks (0-based) of the lower neighbour of each quantile: j - 1 with j = clamp(trunc(m p + 1 - p), 1, m - 1)
    int jq[2];
    double gq[2];
    {
#pragma clang fp contract(off)
        const double ps[2] = {a.lowerQ, a.upperQ};
        for (int q = 0; q < 2; ++q) {
            const double aleph = (double)m * ps[q] + (1.0 - ps[q]);
            int j = (int)aleph;
            j = min(max(j, 1), m - 1);
            jq[q] = j;
            gq[q] = fmin(fmax(aleph - (double)j, 0.0), 1.0);
        }
    }
    for (int t = tid; t < SUMM_RW * 2 * 256; t += 256) (&hist[0][0][0])[t] = 0u;
    if (tid < SUMM_RW * 2) {
        const int rr = tid & (SUMM_RW - 1), q = tid >> 4;
        prefix[rr][q] = 0ull; rank[rr][q] = (unsigned)(jq[q] - 1); cntle[rr][q] = 0u; mingt[rr][q] = ~0ull;
    }
    // mean: per-thread strided partial, then the 4 samples-lanes of a wave, then the 4 waves
    double acc = 0.0;
    if (live)
        for (int j = jj; j < m; j += 16) acc += row[(long long)j * a.cs];
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if ((tid & 63) < SUMM_RW) rsum[tid >> 6][r] = acc;
    __syncthreads();
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (live) {
            const unsigned long long p0 = prefix[r][0], p1 = prefix[r][1];
            for (int j = jj; j < m; j += 16) {
                const unsigned long long k = summ_key(row[(long long)j * a.cs]);
                const unsigned long long hi = pass == 0 ? 0ull : (k >> (shift + 8));
                const unsigned d = (unsigned)(k >> shift) & 255u;
                if (hi == p0) atomicAdd(&hist[r][0][d], 1u);
                if (hi == p1) atomicAdd(&hist[r][1][d], 1u);
            }
        }
        __syncthreads();
        if (tid < SUMM_RW * 2) {
            const int rr = tid & (SUMM_RW - 1), q = tid >> 4;
            unsigned cum = 0, want = rank[rr][q], d = 0;
            for (; d < 255u; ++d) {
                const unsigned cnt = hist[rr][q][d];
                if (want < cum + cnt) break;
                cum += cnt;
            }
            prefix[rr][q] = (prefix[rr][q] << 8) | d;
            rank[rr][q] = want - cum;
        }
        __syncthreads();
        for (int t = tid; t < SUMM_RW * 2 * 256; t += 256) (&hist[0][0][0])[t] = 0u;
        __syncthreads();
    }
    // successor of each selected element
    if (live) {
        const unsigned long long k0 = prefix[r][0], k1 = prefix[r][1];
        unsigned c0 = 0, c1 = 0;
        unsigned long long g0 = ~0ull, g1 = ~0ull;
        for (int j = jj; j < m; j += 16) {
            const unsigned long long k = summ_key(row[(long long)j * a.cs]);
            if (k <= k0) ++c0; else g0 = k < g0 ? k : g0;
            if (k <= k1) ++c1; else g1 = k < g1 ? k : g1;
        }
        atomicAdd(&cntle[r][0], c0); atomicAdd(&cntle[r][1], c1);
        atomicMin(&mingt[r][0], g0); atomicMin(&mingt[r][1], g1);
    }
    __syncthreads();
    if (tid < SUMM_RW && i < a.n) {
#pragma clang fp contract(off)
        double qv[2];
        for (int q = 0; q < 2; ++q) {
            const double av = summ_unkey(prefix[r][q]);                  // sorted[j - 1]
            const double bv = cntle[r][q] > (unsigned)jq[q] ? av : summ_unkey(mingt[r][q]);   // sorted[j]
            qv[q] = m == 1 ? av : av + gq[q] * (bv - av);
        }
        a.mean[i] = ((rsum[0][r] + rsum[1][r]) + (rsum[2][r] + rsum[3][r])) / (double)m;
        a.lower[i] = qv[0];
        a.upper[i] = qv[1];
    }
}

void launch_summarize(const SummArgs& a, hipStream_t st) {
    if (a.m > 16384) {
        hipLaunchKernelGGL(summarize_select_kernel, dim3((a.n + SUMM_RW - 1) / SUMM_RW), dim3(256), 0, st, a);
        return;
    }
    static DeviceOnce attr_set;
    lds_opt_in(attr_set, (const void*)summarize_kernel, 16384 * 8);
    hipLaunchKernelGGL(summarize_kernel, dim3(a.n), dim3(256), (size_t)a.mpad * 8, st, a);
}
