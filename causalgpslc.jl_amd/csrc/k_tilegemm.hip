// f64-MFMA tile update:  C(i,j) (-)= sum_kk A(i,kk) * B(j,kk)^T   on 128 x 128 fp64 tiles.
//
// This one kernel carries every O(N^3) term of the path: the Cholesky panel multiply by the
// inverted diagonal block, the left-looking column update inside a panel, the trailing SYRK
// update, the "D L^-T" solve of the ITE covariance and its SYRK.
//
// CDNA4 mapping
//   * PERSISTENT workgroups: 2 per CU are launched once and each walks its share of the output
//     tiles.  Measured with in-kernel stamps (profiles/r01_tilegemm_stamps.md): with one workgroup per
//     tile the dispatcher leaves a CU slot empty for 25-30 us between two 130 us workgroups — 20 % of
//     the MFMA time; a resident workgroup starts its next tile immediately.
//   * one workgroup = 4 wave64 = one 128 x 128 output tile at a time; each wave owns a 64 x 64
//     quadrant as 4 x 4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs), 2 workgroups per CU;
//   * K is streamed in slabs of 16 tile columns (16 KiB contiguous per operand, tiles are
//     column-major), global_load_dwordx4 -> registers -> ds_write_b128, loads issued two slabs ahead
//     of the MFMAs (two register staging sets), double-buffered in LDS, one barrier per slab;
//   * LDS rows are padded to 144 doubles so the two k-rows a 32-lane group reads with ds_read_b64
//     fall in disjoint halves of the 64-bank row (conflict-free: SQ_LDS_BANK_CONFLICT = 0);
//   * operands are swapped (MFMA "A" = B tile rows, MFMA "B" = A tile rows) so that lane&15 runs
//     along the contiguous row index of the column-major C tile: every C load/store instruction
//     touches four full 128-byte lines;
//   * C is pre-loaded into the accumulators and the subtraction is done by the MFMA's NEG-A modifier
//     (the BLGP field of v_mfma_f64), so the epilogue is stores only;
//   * diagonal tiles of a symmetric update (GemmArgs::sym) need only their lower triangle and go to their own
//     kernel (tile_syrk_diag_kernel below): of the 8 x 8 grid of 16 x 16 sub-tiles the 36 with column <= row
//     are dealt 9 per wave (sub-tile rows w and 7 - w) — 9/16 of the MFMAs of a full tile, one operand
//     stream instead of two.  (Skipping the upper-right quadrant inside this kernel saves nothing: the
//     workgroup still waits for its three other waves.)
//   * XCD-aware work split: XCD x (blockIdx % 8) owns the x-th contiguous run of work items, so the
//     workgroups that share an L2 work on neighbouring tiles of the same matrices.
#include "gpslc_internal.h"
#include "diag_block.h"
#include "back_block.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// measurement-only paths (in-kernel stamps, timing-only variants) exist in the -DGPSLC_DIAG build only
#ifdef GPSLC_DIAG
#define GP_DBG_ON(g) ((g).dbg != nullptr)
#define GP_DIAG_SKIP(g) ((g).diag_skip)
#define GP_FENCE_MODE(a) ((a).fence_mode & 1)
#define GP_TASK_PRIO(a) (((a).fence_mode >> 4) & 3)      // GPSLC_TASK_FENCE bits 4..5: priority of the diagonal tasks
// bit 6: diag(1) of matrix 0 never publishes its progress — a deliberately broken hand-off, for the test of the time-out word
// (tests/test_gpu_tasks.py: the launch must drain and the call must come back with GPSLC_ERR_INTERNAL, not hang)
#define GP_TASK_WITHHOLD(a, b, is_diag, k) ((((a).fence_mode >> 6) & 1) && (b) == 0 && (is_diag) && (k) == 1)
#else
#define GP_DBG_ON(g) false
#define GP_DIAG_SKIP(g) 0
#define GP_FENCE_MODE(a) 0
#define GP_TASK_PRIO(a) 3
#define GP_TASK_WITHHOLD(a, b, is_diag, k) false
#endif

// The trailing-update kernel raises its waves' issue priority around the 64 MFMAs of a slab (s_setprio 1) and drops it for
// the staging stores, the barrier and the next loads: the wave that HAS MFMAs ready wins the issue slot over its partner's
// staging instructions.  Measured (profiles/r04_ab_experiments.md §11, three alternating pairs): trailing update 66.4 ->
// 67.2 TFLOP/s, unit A +0.55 %; priority 3 the same; on the strip kernel -0.3 % (its second phase wants the partner's
// staging to proceed) -> applied to tile_gemm_nt_kernel only.
// K-loop barriers of the trailing and strip kernels order LDS traffic only: __syncthreads() carries a fence that also waits
// vmcnt(0), i.e. for the slab requested at the top of the iteration — the register staging then ran ONE slab ahead, not two.
// The barrier needs exactly two things, both LDS: this wave's ds_writes of the next slab have landed (lgkmcnt(0)) and every
// wave has issued the MFMAs that consumed its ds_reads of the buffer about to be overwritten (program order before s_barrier).
// Measured (profiles/r04_ab_experiments.md §13): trailing update 66.9 -> 67.2 TFLOP/s, unit A +0.3 %, N = 1024 0..+1.7 %.
#define KLOOP_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define MFMA_PRIO_UP __builtin_amdgcn_s_setprio(1)
#define MFMA_PRIO_DOWN __builtin_amdgcn_s_setprio(0)

#define KS 16
#define LROW 144                      // padded k-row (doubles)
#define OPER_LDS (KS * LROW)          // doubles per operand per stage
#define GEMM_LDS_BYTES (2 * 2 * OPER_LDS * 8)
#ifndef SYRK_PF
#define SYRK_PF 2                      // staging depth (register sets = slabs in flight) of the diagonal-tile update loops
#endif
#define FUSE_WD 8                      // inv(L) fragments in flight ahead of their MFMAs (strip kernel)

__device__ __forceinline__ void tri_decode(int t, int& ii, int& jj) {
    // t = ii(ii+1)/2 + jj, 0 <= jj <= ii
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long long)(r + 1) * (r + 2) / 2 <= t) ++r;
    while ((long long)r * (r + 1) / 2 > t) --r;
    ii = r;
    jj = t - r * (r + 1) / 2;
}

template <int NEG>
__device__ __forceinline__ d4 mfma_step(double a, double b, d4 c) {
    // blgp bit 0 = negate the MFMA A operand (f64 MFMA re-uses BLGP as NEG[2:0])
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, NEG);
}

template <int ACC, int DIAG>
__global__ __launch_bounds__(256, 2) void tile_gemm_nt_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    double* lA = smem;                       // [2][KS][LROW]
    double* lB = smem + 2 * OPER_LDS;        // [2][KS][LROW]

    // ---- persistent, XCD-aware work split
    const long long W = (long long)g.ntiles * g.nbatch;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);                 // workgroups on this XCD
    const long long wq = W >> 3, wrm = W & 7;
    const long long x0 = xcd * wq + (xcd < wrm ? xcd : wrm);           // first item of this XCD's run
    const long long xc = wq + (xcd < wrm ? 1 : 0);                     // items in the run

    const int crow = wr * 64 + (lane & 15);
    const int ccol = wc * 64 + (lane >> 4);
    const int frow_a = (lane >> 4) * LROW + wr * 64 + (lane & 15);
    const int frow_b = (lane >> 4) * LROW + wc * 64 + (lane & 15);
    int loff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = tid + 256 * u;
        loff[u] = (q >> 6) * LROW + (q & 63) * 2;
    }
    const int nslab = (g.k1 - g.k0) * (GP_TS / KS);

    // Work distribution inside the XCD's run: the first item of a workgroup is `local`; the following ones come
    // from a per-XCD ticket counter (GemmArgs::queue) when the launch provides one — items differ in cost
    // (augmented-row tiles, skipped diagonal tiles), a static stride leaves the slowest workgroup ~5 items
    // behind the mean — or from the static stride otherwise.  The ticket for the NEXT item is requested at the
    // start of the current one, so its latency hides behind the tile.
    __shared__ int s_ticket;
    long long it = local;
    while (it < xc) {
        int ticket = 0;
        if (g.queue && tid == 0) ticket = atomicAdd(&g.queue[xcd], 1);
        do {
        const long long item = x0 + it;
        const int b = (int)(item / g.ntiles);
        const int t = (int)(item - (long long)b * g.ntiles);
        int ii, jj;
        if (g.order) { ii = g.order[2 * t]; jj = g.order[2 * t + 1]; }
        else if (g.shape == 0) tri_decode(t, ii, jj);
        else { ii = t / g.mj; jj = t - ii * g.mj; }
        const int ti = g.i0 + ii, tj = g.j0 + jj;
        // sym == 2: the full-size diagonal tiles of this launch belong to tile_syrk_diag_kernel
        if (g.sym >= 2 && ti == tj && !(g.short_rows > 0 && ti >= g.short_row0)) break;
        // sym == 3: the augmented-row tiles of the columns that have a full-size diagonal tile ride with it
        if (g.sym == 3 && g.short_rows > 0 && ti >= g.short_row0 && tj < g.short_row0) break;
        if (g.skip_gdiag && ti == tj && g.short_rows > 0 && ti >= g.short_row0) break;   // -R R^T is not needed (EpiArgs::from_rows)
        if (GP_DIAG_SKIP(g) == 3 && g.short_rows > 0 && ti >= g.short_row0) break;   // timing-only: price of the short tiles

        double* __restrict__ Ct = tref_tile(g.C, b, ti, tj);
        // augmented right-hand-side rows hold only `short_rows` live rows: this wave's number of live
        // 16-row sub-tiles (wave-uniform); the dead ones keep their (zero) C values untouched
        int mlive = 4;
        if (g.short_rows > 0 && ti >= g.short_row0) {
            const int mt = (g.short_rows + 15) >> 4;
            mlive = min(4, max(0, mt - 4 * wr));
        }
        // Panel product (ACC == 0): B = inv(L_kk) is lower triangular, so output column block cb needs only the
        // slabs s <= cb.  With the quadrant layout the two right-hand waves carry 416 of the 1152 live MFMAs
        // each (critical path 81 % of a full tile for 56 % of its work); here every wave owns ALL 8 row
        // blocks of the column blocks pw and 7 - pw instead: 32 x ((pw + 1) + (8 - pw)) = 288 MFMAs per wave.
        // acc[m >> 1][2 (m & 1) + n]: row block m, column block n ? 7 - pw : pw.
        const int pw = __builtin_amdgcn_readfirstlane(wave);
        int prow = 8;                                     // live 16-row blocks of this tile
        if (!ACC && g.short_rows > 0 && ti >= g.short_row0) prow = min(8, (g.short_rows + 15) >> 4);
        unsigned long long st0 = 0, st1 = 0, st2 = 0;
        if (GP_DBG_ON(g)) st0 = __builtin_amdgcn_s_memtime();

        // ---- accumulators: acc[m][n][v] = C[wr*64 + 16m + (lane&15)][wc*64 + 16n + (lane>>4) + 4v]
        d4 acc[4][4];
        if (ACC) {
            const double* __restrict__ Cl = Ct + (ccol * GP_TS + crow);
            if (g.nt_c) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            acc[m][n][v] = __builtin_nontemporal_load(Cl + (16 * n + 4 * v) * GP_TS + 16 * m);
            } else {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            acc[m][n][v] = Cl[(16 * n + 4 * v) * GP_TS + 16 * m];
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
        }

        if (nslab > 0) {
            // staging: 16 KiB per operand per slab = 1024 16-byte chunks, 4 per thread, contiguous in HBM
            d2 ra[4], rb[4];     // staging set A
            d2 ra2[4], rb2[4];   // staging set B: loads run two slabs ahead of the MFMAs
            auto gload = [&](int s, d2 (&xa)[4], d2 (&xb)[4]) {
                const int kk = g.k0 + (s >> 3);
                const int so = (s & 7) * (KS * GP_TS);
                const double* pa = tref_tile(g.A, b, ti, kk) + so;
                const double* pb = tref_tile(g.B, b, tj, kk) + so;
                // panel product of a short (augmented-row) tile: the chunks of A below its live rows are never
                // used — point them at chunk 0 of their k-column's line instead of streaming zeros
                const int ao = (!ACC && ((tid & 63) * 2) >= 16 * prow) ? (tid & ~63) * 2 : tid * 2;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xa[u] = *reinterpret_cast<const d2*>(pa + ao + 512 * u);
                    xb[u] = *reinterpret_cast<const d2*>(pb + (tid + 256 * u) * 2);
                }
            };
            auto lstore = [&](int buf, const d2 (&xa)[4], const d2 (&xb)[4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    *reinterpret_cast<d2*>(lA + buf * OPER_LDS + loff[u]) = xa[u];
                    *reinterpret_cast<d2*>(lB + buf * OPER_LDS + loff[u]) = xb[u];
                }
            };
            auto compute = [&](int buf, int s) {
                const double* pa = lA + buf * OPER_LDS + frow_a;
                const double* pb = lB + buf * OPER_LDS + frow_b;
                // Panel product (ACC == 0): B is the lower-triangular inverse inv(L_kk)[c][c'], zero for
                // c' > c.  Slab s carries c' in [16 (s&7), 16 (s&7) + 16), so only the output column
                // sub-tiles n with 64 wc + 16 n + 15 >= 16 (s&7) contribute: 44 % fewer MFMAs.
                const int nlo = ACC ? 0 : min(4, max(0, (s & 7) - 4 * wc));
                // trailing updates (no second phase, registers to spare): the fragments of k-step ks + 1 are read from LDS
                // before the 16 MFMAs of k-step ks are issued, so that a wave never starts a k-step with an LDS round trip
                constexpr bool PIPE = ACC && DIAG == 0;
                double afn[4], bfn[4];
                if (PIPE) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) afn[m] = pa[16 * m];
#pragma unroll
                    for (int n = 0; n < 4; ++n) bfn[n] = pb[16 * n];
                }
#pragma unroll
                for (int ks = 0; ks < KS / 4; ++ks) {
                    double af[4], bf[4];
                    if (PIPE) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) { af[m] = afn[m]; bf[m] = bfn[m]; }
                        if (ks + 1 < KS / 4) {
#pragma unroll
                            for (int m = 0; m < 4; ++m) afn[m] = pa[(ks + 1) * 4 * LROW + 16 * m];
#pragma unroll
                            for (int n = 0; n < 4; ++n) bfn[n] = pb[(ks + 1) * 4 * LROW + 16 * n];
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < 4; ++m) af[m] = pa[ks * 4 * LROW + 16 * m];
#pragma unroll
                        for (int n = 0; n < 4; ++n) bf[n] = pb[ks * 4 * LROW + 16 * n];
                    }
                    if (mlive == 4 && nlo == 0) {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int n = 0; n < 4; ++n)
                                acc[m][n] = mfma_step<ACC>(bf[n], af[m], acc[m][n]);
                    } else {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            if (m < mlive) {
#pragma unroll
                                for (int n = 0; n < 4; ++n)
                                    if (n >= nlo) acc[m][n] = mfma_step<ACC>(bf[n], af[m], acc[m][n]);
                            }
                    }
                }
            };

            auto compute_p = [&](int buf, int s) {
                const int sl = s & 7;
                if (sl > 7 - pw) return;                   // neither column block needs this slab
                const bool use0 = sl <= pw;
                const double* pa = lA + buf * OPER_LDS + (lane >> 4) * LROW + (lane & 15);
                const double* pb = lB + buf * OPER_LDS + (lane >> 4) * LROW + (lane & 15);
#pragma unroll
                for (int ks = 0; ks < KS / 4; ++ks) {
                    const double b0 = pb[ks * 4 * LROW + 16 * pw];
                    const double b1 = pb[ks * 4 * LROW + 16 * (7 - pw)];
                    double af[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) af[m] = pa[ks * 4 * LROW + 16 * m];
#pragma unroll
                    for (int m = 0; m < 8; ++m)
                        if (m < prow) {
                            if (use0) acc[m >> 1][2 * (m & 1)] = mfma_step<0>(b0, af[m], acc[m >> 1][2 * (m & 1)]);
                            acc[m >> 1][2 * (m & 1) + 1] = mfma_step<0>(b1, af[m], acc[m >> 1][2 * (m & 1) + 1]);
                        }
                }
            };

            gload(0, ra, rb);
            lstore(0, ra, rb);
            gload(1, ra, rb);          // nslab is a multiple of 8
            __syncthreads();
            if (GP_DBG_ON(g)) st1 = __builtin_amdgcn_s_memtime();
            // unrolled by two, the staging sets swapping roles: set A holds slab s+1 while slab s+2
            // streams into set B.  The barrier that ends the last slab also protects the LDS buffers
            // against the next work item's first lstore.
            if (DIAG == 0) {
                for (int s = 0; s < nslab; s += 2) {
                    if (s + 2 < nslab) gload(s + 2, ra2, rb2);
                    MFMA_PRIO_UP;
                    if (ACC) compute(0, s); else compute_p(0, s);
                    MFMA_PRIO_DOWN;
                    lstore(1, ra, rb);
                    KLOOP_BARRIER;
                    if (s + 3 < nslab) gload(s + 3, ra, rb);
                    MFMA_PRIO_UP;
                    if (ACC) compute(1, s + 1); else compute_p(1, s + 1);
                    MFMA_PRIO_DOWN;
                    if (s + 2 < nslab) lstore(0, ra2, rb2);
                    KLOOP_BARRIER;
                }
            } else {   // timing-only diagnostic: same MFMA / ds_read stream, operand traffic removed
                for (int s = 0; s < nslab; s += 2) {
                    compute(0, s);
                    if (DIAG == 1) lstore(1, ra, rb);
                    __syncthreads();
                    compute(1, s + 1);
                    if (DIAG == 1) lstore(0, ra, rb);
                    __syncthreads();
                }
            }
        }
        if (GP_DBG_ON(g)) st2 = __builtin_amdgcn_s_memtime();

        {
        // recompute the store addresses from one opaque offset instead of keeping the 64 preload
        // addresses alive (and spilled) across the K loop
        if (!ACC) {
            // panel layout: row block m, column blocks pw (n = 0) and 7 - pw (n = 1); dead rows of a short tile
            // hold zeros (never accumulated, never written)
            double* __restrict__ Cp = Ct + ((lane >> 4) * GP_TS + (lane & 15));
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (m < prow) {     // the dead row blocks of a short tile keep their zeros
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        Cp[(16 * pw + 4 * v) * GP_TS + 16 * m] = acc[m >> 1][2 * (m & 1)][v];
                        Cp[(16 * (7 - pw) + 4 * v) * GP_TS + 16 * m] = acc[m >> 1][2 * (m & 1) + 1][v];
                    }
                }
        } else {
        int soff = ccol * GP_TS + crow;
        asm volatile("" : "+v"(soff));
        double* __restrict__ Cs = Ct + soff;
        if (g.nt_c) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        __builtin_nontemporal_store(acc[m][n][v], Cs + (16 * n + 4 * v) * GP_TS + 16 * m);
        } else {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    Cs[(16 * n + 4 * v) * GP_TS + 16 * m] = acc[m][n][v];
        }
        }
        }
        if (GP_DBG_ON(g)) {   // diagnostic stamps: go to a buffer nothing else reads
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long st3 = __builtin_amdgcn_s_memtime();
            if (tid == 0) {
                unsigned long long* d = g.dbg + (size_t)item * 8;
                d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3;
                d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
                d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
                d[6] = __builtin_amdgcn_s_memrealtime();
                d[7] = blockIdx.x;
            }
        }
        } while (0);
        if (g.queue) {
            if (tid == 0) s_ticket = ticket;
            __syncthreads();
            it = (long long)gx + s_ticket;
            __syncthreads();
        } else {
            it += gx;
        }
    }
    // the last workgroup of the XCD to leave re-arms the counters for the next launch on this stream
    if (g.queue && tid == 0) {
        __threadfence();
        if (atomicAdd(&g.queue[8 + xcd], 1) == gx - 1) {
            g.queue[xcd] = 0;
            g.queue[8 + xcd] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------
// In-panel column update fused with the panel solve, STRIP layout (round 3).
//     X = C(i,k) - sum_kk A(i,kk) B(k,kk)^T          (K loop, X stays in the accumulators)
//     L(i,k) = X * inv(L_kk)^T                          (second phase, GemmArgs::F = the inverted diagonal blocks)
// Wave w owns the 32-row strip [32 w, 32 w + 32) x ALL 128 columns of the tile: acc[m][n][v] =
// X[32 w + 16 m + (lane & 15)][16 n + 4 v + (lane >> 4)] (2 x 8 accumulators, the same 128 registers as a 64 x 64
// quadrant).  A lane's accumulator registers are exactly its k-operands of an MFMA over the columns of X, so the
// second product needs nothing but the wave's own registers and the fragments of inv(L_kk) (L2-resident, shared by the
// launch): 36 live 16 x 16 blocks of the lower-triangular inverse x 4 k-steps x 2 row blocks = 288 MFMAs on EVERY wave,
// no LDS hand-over and no barrier.  (With 64 x 64 quadrants the product X(:, 0:64) inv(L)(64:128, 0:64)^T crossed from
// the left-hand to the right-hand waves through 64 KiB of LDS between two barriers and the waves ran 416 / 160 MFMAs:
// 64 k clocks of second phase per item against 37 k of MFMA time, profiles/r02_fused_kernel_stamps.md.)
// Summation order per output element: ascending k in the K loop, ascending column of X in the second phase — the
// order of the quadrant kernel and of the separate panel product, so the factor is bit-identical.
// The A(i, kk) slabs and the C tile are streamed exactly once per launch: they are loaded / stored NON-TEMPORAL so that
// they do not displace the B(k, kk) panel (shared by the ~24 items of a matrix) and inv(L_kk) in L2 (+1.8 % on the
// kernel, profiles/r03_ab_experiments.md §3).
// ---------------------------------------------------------------------------------------
// One work item of the strip kernel: tile (ti, tj) of batch element b.  no_update: the tile needs the panel product
// only (its column update was done elsewhere, or there is none: first column of a panel).
// AUGEP (task kernel only): the item also applies the panel product to the augmented right-hand-side tile (short_row0, tj) of
// its column — the tile diag(tj) has already updated — instead of that tile being a work item of its own: all four waves load
// its live rows, wave w computes the column blocks w and 7 - w.  Per output element the same MFMA chain as the stand-alone
// item (ascending column block of X, then v): bit-identical.
template <int WD, bool AUGEP = false, bool WT = false>
__device__ __forceinline__ void strip_item(const GemmArgs& g, const int b, const int ti, const int tj, const bool no_update,
                                           const int nslab, double* lA, double* lB, const int tid, const int lane,
                                           const int wave, const int li, const int lg, const int frow_a, const int frow_b,
                                           const int (&loff)[4], const long long item, const bool with_aug = false) {
    double* __restrict__ Ct = tref_tile(g.C, b, ti, tj);
    // augmented right-hand-side tiles hold `short_rows` live rows: a wave whose strip lies below them only helps
    // with the staging (its rows keep their zeros: never loaded, never stored)
    const bool live = !(g.short_rows > 0 && ti >= g.short_row0 && 32 * wave >= g.short_rows);
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (GP_DBG_ON(g)) st0 = __builtin_amdgcn_s_memtime();

    d4 acc[2][8];
    if (live) {
        const double* __restrict__ Cl = Ct + (lg * GP_TS + 32 * wave + li);
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][n][v] = __builtin_nontemporal_load(Cl + (16 * n + 4 * v) * GP_TS + 16 * m);
    }

    if (nslab > 0 && !no_update) {
        d2 ra[4], rb[4], ra2[4], rb2[4];
        auto gload = [&](int s, d2 (&xa)[4], d2 (&xb)[4]) {
            const int kk = g.k0 + (s >> 3);
            const int so = (s & 7) * (KS * GP_TS);
            const double* pa = tref_tile(g.A, b, ti, kk) + so;
            const double* pb = tref_tile(g.B, b, tj, kk) + so;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xa[u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(pa + (tid + 256 * u) * 2));
                xb[u] = *reinterpret_cast<const d2*>(pb + (tid + 256 * u) * 2);
            }
        };
        auto lstore = [&](int buf, const d2 (&xa)[4], const d2 (&xb)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                *reinterpret_cast<d2*>(lA + buf * OPER_LDS + loff[u]) = xa[u];
                *reinterpret_cast<d2*>(lB + buf * OPER_LDS + loff[u]) = xb[u];
            }
        };
        auto compute = [&](int buf) {
            if (!live) return;
            const double* pa = lA + buf * OPER_LDS + frow_a;
            const double* pb = lB + buf * OPER_LDS + frow_b;
#pragma unroll
            for (int ks = 0; ks < KS / 4; ++ks) {
                double af[2], bf[8];
#pragma unroll
                for (int m = 0; m < 2; ++m) af[m] = pa[ks * 4 * LROW + 16 * m];
#pragma unroll
                for (int n = 0; n < 8; ++n) bf[n] = pb[ks * 4 * LROW + 16 * n];
#pragma unroll
                for (int n = 0; n < 8; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][n] = mfma_step<1>(bf[n], af[m], acc[m][n]);
            }
        };
        gload(0, ra, rb);
        lstore(0, ra, rb);
        gload(1, ra, rb);          // nslab is a multiple of 8
        __syncthreads();
        if (GP_DBG_ON(g)) st1 = __builtin_amdgcn_s_memtime();
        for (int s = 0; s < nslab; s += 2) {
            if (s + 2 < nslab) gload(s + 2, ra2, rb2);
            compute(0);
            lstore(1, ra, rb);
            KLOOP_BARRIER;
            if (s + 3 < nslab) gload(s + 3, ra, rb);
            compute(1);
            if (s + 2 < nslab) lstore(0, ra2, rb2);
            KLOOP_BARRIER;
        }
    }
    if (GP_DBG_ON(g)) st2 = __builtin_amdgcn_s_memtime();

    if (live) {
        // fragment (nc, n, v) of W = inv(L_kk): W[16 nc + li][16 n + 4 v + lg], element (c, c') at c' * 128 + c
        const double* __restrict__ Wl = tref_tile(g.F, b, 0, g.fk) + (lg * GP_TS + li);
        double* __restrict__ Co = Ct + (lg * GP_TS + 32 * wave + li);
        // the 144 live fragments in the order of use, WD groups (WD x 128 MFMA clocks) ahead through a register ring
        double wn[WD];
        int pc = 0, pn = 0, pv = 0;        // next fragment to request (compile-time after unrolling)
#pragma unroll
        for (int q = 0; q < WD; ++q) {
            wn[q] = Wl[(16 * pn + 4 * pv) * GP_TS + 16 * pc];
            if (++pv == 4) { pv = 0; if (++pn > pc) { pn = 0; ++pc; } }
        }
        int q = 0;
#pragma unroll
        for (int nc = 0; nc < 8; ++nc) {
            d4 st[2];
            st[0] = (d4){0.0, 0.0, 0.0, 0.0};
            st[1] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int n = 0; n < 8; ++n)
                if (n <= nc) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const double w = wn[q % WD];
                        if (pc < 8) {
                            wn[q % WD] = Wl[(16 * pn + 4 * pv) * GP_TS + 16 * pc];
                            if (++pv == 4) { pv = 0; if (++pn > pc) { pn = 0; ++pc; } }
                        }
                        ++q;
                        st[0] = mfma_step<0>(w, acc[0][n][v], st[0]);
                        st[1] = mfma_step<0>(w, acc[1][n][v], st[1]);
                    }
                }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (WT) {       // task launch: write-through, so that the consumer workgroup needs no L2 write-back (out_store)
                    out_store<true>(Co + (16 * nc + 4 * v) * GP_TS, st[0][v]);
                    out_store<true>(Co + (16 * nc + 4 * v) * GP_TS + 16, st[1][v]);
                } else {
                    __builtin_nontemporal_store(st[0][v], Co + (16 * nc + 4 * v) * GP_TS);
                    __builtin_nontemporal_store(st[1][v], Co + (16 * nc + 4 * v) * GP_TS + 16);
                }
            }
        }
    }
    if (AUGEP && with_aug) {
        const int mrows = (g.short_rows + 15) >> 4;           // live 16-row blocks of the augmented tile (1 or 2)
        double* __restrict__ Cg = tref_tile(g.C, b, g.short_row0, tj) + (lg * GP_TS + li);
        const double* __restrict__ Wl = tref_tile(g.F, b, 0, g.fk) + (lg * GP_TS + li);
        d4 ag[2][8];
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (m < mrows) ag[m][n][v] = Cg[(16 * n + 4 * v) * GP_TS + 16 * m];
        auto colblock = [&](auto ncc) {
            constexpr int nc = decltype(ncc)::value;
            double wf[nc + 1][4];
#pragma unroll
            for (int n = 0; n <= nc; ++n)
#pragma unroll
                for (int v = 0; v < 4; ++v) wf[n][v] = Wl[(16 * n + 4 * v) * GP_TS + 16 * nc];
            d4 st[2];
            st[0] = (d4){0.0, 0.0, 0.0, 0.0};
            st[1] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int n = 0; n <= nc; ++n)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    st[0] = mfma_step<0>(wf[n][v], ag[0][n][v], st[0]);
                    if (mrows > 1) st[1] = mfma_step<0>(wf[n][v], ag[1][n][v], st[1]);
                }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                out_store<WT>(Cg + (16 * nc + 4 * v) * GP_TS, st[0][v]);
                if (mrows > 1) out_store<WT>(Cg + (16 * nc + 4 * v) * GP_TS + 16, st[1][v]);
            }
        };
        // every wave has read the whole tile before any wave overwrites a column block of it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        switch (wave) {
            case 0: colblock(std::integral_constant<int, 0>()); colblock(std::integral_constant<int, 7>()); break;
            case 1: colblock(std::integral_constant<int, 1>()); colblock(std::integral_constant<int, 6>()); break;
            case 2: colblock(std::integral_constant<int, 2>()); colblock(std::integral_constant<int, 5>()); break;
            default: colblock(std::integral_constant<int, 3>()); colblock(std::integral_constant<int, 4>()); break;
        }
    }
    if (GP_DBG_ON(g)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long* d = g.dbg + (size_t)item * 8;
            d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3;
            d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            d[6] = __builtin_amdgcn_s_memrealtime();
            d[7] = blockIdx.x;
        }
    }
}

template <int WD>
__global__ __launch_bounds__(256, 2) void tile_fused_strip_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    double* lA = smem;                       // [2][KS][LROW]
    double* lB = smem + 2 * OPER_LDS;        // [2][KS][LROW]

    const long long W = (long long)g.ntiles * g.nbatch;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const long long wq = W >> 3, wrm = W & 7;
    const long long x0 = xcd * wq + (xcd < wrm ? xcd : wrm);
    const long long xc = wq + (xcd < wrm ? 1 : 0);

    const int frow_a = lg * LROW + 32 * wave + li;     // + 16 m
    const int frow_b = lg * LROW + li;                 // + 16 n
    int loff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = tid + 256 * u;
        loff[u] = (q >> 6) * LROW + (q & 63) * 2;
    }
    const int nslab = (g.k1 - g.k0) * (GP_TS / KS);

    __shared__ int s_ticket;
    long long it = local;
    while (it < xc) {
        int ticket = 0;
        if (g.queue && tid == 0) ticket = atomicAdd(&g.queue[xcd], 1);
        do {
        const long long item = x0 + it;
        const int b = (int)(item / g.ntiles);
        const int t = (int)(item - (long long)b * g.ntiles);
        int ii, jj;
        if (g.shape == 0) tri_decode(t, ii, jj);
        else { ii = t / g.mj; jj = t - ii * g.mj; }
        const int ti = g.i0 + ii, tj = g.j0 + jj;
        if (g.sym >= 2 && ti == tj && !(g.short_rows > 0 && ti >= g.short_row0)) break;   // tile_syrk_diag_kernel's
        // sym == 3: the augmented tile was already updated (it rode with the diagonal item); panel product only
        const bool no_update = g.sym == 3 && g.short_rows > 0 && ti >= g.short_row0 && tj < g.short_row0;
        strip_item<WD>(g, b, ti, tj, no_update, nslab, lA, lB, tid, lane, wave, li, lg, frow_a, frow_b, loff, item);
        } while (0);
        if (g.queue) {
            if (tid == 0) s_ticket = ticket;
            __syncthreads();
            it = (long long)gx + s_ticket;
            __syncthreads();
        } else {
            it += gx;
        }
    }
    if (g.queue && tid == 0) {
        __threadfence();
        if (atomicAdd(&g.queue[8 + xcd], 1) == gx - 1) {
            g.queue[xcd] = 0;
            g.queue[8 + xcd] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Update of a diagonal tile INSIDE a panel, one wave's share: tile (td, td) -= sum_{kk in [g.k0, kd1)} A(td, kk) A(td, kk)^T,
// lower triangle, 9 sub-tiles per wave, the augmented right-hand-side rows (MT row blocks) riding along — and the result left
// as the PACKED LDS IMAGE the factorisation of diag_block.h works on, so that diag_update_potrf_kernel below goes from the
// update to the Cholesky + inverse without the tile's HBM round trip.  Same arithmetic and summation order as
// tile_syrk_diag_kernel (syrk_diag_wave below): the factor is bit-identical to the two-launch form.
// (The round-4 experiment that chained this INTO the strip kernel's launch — one launch per column, measured slower at every
// size — lives in profiles/r04_chain_experiment.patch, not in the sources.)
// ---------------------------------------------------------------------------------------
template <int W, int MT, bool WT = false>
__device__ __forceinline__ void syrk_chain_wave(const GemmArgs& g, const int b, const int td, const int kd1, double* smem,
                                                const int tid, const int lane) {
    // diagonal tile (td, td) -= sum_{kk in [g.k0, kd1)} A(td, kk) A(td, kk)^T -> packed LDS image (smem); augmented tile
    // (g.short_row0, td) likewise -> HBM.  Layouts and MFMA order: syrk_diag_wave below.
    double* lA = smem;
    double* lG = smem + 2 * OPER_LDS;
    int loff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = tid + 256 * u;
        loff[u] = (q >> 6) * LROW + (q & 63) * 2;
    }
    const bool glive = MT > 0 && tid < 128 * MT;     // one row pair of the augmented slab per thread
    const int nslab = (kd1 - g.k0) * (GP_TS / KS);
    const int li = lane & 15, lg = lane >> 4;
    const int fbase = lg * LROW + li;
    double* __restrict__ Cd = tref_tile(g.C, b, td, td) + (lg * GP_TS + li);
    double* __restrict__ Cg = MT > 0 ? tref_tile(g.C, b, g.short_row0, td) + (lg * GP_TS + li) : nullptr;

    d4 a0c[W + 1], a1c[8 - W];     // sub-tiles (W, cb) and (7 - W, cb)
#pragma unroll
    for (int cb = 0; cb <= W; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) a0c[cb][v] = Cd[(16 * cb + 4 * v) * GP_TS + 16 * W];
#pragma unroll
    for (int cb = 0; cb < 8 - W; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) a1c[cb][v] = Cd[(16 * cb + 4 * v) * GP_TS + 16 * (7 - W)];
    d4 ag[MT > 0 ? MT : 1][2];
    if (MT > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                ag[m][0][v] = Cg[(16 * W + 4 * v) * GP_TS + 16 * m];
                ag[m][1][v] = Cg[(16 * (7 - W) + 4 * v) * GP_TS + 16 * m];
            }
    }
    // Staging: SYRK_PF register sets, loads SYRK_PF slabs ahead of the MFMAs.  A slab of this update is 9 MFMAs per wave and
    // k-step — 576 pipe clocks — so its loads need more than one slab time to arrive; the barriers order LDS traffic only
    // (__syncthreads() waits vmcnt(0): with it the "two slabs ahead" of round 3 was in fact less than one).  The augmented
    // rows' loads travel with their slab (vmcnt counts in order), one row pair per thread.  Measured (profiles/
    // r04_ab_experiments.md §12): 2 sets + LDS-only barriers +1.1..2.0 % at N = 1024, 4 sets no better, 8 sets spill.
    // augmented slab: 16 k-columns x 16 MT live rows = 128 MT row pairs, ONE per thread (tid < 128 MT)
    d2 rs[SYRK_PF][4], rgs[MT > 0 ? SYRK_PF : 1];
    const int gcol = tid / (8 * (MT > 0 ? MT : 1)), grp = tid % (8 * (MT > 0 ? MT : 1));
    auto gload = [&](int s, int set) {
        const int kk = g.k0 + (s >> 3);
        const double* pa = tref_tile(g.A, b, td, kk) + (s & 7) * (KS * GP_TS);
#pragma unroll
        for (int u = 0; u < 4; ++u) rs[set][u] = *reinterpret_cast<const d2*>(pa + (tid + 256 * u) * 2);
        if (MT > 0 && glive) {
            const double* pg = tref_tile(g.A, b, g.short_row0, kk) + (s & 7) * (KS * GP_TS);
            rgs[MT > 0 ? set : 0] = *reinterpret_cast<const d2*>(pg + gcol * GP_TS + 2 * grp);
        }
    };
    auto lstore = [&](int buf, int set) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<d2*>(lA + buf * OPER_LDS + loff[u]) = rs[set][u];
        if (MT > 0 && glive)
            *reinterpret_cast<d2*>(lG + buf * OPER_LDS + gcol * LROW + 2 * grp) = rgs[MT > 0 ? set : 0];
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto compute = [&](int buf) {
        const double* pa = lA + buf * OPER_LDS + fbase;
        const double* pg = lG + buf * OPER_LDS + fbase;
#pragma unroll
        for (int ks = 0; ks < KS / 4; ++ks) {
            double bf[8 - W];
#pragma unroll
            for (int cb = 0; cb < 8 - W; ++cb) bf[cb] = pa[ks * 4 * LROW + 16 * cb];
            const double r0 = pa[ks * 4 * LROW + 16 * W];
            const double r1 = pa[ks * 4 * LROW + 16 * (7 - W)];
#pragma unroll
            for (int cb = 0; cb <= W; ++cb) a0c[cb] = mfma_step<1>(bf[cb], r0, a0c[cb]);
#pragma unroll
            for (int cb = 0; cb < 8 - W; ++cb) a1c[cb] = mfma_step<1>(bf[cb], r1, a1c[cb]);
            if (MT > 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const double gf = pg[ks * 4 * LROW + 16 * m];
                    ag[m][0] = mfma_step<1>(bf[W], gf, ag[m][0]);
                    ag[m][1] = mfma_step<1>(bf[7 - W], gf, ag[m][1]);
                }
            }
        }
    };
#pragma unroll
    for (int u = 0; u < SYRK_PF; ++u) gload(u, u);          // nslab is a multiple of 8 >= SYRK_PF
    lstore(0, 0);
    lds_barrier();
    for (int s = 0; s < nslab; s += SYRK_PF) {
#pragma unroll
        for (int u = 0; u < SYRK_PF; ++u) {                  // slab s + u sits in LDS buffer u & 1
            if (s + u + SYRK_PF < nslab) gload(s + u + SYRK_PF, u);      // set u went to LDS one step ago
            compute(u & 1);
            if (s + u + 1 < nslab) lstore((u + 1) & 1, (u + 1) % SYRK_PF);
            lds_barrier();
        }
    }
    {
        // the staging buffers are dead (last barrier): the updated lower blocks become the packed image of the factorisation
        // — block (i, j) at ((i (i + 1) / 2 + j) << 8), element (r, c) at c * 16 + r; acc[v] = element (li, lg + 4 v)
#pragma unroll
        for (int cb = 0; cb <= W; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) smem[(((W * (W + 1)) / 2 + cb) << 8) + (lg + 4 * v) * 16 + li] = a0c[cb][v];
#pragma unroll
        for (int cb = 0; cb < 8 - W; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) smem[((((7 - W) * (8 - W)) / 2 + cb) << 8) + (lg + 4 * v) * 16 + li] = a1c[cb][v];
    }
    if (MT > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                out_store<WT>(Cg + (16 * W + 4 * v) * GP_TS + 16 * m, ag[m][0][v]);
                out_store<WT>(Cg + (16 * (7 - W) + 4 * v) * GP_TS + 16 * m, ag[m][1][v]);
            }
    }
}

// ---------------------------------------------------------------------------------------
// Diagonal tile of an in-panel column in ONE launch (round 4): update of tile (k, k) over [k0, k1) with the
// augmented rows riding along (what tile_syrk_diag_kernel<MT> does for it), then — on the packed image the update leaves in
// LDS, without the HBM round trip of the tile — its Cholesky + inverse (what diag_potrf_inv_la_kernel does).  One workgroup
// per matrix, two per CU.  This pairs the factorisation's fp64 pivot chains with the
// diagonal update's sparse MFMA stream (9 MFMAs per wave and k-step between staging waits), not with a dense one.
// ---------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256, 2) void diag_update_potrf_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.x;
    const int td = g.i0;
    switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
        case 0: syrk_chain_wave<0, MT>(g, b, td, g.k1, smem, tid, lane); break;
        case 1: syrk_chain_wave<1, MT>(g, b, td, g.k1, smem, tid, lane); break;
        case 2: syrk_chain_wave<2, MT>(g, b, td, g.k1, smem, tid, lane); break;
        default: syrk_chain_wave<3, MT>(g, b, td, g.k1, smem, tid, lane); break;
    }
    diag_potrf_inv_la_body(smem, tref_tile(g.C, b, td, td), tref_tile(g.F, b, 0, td), g.info + b, g.info_base + GP_TS * td,
                           tid, true);
}

template <int MT>
static void launch_diag_update_potrf_t(const GemmArgs& g, hipStream_t st) {
    static DeviceOnce once;
    constexpr int bytes = DIAG3_LDS_BYTES > GEMM_LDS_BYTES ? DIAG3_LDS_BYTES : GEMM_LDS_BYTES;
    lds_opt_in(once, (const void*)diag_update_potrf_kernel<MT>, bytes);
    hipLaunchKernelGGL(diag_update_potrf_kernel<MT>, dim3(g.nbatch), dim3(256), bytes, st, g);
}

// g: A = B = C = the tile matrix, i0 = the column k, [k0, k1) = the panel columns left of it (k1 - k0 >= 1), F / info /
// info_base as for launch_diag, short_row0 / short_rows = the augmented row when carry_aug
void launch_diag_update_potrf(const GemmArgs& g, int carry_aug, hipStream_t st) {
    if (g.nbatch <= 0) return;
    const int mt = carry_aug ? (g.short_rows + 15) / 16 : 0;
    if (mt == 0) launch_diag_update_potrf_t<0>(g, st);
    else if (mt == 1) launch_diag_update_potrf_t<1>(g, st);
    else launch_diag_update_potrf_t<2>(g, st);
}

// ---------------------------------------------------------------------------------------
// The whole left-looking factorisation of a batch in ONE persistent launch (round 6).
// A workgroup draws a ticket from its XCD's queue (the queue is chosen by the hardware XCC_ID, not by blockIdx: the
// strips of one matrix column then really share one L2; a workgroup whose queue has run dry goes on with the next
// queue, so every task is executed whatever the placement), reads the task descriptor the host laid out
// (PotrfTaskArgs::list), waits for the task's producers on the matrix's progress words, runs the body — exactly the
// device functions of the per-column launches: strip_item, syrk_chain_wave + diag_potrf_inv_la_body, so every tile
// receives the same MFMA chains in the same order and the factor is bit-identical — and publishes its own progress.
// Hand-off between workgroups (MI355X_MICROARCH.md "inter-workgroup visibility", form R1): every tile another workgroup of the
// launch will read is stored WRITE-THROUGH (agent-scope relaxed atomic stores = global_store ... sc1: out_store, diag_block.h),
// every storing wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane stores the progress word (relaxed,
// agent scope); consumer: ONE lane polls relaxed (bounded, with s_sleep), ONE agent-scope acquire fence, vmcnt(0), workgroup
// barrier, then ordinary loads.  (The plain-store form — nt stores + an agent-scope release fence by one lane — is the
// measurement build's GPSLC_TASK_FENCE=2 variant: 0.8 % slower at N = 1024, 3-4 % at N = 640 / 768, every fence writes the
// XCD's whole L2 back; with NO release at all the consumers read stale tiles at once, profiles/r06_ab_experiments.md §1.)
// No deadlock: inside a queue every task follows its producers (host order), tickets are handed out in that order, and
// a workgroup that holds a ticket is running — the oldest unfinished ticket never waits for anything unfinished.
// A poll that exceeds its bound (a bug, never a schedule) sets the time-out word: every workgroup then stops waiting, the
// launch drains, and the host reports GPSLC_ERR_INTERNAL instead of a hung GPU.
// ---------------------------------------------------------------------------------------
#define TASK_POLL_LIMIT (1 << 20)

__device__ __forceinline__ int task_progress_load(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane: wait until *p >= need (or the launch has timed out)
__device__ __forceinline__ void task_wait(const int* p, int need, int* tmo) {
    if (task_progress_load(p) >= need) return;
    for (int spins = 0;; ++spins) {
        __builtin_amdgcn_s_sleep(8);
        if (task_progress_load(p) >= need) return;
        if ((spins & 63) == 63 && task_progress_load(tmo) != 0) return;
        if (spins > TASK_POLL_LIMIT) {
            __hip_atomic_store(tmo, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

// lane 0 of a workgroup: the next descriptor of queue `q` (or of the queues after it once it has run dry: `visited` counts
// the empty ones), TASK_NONE when all eight are exhausted
__device__ __forceinline__ unsigned task_fetch(const PotrfTaskArgs& a, int& q, int& visited) {
    while (visited < 8) {
        const int len = (int)a.list[8 + q];
        const int t = len > 0 ? atomicAdd(&a.sync[q], 1) : len;
        if (t < len) return a.list[TASK_LIST_HDR + a.list[q] + t];
        q = (q + 1) & 7;
        ++visited;
    }
    return TASK_NONE;
}

template <int MT, bool WT>
__global__ __launch_bounds__(256, 2) void potrf_tasks_kernel(PotrfTaskArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // the fetching lane's state lives in LDS, and everything a body needs is recomputed from the thread index inside its
    // branch: the strip body and the diagonal body each fill the register file on their own (248 / 254 VGPRs), so nothing
    // but the thread index (and lane 0's prefetched ticket) may stay live across them
    __shared__ unsigned s_desc, s_next;
    __shared__ int s_q, s_visited;
    __shared__ unsigned long long s_ready;      // measurement build only
    if (threadIdx.x == 0) {
        int q = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7);     // HW_REG_XCC_ID: this workgroup's XCD
        int visited = 0;                                                    // queues found empty so far
        s_next = task_fetch(a, q, visited);
        s_q = q; s_visited = visited;
    }
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));          // opaque: nothing derived from it is hoisted out of the task loop
        unsigned long long stf = 0;
        if (GP_DBG_ON(a)) stf = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            int* const tmo = a.sync + 8;
            const unsigned d = s_next;
            if (d != TASK_NONE) {
                const int nt = a.nt;
                const int b = TASK_B(d), k = TASK_K(d), i = TASK_I(d), rows = TASK_ROWS(d);
                const int* prog = a.sync + TASK_SYNC_HDR + (long long)b * TASK_SYNC_STRIDE;
                if ((d >> 30) == 3) {               // back-substitution: the whole factor and the solved augmented row
                    task_wait(prog, nt, tmo);
                    task_wait(prog + 1 + nt, nt, tmo);
                } else if ((d >> 30) & 1) {         // diag(k): tile row k (and the augmented row) final up to column k - 1
                    if (k > 0) {
                        task_wait(prog + 1 + k, k, tmo);
                        if (MT > 0) task_wait(prog + 1 + nt, k, tmo);
                        if (rows > 1) task_wait(prog + 2 + k, k, tmo);      // ... and tile row k + 1 for the strip it goes on with
                    }
                } else {                            // strip(i.., k): inv(L_kk) and tile row k (diag(k)), the tile rows up to column k - 1
                    task_wait(prog, k + 1, tmo);
                    // MT == 0: the augmented row (i == nt) is an ordinary tile row whose strips carry their own column update
                    if ((i < nt || MT == 0) && k > 0)
                        for (int r = 0; r < rows; ++r) task_wait(prog + 1 + i + r, k, tmo);
                }
                if (GP_DBG_ON(a)) s_ready = __builtin_amdgcn_s_memtime();       // producers done (before the acquire)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            s_desc = d;
        }
        __syncthreads();
        const unsigned d = s_desc;
        if (d == TASK_NONE) break;
        const int b = TASK_B(d), k = TASK_K(d), i = TASK_I(d), rows = TASK_ROWS(d);
        const bool is_back = (d >> 30) == 3;
        const bool is_diag = (d >> 30) == 1;
        const bool with_aug = (d >> 30) == 2;       // strip(k + 1, k) also carries the augmented tile (nt, k)
        unsigned long long st0 = 0, st1 = 0;
        if (GP_DBG_ON(a)) st0 = __builtin_amdgcn_s_memtime();
        // the ticket of the NEXT task is drawn now and read after the body: its round trip hides behind the tile
        int tnext = 0, qn = 0, lenn = 0;
        if (tid == 0) {
            qn = s_q;
            lenn = s_visited < 8 ? (int)a.list[8 + qn] : 0;
            tnext = lenn > 0 ? atomicAdd(&a.sync[qn], 1) : 0;
        }
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (is_back) {
            // alpha = L^-T z of this matrix (launch_backsolve's kernels as one task: an HBM stream of the factor beside the
            // other workgroups' MFMA work); nobody in this launch reads alpha: no progress word to move
            back_task_body(a.g.C, a.g.F, b, a.nt, a.alpha + (long long)b * a.nt * GP_TS, smem, tid);
        } else if (is_diag) {
            // the pivot chains of a diagonal task are dependent fp64 operations that share the SIMD's datapath with the partner
            // workgroup's MFMA stream: at raised priority its instructions are issued first whenever they are ready
            if (GP_TASK_PRIO(a) == 3) __builtin_amdgcn_s_setprio(3);
            else if (GP_TASK_PRIO(a) == 1) __builtin_amdgcn_s_setprio(1);
            double* tile = tref_tile(a.g.C, b, k, k);
            double* invt = tref_tile(a.g.F, b, 0, k);
            if (k == 0) {
                diag_potrf_inv_la_body<WT>(smem, tile, invt, a.g.info + b, a.g.info_base, tid, false);
            } else {
                switch (wave) {
                    case 0: syrk_chain_wave<0, MT, WT>(a.g, b, k, k, smem, tid, lane); break;
                    case 1: syrk_chain_wave<1, MT, WT>(a.g, b, k, k, smem, tid, lane); break;
                    case 2: syrk_chain_wave<2, MT, WT>(a.g, b, k, k, smem, tid, lane); break;
                    default: syrk_chain_wave<3, MT, WT>(a.g, b, k, k, smem, tid, lane); break;
                }
                diag_potrf_inv_la_body<WT>(smem, tile, invt, a.g.info + b, a.g.info_base + GP_TS * k, tid, true);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        // a diagonal task with the row bit set goes on with strip(k + 1, k) and the augmented tile of its column — the two tiles
        // the NEXT diagonal task waits for — on the inverse it has just written (its own stores: drained, then a barrier); one
        // fetch, acquire and release for both
        const bool diag_then_strip = is_diag && rows > 1;
        if (diag_then_strip) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if ((!is_diag && !is_back) || diag_then_strip) {
            const int li = lane & 15, lg = lane >> 4;
            int loff[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = tid + 256 * u;
                loff[u] = (q >> 6) * LROW + (q & 63) * 2;
            }
            GemmArgs gl = a.g;
            gl.k1 = k;
            gl.fk = k;
            // `rows` consecutive tiles of the column, one after the other: they stream the same B panel, and the task's fetch,
            // acquire and release are paid once
            if (diag_then_strip)
                strip_item<FUSE_WD, (MT > 0), WT>(gl, b, k + 1, k, false, 8 * k, smem, smem + 2 * OPER_LDS, tid, lane, wave, li, lg,
                                              lg * LROW + 32 * wave + li, lg * LROW + li, loff, 0, true);
            else
                for (int r = 0; r < rows; ++r)
                    strip_item<FUSE_WD, (MT > 0), WT>(gl, b, i + r, k, /*no_update=*/MT > 0 && i >= a.nt, 8 * k, smem, smem + 2 * OPER_LDS, tid,
                                                  lane, wave, li, lg, lg * LROW + 32 * wave + li, lg * LROW + li, loff, 0,
                                                  with_aug && r == 0);
        }
        // publish: every wave's stores have left the CU, then one lane releases and moves the matrix's progress word(s)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (GP_DBG_ON(a)) st1 = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            // the next descriptor travels under the release fence
            unsigned dn = TASK_NONE;
            const bool hit = tnext < lenn;
            if (hit) dn = a.list[TASK_LIST_HDR + a.list[qn] + tnext];
            int* prog = a.sync + TASK_SYNC_HDR + (long long)b * TASK_SYNC_STRIDE;
            if (!is_back && !GP_TASK_WITHHOLD(a, b, is_diag, k)) {
                if (GP_FENCE_MODE(a) == 0 && !WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (is_diag) {
                    __hip_atomic_store(prog, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (diag_then_strip) {
                        __hip_atomic_store(prog + 2 + k, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (MT > 0) __hip_atomic_store(prog + 1 + a.nt, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else
                    for (int r = 0; r < rows; ++r)
                        __hip_atomic_store(prog + 1 + i + r, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (with_aug) __hip_atomic_store(prog + 1 + a.nt, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!hit) {                      // this queue has run dry (or had long before): go on with the others
                int q = qn, visited = s_visited;
                if (visited < 8) { q = (q + 1) & 7; ++visited; }
                dn = task_fetch(a, q, visited);
                s_q = q; s_visited = visited;
            }
            s_next = dn;
            if (GP_DBG_ON(a)) {     // measurement build: fetch start, body start, body end, published, descriptor, workgroup, ready
                unsigned long long* dd = a.dbg + 8 * (size_t)atomicAdd(a.sync + 9, 1);
                dd[0] = stf; dd[1] = st0; dd[2] = st1; dd[3] = __builtin_amdgcn_s_memtime(); dd[4] = d; dd[5] = blockIdx.x;
                dd[6] = s_ready; dd[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            }
        }
    }
}

template <int MT, bool WT>
static void launch_potrf_tasks_t(const PotrfTaskArgs& a, unsigned grid, hipStream_t st) {
    static DeviceOnce once;
    constexpr int bytes = DIAG3_LDS_BYTES > GEMM_LDS_BYTES ? DIAG3_LDS_BYTES : GEMM_LDS_BYTES;
    lds_opt_in(once, (const void*)potrf_tasks_kernel<MT, WT>, bytes);
    hipLaunchKernelGGL((potrf_tasks_kernel<MT, WT>), dim3(grid), dim3(256), bytes, st, a);
}

// mt: 16-row blocks (1 or 2) of the augmented right-hand-side row that ride with the diagonal tasks; 0: more than 32 right-hand
// sides (a level sweep: src/prediction.jl:24-33 runs ~100 levels per posterior sample) — the augmented row is an ordinary tile
// row of the task list, strip(nt, k) with its own column update over the live rows, as in the per-column launches
void launch_potrf_tasks(const PotrfTaskArgs& a, long long ntasks, int mt, hipStream_t st) {
    if (ntasks <= 0) return;
    int slots = 2 * device_cus();
#ifdef GPSLC_DIAG
    slots = diag_env("GPSLC_GEMM_SLOTS", slots);
#endif
    const unsigned grid = (unsigned)(ntasks < slots ? ntasks : slots);
#ifdef GPSLC_DIAG
    if (a.fence_mode & 2) {      // measurement build, GPSLC_TASK_FENCE bit 1: plain / nt payload stores + an agent-scope release fence
        if (mt == 0) launch_potrf_tasks_t<0, false>(a, grid, st);
        else if (mt <= 1) launch_potrf_tasks_t<1, false>(a, grid, st);
        else launch_potrf_tasks_t<2, false>(a, grid, st);
        return;
    }
#endif
    if (mt == 0) launch_potrf_tasks_t<0, true>(a, grid, st);
    else if (mt <= 1) launch_potrf_tasks_t<1, true>(a, grid, st);
    else launch_potrf_tasks_t<2, true>(a, grid, st);
}

// ---------------------------------------------------------------------------------------
// Diagonal tiles of a symmetric update:  C(t, t) -= sum_kk A(t, kk) A(t, kk)^T, lower triangle only.
// Work item = (diagonal tile t in [0, mi), batch element).  Wave w owns sub-tile rows w and 7 - w of the
// 8 x 8 grid of 16 x 16 sub-tiles: (w, 0..w) and (7 - w, 0..7 - w), 9 sub-tiles, 9 accumulators.  Both MFMA
// operands come from the one staged A slab.  The strictly-upper sub-tiles of C are never touched (the
// diagonal-block kernel reads the lower triangle only).
//
// MT > 0: the item also carries the augmented (right-hand-side) row of its column,
//     C(aug, t) -= sum_kk A(aug, kk) A(t, kk)^T      for the first 16 MT rows of the tile,
// because it streams exactly the operand that update needs (A(t, kk)): as work items of the general kernel the
// augmented-row tiles cost 3.9 % of its time (measured, GPSLC_GEMM_DIAG=3) for 0.1 % of the flops — each one
// streams two full operand panels for two live rows.  Wave w takes column blocks w and 7 - w of those rows.
// ---------------------------------------------------------------------------------------
#define DG_LDS_BYTES(MT) ((2 * OPER_LDS + ((MT) > 0 ? 2 * OPER_LDS : 0)) * 8)

template <int W, int MT>
__device__ __forceinline__ void syrk_diag_wave(const GemmArgs& g, double* lA, int tid, int lane) {
    double* lG = lA + 2 * OPER_LDS;      // augmented-row slabs (MT > 0)
    const long long Wk = (long long)g.mi * g.nbatch;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const long long wq = Wk >> 3, wrm = Wk & 7;
    const long long x0 = xcd * wq + (xcd < wrm ? xcd : wrm);
    const long long xc = wq + (xcd < wrm ? 1 : 0);
    int loff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = tid + 256 * u;
        loff[u] = (q >> 6) * LROW + (q & 63) * 2;
    }
    // rows [0, 16 MT) of an augmented slab: the chunks of this thread whose row pair is live
    const bool glive = MT > 0 && tid < 128 * MT;     // one row pair of the augmented slab per thread
    const int nslab = (g.k1 - g.k0) * (GP_TS / KS);
    const int fbase = (lane >> 4) * LROW + (lane & 15);

    for (long long it = local; it < xc; it += gx) {
        const long long item = x0 + it;
        const int b = (int)(item / g.mi);
        const int t = (int)(item - (long long)b * g.mi);
        const int ti = g.i0 + t;
        double* __restrict__ Cd = tref_tile(g.C, b, ti, ti) + ((lane >> 4) * GP_TS + (lane & 15));
        double* __restrict__ Cg = MT > 0 ? tref_tile(g.C, b, g.short_row0, ti) + ((lane >> 4) * GP_TS + (lane & 15)) : nullptr;

        d4 a0c[W + 1], a1c[8 - W];     // sub-tiles (W, cb) and (7 - W, cb)
#pragma unroll
        for (int cb = 0; cb <= W; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) a0c[cb][v] = Cd[(16 * cb + 4 * v) * GP_TS + 16 * W];
#pragma unroll
        for (int cb = 0; cb < 8 - W; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) a1c[cb][v] = Cd[(16 * cb + 4 * v) * GP_TS + 16 * (7 - W)];
        d4 ag[MT > 0 ? MT : 1][2];     // augmented rows: row block m, column blocks W (0) and 7 - W (1)
        if (MT > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    ag[m][0][v] = Cg[(16 * W + 4 * v) * GP_TS + 16 * m];
                    ag[m][1][v] = Cg[(16 * (7 - W) + 4 * v) * GP_TS + 16 * m];
                }
        }

        // staging as in syrk_chain_wave above: SYRK_PF register sets, LDS-only barriers
        d2 rs[SYRK_PF][4], rgs[MT > 0 ? SYRK_PF : 1];
        const int gcol = tid / (8 * (MT > 0 ? MT : 1)), grp = tid % (8 * (MT > 0 ? MT : 1));
        auto gload = [&](int s, int set) {
            const int kk = g.k0 + (s >> 3);
            const double* pa = tref_tile(g.A, b, ti, kk) + (s & 7) * (KS * GP_TS);
#pragma unroll
            for (int u = 0; u < 4; ++u) rs[set][u] = *reinterpret_cast<const d2*>(pa + (tid + 256 * u) * 2);
            if (MT > 0 && glive) {
                const double* pg = tref_tile(g.A, b, g.short_row0, kk) + (s & 7) * (KS * GP_TS);
                rgs[MT > 0 ? set : 0] = *reinterpret_cast<const d2*>(pg + gcol * GP_TS + 2 * grp);
            }
        };
        auto lstore = [&](int buf, int set) {
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<d2*>(lA + buf * OPER_LDS + loff[u]) = rs[set][u];
            if (MT > 0 && glive)
                *reinterpret_cast<d2*>(lG + buf * OPER_LDS + gcol * LROW + 2 * grp) = rgs[MT > 0 ? set : 0];
        };
        auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        auto compute = [&](int buf) {
            const double* pa = lA + buf * OPER_LDS + fbase;
            const double* pg = lG + buf * OPER_LDS + fbase;
#pragma unroll
            for (int ks = 0; ks < KS / 4; ++ks) {
                double bf[8 - W];
#pragma unroll
                for (int cb = 0; cb < 8 - W; ++cb) bf[cb] = pa[ks * 4 * LROW + 16 * cb];
                const double r0 = pa[ks * 4 * LROW + 16 * W];
                const double r1 = pa[ks * 4 * LROW + 16 * (7 - W)];
#pragma unroll
                for (int cb = 0; cb <= W; ++cb) a0c[cb] = mfma_step<1>(bf[cb], r0, a0c[cb]);
#pragma unroll
                for (int cb = 0; cb < 8 - W; ++cb) a1c[cb] = mfma_step<1>(bf[cb], r1, a1c[cb]);
                if (MT > 0) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const double gf = pg[ks * 4 * LROW + 16 * m];
                        ag[m][0] = mfma_step<1>(bf[W], gf, ag[m][0]);
                        ag[m][1] = mfma_step<1>(bf[7 - W], gf, ag[m][1]);
                    }
                }
            }
        };
        if (nslab > 0) {
#pragma unroll
            for (int u = 0; u < SYRK_PF; ++u) gload(u, u);          // nslab is a multiple of 8 >= SYRK_PF
            lstore(0, 0);
            lds_barrier();
            for (int s = 0; s < nslab; s += SYRK_PF) {
#pragma unroll
                for (int u = 0; u < SYRK_PF; ++u) {
                    if (s + u + SYRK_PF < nslab) gload(s + u + SYRK_PF, u);
                    compute(u & 1);
                    if (s + u + 1 < nslab) lstore((u + 1) & 1, (u + 1) % SYRK_PF);
                    lds_barrier();
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb <= W; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) Cd[(16 * cb + 4 * v) * GP_TS + 16 * W] = a0c[cb][v];
#pragma unroll
        for (int cb = 0; cb < 8 - W; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) Cd[(16 * cb + 4 * v) * GP_TS + 16 * (7 - W)] = a1c[cb][v];
        if (MT > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    Cg[(16 * W + 4 * v) * GP_TS + 16 * m] = ag[m][0][v];
                    Cg[(16 * (7 - W) + 4 * v) * GP_TS + 16 * m] = ag[m][1][v];
                }
        }
    }
}

template <int MT>
__global__ __launch_bounds__(256, 2) void tile_syrk_diag_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // every wave runs the same item loop and the same barriers; only the sub-tile rows differ
    switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
        case 0: syrk_diag_wave<0, MT>(g, smem, tid, lane); break;
        case 1: syrk_diag_wave<1, MT>(g, smem, tid, lane); break;
        case 2: syrk_diag_wave<2, MT>(g, smem, tid, lane); break;
        default: syrk_diag_wave<3, MT>(g, smem, tid, lane); break;
    }
}

template <int MT>
static void launch_syrk_diag_t(const GemmArgs& g, unsigned grid, hipStream_t st) {
    static DeviceOnce once;
    lds_opt_in(once, (const void*)tile_syrk_diag_kernel<MT>, DG_LDS_BYTES(MT));
    hipLaunchKernelGGL(tile_syrk_diag_kernel<MT>, dim3(grid), dim3(256), DG_LDS_BYTES(MT), st, g);
}

// g.mi = number of full-size diagonal tiles (i0 + t, i0 + t), t < mi; A == B by contract (sym).
// carry_aug: the items also update the augmented-row tiles (short_row0, i0 + t) (live rows g.short_rows).
void launch_syrk_diag(const GemmArgs& g, int carry_aug, hipStream_t st) {
    if (g.mi <= 0 || g.nbatch <= 0 || g.k1 <= g.k0) return;
    int slots = 2 * device_cus();
#ifdef GPSLC_DIAG
    slots = diag_env("GPSLC_GEMM_SLOTS", slots);
#endif
    const long long Wk = (long long)g.mi * g.nbatch;
    const unsigned grid = (unsigned)(Wk < slots ? Wk : slots);
    const int mt = carry_aug ? (g.short_rows + 15) / 16 : 0;   // callers pass carry_aug only for mt <= 2
    if (mt == 0) launch_syrk_diag_t<0>(g, grid, st);
    else if (mt == 1) launch_syrk_diag_t<1>(g, grid, st);
    else launch_syrk_diag_t<2>(g, grid, st);
}

template <int ACC, int DIAG>
static void launch_one(const GemmArgs& g, unsigned grid, hipStream_t st) {
    static DeviceOnce once;
    lds_opt_in(once, (const void*)tile_gemm_nt_kernel<ACC, DIAG>, GEMM_LDS_BYTES);
    hipLaunchKernelGGL((tile_gemm_nt_kernel<ACC, DIAG>), dim3(grid), dim3(256), GEMM_LDS_BYTES, st, g);
}

void launch_tile_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.ntiles <= 0 || g.nbatch <= 0) return;
    int slots = 2 * device_cus();   // 2 workgroups per CU (227 VGPRs, 72 KiB LDS each)
    const long long W = (long long)g.ntiles * g.nbatch;
#ifdef GPSLC_DIAG
    slots = diag_env("GPSLC_GEMM_SLOTS", slots);
    const int panel_slots = diag_env("GPSLC_PANEL_SLOTS", 0);
    if (!g.accumulate && panel_slots > 0) slots = panel_slots;
#endif
    const unsigned grid = (unsigned)(W < slots ? W : slots);
#ifdef GPSLC_DIAG
    if (g.diag_skip == 1) {        // timing-only diagnostics (GPSLC_GEMM_DIAG), separate instantiations
        if (g.accumulate) launch_one<1, 1>(g, grid, st); else launch_one<0, 1>(g, grid, st);
        return;
    }
    if (g.diag_skip == 2) {
        if (g.accumulate) launch_one<1, 2>(g, grid, st); else launch_one<0, 2>(g, grid, st);
        return;
    }
#endif
    if (g.fuse && g.accumulate) {
        static DeviceOnce once;
        lds_opt_in(once, (const void*)tile_fused_strip_kernel<FUSE_WD>, GEMM_LDS_BYTES);
        hipLaunchKernelGGL((tile_fused_strip_kernel<FUSE_WD>), dim3(grid), dim3(256), GEMM_LDS_BYTES, st, g);
    } else {
        if (g.accumulate) launch_one<1, 0>(g, grid, st); else launch_one<0, 0>(g, grid, st);
    }
}
