// 128 x 128 diagonal-tile Cholesky + inverse on a PACKED LDS image, as a device function: the diagonal-block kernel
// (diag_potrf_inv_la_kernel, k_diag.hip) runs it on a tile it loads itself, diag_update_potrf_kernel (k_tilegemm.hip) on the
// image its own diagonal-tile update has just left in LDS.
// Tried on it in round 4 and not kept (profiles/r04_ab_experiments.md §4): barriers that wait for LDS traffic only
// (__syncthreads() also waits for the acknowledgement of the row-p stores to HBM) and two trailing blocks per pass.
// Round 5 (profiles/r05_ab_experiments.md §6): one-block lookahead, factor + inverse of the 16 x 16 block in one sweep of column
// operations, output rows as flat index ranges — 177 k -> 140 k clocks per matrix in the stamps (cold workgroup), N = 512 +4.5 %,
// N = 1024 +2.1 %, N = 4096 +0.55 % on one box.  The loop without lookahead is profiles/r05_potrf_no_lookahead.patch.
#pragma once
#include "gpslc_internal.h"
#include "sm_blocks.h"
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));

#define NSB (GP_TS / SB)

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
// matrices (round 2; round 5: ~120 us).  What makes the packing possible:
//   * block row p of L is final after step p: it is written to the output tile right away and its LDS slots
//     are then overwritten, in place, by block row p of inv(L):
//         W_pq = -W_pp * sum_{m=q}^{p-1} L_pm W_mq        (q < p; needs rows < p of W only),
//     i.e. the inversion runs row-wise inside the factorisation loop instead of column-wise after it;
//   * W blocks are stored transposed (element (r, c) at r*16 + c) so that they are read as MFMA operands with
//     the conflict-free fragment pattern; slot (p, p) receives W_pp^T once L_pp has been written out.
// Summation orders of the block products are those of versions 1 and 2 (same MFMA chains).
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
// (p+1, p), applies it to the diagonal block (p+1, p+1) — the only update of step p that block needs — and factorises +
// inverts it (one sweep: sb_factor_inv) into the SECOND W slot, all in phase 1, while waves 1..3 do the rest of phase 1: the
// remaining panel blocks, block row p of inv(L), the output rows.  The trailing blocks of phase 2 are dealt to all four waves.
// Step p + 1 then starts with its diagonal block already factorised.  Arithmetic: every block receives the same MFMA chains in the same order as
// the loop without lookahead; the 16 x 16 factor and its inverse come from sb_factor_inv below (same algorithm class, different
// rounding in the last bits than rounds 1-4).  LDS: one more 2 KiB W slot and a 128-byte broadcast line.
// ---------------------------------------------------------------------------------------
#define DIAG3_LDS_BYTES ((36 * 256 + 512 + 16) * 8)

// wave-level: Cholesky of the 16 x 16 block at Dpp (column-major, ld 16) AND its inverse in ONE sweep of column operations
// (sm_factor_rows_lds, sm_blocks.h: the routine of the single-workgroup node kernels, ~6 k clocks per block where the
// readlane-only Cholesky + separate triangular inverse of rounds 1-4 took 19-23 k, in-kernel stamps of round 5).  Lanes 0-15 hold
// the block's rows (row i in lane i) and end up holding L_pp; lanes 16-31 hold the rows of the IDENTITY and end up holding
// I L_pp^-T — the column operations that turn A_pp into L_pp turn e_i' into row i of L_pp^-T, i.e. column i of W = L_pp^-1: the
// transposed image the MFMAs read.  Results: Dpp <- L_pp (exact diagonal, zeros above), Wdst[c' * 16 + c] = W[c][c'].
// bc: this wave's 16-double LDS line; bad: 1-based failing pivot of the tile (base = 16 p), wave-uniform.
__device__ __forceinline__ void sb_factor_inv(double* Dpp, double* Wdst, double* bc, int lane, int li, int base, int& bad) {
    double r[SB];
#pragma unroll
    for (int c = 0; c < SB; ++c) r[c] = (lane < SB) ? Dpp[c * SB + li] : ((lane < 2 * SB && c == li) ? 1.0 : 0.0);
    double lcc;
    sm_factor_rows_lds(r, li, lane, base, bad, lcc, bc);
    if (lane < SB) {
#pragma unroll
        for (int c = 0; c < SB; ++c) Dpp[c * SB + li] = (li > c) ? r[c] : (li == c ? lcc : 0.0);
    } else if (lane < 2 * SB) {
#pragma unroll
        for (int c = 0; c < SB; ++c) Wdst[li * SB + c] = r[c];
    }
}

// X = A W^T for one or two panel blocks, in place.  The two MFMA chains are independent and the code is branch-free up to the
// final stores (a lone block is computed twice: a wave-uniform `if` around every MFMA would put the two chains into separate
// basic blocks and serialise their LDS round trips — which is what the block updates of this kernel are bound by).
__device__ __forceinline__ void sb_panel2(double* A0, double* A1, const double* W, int lane, int li, int lg) {
    const double* B1 = A1 ? A1 : A0;
    d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0;
    double wf[4], f0[4], f1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { wf[kk] = bfrag(W, kk, lane); f0[kk] = bfrag(A0, kk, lane); f1[kk] = bfrag(B1, kk, lane); }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a0 = mma(wf[kk], f0[kk], a0);
        a1 = mma(wf[kk], f1[kk], a1);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) A0[(lg + 4 * v) * SB + li] = a0[v];
    if (A1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) A1[(lg + 4 * v) * SB + li] = a1[v];
    }
}
// A_ij -= X_i X_j^T for trailing block t of step p (t = ii (ii + 1) / 2 + rem over the blocks below / right of (p, p)), and for
// block t2 as well when t2 < nt_ (two independent accumulators: the LDS round trips and the 4-MFMA chains of the two overlap)
__device__ __forceinline__ void sb_trail2(double* P, int p, int t, int t2, int nt_, int lane, int li, int lg) {
    auto decode = [&](int tt, int& i, int& j) {
        int ii = 0, rem = tt;
        while (rem > ii) { rem -= ii + 1; ++ii; }
        i = p + 1 + ii; j = p + 1 + rem;
    };
    int i0, j0, i1, j1;
    decode(t, i0, j0);
    const bool two = t2 < nt_;
    decode(two ? t2 : t, i1, j1);                  // a lone block is computed twice (branch-free: see sb_panel2)
    double* A0 = BLK(i0, j0);
    const double* Xi0 = BLK(i0, p);
    const double* Xj0 = BLK(j0, p);
    double* A1 = BLK(i1, j1);
    const double* Xi1 = BLK(i1, p);
    const double* Xj1 = BLK(j1, p);
    d4 c0, c1;
    double fi0[4], fj0[4], fi1[4], fj1[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { c0[v] = A0[(lg + 4 * v) * SB + li]; c1[v] = A1[(lg + 4 * v) * SB + li]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        fi0[kk] = bfrag(Xi0, kk, lane); fj0[kk] = bfrag(Xj0, kk, lane);
        fi1[kk] = bfrag(Xi1, kk, lane); fj1[kk] = bfrag(Xj1, kk, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        c0 = mma_neg(fj0[kk], fi0[kk], c0);
        c1 = mma_neg(fj1[kk], fi1[kk], c1);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) A0[(lg + 4 * v) * SB + li] = c0[v];
    if (two) {
#pragma unroll
        for (int v = 0; v < 4; ++v) A1[(lg + 4 * v) * SB + li] = c1[v];
    }
}

// The barriers inside the loop order LDS traffic only: __syncthreads() carries a fence that also waits vmcnt(0), i.e. for the
// ACKNOWLEDGEMENT of the output-row stores of this step (22 per thread, to HBM, never read back by this kernel) — in-kernel stamps
// of round 5 showed waves 1..3 spending ~10 k clocks per step in phase 1 for two panel blocks and those stores.  (Round 4 tried
// the same barrier and saw nothing: the pivot chain, then serialised with everything else, hid it.)  The first barrier after
// the image load stays a full __syncthreads(): it waits for global LOADS.
#define DIAG_LDS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// stamps (measurement build only, else null): s_memtime of waves 0 and 1 at the phase boundaries of every step
#ifdef GPSLC_DIAG
#define DIAG_STAMP(slot) do { if (stamps && lane == 0 && wave < 2) stamps[(slot) * 2 + wave] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DIAG_STAMP(slot) do { } while (0)
#endif
// P: 36 packed blocks + TWO 256-double W slots + one 16-double broadcast line (DIAG3_LDS_BYTES of LDS); tile / invt: the output
// tiles (L_kk, inv(L_kk)) in HBM; image_ready: the lower blocks are already in P (the caller has NOT yet synchronised: the
// first barrier is in here); a non-positive pivot c (0-based) is reported as info_code0 + c + 1 through an atomicCAS on
// *info_word.  Called by all 256 threads of the workgroup (it contains barriers).
// WT (the persistent task launch, potrf_tasks_kernel): the output tiles are stored WRITE-THROUGH (agent-scope relaxed atomic
// stores = global_store ... sc1), so that another workgroup of the same launch can be handed them without an L2 write-back
template <bool WT>
__device__ __forceinline__ void out_store(double* p, double v) {
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool WT = false>
__device__ __forceinline__ void diag_potrf_inv_la_body(double* P, double* tile, double* invt, int* info_word,
                                                       int info_code0, int tid, bool image_ready,
                                                       unsigned long long* stamps = nullptr) {
    double* Wslot = P + 36 * 256;    // [2][256]
    double* bc = Wslot + 512;        // wave 0's broadcast line of the pivot chain
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
    // the strictly-upper 16 x 16 blocks of both output tiles are zeros: stored here, where they overlap the latency of the image
    // load (28 blocks x 2 tiles, one element per thread and block)
    for (int bi = 0; bi < NSB; ++bi)
        for (int bj = bi + 1; bj < NSB; ++bj) {
            out_store<WT>(tile + (SB * bj + ec) * GP_TS + SB * bi + er, 0.0);
            out_store<WT>(invt + (SB * bj + ec) * GP_TS + SB * bi + er, 0.0);
        }
    int bad = 0;
    DIAG_STAMP(0);
    __syncthreads();
    DIAG_STAMP(1);
    if (wave == 0) sb_factor_inv(BLK(0, 0), Wslot, bc, lane, li, 0, bad);       // block (0, 0): nothing to overlap it with
    DIAG_STAMP(2);
    DIAG_LDS_BARRIER;
    DIAG_STAMP(3);

    for (int p = 0; p < NSB; ++p) {
        double* Dpp = BLK(p, p);
        const double* Wcur = Wslot + (p & 1) * 256;          // inv(L_pp), transposed image
        double* Wnext = Wslot + ((p + 1) & 1) * 256;
        // ---- phase 1: reads of block row p (final L) and of the rows < p of W; writes to column p and to HBM
        d4 wq[3];                                // waves 1..3: their blocks of row p of inv(L) between the two phases
        if (wave == 0) {
            if (p + 1 < NSB) {
                // panel block (p+1, p), the one update the next diagonal block needs, then its factor and inverse
                double* Aip = BLK(p + 1, p);
                sb_panel2(Aip, nullptr, Wcur, lane, li, lg);
                // the wave reads back its own LDS stores (and, in the factorisation, its other lanes'): the LDS queue keeps a
                // wave's operations in order, the wavefront-scope fence (no instruction) makes the compiler honour that order
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                double* Ann = BLK(p + 1, p + 1);
                d4 ad;
#pragma unroll
                for (int v = 0; v < 4; ++v) ad[v] = Ann[(lg + 4 * v) * SB + li];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) ad = mma_neg(bfrag(Aip, kk, lane), bfrag(Aip, kk, lane), ad);
#pragma unroll
                for (int v = 0; v < 4; ++v) Ann[(lg + 4 * v) * SB + li] = ad[v];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                sb_factor_inv(Ann, Wnext, bc, lane, li, SB * (p + 1), bad);
            }
        } else {
            // (b) panel: X_i = A_i * W_pp^T for the sub-blocks below (p+1, p): at most two per wave, as one pair
            {
                const int ia = p + 2 + (wave - 1), ib = ia + 3;
                if (ia < NSB) sb_panel2(BLK(ia, p), ib < NSB ? BLK(ib, p) : nullptr, Wcur, lane, li, lg);
            }
            // (b') block row p of inv(L): W_pq = -W_pp sum_{m=q}^{p-1} L_pm W_mq for this wave's q = w-1, w+2, w+5 — the three
            // chains run interleaved over m (each still ascending in m) and share the fragments of L_pm; kept in registers
            // until phase 2
            {
                const int q0 = wave - 1;
                d4 accT[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) { accT[u] = (d4){0.0, 0.0, 0.0, 0.0}; wq[u] = accT[u]; }
                // m runs over [q0, p) in up to three segments with 1, 2 and 3 live chains: inside a segment the body is
                // branch-free, so the chains' LDS reads and MFMAs interleave
                auto seg = [&](int m0, int m1, auto nlive) {
                    constexpr int NLV = decltype(nlive)::value;
                    for (int m = m0; m < m1; ++m) {
                        const double* Lpm = BLK(p, m);
                        double lf[4], wf[NLV][4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) lf[kk] = bfrag(Lpm, kk, lane);
#pragma unroll
                        for (int u = 0; u < NLV; ++u)
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) wf[u][kk] = bfrag(BLK(m, q0 + 3 * u), kk, lane);   // transposed image
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int u = 0; u < NLV; ++u) accT[u] = mma(lf[kk], wf[u][kk], accT[u]);
                    }
                };
                seg(q0, min(q0 + 3, p), std::integral_constant<int, 1>());
                seg(q0 + 3, min(q0 + 6, p), std::integral_constant<int, 2>());
                seg(q0 + 6, p, std::integral_constant<int, 3>());
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int q = q0 + 3 * u;
                    if (q < p) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) wq[u] = mma_neg(accT[u][kk], bfrag(Wcur, kk, lane), wq[u]);
                        // W_pq[r = lane&15][c = (lane>>4)+4v] -> output tile of the inverse
#pragma unroll
                        for (int v = 0; v < 4; ++v) out_store<WT>(invt + (SB * q + lg + 4 * v) * GP_TS + SB * p + li, wq[u][v]);
                    }
                }
            }
            // block row p of the outputs: L (slots (p, 0..p)) and the diagonal block of inv(L) — p + 2 blocks of 256 elements as
            // ONE flat index range over the 192 threads (a wave's 64 consecutive indices lie in one block: 256 = 4 x 64).  The
            // zeros right of them were stored before the loop.  (Until round 5 this was a per-block loop with the zero blocks in
            // it, and in the lookahead arrangement wave 1 carried twice the iterations of waves 2 and 3: in-kernel stamps showed
            // ~10 k clocks per step in it — as long as the pivot chain it runs beside.)
            const int nel = (p + 2) * 256;
            for (int idx = t3; idx < nel; idx += 192) {
                const int jb = idx >> 8, e = idx & 255, rr = e & 15, cc = e >> 4;
                if (jb < p) out_store<WT>(tile + (SB * jb + cc) * GP_TS + SB * p + rr, BLK(p, jb)[e]);
                else if (jb == p) out_store<WT>(tile + (SB * p + cc) * GP_TS + SB * p + rr, (rr >= cc) ? Dpp[e] : 0.0);
                else out_store<WT>(invt + (SB * p + cc) * GP_TS + SB * p + rr, (rr >= cc) ? Wcur[e] : 0.0);
            }
        }
        DIAG_STAMP(4 + 4 * p);
        DIAG_LDS_BARRIER;
        DIAG_STAMP(5 + 4 * p);
        // ---- phase 2: block row p of W into its slots (transposed); the trailing update of the blocks (i, j > p) other than
        // (p+1, p+1), two blocks at a time, on all four waves (wave 0's chain ended in phase 1)
        if (wave != 0) {
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
        }
        {
            const int m = NSB - p - 1;
            const int nt_ = m * (m + 1) / 2;
            for (int t = 1 + wave; t < nt_; t += 8) sb_trail2(P, p, t, t + 4, nt_, lane, li, lg);   // t = 0: wave 0, phase 1
        }
        DIAG_STAMP(6 + 4 * p);
        DIAG_LDS_BARRIER;
        DIAG_STAMP(7 + 4 * p);
    }
    if (wave == 0 && lane == 0 && bad != 0) atomicCAS(info_word, 0, info_code0 + bad);
}
