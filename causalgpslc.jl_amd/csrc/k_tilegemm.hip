// f64-MFMA tile update:  C(i,j) (-)= sum_kk A(i,kk) * B(j,kk)^T   on 128 x 128 fp64 tiles.
//
// This one kernel carries every O(N^3) term of the path: the Cholesky panel multiply by the
// inverted diagonal block, the left-looking column update inside a panel, the trailing SYRK
// update, the "D L^-T" solve of the ITE covariance and its SYRK.
//
// CDNA4 mapping
//   * one workgroup = 4 wave64 = one 128 x 128 output tile; each wave owns a 64 x 64 quadrant as
//     4 x 4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs), 2 workgroups per CU;
//   * K is streamed in slabs of 16 tile columns (16 KiB contiguous per operand, tiles are
//     column-major), global_load_dwordx4 -> registers -> ds_write_b128 one slab ahead of the MFMAs
//     (issue-early / write-late), double-buffered in LDS, one barrier per slab;
//   * LDS rows are padded to 144 doubles so the two k-rows a 32-lane group reads with ds_read_b64
//     fall in disjoint halves of the 64-bank row (conflict-free);
//   * operands are swapped (MFMA "A" = B tile rows, MFMA "B" = A tile rows) so that lane&15 runs
//     along the contiguous row index of the column-major C tile: every C load/store instruction
//     touches four full 128-byte lines;
//   * C is pre-loaded into the accumulators and the subtraction is done by the MFMA's NEG-A modifier
//     (the BLGP field of v_mfma_f64), so the epilogue is stores only;
//   * blockIdx is remapped so that each XCD (private 4 MiB L2) works through a contiguous run of
//     output tiles, i.e. neighbouring tiles that share operand tile-rows hit the same L2.
#include "gpslc_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define KS 16
#define LROW 144                      // padded k-row (doubles)
#define OPER_LDS (KS * LROW)          // doubles per operand per stage
#define GEMM_LDS_BYTES (2 * 2 * OPER_LDS * 8)

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

template <int ACC>
__global__ __launch_bounds__(256, 2) void tile_gemm_nt_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;

    // ---- XCD-aware, bijective block remap: XCD x gets the x-th contiguous run of work items
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int b = bid / g.ntiles;
    const int t = bid - b * g.ntiles;
    int ii, jj;
    if (g.shape == 0) tri_decode(t, ii, jj);
    else { ii = t / g.mj; jj = t - ii * g.mj; }
    const int ti = g.i0 + ii, tj = g.j0 + jj;

    double* __restrict__ Ct = tref_tile(g.C, b, ti, tj);

    // ---- accumulators: acc[m][n][v] = C[wr*64 + 16m + (lane&15)][wc*64 + 16n + (lane>>4) + 4v]
    d4 acc[4][4];
    const int crow = wr * 64 + (lane & 15);
    const int ccol = wc * 64 + (lane >> 4);
    if (ACC) {
        const double* __restrict__ Cl = Ct + (ccol * GP_TS + crow);
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m][n][v] = Cl[(16 * n + 4 * v) * GP_TS + 16 * m];
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
    }

    const int nslab = (g.k1 - g.k0) * (GP_TS / KS);
    if (nslab > 0) {
        // staging: 16 KiB per operand per slab = 1024 16-byte chunks, 4 per thread, contiguous in HBM
        d4* sA = reinterpret_cast<d4*>(smem);   // viewed per 16 B only for address arithmetic
        (void)sA;
        double* lA = smem;                       // [2][KS][LROW]
        double* lB = smem + 2 * OPER_LDS;        // [2][KS][LROW]
        int loff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + 256 * u;
            loff[u] = (q >> 6) * LROW + (q & 63) * 2;
        }
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 ra[4], rb[4];

        auto slab_ptrs = [&](int s, const double*& pa, const double*& pb) {
            const int kk = g.k0 + (s >> 3);
            const int so = (s & 7) * (KS * GP_TS);
            pa = tref_tile(g.A, b, ti, kk) + so;
            pb = tref_tile(g.B, b, tj, kk) + so;
        };
        auto gload = [&](int s) {
            const double *pa, *pb;
            slab_ptrs(s, pa, pb);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ra[u] = *reinterpret_cast<const d2*>(pa + (tid + 256 * u) * 2);
                rb[u] = *reinterpret_cast<const d2*>(pb + (tid + 256 * u) * 2);
            }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                *reinterpret_cast<d2*>(lA + buf * OPER_LDS + loff[u]) = ra[u];
                *reinterpret_cast<d2*>(lB + buf * OPER_LDS + loff[u]) = rb[u];
            }
        };

        gload(0);
        lstore(0);
        __syncthreads();

        const int frow_a = (lane >> 4) * LROW + wr * 64 + (lane & 15);
        const int frow_b = (lane >> 4) * LROW + wc * 64 + (lane & 15);

        for (int s = 0; s < nslab; ++s) {
            const int buf = s & 1;
            if (s + 1 < nslab) gload(s + 1);
            const double* pa = lA + buf * OPER_LDS + frow_a;
            const double* pb = lB + buf * OPER_LDS + frow_b;
#pragma unroll
            for (int ks = 0; ks < KS / 4; ++ks) {
                double af[4], bf[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = pa[ks * 4 * LROW + 16 * m];
#pragma unroll
                for (int n = 0; n < 4; ++n) bf[n] = pb[ks * 4 * LROW + 16 * n];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = mfma_step<ACC>(bf[n], af[m], acc[m][n]);
            }
            if (s + 1 < nslab) lstore(buf ^ 1);
            __syncthreads();
        }
    }

    // recompute the store addresses from one opaque offset instead of keeping the 64 preload
    // addresses alive (and spilled) across the K loop
    int soff = ccol * GP_TS + crow;
    asm volatile("" : "+v"(soff));
    double* __restrict__ Cs = Ct + soff;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                Cs[(16 * n + 4 * v) * GP_TS + 16 * m] = acc[m][n][v];
}

void launch_tile_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.ntiles <= 0 || g.nbatch <= 0) return;
    const long long grid = (long long)g.ntiles * g.nbatch;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tile_gemm_nt_kernel<1>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)tile_gemm_nt_kernel<0>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        attr_set = true;
    }
    if (g.accumulate)
        hipLaunchKernelGGL(tile_gemm_nt_kernel<1>, dim3((unsigned)grid), dim3(256), GEMM_LDS_BYTES, st, g);
    else
        hipLaunchKernelGGL(tile_gemm_nt_kernel<0>, dim3((unsigned)grid), dim3(256), GEMM_LDS_BYTES, st, g);
}
