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

#ifdef GPSLC_DIAG
// measurement build: phase stamps of workgroup 0 of the last launch (tools/potrf_stamps.py reads them through the symbol below)
__device__ unsigned long long g_potrf_stamps[2 * 40];
extern "C" int gpslc_diag_potrf_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_potrf_stamps), sizeof(unsigned long long) * 2 * 40);
}
#endif
__global__ __launch_bounds__(256, 2) void diag_potrf_inv_la_kernel(TRef M, int k, double* inv,
                                                                    long long inv_bstride, int* info,
                                                                    int info_base) {
    extern __shared__ __attribute__((aligned(16))) double P[];   // 36 blocks x 256 + two W slots
    const long long b = blockIdx.x;
#ifdef GPSLC_DIAG
    unsigned long long* stamps = b == 0 ? g_potrf_stamps : nullptr;
#else
    unsigned long long* stamps = nullptr;
#endif
    diag_potrf_inv_la_body(P, tref_tile(M, b, k, k), inv + b * inv_bstride + (long long)k * GP_TSQ, info + b,
                           info_base + GP_TS * k, (int)threadIdx.x, false, stamps);
}

void launch_diag(const TRef& M, int k, double* inv, long long inv_bstride, int* info,
                 int info_base, int nbatch, hipStream_t st) {
    static DeviceOnce attr3;
    lds_opt_in(attr3, (const void*)diag_potrf_inv_la_kernel, DIAG3_LDS_BYTES);
    hipLaunchKernelGGL(diag_potrf_inv_la_kernel, dim3(nbatch), dim3(256), DIAG3_LDS_BYTES, st, M, k, inv,
                       inv_bstride, info, info_base);
}
