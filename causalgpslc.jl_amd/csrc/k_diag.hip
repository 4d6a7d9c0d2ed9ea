// Diagonal-block kernel of the blocked Cholesky: factor the 128 x 128 tile (k, k) in LDS
// (one workgroup per matrix of the batch) and produce inv(L_kk), so that the panel solve
// L(i,k) = A(i,k) * inv(L_kk)^T becomes one more f64-MFMA tile product (k_tilegemm.hip).
//
// Failure semantics mirror LAPACK potrf / PDMats: the first non-positive (or NaN) pivot is
// reported as info = info_base + 128*k + c + 1 (1-based) through an atomicCAS on the batch
// element's info word; the kernel always terminates (no data-dependent loops).
#include "gpslc_internal.h"

#define DLD 129   // LDS leading dimension (doubles): conflict-free both along and across columns

__global__ __launch_bounds__(256) void diag_potrf_inv_kernel(TRef M, int k, double* inv,
                                                             long long inv_bstride, int* info,
                                                             int info_base) {
    extern __shared__ __attribute__((aligned(16))) double S[];   // [128][DLD] + col[128] + dg[128]
    double* col = S + GP_TS * DLD;
    double* dg = col + GP_TS;
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    double* tile = tref_tile(M, b, k, k);

    for (int idx = tid; idx < GP_TSQ; idx += 256) S[(idx >> 7) * DLD + (idx & 127)] = tile[idx];

    const int r = tid & 127, h = tid >> 7;
    int bad = 0;
    for (int c = 0; c < GP_TS; ++c) {
        __syncthreads();
        const double d = S[c * DLD + c];
        if (!(d > 0.0) && bad == 0) bad = c + 1;
        const double s = sqrt(d);
        const double is = 1.0 / s;
        if (h == 0) {
            if (r > c) {
                const double v = S[c * DLD + r] * is;
                col[r] = v;
                S[c * DLD + r] = v;
            } else if (r == c) {
                dg[c] = s;
            }
        }
        __syncthreads();
        const double lr = (r > c) ? col[r] : 0.0;
        for (int cc = c + 1 + h; cc < GP_TS; cc += 2) {
            if (r >= cc) S[cc * DLD + r] -= lr * col[cc];
        }
    }
    __syncthreads();
    if (tid == 0 && bad != 0) atomicCAS(&info[b], 0, info_base + GP_TS * k + bad);

    // ---- inverse of L (lower triangular) by column-parallel forward substitution.
    // Thread j (< 128) owns column j of W = inv(L); W[m][j] (m > j) is kept at S[m*DLD + j], i.e. in
    // the strictly-upper part of the LDS image, which the factor does not use.
    if (tid < GP_TS) {
        const int j = tid;
        const double wjj = 1.0 / dg[j];
        for (int i = 1; i < GP_TS; ++i) {
            // acc = sum_{m=j}^{i-1} L[i][m] W[m][j]   (uniform loop, masked below j)
            double acc = 0.0;
            for (int m = 0; m < i; ++m) {
                const double lim = S[m * DLD + i];                     // L[i][m], broadcast
                const double wmj = (m > j) ? S[m * DLD + j] : (m == j ? wjj : 0.0);
                acc += lim * wmj;
            }
            if (i > j) S[i * DLD + j] = -acc / dg[i];
        }
    }
    __syncthreads();

    // ---- write back: factor (lower, zero strictly-upper) and inverse (lower, zero strictly-upper)
    double* invt = inv + b * inv_bstride + (long long)k * GP_TSQ;
    for (int idx = tid; idx < GP_TSQ; idx += 256) {
        const int c = idx >> 7, rr = idx & 127;
        double lv, wv;
        if (rr > c) { lv = S[c * DLD + rr]; wv = S[rr * DLD + c]; }
        else if (rr == c) { lv = dg[c]; wv = 1.0 / dg[c]; }
        else { lv = 0.0; wv = 0.0; }
        tile[idx] = lv;
        invt[idx] = wv;
    }
}

#define DIAG_LDS_BYTES ((GP_TS * DLD + 2 * GP_TS) * 8)

void launch_diag(const TRef& M, int k, double* inv, long long inv_bstride, int* info,
                 int info_base, int nbatch, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)diag_potrf_inv_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(diag_potrf_inv_kernel, dim3(nbatch), dim3(256), DIAG_LDS_BYTES, st, M, k, inv,
                       inv_bstride, info, info_base);
}
