// Micro-benchmark + layout check of v_mfma_f64_16x16x4_f64 on gfx950.
//  (1) verifies the operand / accumulator lane maps and the NEG-A use of the BLGP field that
//      k_tilegemm.hip relies on, against a host computation with asymmetric data;
//  (2) measures the sustained f64 MFMA rate (the guides list no f64 rate; BASELINE.md asks for it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

__global__ void layout_kernel(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/,
                              const double* C /*16x16 row-major*/, double* D0, double* D1) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];      // A[i = l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];     // B[k = l>>4][j = l&15]
    d4 c;
    for (int v = 0; v < 4; ++v) c[v] = C[((l >> 4) + 4 * v) * 16 + (l & 15)];   // row = (l>>4)+4v, col = l&15
    d4 d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    d4 d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
    for (int v = 0; v < 4; ++v) {
        D0[((l >> 4) + 4 * v) * 16 + (l & 15)] = d0[v];
        D1[((l >> 4) + 4 * v) * 16 + (l & 15)] = d1[v];
    }
}

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, double seed, unsigned long long* clk) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){seed, 0.0, seed, 1.0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    // 32 x NACC MFMAs per trip so that the compiler's accumulator shuffling at the loop back-edge
    // (v_accvgpr moves) is amortised to < 1 % of the issue slots
    for (int it = 0; it < iters; it += 32) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// sustained variant: operands with random mantissas (data-dependent switching power), launched back to back
// for > 1 s so that the power manager settles — the ceiling a long-running MFMA kernel can actually hold
__global__ __launch_bounds__(256) void rate_random_kernel(double* out, int iters, unsigned seed, unsigned long long* clk) {
    d4 acc[8];
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
    auto rnd = [&h]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5;
                        const unsigned lo = h * 2246822519u;
                        return __longlong_as_double(0x3FF0000000000000ll | ((long long)(h & 0xFFFFF) << 32) | lo) - 1.5; };
    for (int i = 0; i < 8; ++i) acc[i] = (d4){rnd(), rnd(), rnd(), rnd()};
    const double a = rnd(), b = rnd();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 32) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 1);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// plain vector FMA rate for comparison (spec: vector fp64 = matrix fp64 = 78.6 TF)
__global__ __launch_bounds__(256) void valu_kernel(double* out, int iters, double seed, unsigned long long* clk) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i + threadIdx.x * 1e-6;
    const double a = 1.0000001, b = 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    // ---- layout
    std::vector<double> A(64), B(64), C(256), D0(256), D1(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 0.5 + k * 7.25;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = -3 + j * 1.75 + k * k * 0.125;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) C[i * 16 + j] = 100.0 * i + j;
    double *dA, *dB, *dC, *dD0, *dD1;
    HC(hipMalloc(&dA, 64 * 8)); HC(hipMalloc(&dB, 64 * 8)); HC(hipMalloc(&dC, 256 * 8));
    HC(hipMalloc(&dD0, 256 * 8)); HC(hipMalloc(&dD1, 256 * 8));
    HC(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    HC(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    HC(hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD0, dD1);
    HC(hipDeviceSynchronize());
    HC(hipMemcpy(D0.data(), dD0, 256 * 8, hipMemcpyDeviceToHost));
    HC(hipMemcpy(D1.data(), dD1, 256 * 8, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ab = 0;
        for (int k = 0; k < 4; ++k) ab += A[i * 4 + k] * B[k * 16 + j];
        e0 = fmax(e0, fabs(D0[i * 16 + j] - (C[i * 16 + j] + ab)));
        e1 = fmax(e1, fabs(D1[i * 16 + j] - (C[i * 16 + j] - ab)));
    }
    printf("layout: max|D - (C + A*B)| = %.3e (blgp=0), max|D - (C - A*B)| = %.3e (blgp=1, NEG-A)  -> %s\n",
           e0, e1, (e0 < 1e-9 && e1 < 1e-9) ? "LAYOUT_OK" : "LAYOUT_MISMATCH");

    // ---- rate
    hipDeviceProp_t prop; HC(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* dout; HC(hipMalloc(&dout, (size_t)cus * 8 * 256 * 8));
    hipEvent_t e_a, e_b; HC(hipEventCreate(&e_a)); HC(hipEventCreate(&e_b));
    unsigned long long* dclk; HC(hipMalloc(&dclk, 16));
    unsigned long long hclk[2];
    for (int wpc = 1; wpc <= 4; wpc *= 2) {
        const int blocks = cus * wpc;   // 256-thread blocks: wpc waves per SIMD
        const int iters = 20480;
        for (int rep = 0; rep < 2; ++rep) {
            HC(hipEventRecord(e_a));
            hipLaunchKernelGGL(rate_kernel<8>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1.0 + rep, dclk);
            HC(hipEventRecord(e_b));
            HC(hipEventSynchronize(e_b));
            float ms; HC(hipEventElapsedTime(&ms, e_a, e_b));
            HC(hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)hclk[0] / (double)hclk[1] * 0.1;
            const double flop = (double)blocks * 4 /*waves*/ * iters * 8 * 2.0 * 16 * 16 * 4;
            const double cyc = (double)hclk[0] / ((double)iters * 8 * wpc);
            printf("mfma_f64_16x16x4: %d CUs, %d wave/SIMD, 8 acc: %.2f ms  %.2f TFLOP/s  in-kernel clock %.2f GHz, %.1f cycles/MFMA/SIMD\n",
                   cus, wpc, ms, flop / (ms * 1e-3) / 1e12, ghz, cyc);
        }
    }
    {   // sustained: 150 back-to-back launches at 2 waves/SIMD, the last 100 timed
        const int blocks = cus * 2, iters = 20480, total = 150, timed = 100;
        for (int l = 0; l < total; ++l) {
            if (l == total - timed) HC(hipEventRecord(e_a));
            hipLaunchKernelGGL(rate_random_kernel, dim3(blocks), dim3(256), 0, 0, dout, iters, 17u + l, dclk);
        }
        HC(hipEventRecord(e_b));
        HC(hipEventSynchronize(e_b));
        float ms; HC(hipEventElapsedTime(&ms, e_a, e_b));
        HC(hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost));
        const double flop = (double)timed * blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
        printf("mfma_f64_16x16x4 SUSTAINED (random operands, alternating NEG, %d launches = %.0f ms timed): %.2f TFLOP/s  "
               "in-kernel clock %.2f GHz (last launch)\n", timed, ms, flop / (ms * 1e-3) / 1e12,
               (double)hclk[0] / (double)hclk[1] * 0.1);
    }
    for (int sub = 0; sub < 2; ++sub) {   // a subset of the CUs: is the rate chip-power limited?
        const int blocks = sub == 0 ? 32 : 128;
        const int iters = 20480;
        HC(hipEventRecord(e_a));
        hipLaunchKernelGGL(rate_kernel<8>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1.5, dclk);
        HC(hipEventRecord(e_b));
        HC(hipEventSynchronize(e_b));
        float ms; HC(hipEventElapsedTime(&ms, e_a, e_b));
        HC(hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost));
        const double flop = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
        printf("mfma_f64_16x16x4: only %d blocks (1 wave/SIMD): %.2f ms %.2f TFLOP/s = %.3f TF per CU, clock %.2f GHz, %.1f cycles/MFMA\n",
               blocks, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / blocks,
               (double)hclk[0] / (double)hclk[1] * 0.1, (double)hclk[0] / ((double)iters * 8));
    }
    for (int wpc = 1; wpc <= 4; wpc *= 2) {
        const int blocks = cus * wpc;
        const int iters = 40000;
        HC(hipEventRecord(e_a));
        hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(256), 0, 0, dout, iters, 1.0, dclk);
        HC(hipEventRecord(e_b));
        HC(hipEventSynchronize(e_b));
        float ms; HC(hipEventElapsedTime(&ms, e_a, e_b));
        HC(hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost));
        const double flop = (double)blocks * 256 * iters * 16 * 2.0;
        printf("v_fma_f64: %d wave/SIMD: %.2f ms  %.2f TFLOP/s  clock %.2f GHz, %.2f cycles per wave-FMA per SIMD\n", wpc, ms,
               flop / (ms * 1e-3) / 1e12, (double)hclk[0] / (double)hclk[1] * 0.1,
               (double)hclk[0] / ((double)iters * 16 * wpc));
    }
    return 0;
}
