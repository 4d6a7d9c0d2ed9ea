// Do fp64 MFMA and fp64 VALU instructions of two waves on the same SIMD execute concurrently on gfx950?
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run a register-only chain-free stream of
// v_mfma_f64_16x16x4_f64, waves 4-7 (the second wave of every SIMD) a stream of independent v_fma_f64.
// Modes: 0 = MFMA waves only (the others exit), 1 = VALU waves only, 2 = both.  If the two pipes were independent,
// mode 2 would take max(t0, t1); if they share the fp64 datapath, about t0 + t1.
// build: hipcc -O3 --offload-arch=gfx950 coexec_f64_bench.hip -o coexec_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void coexec_kernel(int mode, int iters, double* out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave < 4) {
        if (mode == 1) return;
        d4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
        const double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);   // 8 independent accumulators
        }
        double s = 0.0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 123.456) out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (mode == 0) return;
        double x[16];
        for (int i = 0; i < 16; ++i) x[i] = 1.0 + (lane + i) * 1e-9;
        const double m = 1.0 - 1e-12, c = 1e-13;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)                 // 8 x 16 = 128 independent-enough FMAs per iteration: 128 x 4 cycles = the 8 MFMAs' 512
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fma(x[i], m, c);
        }
        double s = 0.0;
        for (int i = 0; i < 16; ++i) s += x[i];
        if (s == 123.456) out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

int main() {
    double* out;
    (void)hipMalloc(&out, 1024 * 512 * sizeof(double));
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 200000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(coexec_kernel, dim3(cus), dim3(512), 0, 0, mode, iters, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double mfma_tf = mode != 1 ? (double)cus * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12 : 0.0;
            const double valu_tf = mode != 0 ? (double)cus * 4 * iters * 128 * 64 * 2.0 / (ms * 1e-3) / 1e12 : 0.0;
            printf("mode %d (%s): %.2f ms  MFMA %.1f TFLOP/s  VALU %.1f TFLOP/s  sum %.1f\n", mode,
                   mode == 0 ? "MFMA waves only" : mode == 1 ? "VALU waves only" : "both", ms, mfma_tf, valu_tf, mfma_tf + valu_tf);
        }
    return 0;
}
