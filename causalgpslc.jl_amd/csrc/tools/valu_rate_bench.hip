// Issue-rate micro-benchmark for the fp64 VALU instructions of the fused RBF kernels (gfx950).
// Each kernel runs 8 independent dependency chains of one instruction; cycles = s_memtime delta of
// one wave / (iterations x 8 x waves per SIMD).  Build: make tools/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double seed, unsigned long long* clk) {
    double x[8];
    int e[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed + i * 0.37 + threadIdx.x * 1e-3; e[i] = (threadIdx.x + i) & 3; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = __builtin_fma(x[i], 0.999999, 1e-7);
            else if (OP == 1) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[i]));
            else if (OP == 2) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(e[i] - 1));
            else if (OP == 3) { int r; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x[i])); e[i] ^= r; }
            else if (OP == 4) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
            else if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
            else if (OP == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
            else if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(e[i]) : "v"(it));
            else if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(e[i]) : "v"(e[(i + 1) & 7]));
            else if (OP == 9) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + e[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int OP>
int run(const char* name, double* dout, unsigned long long* dclk, int cus) {
    const int iters = 20000, wps = 4;
    hipLaunchKernelGGL(rate<OP>, dim3(cus * wps), dim3(256), 0, 0, dout, iters, 1.25, dclk);
    HC(hipDeviceSynchronize());
    hipLaunchKernelGGL(rate<OP>, dim3(cus * wps), dim3(256), 0, 0, dout, iters, 1.25, dclk);
    HC(hipDeviceSynchronize());
    unsigned long long h;
    HC(hipMemcpy(&h, dclk, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz on this part; report relative to v_fma_f64 instead of absolute cycles
    printf("%-16s %10.1f memtime ticks per 1000 wave-instructions per SIMD\n", name, 1000.0 * (double)h / ((double)iters * 8 * wps));
    return 0;
}

int main() {
    hipDeviceProp_t prop; HC(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* dout; HC(hipMalloc(&dout, (size_t)cus * 4 * 256 * 8));
    unsigned long long* dclk; HC(hipMalloc(&dclk, 16));
    run<0>("v_fma_f64", dout, dclk, cus);
    run<5>("v_mul_f64", dout, dclk, cus);
    run<6>("v_add_f64", dout, dclk, cus);
    run<4>("v_max_f64", dout, dclk, cus);
    run<1>("v_rndne_f64", dout, dclk, cus);
    run<2>("v_ldexp_f64", dout, dclk, cus);
    run<3>("v_cvt_i32_f64", dout, dclk, cus);
    run<9>("v_rcp_f64", dout, dclk, cus);
    run<7>("v_cndmask_b32", dout, dclk, cus);
    run<8>("v_mov_b32", dout, dclk, cus);
    return 0;
}
