// Issue rates of the f64 vector instructions of contains_kernel's inner loop on gfx950: v_fma_f64, v_mul_f64, v_cmp_lt_f64
// (to an SGPR pair), v_add_f64, v_max_f64 -- wave-instructions per second over the whole chip, and the same as cycles per
// wave-instruction on one SIMD.  hipcc --offload-arch=gfx950 -O3 scripts/microbench/valu_f64_rates.hip -o /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X
__global__ void k_fma(double* out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void k_mul(double* out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0000001;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                     "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void k_add(double* out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                     "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void k_max(double* out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 3.5;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
                     "v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void k_cmp(double* out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, b = 3.5;
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_cmp_lt_f64 %0, %8, %12\n v_cmp_lt_f64 %1, %9, %12\n v_cmp_lt_f64 %2, %10, %12\n v_cmp_lt_f64 %3, %11, %12\n"
                     "v_cmp_lt_f64 %4, %8, %12\n v_cmp_lt_f64 %5, %9, %12\n v_cmp_lt_f64 %6, %10, %12\n v_cmp_lt_f64 %7, %11, %12\n"
                     : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3), "=s"(m4), "=s"(m5), "=s"(m6), "=s"(m7)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(m0 ^ m1 ^ m2 ^ m3 ^ m4 ^ m5 ^ m6 ^ m7);
}
__global__ void k_and32(double* out, int iters) {  // a 32-bit VALU op, for scale
    unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 0x7fffffffu;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                     "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <class K>
void run(const char* name, K k) {
    double* out;
    const int blocks = 256 * 8, threads = 256, iters = 20000;   // 8 wavefronts per SIMD
    hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    const double winst = 8.0 * iters * blocks * (threads / 64);       // wave-instructions
    const double per_simd = winst / (256.0 * 4.0);                      // per SIMD
    printf("%-14s %.3f ms  %.3g wave-instructions/s  = %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms,
           winst / (ms * 1e-3), ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    run("v_fma_f64", k_fma);
    run("v_mul_f64", k_mul);
    run("v_add_f64", k_add);
    run("v_max_f64", k_max);
    run("v_cmp_lt_f64", k_cmp);
    run("v_and_b32", k_and32);
    return 0;
}
