// Peak rate of v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_f64 against v_fma_f64 on this GPU (gfx950).
// hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_f64_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k_mfma16(double* out, int iters) {
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mfma4(double* out, int iters) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma(double* out, int iters) {
    double acc[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K>
double run(K k, double flops_per_thread_iter, int iters) {
    double* out;
    const int blocks = 256 * 8, threads = 256;
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
    return flops_per_thread_iter * iters * blocks * threads / (ms * 1e-3) / 1e12;
}
int main() {
    const int iters = 20000;
    // per wave64 instruction: 16x16x4 = 2048 flops -> 32 per lane; 4x4x4 (4 blocks) = 512 flops -> 8 per lane; fma 2 per lane
    printf("v_mfma_f64_16x16x4_f64 : %.1f TFLOP/s\n", run(k_mfma16, 8 * 32.0, iters));
    printf("v_mfma_f64_4x4x4_f64   : %.1f TFLOP/s\n", run(k_mfma4, 8 * 8.0, iters));
    printf("v_fma_f64              : %.1f TFLOP/s\n", run(k_fma, 8 * 2.0, iters));
    return 0;
}
