// Do the f64 matrix pipe and the f64 vector pipe of gfx950 add up?  In every workgroup of 256 threads, `nm` of the four
// wavefronts run chains of v_mfma_f64_16x16x4_f64 and the others chains of v_fma_f64 (one wavefront of each kind per SIMD
// pair at nm = 2); the aggregate rate is compared with each kind alone.  If the two share their FP64 multipliers the mixed
// kernel cannot beat the better of the two.
// hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mix(double* out, int iters_m, int iters_v, int nm) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (((wave + (int)blockIdx.x) & 3) < nm) {   // (rotated per workgroup: every SIMD hosts wavefronts of both kinds)
        v4d acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int it = 0; it < iters_m; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
        for (int it = 0; it < iters_v; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
        for (int i = 0; i < 8; ++i) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* out;
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // iteration counts chosen so that each kind alone runs ~equally long (an MFMA iteration = 8 x 32 flops per lane,
    // a VALU iteration = 8 x 2): balance by the measured stand-alone rates
    const int it_m = 20000;
    for (int nm = 0; nm <= 4; ++nm) {
        for (int it_v : {0, it_m * 6, it_m * 12, it_m * 16}) {
            if (nm == 4 && it_v) continue;
            if (nm == 0 && !it_v) continue;
            hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(threads), 0, 0, out, 10, 10, nm);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(threads), 0, 0, out, it_m, it_v, nm);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double fm = 8 * 32.0 * it_m * (double)blocks * nm * 64, fv = 8 * 2.0 * (double)it_v * blocks * (4 - nm) * 64;
            printf("mfma waves %d/4 (%d iters)  valu waves %d/4 (%d iters): %.3f ms  mfma %.1f + valu %.1f = %.1f TFLOP/s\n", nm, it_m,
                   4 - nm, it_v, ms, fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
        }
    }
    hipFree(out);
    return 0;
}
