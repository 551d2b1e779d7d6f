// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on MI355X vs waves/SIMD, accumulators, duration.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, int iters) {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
    printf("NACC=%d waves/SIMD=%d iters=%d: %.3f ms  %.1f TF\n", NACC, blocks_per_cu, iters, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(1, 2000); run<4>(1, 20000); run<8>(1, 10000); run<8>(2, 5000); run<8>(2, 20000); run<4>(2, 20000);
    }
    return 0;
}
