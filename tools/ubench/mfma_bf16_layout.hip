// Probe: operand layout of v_mfma_f32_32x32x16_bf16 on gfx950 (asymmetric A, B) + rate check.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ unsigned short f2bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }
__global__ void k(const float *A, const float *B, float *D) {   // A[32][16], B[16][32], D[32][32]
    const int l = threadIdx.x, i = l & 31, kg = l >> 5;
    union { bf16x8 v; unsigned short s[8]; } a, b;
    for (int j = 0; j < 8; ++j) { a.s[j] = f2bf(A[i * 16 + kg * 8 + j]); b.s[j] = f2bf(B[(kg * 8 + j) * 32 + i]); }
    f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + i] = c[r];
}
template <int NACC> __global__ void __launch_bounds__(256) rate(float *out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    union { bf16x8 v; unsigned short s[8]; } a, b;
    for (int j = 0; j < 8; ++j) { a.s[j] = 0x3f80 + threadIdx.x % 7; b.s[j] = 0x3f00 + j; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[i], 0, 0, 0);
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float hA[512], hB[512], hD[1024], ref[1024];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5) * 0.5f; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
    printf("layout check (A[i=l&31][k=8*(l>>5)+j], B[k][n=l&31], D rows (r&3)+8(r>>2)+4(l>>5)): max err %g\n", err);
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        int grid = 256 * bpc, iters = 40000;
        hipLaunchKernelGGL(rate<8>, dim3(grid), dim3(256), 0, 0, out, 10); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(rate<8>, dim3(grid), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("bf16 32x32x16 rate, %d wave/SIMD: %.1f TF\n", bpc, (double)grid * 4 * iters * 8 * 2.0 * 32 * 32 * 16 / ms / 1e9);
    }
    return 0;
}
