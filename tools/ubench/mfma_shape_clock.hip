// Development aid (round 4, VERDICT r3 item 8 i): does the MFMA *shape* change what the power management gives back?
// Register-only loops of v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 (same flops per cycle: 1024 per SIMD), and of
// the bf16 forms, at 1 / 2 waves per SIMD; reports sustained TFLOP/s (hipEvents) = 2.5 PFLOP/s x (clock / 2.4 GHz) x issue density.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_clock tools/ubench/mfma_shape_clock.hip && /tmp/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// SHAPE 4: 32x32x16 f16 with EIGHT operand pairs of pseudo-random values cycled through (the multiplier inputs toggle on every
// instruction, as in a real kernel; SHAPE 0 feeds the same registers every time)
template <int SHAPE>   // 0: 32x32x16 f16, 1: 16x16x32 f16, 2: 32x32x16 bf16, 3: 16x16x32 bf16
__global__ void __launch_bounds__(256) mfma_loop(float *out, int iters) {
    h8 a, b; b8 ab, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); ab[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i]; }
    if constexpr (SHAPE == 5 || SHAPE == 6 || SHAPE == 7) {
        // operand reuse between consecutive instructions, random values: 5 = A held for four instructions while B walks four registers
        // (a register-blocked GEMM's natural order); 6 = snake over a 2 x 2 block (one operand changes per instruction);
        // 7 = as 4 (both change every instruction) but out of only TWO registers per operand
        h8 ra[4], rb[4];
        unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
        for (int p = 0; p < 4; ++p)
            for (int i = 0; i < 8; ++i) {
                x = x * 1664525u + 1013904223u; ra[p][i] = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 0.001f);
                x = x * 1664525u + 1013904223u; rb[p][i] = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 0.001f);
            }
        f16v acc[4];
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (SHAPE == 5) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[u], rb[k], acc[k], 0, 0, 0);
                    else if constexpr (SHAPE == 6) {
                        constexpr int ai[4] = {0, 0, 1, 1}, bi[4] = {0, 1, 1, 0};
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[2 * (u & 1) + ai[k]], rb[2 * (u >> 1) + bi[k]], acc[k], 0, 0, 0);
                    } else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[k & 1], rb[(k + u) & 1], acc[k], 0, 0, 0);
                }
        }
        float s = 0.f;
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
        if (s == 12345.f) out[threadIdx.x] = s;
        return;
    }
    if constexpr (SHAPE == 4) {
        h8 ra[8], rb[8];
        unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
        for (int p = 0; p < 8; ++p)
            for (int i = 0; i < 8; ++i) {
                x = x * 1664525u + 1013904223u; ra[p][i] = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 0.001f);
                x = x * 1664525u + 1013904223u; rb[p][i] = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 0.001f);
            }
        f16v acc[4];
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[(4 * u + k) & 7], rb[(4 * u + k + 3) & 7], acc[k], 0, 0, 0);
        }
        float s = 0.f;
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
        if (s == 12345.f) out[threadIdx.x] = s;
        return;
    }
    if constexpr (SHAPE == 0 || SHAPE == 2) {
        f16v acc[4];
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (SHAPE == 0) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
                    else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[k], 0, 0, 0);
                }
        }
        float s = 0.f;
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
        if (s == 12345.f) out[threadIdx.x] = s;
    } else {
        f4v acc[8];
        for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if constexpr (SHAPE == 1) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
                    else acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[k], 0, 0, 0);
                }
        }
        float s = 0.f;
        for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) s += acc[k][r];
        if (s == 12345.f) out[threadIdx.x] = s;
    }
}

template <int SHAPE> static void run(const char *name, int wgs_per_cu, float *d, int cus) {
    const int iters = 20000;
    const double flop_per_wave_iter = (SHAPE % 2 == 0) ? 16.0 * 32768.0 : 32.0 * 16384.0;    // 16 x 32x32x16 or 32 x 16x16x32 per iteration
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * wgs_per_cu;
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(grid), dim3(256), 0, 0, d, 200);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(grid), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double tf = flop_per_wave_iter * iters * 4.0 * grid / (ms * 1e-3) / 1e12;
        printf("%-22s %d wave(s)/SIMD  %8.2f ms  %7.1f TFLOP/s  (= clock %.2f GHz if the pipe never idles)\n", name, wgs_per_cu, ms, tf, tf / 2500.0 * 2.4);
    }
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float *d; hipMalloc(&d, 4096);
    printf("%s, %d CUs\n", pr.name, cus);
    for (int w : {1, 2}) {
        run<0>("f16  32x32x16", w, d, cus);
        run<1>("f16  16x16x32", w, d, cus);
        run<2>("bf16 32x32x16", w, d, cus);
        run<3>("bf16 16x16x32", w, d, cus);
        run<4>("f16  32x32x16 random", w, d, cus);
        run<5>("f16 random, A held x4", w, d, cus);
        run<6>("f16 random, snake 2x2", w, d, cus);
        run<7>("f16 random, 2 regs", w, d, cus);
    }
    return 0;
}
