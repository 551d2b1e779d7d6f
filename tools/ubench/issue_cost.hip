// issue_cost.hip -- what one wave pays per instruction (cycles between two s_memtime stamps / instructions) for the
// instruction classes the conv_pf3 epilogue uses, at 1 and 2 waves per SIMD, one workgroup per CU.  Development aid.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int TEST> __global__ void __launch_bounds__(512) k(float *g, unsigned long long *out, int reps) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    float a[32], acc = 0.f;
    for (int i = 0; i < 32; ++i) a[i] = g[threadIdx.x + i * 512];
    f32x4 v4[8];
    for (int i = 0; i < 8; ++i) v4[i] = *reinterpret_cast<f32x4 *>(g + threadIdx.x * 4 + i * 4096);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (TEST == 0) {          // 32 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
        } else if constexpr (TEST == 1) {   // 16 v_pk_fma_f32 (32 values)
#pragma unroll
            for (int i = 0; i < 32; i += 2) { f32x2 v = {a[i], a[i + 1]}; v = __builtin_elementwise_fma(v, (f32x2){1.0001f, 1.0001f}, (f32x2){0.5f, 0.5f}); a[i] = v[0]; a[i + 1] = v[1]; }
        } else if constexpr (TEST == 2) {   // dependent chain of 32 v_fma
#pragma unroll
            for (int i = 0; i < 32; ++i) acc = __builtin_fmaf(acc, 1.0001f, a[i]);
        } else if constexpr (TEST == 3) {   // 32 ds_write_b32
#pragma unroll
            for (int i = 0; i < 32; ++i) lds[(threadIdx.x >> 6) * 2048 + i * 64 + lane] = a[i];
        } else if constexpr (TEST == 4) {   // 8 ds_read_b128 + use
#pragma unroll
            for (int i = 0; i < 8; ++i) v4[i] += *reinterpret_cast<f32x4 *>(lds + (threadIdx.x >> 6) * 2048 + i * 256 + lane * 4);
        } else if constexpr (TEST == 5) {   // 32 ds_read_b32
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] += lds[(threadIdx.x >> 6) * 2048 + i * 64 + lane];
        } else if constexpr (TEST == 6) {   // 8 global_store_dwordx4 (1 KiB each, own region)
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4 *>(g + ((size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + i) * 65536 + lane * 4 + (rep & 255) * 256) = v4[i];
        } else if constexpr (TEST == 7) {   // 8 global_load_dwordx4, consumed at the end of the batch
            f32x4 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = *reinterpret_cast<f32x4 *>(g + ((size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + i) * 65536 + lane * 4 + (rep & 255) * 256);
#pragma unroll
            for (int i = 0; i < 8; ++i) v4[i] += t[i];
        } else if constexpr (TEST == 8) {   // 32 v_max_f32
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] = fmaxf(a[i], 0.25f);
        } else if constexpr (TEST == 9) {   // 32 global_store_dword (128-byte rows)
#pragma unroll
            for (int i = 0; i < 32; ++i) g[((size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + i) * 65536 + lane + (rep & 255) * 64] = a[i];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = acc;
    for (int i = 0; i < 32; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += v4[i][0] + v4[i][3];
    if (s == 12345.678f) g[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int TEST> static void run(const char *name, int nops, float *g, unsigned long long *out, int threads) {
    const int reps = 200, wgs = 256;
    hipLaunchKernelGGL(k<TEST>, dim3(wgs), dim3(threads), 65536, 0, g, out, reps);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<TEST>, dim3(wgs), dim3(threads), 65536, 0, g, out, reps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(wgs * 8);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    double m = 0; int n = 0;
    for (int b = 0; b < wgs; ++b) for (int w = 0; w < threads / 64; ++w) { m += (double)h[b * 8 + w]; ++n; }
    printf("%-44s %d waves/CU: %7.1f cycles per instruction (per wave)\n", name, threads / 64, m / n / reps / nops);
}

int main() {
    float *g; unsigned long long *out;
    CK(hipMalloc(&g, (size_t)256 * 8 * 32 * 65536 * 4 + (1 << 24)));
    CK(hipMemset(g, 0, (size_t)256 * 8 * 32 * 65536 * 4 + (1 << 24)));
    CK(hipMalloc(&out, 256 * 8 * 8));
    for (int threads : {256, 512}) {
        run<0>("v_fma_f32 x32 independent", 32, g, out, threads);
        run<1>("v_pk_fma_f32 x16", 16, g, out, threads);
        run<2>("v_fma_f32 x32 dependent chain", 32, g, out, threads);
        run<8>("v_max_f32 x32", 32, g, out, threads);
        run<3>("ds_write_b32 x32", 32, g, out, threads);
        run<4>("ds_read_b128 x8 (+ use)", 8, g, out, threads);
        run<5>("ds_read_b32 x32 (+ use)", 32, g, out, threads);
        run<6>("global_store_dwordx4 x8", 8, g, out, threads);
        run<9>("global_store_dword x32", 32, g, out, threads);
        run<7>("global_load_dwordx4 x8 (+ use)", 8, g, out, threads);
    }
    return 0;
}
