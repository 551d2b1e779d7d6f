// Pricing Winograd F(2x2, 3x3) for the split-fp16 matrix-core arithmetic, speed half.
// The transformed-domain work of a 3x3 layer is 16 independent GEMMs (one per position of the 4x4 tile) with K = Cin: 2.25x fewer
// products than the 9 taps, but 16 accumulator sets per (cout, tile) block instead of one.  With fp32 accumulators in registers that
// bounds a wave's tile at 32 cout x 32 tiles (16 x 16 VGPRs, single accumulator set => three weight planes), so every three MFMAs
// fetch 3 A + 2 B operands from LDS (5 KB per 98 kflop; the 9-tap kernels: 64 x 64 wave tiles, 10 KB per 393 kflop).
// This file measures, on the real part:
//   main  : the 16-position MFMA loop of 128 -> 128 channels @128^2, batch 32 (131 072 Winograd tiles), operands from LDS only
//           (no DMA, no transforms, no epilogue): the FLOOR of a Winograd kernel's time, LDS-read-bound or matrix-bound;
//   xform : the VALU work of the input transform + two-plane split for the same layer (B^T d B on 4x4 patches from LDS, split2h,
//           planes written back to LDS), alone;
//   both  : the two in one kernel (per chunk: transform, then multiply; one / two waves per SIMD) -- do they overlap?
// against conv_pf3_kernel<2,2,2,2,.> on the same layer: 0.36 - 0.39 ms (profiles/per_op_r05*.txt).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wino_lab tools/ubench/wino_lab.hip && /tmp/wino_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split2h(float a, _Float16 &h, _Float16 &l) { h = (_Float16)a; l = (_Float16)((a - (float)h) * 2048.0f); }

// MODE 1 main only, 2 xform only, 3 both.  NPOS positions per wave: 16 (four waves per workgroup: 4 cout blocks) or 8 (eight waves:
// 4 cout blocks x 2 position halves, two waves per SIMD).  A workgroup owns 32 Winograd tiles x 128 cout; nchunk 16-channel chunks.
template <int MODE, int NPOS>
__global__ void __launch_bounds__(NPOS == 16 ? 256 : 512) wino_kernel(float *out, int nchunk, int n_wg_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    // LDS map: [A stage: 3 planes x 2 k-halves x 128 cout] x 2 positions (24 KB); [V: 16 positions x 2 planes x 2 k-halves x 32 tiles] (32 KB);
    // [input patches of a chunk: 2 planes x 2 k-halves x (4 x 34 px rows ... ) ~ 2 x 2 x 6 x 66] (25 KB)
    uint4 *As = smem, *Vs = smem + 2 * 768, *Xs = Vs + 16 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave & 3, ph = NPOS == 16 ? 0 : wave >> 2;
    const int n = lane & 31, kg = lane >> 5;
    for (int i = tid; i < 2 * 768 + 16 * 128 + 2 * 2 * 6 * 66; i += blockDim.x) {
        unsigned x = 0x9E3779B9u * (i + 1);
        smem[i] = make_uint4(0x3c003800u ^ (x & 0x03ff03ffu), 0x38003c00u ^ ((x >> 3) & 0x03ff03ffu), 0x3a003900u ^ ((x >> 5) & 0x03ff03ffu), 0x39003a00u ^ ((x >> 7) & 0x03ff03ffu));
    }
    __syncthreads();
    f16v acc[NPOS];
    for (int p = 0; p < NPOS; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float sink = 0.f;
    for (int tile = blockIdx.x; tile < n_wg_tiles; tile += gridDim.x) {
        for (int c = 0; c < nchunk; ++c) {
            if constexpr (MODE & 2) {
                // input transform of this chunk: 32 tiles x 16 channels x 16 positions = 8192 values by 256 (512) threads: thread = (tile, 8-channel
                // unit, half of the positions for 512 threads): reads the 4x4 patch (planes h, l' -> fp32), B^T d B, split, writes V planes
                const int t = tid & 31, ku = (tid >> 5) & 1, part = tid >> 6;         // part: which rows of the 4x4 result this thread produces
                const int nparts = blockDim.x >> 6;                                    // 4 or 8 threads per (tile, k-half)
                const int ty = t >> 4, tx = t & 15;                                    // 2 x 16 tiles of a 4 x 32-pixel output band
                // a thread produces ONE row pr of the 4x4 result (eight waves: one half of that row): it needs two rows of the patch
                const int pr = part & 3;
                const int ra = pr == 0 ? 0 : (pr == 1 ? 1 : (pr == 2 ? 2 : 1)), rb = pr == 0 ? 2 : (pr == 1 ? 2 : (pr == 2 ? 1 : 3));
                const float sgn = pr == 1 ? 1.f : -1.f;
                float r4[4][8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const h8 ha = __builtin_bit_cast(h8, Xs[(ku * 2 + 0) * 6 * 66 + (2 * ty + ra) * 66 + 2 * tx + j]);
                    const h8 la = __builtin_bit_cast(h8, Xs[(ku * 2 + 1) * 6 * 66 + (2 * ty + ra) * 66 + 2 * tx + j]);
                    const h8 hb = __builtin_bit_cast(h8, Xs[(ku * 2 + 0) * 6 * 66 + (2 * ty + rb) * 66 + 2 * tx + j]);
                    const h8 lb = __builtin_bit_cast(h8, Xs[(ku * 2 + 1) * 6 * 66 + (2 * ty + rb) * 66 + 2 * tx + j]);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        r4[j][q] = ((float)ha[q] + (float)la[q] * (1.0f / 2048.0f)) + sgn * ((float)hb[q] + (float)lb[q] * (1.0f / 2048.0f));
                }
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) {
                    if (nparts > 4 && (pc >> 1) != (part >> 2)) continue;            // eight waves: the column pairs are split as well
                    h8 vh, vl;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float v = pc == 0 ? r4[0][q] - r4[2][q] : (pc == 1 ? r4[1][q] + r4[2][q] : (pc == 2 ? r4[2][q] - r4[1][q] : r4[1][q] - r4[3][q]));
                        _Float16 x, y;
                        split2h(v, x, y);
                        vh[q] = x; vl[q] = y;
                    }
                    Vs[((pr * 4 + pc) * 2 + 0) * 64 + ku * 32 + t] = __builtin_bit_cast(uint4, vh);
                    Vs[((pr * 4 + pc) * 2 + 1) * 64 + ku * 32 + t] = __builtin_bit_cast(uint4, vl);
                }
                __syncthreads();
            }
            if constexpr (MODE & 1) {
#pragma unroll
                for (int p = 0; p < NPOS; ++p) {
                    const int pos = ph * NPOS + p;
                    const h8 ah = __builtin_bit_cast(h8, As[(pos & 1) * 768 + (0 * 2 + kg) * 128 + cb * 32 + n]);
                    const h8 al = __builtin_bit_cast(h8, As[(pos & 1) * 768 + (1 * 2 + kg) * 128 + cb * 32 + n]);
                    const h8 a2 = __builtin_bit_cast(h8, As[(pos & 1) * 768 + (2 * 2 + kg) * 128 + cb * 32 + n]);
                    const h8 bh = __builtin_bit_cast(h8, Vs[(pos * 2 + 0) * 64 + kg * 32 + n]);
                    const h8 bl = __builtin_bit_cast(h8, Vs[(pos * 2 + 1) * 64 + kg * 32 + n]);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, bl, acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[p], 0, 0, 0);
                }
            }
            if constexpr (MODE == 3) __syncthreads();
        }
        if constexpr (MODE & 1) {          // (an epilogue would transform and store here; keep the accumulators alive and bounded)
            for (int p = 0; p < NPOS; ++p) for (int r = 0; r < 16; ++r) { sink += acc[p][r]; acc[p][r] = 0.f; }
        }
    }
    if (sink == 12345.678f) out[tid] = sink;
}

template <int MODE, int NPOS> static void run(const char *name, int cus, float *d) {
    const int nchunk = 8, n_wg_tiles = 131072 / 32;           // 128 -> 128 @128^2, batch 32
    const size_t lds = (2 * 768 + 16 * 128 + 2 * 2 * 6 * 66) * 16;
    hipFuncSetAttribute((const void *)wino_kernel<MODE, NPOS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = NPOS == 16 ? 256 : 512;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int grid = cus * wgs_per_cu;
        hipLaunchKernelGGL((wino_kernel<MODE, NPOS>), dim3(grid), dim3(threads), lds, 0, d, nchunk, n_wg_tiles);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((wino_kernel<MODE, NPOS>), dim3(grid), dim3(threads), lds, 0, d, nchunk, n_wg_tiles);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double gf = 2.0 * 16 * 128 * 128 * 131072.0 * 3 / 1e9;    // executed fp16 flops of the 16 GEMMs, three products each
        printf("%-44s %d workgroup(s)/CU x %d waves  %7.3f ms%s\n", name, wgs_per_cu, threads / 64, best,
               (MODE & 1) ? "" : "   (no MFMA)");
        if (MODE & 1) printf("%-44s   = %.0f TFLOP/s of fp16 products (%.2f of 2500)\n", "", gf / best, gf / best / 2500.0);
    }
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float *d; hipMalloc(&d, 1 << 20);
    printf("%s, %d CUs; layer 128 -> 128 @128x128, batch 32: 131072 Winograd tiles, 206 GFLOP of fp16 products (direct form: 464)\n", pr.name, cus);
    run<1, 16>("main loop, 16 positions per wave", cus, d);
    run<1, 8>("main loop, 8 positions per wave", cus, d);
    run<2, 16>("input transform + split only (4 waves)", cus, d);
    run<2, 8>("input transform + split only (8 waves)", cus, d);
    run<3, 16>("transform, then multiply (4 waves)", cus, d);
    run<3, 8>("transform, then multiply (8 waves)", cus, d);
    return 0;
}
