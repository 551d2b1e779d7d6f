// pf_lab.hip -- development bench for the DMA-fed convolution kernels (conv_pf_kernel.h and successors) on one
// layer shape, outside the library: times kernel variants back to back on the same random operands, checks them
// against each other, and runs the compile-time ablations.  Not shipped, not part of the library.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCDC_PF_ABLATE=1 -I cdc_compression_amd/csrc -I include \
//                tools/ubench/pf_lab.hip -o tools/ubench/pf_lab
//   run:   pf_lab [B=32] [C=64] [H=256] [reps=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <string>

#include "cdc_internal.h"
#include "conv_pf_kernel.h"
#include "pfpp_lab.h"
#ifdef LAB_PF3
#include "conv_pf3_kernel.h"
#endif

using namespace cdc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float urand(unsigned long long i, unsigned seed) {   // (-1, 1)
    return (float)(int)(hash32((unsigned)i * 2654435761u + seed + (unsigned)(i >> 32) * 97u) >> 8) * (1.0f / 8388608.0f) - 1.0f;
}
// PF tensor [B][C/8][2][H+2][W+2] units: interior = split2h of a random activation, halo zero
__global__ void fill_pf(uint4 *pf, int C, int H, int W, unsigned seed) {
    const int b = blockIdx.y;
    const long long n = (long long)(C / 8) * (H + 2) * (W + 2);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % (W + 2));
    const long long t = i / (W + 2);
    const int y = (int)(t % (H + 2)), g = (int)(t / (H + 2));
    f16x8 hv, lv;
    const bool in = x >= 1 && x <= W && y >= 1 && y <= H;
    for (int q = 0; q < 8; ++q) {
        _Float16 h = 0, l = 0;
        if (in) split2h(1.5f * urand(((long long)(b * C + g * 8 + q) * H + (y - 1)) * W + (x - 1), seed), h, l);
        hv[q] = h; lv[q] = l;
    }
    const long long ps = (long long)(H + 2) * (W + 2);
    uint4 *dp = pf + (long long)b * (C / 8) * 2 * ps + (long long)g * 2 * ps + (long long)y * (W + 2) + x;
    dp[0] = __builtin_bit_cast(uint4, hv);
    dp[ps] = __builtin_bit_cast(uint4, lv);
}
__global__ void fill_f32(float *p, long long n, unsigned seed, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = scale * urand(i, seed);
}
// weight planes [tap][Cin/16][6 rows = plane*2 + khalf][COP] units of 8 cin: w 2^s = WH + WL, WH2 = WH 2^-11
__global__ void fill_w(uint4 *w, int taps, int nchunk, int COP, unsigned seed) {
    const long long n = (long long)taps * nchunk * 2 * COP;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % COP);
    long long t = i / COP;
    const int kh = (int)(t % 2); t /= 2;
    const int ch = (int)(t % nchunk);
    const int tap = (int)(t / nchunk);
    f16x8 wh, wl, wh2;
    for (int q = 0; q < 8; ++q) {
        const float v = 8192.f * urand(i * 8 + q, seed);
        const _Float16 h = (_Float16)v;
        wh[q] = h; wl[q] = (_Float16)(v - (float)h); wh2[q] = (_Float16)((float)h * (1.0f / 2048.0f));
    }
    uint4 *base = w + ((long long)(tap * nchunk + ch) * 6) * COP + co;
    base[(0 + kh) * (long long)COP] = __builtin_bit_cast(uint4, wh);
    base[(2 + kh) * (long long)COP] = __builtin_bit_cast(uint4, wl);
    base[(4 + kh) * (long long)COP] = __builtin_bit_cast(uint4, wh2);
}
__global__ void diff_kernel(const float *a, const float *b, long long n, float *out) {   // out[0] = max |a-b|, out[1] = max |a|
    float m = 0.f, ma = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = fabsf(a[i] - b[i]);
        m = fmaxf(m, d != d ? 1e30f : d); ma = fmaxf(ma, fabsf(a[i]));
    }
    atomicMax(reinterpret_cast<int *>(out), __float_as_int(m));
    atomicMax(reinterpret_cast<int *>(out) + 1, __float_as_int(ma));
}
__global__ void diff_where(const float *a, const float *b, long long n, float tol, int *cnt, long long *idx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = fabsf(a[i] - b[i]);
        if (!(d <= tol)) { const int k = atomicAdd(cnt, 1); if (k < 64) idx[k] = i; }
    }
}
// PF tensors compared by VALUE: h + l' 2^-11 per element (halo included: it must stay zero in both)
__global__ void diff_pf(const _Float16 *a, const _Float16 *b, long long nunits_per_plane_pair, long long ps, float *out) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nunits_per_plane_pair * 8; i += (long long)gridDim.x * 256) {
        const long long pair = i / (ps * 8), rem = i - pair * ps * 8;          // (group, position, channel-in-unit)
        const long long ih = (pair * 2 * ps) * 8 + rem, il = ih + ps * 8;
        const float va = (float)a[ih] + (float)a[il] * (1.0f / 2048.0f), vb = (float)b[ih] + (float)b[il] * (1.0f / 2048.0f);
        const float d = fabsf(va - vb);
        m = fmaxf(m, d != d ? 1e30f : d);
    }
    atomicMax(reinterpret_cast<int *>(out), __float_as_int(m));
}
__global__ void diff_pf_where(const _Float16 *a, const _Float16 *b, long long n8, long long ps, int *cnt, float *rec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const long long pair = i / (ps * 8), rem = i - pair * ps * 8;
        const long long ih = (pair * 2 * ps) * 8 + rem, il = ih + ps * 8;
        const float va = (float)a[ih] + (float)a[il] * (1.0f / 2048.0f), vb = (float)b[ih] + (float)b[il] * (1.0f / 2048.0f);
        if (!(fabsf(va - vb) <= 1e-3f)) {
            const int k = atomicAdd(cnt, 1);
            if (k < 12) { rec[k * 8] = (float)pair; rec[k * 8 + 1] = (float)(rem / 8); rec[k * 8 + 2] = (float)(rem % 8); rec[k * 8 + 3] = (float)a[ih]; rec[k * 8 + 4] = (float)a[il]; rec[k * 8 + 5] = (float)b[ih]; rec[k * 8 + 6] = (float)b[il]; }
        }
    }
}
__global__ void diff_u16(const unsigned short *a, const unsigned short *b, long long n, int *out) {
    int c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}

struct Lab {
    int B, C, Cout, H, W;
    uint4 *x_pf, *w, *y_pf[2];
    float *bias, *g, *bb, *resid, *out[2], *scratch, *stat[2][2];
    long long pf_bs, pf_ps;
    hipEvent_t e0, e1;
};

static PfArgs base_args(const Lab &L, bool ln, bool resid, bool f32out, bool pfout, int slot) {
    PfArgs a{};
    a.src0 = L.x_pf; a.src1 = nullptr; a.src0_bs = L.pf_bs; a.src1_bs = 0;
    a.C0 = L.C; a.Cin = L.C; a.H = L.H; a.W = L.W;
    a.w = L.w; a.w_zs = 0; a.KH = 3; a.KW = 3; a.nz = 1;
    a.pad_y[0] = 1; a.pad_x[0] = 1;
    a.nchunk = L.C / 16; a.COP = L.Cout; a.Cout = L.Cout;
    a.acc_scale = 1.0f / 8192.f / 24.f;
    a.out = f32out ? L.out[slot] : nullptr;
    a.out_bs = (long long)L.Cout * L.H * L.W; a.out_cs = (long long)L.H * L.W; a.out_ys = L.W; a.out_xs = 1; a.out_zoff[0] = 0;
    a.out_pf = pfout ? L.y_pf[slot] : nullptr;
    a.pf_bs = (long long)(L.Cout / 8) * 2 * L.pf_ps; a.pf_ps = L.pf_ps; a.pf_ys = L.W + 2; a.pf_xs = 1; a.pf_zoff[0] = (L.W + 2) + 1;
    a.Ho = L.H; a.Wo = L.W; a.lognbw = 5; a.B = L.B;
    a.bias = L.bias; a.ep_g = ln ? L.g : nullptr; a.ep_b = ln ? L.bb : nullptr; a.eps = 1e-5f;
    a.relu = ln ? 1 : 0; a.relu_slope = 0.f;
    a.resid = resid ? L.resid : nullptr; a.resid_bs = a.out_bs; a.resid_cs = a.out_cs;
    return a;
}

template <class F> static float time_it(Lab &L, int reps, F &&launch) {
    for (int i = 0; i < 10; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(L.e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(L.e1));
    CK(hipEventSynchronize(L.e1));
    float ms; CK(hipEventElapsedTime(&ms, L.e0, L.e1));
    CK(hipGetLastError());
    return ms / reps;
}

static void report(const char *name, const Lab &L, float ms) {
    const double fl = 2.0 * L.B * L.H * L.W * (double)L.Cout * L.C * 9;
    printf("%-58s %8.4f ms  %7.1f TF (x3 executed: %6.1f = %4.1f %% of 2500)\n", name, ms, fl / ms / 1e9, 3 * fl / ms / 1e9, 3 * fl / ms / 1e9 / 25.0);
    fflush(stdout);
}

static void compare(Lab &L, const char *what, bool f32, bool pf) {
    if (f32) {
        CK(hipMemset(L.scratch, 0, 16));
        const long long n = (long long)L.B * L.Cout * L.H * L.W;
        hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, L.out[0], L.out[1], n, L.scratch);
        float h[2]; CK(hipMemcpy(h, L.scratch, 8, hipMemcpyDeviceToHost));
        printf("    check %-40s fp32 max |diff| %.3e (max |ref| %.3e)\n", what, h[0], h[1]);
        if (h[0] > 1e-3f) {
            CK(hipMemset(L.scratch, 0, 1024));
            hipLaunchKernelGGL(diff_where, dim3(2048), dim3(256), 0, 0, L.out[0], L.out[1], n, 1e-3f, (int *)L.scratch, (long long *)L.scratch + 8);
            long long hi[72]; CK(hipMemcpy(hi, L.scratch, sizeof hi, hipMemcpyDeviceToHost));
            const int cnt = (int)hi[0] & 0x7fffffff;
            printf("      %d elements off; first: ", *(int *)hi);
            for (int k = 0; k < 12 && k < *(int *)hi; ++k) { const long long i = hi[8 + k]; printf("(b%lld c%lld y%lld x%lld) ", i / ((long long)L.Cout * L.H * L.W), i / (L.H * L.W) % L.Cout, i / L.W % L.H, i % L.W); }
            printf("\n"); (void)cnt;
        }
    }
    if (pf) {
        CK(hipMemset(L.scratch, 0, 16));
        const long long n = (long long)L.B * (L.Cout / 8) * 2 * L.pf_ps * 8;
        hipLaunchKernelGGL(diff_u16, dim3(2048), dim3(256), 0, 0, (const unsigned short *)L.y_pf[0], (const unsigned short *)L.y_pf[1], n, (int *)L.scratch);
        int h; CK(hipMemcpy(&h, L.scratch, 4, hipMemcpyDeviceToHost));
        printf("    check %-40s planes: %d differing fp16 values of %lld\n", what, h, n);
        CK(hipMemset(L.scratch, 0, 16));
        const long long pairs = (long long)L.B * (L.Cout / 8);
        hipLaunchKernelGGL(diff_pf, dim3(2048), dim3(256), 0, 0, (const _Float16 *)L.y_pf[0], (const _Float16 *)L.y_pf[1], pairs * L.pf_ps, L.pf_ps, L.scratch);
        float hv; CK(hipMemcpy(&hv, L.scratch, 4, hipMemcpyDeviceToHost));
        printf("    check %-40s planes by value (h + l/2048, halo included): max |diff| %.3e\n", what, hv);
        if (hv > 1e-3f) {
            CK(hipMemset(L.scratch, 0, 1024));
            hipLaunchKernelGGL(diff_pf_where, dim3(64), dim3(256), 0, 0, (const _Float16 *)L.y_pf[0], (const _Float16 *)L.y_pf[1], pairs * L.pf_ps * 8, L.pf_ps, (int *)L.scratch, L.scratch + 16);
            float r[16 + 96]; CK(hipMemcpy(r, L.scratch, sizeof r, hipMemcpyDeviceToHost));
            printf("      %d values off (of the first blocks scanned); (pair pos ch | ref h l | got h l):\n", *(int *)r);
            for (int k = 0; k < 12 && k < *(int *)r; ++k) { const float *q = r + 16 + k * 8; const long long pos = (long long)q[1]; printf("        pair %d y %lld x %lld ch %d | %g %g | %g %g\n", (int)q[0], pos / (L.W + 2), pos % (L.W + 2), (int)q[2], q[3], q[4], q[5], q[6]); }
        }
    }
    fflush(stdout);
}

static void launch_pf(const PfArgs &a0, int MB, int NPW, int WM, int WP, int B, int H, int W, pf_kernel_fn fn, int ring, size_t lds) {
    PfArgs a = a0;
    const int TH = WP * NPW;
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + TH - 1) / TH; a.ring = ring;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * B), (unsigned)(a.Cout / (WM * MB * 32)), 1);
    a.xcd_remap = (grid.x % 8 == 0 && grid.x >= 64) ? 1 : 0;
    hipLaunchKernelGGL(fn, grid, dim3(64 * WM * WP), lds, 0, a);
}

template <int MB, int NPW, int WM, int WP> static void run_base(Lab &L, const char *tag, PfArgs a, int reps) {
    constexpr int ring = pf_ring(MB, NPW, WM, WP, 3, 3);
    const size_t patch = (size_t)2 * pf_patch_units(NPW, WP, 3, 3) * 16, wst = (size_t)pf_rows(MB, NPW) * WM * MB * 32 * 16;
    const size_t lds = std::max(patch + ring * wst, sizeof(float) * (size_t)(4 * WM * MB * 32 + 2 * WM * WP * NPW * 32));
    pf_kernel_fn fn = conv_pf_kernel<MB, NPW, WM, WP, 3, 3>;
    CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    char name[160];
    snprintf(name, sizeof name, "pf<%d,%d,%d,%d> R%d lds %zu %s dbg=%d", MB, NPW, WM, WP, ring, lds, tag, a.dbg);
    const float ms = time_it(L, reps, [&] { launch_pf(a, MB, NPW, WM, WP, L.B, L.H, L.W, fn, ring, lds); });
    report(name, L, ms);
}

// bare matrix-pipe rate in the loop's own pattern: 12 x v_mfma_f32_32x32x16_f16 on 8 accumulators, NWG waves per SIMD
template <int BAR> __global__ void __launch_bounds__(256, 2) mfma_rate(float *out, int iters, float seed) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 8; ++q) { a[i][q] = (_Float16)(seed + i + q + (threadIdx.x & 3)); b[i][q] = (_Float16)(seed * 0.5f + i - q); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 + (k >> 1)], b[k & 1], acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[4 + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k >> 1], b[2 + (k & 1)], acc[4 + k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k >> 1], b[k & 1], acc[k], 0, 0, 0);
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static void run_rate(Lab &L) {
    for (int wgs : {256, 512}) for (int bar = 0; bar < 2; ++bar) {
        const int iters = 4000;
        auto go = [&] { if (bar) hipLaunchKernelGGL(mfma_rate<1>, dim3(wgs), dim3(256), 0, 0, L.scratch, iters, 1.0f); else hipLaunchKernelGGL(mfma_rate<0>, dim3(wgs), dim3(256), 0, 0, L.scratch, iters, 1.0f); };
        const float ms = time_it(L, 5, go);
        printf("mfma_rate f16 32x32x16: %d WGs x 4 waves, barrier %d: %.1f TF (%.1f %% of 2500)\n", wgs, bar, (double)wgs * 4 * iters * 12 * 32768.0 / ms / 1e9, (double)wgs * 4 * iters * 12 * 32768.0 / ms / 1e9 / 25.0);
    }
}

template <int MB, int NPW, int WM, int WP> static void run_pp(Lab &L, const char *tag, PfArgs a, int reps) {
    constexpr int ring = pf_ring(MB, NPW, WM, WP, 3, 3);
    const size_t patch = (size_t)2 * pf_patch_units(NPW, WP, 3, 3) * 16, wst = (size_t)pf_rows(MB, NPW) * WM * MB * 32 * 16;
    const size_t ldsg = std::max(patch + ring * wst, sizeof(float) * (size_t)(4 * WM * MB * 32 + 2 * WM * WP * NPW * 32));
    pf_kernel_fn fn = conv_pfpp_kernel<MB, NPW, WM, WP, 3, 3>;
    CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * ldsg)));
    char name[160];
    snprintf(name, sizeof name, "PINGPONG pf<%d,%d,%d,%d> R%d lds 2x%zu %s dbg=%d", MB, NPW, WM, WP, ring, ldsg, tag, a.dbg);
    const int TH = WP * NPW;
    a.tiles_x = (L.W + 31) / 32; a.tiles_y = (L.H + TH - 1) / TH; a.ring = (int)(ldsg / 16);
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * L.B / 2), (unsigned)(a.Cout / (WM * MB * 32)), 1);
    a.xcd_remap = (grid.x % 8 == 0 && grid.x >= 64) ? 1 : 0;
    const float ms = time_it(L, reps, [&] { hipLaunchKernelGGL(fn, grid, dim3(128 * WM * WP), 2 * ldsg, 0, a); });
    report(name, L, ms);
#ifdef LAB_TL2
    {
        CK(hipMemset(L.scratch, 0, 64 * 8 * 8 * 8));
        a.res3_x = L.scratch;
        hipLaunchKernelGGL(fn, grid, dim3(128 * WM * WP), 2 * ldsg, 0, a);
        std::vector<unsigned long long> h(64 * 8 * 8);
        CK(hipMemcpy(h.data(), L.scratch, h.size() * 8, hipMemcpyDeviceToHost));
        double pro = 0, loop = 0;
        for (int b = 0; b < 64; ++b) for (int w = 0; w < 8; ++w) { pro += (double)h[(b * 8 + w) * 8 + 6] / 512; loop += (double)h[(b * 8 + w) * 8 + 7] / 512; }
        printf("    coarse timeline (first 64 WGs): prologue %.0f cycles, main loop %.0f cycles = %.0f per tap\n", pro, loop, loop / 36);
        a.res3_x = nullptr;
    }
#endif
#ifdef LAB_TL
    {
        CK(hipMemset(L.scratch, 0, 16 * 8 * 8 * 8));
        a.res3_x = L.scratch;
        hipLaunchKernelGGL(fn, grid, dim3(128 * WM * WP), 2 * ldsg, 0, a);
        std::vector<unsigned long long> h(16 * 8 * 8);
        CK(hipMemcpy(h.data(), L.scratch, h.size() * 8, hipMemcpyDeviceToHost));
        const int S = 36;
        printf("    timeline, cycles per tap (mean over 16 WGs): wave: mma+ctl | vmwait | barX | fetch | dma | barY\n");
        for (int w = 0; w < 8; ++w) {
            double m[6] = {0};
            for (int b = 0; b < 16; ++b) for (int q = 0; q < 6; ++q) m[q] += (double)h[(b * 8 + w) * 8 + q] / 16 / S;
            printf("      wave %d: %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f   sum %7.0f\n", w, m[0], m[1], m[2], m[3], m[4], m[5], m[0] + m[1] + m[2] + m[3] + m[4] + m[5]);
        }
    }
#endif
}

#ifdef LAB_PF3
template <int MB, int NPW, int WM, int WP, bool SYNC = false> static void lab_pf3(Lab &L, const char *tag, bool ln, bool resid, bool f32, bool pf, int reps, bool stat = false) {
    // reference result from the baseline kernel into slot 0
    {
        PfArgs a = base_args(L, ln, resid, f32, pf, 0);
        if (stat) { a.stat_mean = L.stat[0][0]; a.stat_rstd = L.stat[0][1]; }
        constexpr int ring = pf_ring(MB, NPW, WM, WP, 3, 3);
        const size_t patch = (size_t)2 * pf_patch_units(NPW, WP, 3, 3) * 16, wst = (size_t)pf_rows(MB, NPW) * WM * MB * 32 * 16;
        const size_t lds = std::max(patch + ring * wst, sizeof(float) * (size_t)(4 * WM * MB * 32 + 2 * WM * WP * NPW * 32));
        pf_kernel_fn fn = conv_pf_kernel<MB, NPW, WM, WP, 3, 3>;
        CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        launch_pf(a, MB, NPW, WM, WP, L.B, L.H, L.W, fn, ring, lds);
    }
    PfArgs a = base_args(L, ln, resid, f32, pf, 1);
    if (stat) { a.stat_mean = L.stat[1][0]; a.stat_rstd = L.stat[1][1]; }
    const int TH = WP * NPW;
    a.tiles_x = L.W / 32; a.tiles_y = L.H / TH;
    const int ntiles = a.tiles_x * a.tiles_y * L.B;
    const int G = getenv("PF3_G") ? atoi(getenv("PF3_G")) : 256;
    if (ntiles % (2 * G)) { printf("pf3: %d tiles not divisible by 2 x %d workgroups\n", ntiles, G); return; }
    a.n_iter = ntiles / (2 * G);
    a.xcd_remap = 1;
    const size_t lds = pf3_lds_bytes();
    if (pf3_lds_used(MB, NPW, WM, WP, L.B) > lds) { printf("pf3: LDS overflow\n"); return; }
    pf_kernel_fn fn = resid ? (pf ? conv_pf3_kernel<MB, NPW, WM, WP, 7, SYNC> : conv_pf3_kernel<MB, NPW, WM, WP, 3, SYNC>) : conv_pf3_kernel<MB, NPW, WM, WP, 4, SYNC>;
    if constexpr (SYNC) if (stat) fn = conv_pf3_kernel<MB, NPW, WM, WP, 11, true>;
    if ((resid && !f32) || (!resid && (f32 || !pf))) { printf("pf3 lab: variant not instantiated\n"); return; }
    CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    char name[160];
    snprintf(name, sizeof name, "PF3%s <%d,%d,%d,%d> D%d G %d iters %d %s", SYNC ? "-SYNC" : "", MB, NPW, WM, WP, CDC_PF3_D, G, a.n_iter, tag);
    dim3 grid((unsigned)G, (unsigned)(a.Cout / (WM * MB * 32)), 1);
    for (int stag : {0}) {
        a.dbg = stag;
        if (stag && !SYNC) break;
        const float ms = time_it(L, reps, [&] { hipLaunchKernelGGL(fn, grid, dim3(512), lds, 0, a); });
        char nm[200]; snprintf(nm, sizeof nm, "%s stagger %d", name, stag);
        report(nm, L, ms);
    }
    compare(L, "pf3 vs baseline", f32, pf);
    if (stat) {
        for (int r = 0; r < 2; ++r) {
            CK(hipMemset(L.scratch, 0, 16));
            hipLaunchKernelGGL(diff_kernel, dim3(512), dim3(256), 0, 0, L.stat[0][r], L.stat[1][r], (long long)L.B * L.H * L.W, L.scratch);
            float h[2]; CK(hipMemcpy(h, L.scratch, 8, hipMemcpyDeviceToHost));
            printf("    check LayerNorm statistics (%s)                 max |diff| %.3e (max |ref| %.3e)\n", r ? "rstd" : "mean", h[0], h[1]);
        }
    }
#ifdef CDC_PF3_TL
    {
        CK(hipMemset(L.scratch, 0, 16 * 8 * 64 * 8));
        a.res3_x = L.scratch;
        CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        hipLaunchKernelGGL(fn, grid, dim3(512), 163840, 0, a);
        std::vector<unsigned long long> h(16 * 8 * 64);
        CK(hipMemcpy(h.data(), L.scratch, h.size() * 8, hipMemcpyDeviceToHost));
        double m[64] = {0};
        for (size_t i = 0; i < 16 * 8; ++i) for (int q = 0; q < 64; ++q) m[q] += (double)h[i * 64 + q] / 128;
        const int nchunk = L.C / 16;
        printf("    slot timeline (cycles per wave): main %.0f total = %.0f per step | epilogue %.0f = %.0f per step | idle %.0f = %.0f per step\n",
               m[60], m[60] / (a.n_iter * nchunk * 9.0), m[61], m[61] / (a.n_iter * 9.0), m[62], m[62] / (((nchunk + 1) / 2) * 9.0));
        printf("    epilogue step t: [piece 2t | dma+wait | barrier | piece 2t+1 | barrier] cycles per tile\n");
        for (int t = 0; t < 9; ++t)
            printf("      t%d: %6.0f %6.0f %6.0f | %6.0f %6.0f\n", t, m[5 * t] / a.n_iter, m[5 * t + 1] / a.n_iter, m[5 * t + 2] / a.n_iter, m[5 * t + 3] / a.n_iter, m[5 * t + 4] / a.n_iter);
    }
#endif
}
#endif

int main(int argc, char **argv) {
    Lab L{};
    L.B = argc > 1 ? atoi(argv[1]) : 32;
    L.C = argc > 2 ? atoi(argv[2]) : 64;
    L.H = L.W = argc > 3 ? atoi(argv[3]) : 256;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    L.Cout = L.C;
    L.pf_ps = (long long)(L.H + 2) * (L.W + 2);
    L.pf_bs = (long long)(L.C / 8) * 2 * L.pf_ps;
    const long long npf = (long long)L.B * L.pf_bs, nf = (long long)L.B * L.Cout * L.H * L.W;
    CK(hipMalloc(&L.x_pf, npf * 16));
    for (int s = 0; s < 2; ++s) { CK(hipMalloc(&L.y_pf[s], npf * 16)); CK(hipMemset(L.y_pf[s], 0, npf * 16)); CK(hipMalloc(&L.out[s], nf * 4)); CK(hipMemset(L.out[s], 0, nf * 4)); }
    CK(hipMalloc(&L.resid, nf * 4));
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 2; ++r) { CK(hipMalloc(&L.stat[q][r], (size_t)L.B * L.H * L.W * 4)); CK(hipMemset(L.stat[q][r], 0, (size_t)L.B * L.H * L.W * 4)); }
    const int taps = 9, nchunk = L.C / 16;
    CK(hipMalloc(&L.w, (size_t)taps * nchunk * 6 * L.Cout * 16));
    CK(hipMalloc(&L.bias, L.Cout * 4)); CK(hipMalloc(&L.g, L.Cout * 4)); CK(hipMalloc(&L.bb, L.Cout * 4));
    CK(hipMalloc(&L.scratch, 1 << 20));
    CK(hipEventCreate(&L.e0)); CK(hipEventCreate(&L.e1));
    {
        const long long n = (long long)(L.C / 8) * (L.H + 2) * (L.W + 2);
        hipLaunchKernelGGL(fill_pf, dim3((unsigned)((n + 255) / 256), L.B), dim3(256), 0, 0, L.x_pf, L.C, L.H, L.W, 11u);
        hipLaunchKernelGGL(fill_f32, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, 0, L.resid, nf, 12u, 1.0f);
        hipLaunchKernelGGL(fill_f32, dim3(1), dim3(256), 0, 0, L.bias, (long long)L.Cout, 13u, 0.1f);
        hipLaunchKernelGGL(fill_f32, dim3(1), dim3(256), 0, 0, L.g, (long long)L.Cout, 14u, 1.0f);
        hipLaunchKernelGGL(fill_f32, dim3(1), dim3(256), 0, 0, L.bb, (long long)L.Cout, 15u, 0.2f);
        const long long nw = (long long)taps * nchunk * 2 * L.Cout;
        hipLaunchKernelGGL(fill_w, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, L.w, taps, nchunk, L.Cout, 16u);
        CK(hipDeviceSynchronize());
    }
    printf("pf_lab: B %d C %d -> %d, %dx%d, 3x3, reps %d\n", L.B, L.C, L.Cout, L.H, L.W, reps);
    if (getenv("LAB_RATE")) run_rate(L);

    // ---- baseline kernel, the three epilogue flavours of the network -------------------------------------------
    if (L.C == 64) {
        run_base<2, 2, 1, 4>(L, "LN+res f32", base_args(L, true, true, true, false, 0), reps);
        run_base<2, 2, 1, 4>(L, "LN nof32 +pf", base_args(L, true, false, false, true, 0), reps);
        run_base<2, 2, 1, 4>(L, "LN+res f32+pf", base_args(L, true, true, true, true, 0), reps);
#if CDC_PF_ABLATE
        for (int dbg : {0, 256, 256 + 32}) {
            PfArgs a = base_args(L, true, true, true, false, 0), a1 = base_args(L, true, true, true, false, 1);
            a.dbg = a1.dbg = dbg;
            run_base<2, 2, 1, 4>(L, "LN+res f32 ABLATED", a, reps);
            run_pp<2, 2, 1, 4>(L, "LN+res f32", a1, reps);
            if (dbg == 0) compare(L, "ping-pong vs baseline", true, false);
        }
#endif
    } else if (L.C == 128) {
        run_base<2, 2, 2, 2>(L, "LN+res f32", base_args(L, true, true, true, false, 0), reps);
        run_base<2, 2, 2, 2>(L, "LN nof32 +pf", base_args(L, true, false, false, true, 0), reps);
        run_base<2, 2, 2, 2>(L, "LN+res f32+pf", base_args(L, true, true, true, true, 0), reps);
    }
#ifdef LAB_PF3
    if (L.C == 64) {
        lab_pf3<2, 2, 1, 4>(L, "LN+res f32", true, true, true, false, reps);
        lab_pf3<2, 2, 1, 4>(L, "LN nof32 +pf", true, false, false, true, reps);
        lab_pf3<2, 2, 1, 4>(L, "LN+res f32+pf", true, true, true, true, reps);
        lab_pf3<2, 2, 1, 4, true>(L, "LN+res f32", true, true, true, false, reps);
        lab_pf3<2, 2, 1, 4, true>(L, "LN+res f32 +stat", true, true, true, false, reps, true);
        lab_pf3<2, 2, 1, 4, true>(L, "LN nof32 +pf", true, false, false, true, reps);
        lab_pf3<2, 2, 1, 4, true>(L, "LN+res f32+pf", true, true, true, true, reps);
    } else if (L.C == 128) {
        lab_pf3<2, 2, 2, 2, true>(L, "LN+res f32", true, true, true, false, reps);
        lab_pf3<2, 2, 2, 2, true>(L, "LN+res f32 +stat", true, true, true, false, reps, true);
        lab_pf3<2, 2, 2, 2, true>(L, "LN nof32 +pf", true, false, false, true, reps);
        lab_pf3<2, 2, 2, 2, true>(L, "LN+res f32+pf", true, true, true, true, reps);
    }
#endif
    return 0;
}
