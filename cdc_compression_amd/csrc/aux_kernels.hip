// aux_kernels.hip -- the non-GEMM kernels of the decode path: standalone channel LayerNorm,
// time-embedding MLPs, k-softmax statistics, linear-attention context (k.v^T on MFMA), DDIM update.
#include <math.h>

#include <algorithm>
#include <stdlib.h>

#include "cdc_internal.h"

namespace cdc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------------------------
// Channel LayerNorm (reference network_components.py:56-66), NCHW: one thread per pixel, channel
// loop strided by HW (coalesced across the wave).  Used where the convolution epilogue cannot own
// all channels (few-pixel levels, Cout % 32 != 0) and for statistics-only passes.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ln_kernel(const LnArgs a) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.HW) return;
    const size_t base = (size_t)b * a.C * a.HW + p;
    const float *x = a.in + base;
    float s = 0.f;
    for (int c = 0; c < a.C; ++c) s += x[(size_t)c * a.HW];
    const float mean = s / (float)a.C;
    float q = 0.f;
    for (int c = 0; c < a.C; ++c) {
        const float d = x[(size_t)c * a.HW] - mean;
        q += d * d;
    }
    const float var = q / (float)a.C;
    if (a.fault && !(var < 3.0e38f)) *a.fault = 1;       // range guard (ConvArgs::fault): before LayerNorm + ReLU can hide it
    if (!a.out) {
        a.stat_mean[(size_t)b * a.HW + p] = mean;
        a.stat_rstd[(size_t)b * a.HW + p] = 1.0f / sqrtf(var + a.eps);
        return;
    }
    const float den = sqrtf(var + a.eps);
    float *y = a.out + base;
    const float *r = a.resid ? a.resid + base : nullptr;
    const float *sh = a.shift ? a.shift + (size_t)b * a.shift_bs : nullptr;
    float s2 = 0.f;
    for (int c = 0; c < a.C; ++c) {
        float v = (x[(size_t)c * a.HW] - mean) / den * a.g[c] + a.b[c];
        if (a.relu) v = fmaxf(v, 0.f);
        if (sh) v += sh[c];
        if (r) v += r[(size_t)c * a.HW];
        y[(size_t)c * a.HW] = v;
        s2 += v;
    }
    if (a.stat_mean) {
        const float m2 = s2 / (float)a.C;
        float q2 = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float d = y[(size_t)c * a.HW] - m2;
            q2 += d * d;
        }
        a.stat_mean[(size_t)b * a.HW + p] = m2;
        a.stat_rstd[(size_t)b * a.HW + p] = 1.0f / sqrtf(q2 / (float)a.C + a.eps);
    }
}

// Same operation, 8 channel slices x 32 pixels per workgroup with the pixel's channels cached in
// registers (C <= 8*kLnCache): the few-pixel levels (8x8, 16x16) otherwise leave the chip idle while
// single threads walk 384 channels three times.
constexpr int kLnCache = 48;

// PL pixels x CS = 256/PL channel slices per workgroup: PL = 32 normally, 8 for the few-pixel levels (more
// workgroups; a thread then walks C/32 channels x nparts split-K slices).
// NP: number of split-K slices summed on load (compile-time: all loads of a thread are issued back to back; with a
// run-time slice loop every slice of every channel row was its own serialized round trip -- 26 us per launch at the
// few-pixel levels); NP = 0: run-time a.nparts (> 4).
template <int PL, int NP>
__global__ void __launch_bounds__(256) ln_kernel_sliced(const LnArgs a) {
    constexpr int CS = 256 / PL, NV = 8 * kLnCache / CS;
    __shared__ float red[CS][PL + 1];
    const int b = blockIdx.y;
    const int pl = threadIdx.x % PL, cs = threadIdx.x / PL;
    const int p = blockIdx.x * PL + pl;
    const bool pv = p < a.HW;
    const size_t base = (size_t)b * a.C * a.HW + (pv ? p : 0);
    const float *x = a.in + base;
    float v[NV];
    float s = 0.f;
    if constexpr (NP > 0) {
        float t[NP][NV];
#pragma unroll
        for (int k = 0; k < NP; ++k)                  // split-K slices of the producing convolution
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = cs + CS * i;
                t[k][i] = (pv && c < a.C) ? x[(size_t)k * a.part_stride + (size_t)c * a.HW] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = t[0][i];
#pragma unroll
            for (int k = 1; k < NP; ++k) v[i] += t[k][i];           // same order as the run-time loop: slice 0, 1, 2, ...
            s += v[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = cs + CS * i;
            v[i] = (pv && c < a.C) ? x[(size_t)c * a.HW] : 0.f;
            for (int k = 1; k < a.nparts; ++k)
                v[i] += (pv && c < a.C) ? x[(size_t)k * a.part_stride + (size_t)c * a.HW] : 0.f;
            s += v[i];
        }
    }
    red[cs][pl] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < CS; ++k) tot += red[k][pl];
    const float mean = tot / (float)a.C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float d = v[i] - mean;
        q += (cs + CS * i < a.C) ? d * d : 0.f;
    }
    red[cs][pl] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int k = 0; k < CS; ++k) tot += red[k][pl];
    const float var = tot / (float)a.C;
    if (a.fault && !(var < 3.0e38f)) *a.fault = 1;       // range guard (ConvArgs::fault)
    if (!a.out) {
        if (pv && cs == 0) {
            a.stat_mean[(size_t)b * a.HW + p] = mean;
            a.stat_rstd[(size_t)b * a.HW + p] = 1.0f / sqrtf(var + a.eps);
        }
        return;
    }
    const float den = sqrtf(var + a.eps);
    float *y = a.out + base;
    const float *r = a.resid ? a.resid + base : nullptr;
    const float *sh = a.shift ? a.shift + (size_t)b * a.shift_bs : nullptr;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = cs + CS * i;
        if (c < a.C) {
            float w = (v[i] - mean) / den * a.g[c] + a.b[c];
            if (a.relu) w = fmaxf(w, 0.f);
            if (sh) w += sh[c];
            if (r && pv) w += r[(size_t)c * a.HW];
            if (pv) y[(size_t)c * a.HW] = w;
            v[i] = w;
            s2 += w;
        } else {
            v[i] = 0.f;
        }
    }
    if (a.stat_mean) {
        __syncthreads();
        red[cs][pl] = s2;
        __syncthreads();
        tot = 0.f;
#pragma unroll
        for (int k = 0; k < CS; ++k) tot += red[k][pl];
        const float m2 = tot / (float)a.C;
        __syncthreads();
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float d = v[i] - m2;
            q2 += (cs + CS * i < a.C) ? d * d : 0.f;
        }
        red[cs][pl] = q2;
        __syncthreads();
        tot = 0.f;
#pragma unroll
        for (int k = 0; k < CS; ++k) tot += red[k][pl];
        if (pv && cs == 0) {
            a.stat_mean[(size_t)b * a.HW + p] = m2;
            a.stat_rstd[(size_t)b * a.HW + p] = 1.0f / sqrtf(tot / (float)a.C + a.eps);
        }
    }
}

// The same pass on 16-byte accesses (round 4; HW % 4 == 0): COLS float4 columns = 4 COLS pixels x CS = 256 / COLS channel slices per
// workgroup.  A 16 x 16 or 8 x 8 level is a latency chain, not a bandwidth problem (60 MB / 16 MB per launch): one quarter of the
// load instructions (all of a thread's loads fit the 64-deep vector-memory queue of its wave: one round trip instead of three), the
// residual and the per-channel parameters fetched in that same round trip instead of after the statistics, and the cross-slice sums as
// an in-wave butterfly + 4 LDS values instead of CS LDS reads per thread.
template <int COLS, int NP>
__global__ void __launch_bounds__(256) ln_kernel_vec(const LnArgs a) {
    constexpr int CS = 256 / COLS, NV = 8 * kLnCache / CS;
    __shared__ float4 red[2][4][COLS];
    const int b = a.img_major ? blockIdx.x : blockIdx.y, bx = a.img_major ? blockIdx.y : blockIdx.x, tid = threadIdx.x;
    const int col = tid % COLS, cs = tid / COLS, wave = tid >> 6;
    const int p = (bx * COLS + col) * 4;
    const bool pv = p < a.HW;
    const size_t base = (size_t)b * a.C * a.HW + (pv ? p : 0);
    const float *x = a.in + base;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 t[NP][NV], rv[NV];
    float gv[NV], bv[NV], sv[NV];
#pragma unroll
    for (int k = 0; k < NP; ++k)                      // split-K slices of the producing convolution
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            // (always a load from a valid address, the value masked afterwards: "cond ? *p : zero4" became a select between p and the
            //  address of a zero4 copy in SCRATCH, every load of the pass a flat load and the masked lanes a round trip to private memory)
            const int c = cs + CS * i, cc = c < a.C ? c : a.C - 1;
            const float4 ld = *reinterpret_cast<const float4 *>(x + (size_t)k * a.part_stride + (size_t)cc * a.HW);
            const bool ok = pv && c < a.C;
            t[k][i] = make_float4(ok ? ld.x : 0.f, ok ? ld.y : 0.f, ok ? ld.z : 0.f, ok ? ld.w : 0.f);
        }
    const float *r = (a.out && a.resid) ? a.resid + base : nullptr;
    const float *sh = (a.out && a.shift) ? a.shift + (size_t)b * a.shift_bs : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = cs + CS * i;
        const bool ok = c < a.C && a.out;
        rv[i] = zero4;
        if (r && pv && ok) rv[i] = *reinterpret_cast<const float4 *>(r + (size_t)c * a.HW);
        gv[i] = ok ? a.g[c] : 0.f;
        bv[i] = ok ? a.b[c] : 0.f;
        sv[i] = (sh && ok) ? sh[c] : 0.f;
    }
    auto wg_sum = [&](float4 s, int slot) {
#pragma unroll
        for (int o = COLS; o < 64; o <<= 1) {
            s.x += __shfl_xor(s.x, o); s.y += __shfl_xor(s.y, o); s.z += __shfl_xor(s.z, o); s.w += __shfl_xor(s.w, o);
        }
        if ((tid & 63) < COLS) red[slot][wave][col] = s;
        __syncthreads();
        float4 q = red[slot][0][col];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const float4 u = red[slot][w][col]; q.x += u.x; q.y += u.y; q.z += u.z; q.w += u.w; }
        return q;
    };
    float4 v[NV], s = zero4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = t[0][i];
#pragma unroll
        for (int k = 1; k < NP; ++k) { v[i].x += t[k][i].x; v[i].y += t[k][i].y; v[i].z += t[k][i].z; v[i].w += t[k][i].w; }   // slice 0, 1, 2, ...
        s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w;
    }
    const float inv_c = 1.0f / (float)a.C;
    float4 mean = wg_sum(s, 0);
    mean.x *= inv_c; mean.y *= inv_c; mean.z *= inv_c; mean.w *= inv_c;
    float4 q = zero4;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (cs + CS * i < a.C) {
            const float dx = v[i].x - mean.x, dy = v[i].y - mean.y, dz = v[i].z - mean.z, dw = v[i].w - mean.w;
            q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
        }
    float4 var = wg_sum(q, 1);
    var.x *= inv_c; var.y *= inv_c; var.z *= inv_c; var.w *= inv_c;
    if (a.fault && !(var.x < 3.0e38f && var.y < 3.0e38f && var.z < 3.0e38f && var.w < 3.0e38f)) *a.fault = 1;   // range guard (ConvArgs::fault)
    if (!a.out) {
        if (pv && cs == 0) {
            *reinterpret_cast<float4 *>(a.stat_mean + (size_t)b * a.HW + p) = mean;
            *reinterpret_cast<float4 *>(a.stat_rstd + (size_t)b * a.HW + p) =
                make_float4(1.0f / sqrtf(var.x + a.eps), 1.0f / sqrtf(var.y + a.eps), 1.0f / sqrtf(var.z + a.eps), 1.0f / sqrtf(var.w + a.eps));
        }
        return;
    }
    const float4 den = make_float4(sqrtf(var.x + a.eps), sqrtf(var.y + a.eps), sqrtf(var.z + a.eps), sqrtf(var.w + a.eps));
    float *y = a.out + base;
    float4 s2 = zero4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = cs + CS * i;
        if (c < a.C) {
            float4 w;
            w.x = (v[i].x - mean.x) / den.x * gv[i] + bv[i];
            w.y = (v[i].y - mean.y) / den.y * gv[i] + bv[i];
            w.z = (v[i].z - mean.z) / den.z * gv[i] + bv[i];
            w.w = (v[i].w - mean.w) / den.w * gv[i] + bv[i];
            if (a.relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
            if (sh) { w.x += sv[i]; w.y += sv[i]; w.z += sv[i]; w.w += sv[i]; }
            if (r) { w.x += rv[i].x; w.y += rv[i].y; w.z += rv[i].z; w.w += rv[i].w; }
            if (pv) *reinterpret_cast<float4 *>(y + (size_t)c * a.HW) = w;
            v[i] = w;
            s2.x += w.x; s2.y += w.y; s2.z += w.z; s2.w += w.w;
        } else {
            v[i] = zero4;
        }
    }
    if (a.stat_mean) {
        float4 m2 = wg_sum(s2, 0);
        m2.x *= inv_c; m2.y *= inv_c; m2.z *= inv_c; m2.w *= inv_c;
        float4 q2 = zero4;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (cs + CS * i < a.C) {
                const float dx = v[i].x - m2.x, dy = v[i].y - m2.y, dz = v[i].z - m2.z, dw = v[i].w - m2.w;
                q2.x += dx * dx; q2.y += dy * dy; q2.z += dz * dz; q2.w += dw * dw;
            }
        const float4 v2 = wg_sum(q2, 1);
        if (pv && cs == 0) {
            *reinterpret_cast<float4 *>(a.stat_mean + (size_t)b * a.HW + p) = m2;
            *reinterpret_cast<float4 *>(a.stat_rstd + (size_t)b * a.HW + p) =
                make_float4(1.0f / sqrtf(v2.x * inv_c + a.eps), 1.0f / sqrtf(v2.y * inv_c + a.eps), 1.0f / sqrtf(v2.z * inv_c + a.eps),
                            1.0f / sqrtf(v2.w * inv_c + a.eps));
        }
    }
}

template <int COLS> static void ln_vec_launch(const LnArgs &a, dim3 grid, hipStream_t st) {
    switch (a.nparts) {
        case 1: hipLaunchKernelGGL((ln_kernel_vec<COLS, 1>), grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((ln_kernel_vec<COLS, 2>), grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((ln_kernel_vec<COLS, 3>), grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((ln_kernel_vec<COLS, 4>), grid, dim3(256), 0, st, a); break;
        default:
            // five / six slices: the 8-pixel form only (three float4 per thread and slice; one image per call slices the 8 x 8 level
            // six ways -- kLnVecMaxParts2)
            if constexpr (COLS == 2) {
                if (a.nparts == 5) hipLaunchKernelGGL((ln_kernel_vec<2, 5>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((ln_kernel_vec<2, 6>), grid, dim3(256), 0, st, a);
            }
            break;
    }
}
constexpr int kLnVecMaxParts2 = 6;      // most slices ln_kernel_vec<2, NP> is instantiated for (other widths: 4)

// The workgroup shape (pixel columns per workgroup, hence the order of the cross-slice channel reduction) is chosen from the RUN-TIME
// batch B, so the last bits of a LayerNorm output may differ between batch sizes (the parity tests bound rows of a batch against
// batch-1 calls at 5e-6).  Nothing with a bit-exactness contract across batch sizes contains a LayerNorm: hyper_dec -- the entropy
// coder's model, include/cdc_hip.h -- is convolutions + LeakyReLU only.
hipError_t ln_launch(const LnArgs &a, int B, hipStream_t st) {
    if (a.nparts > 1 && a.C > 8 * kLnCache) return hipErrorInvalidValue;
    if (a.C <= 8 * kLnCache && a.nparts >= 1 && a.nparts <= kLnVecMaxParts2 && (a.HW & 3) == 0 && (a.part_stride & 3) == 0 &&
        ((((uintptr_t)a.in) | ((uintptr_t)a.out) | ((uintptr_t)a.resid) | ((uintptr_t)a.stat_mean) | ((uintptr_t)a.stat_rstd)) & 15) == 0 &&
        !dev_env("CDC_NO_LN_VEC")) {
        const long long min_wgs = 256;
        int cols = (long long)ceil_div(a.HW, 32) * B >= min_wgs ? 8 : ((long long)ceil_div(a.HW, 16) * B >= min_wgs ? 4 : 2);
        if (const char *e = dev_env("CDC_LN_VEC_COLS")) { const int v = atoi(e); if (v == 8 || v == 4 || v == 2) cols = v; }
        if (a.nparts > 4) cols = 2;                   // (more than four slices exist for the 8-pixel form only)
        // Narrow workgroups (8 / 16 pixels = 32 / 64 bytes of every 128-byte channel row) share each cache line with their neighbours of the
        // same image.  Workgroups go to the XCDs round robin by linear id, and the eight L2s share nothing: with the image as the FAST grid
        // dimension (and B a multiple of 8) all column blocks of an image run on ONE XCD, so a line is fetched from the memory side once
        // instead of once per neighbour (round 6; CDC_LN_NO_IMG_MAJOR=1 brings the old order back for A/B).
        LnArgs la = a;
        la.img_major = (cols < 8 && B % 8 == 0 && a.HW > 4 * cols && !dev_env("CDC_LN_NO_IMG_MAJOR")) ? 1 : 0;
        const unsigned nbx = (unsigned)ceil_div(a.HW, 4 * cols);
        const dim3 grid = la.img_major ? dim3((unsigned)B, nbx) : dim3(nbx, (unsigned)B);
        if (cols == 8) ln_vec_launch<8>(la, grid, st);
        else if (cols == 4) ln_vec_launch<4>(la, grid, st);
        else ln_vec_launch<2>(la, grid, st);
        return hipGetLastError();
    }
    if (a.C <= 8 * kLnCache) {
        static const int pl8_max = 64;
        // 8-pixel workgroups also wherever 32-pixel ones would leave most of the chip idle (small batches)
        static const int min_wgs = 256;    // (measured at batch 32: 16x16 maps are faster on 32-pixel workgroups, 128-byte rows)
        const bool pl8 = a.HW <= pl8_max || (long long)ceil_div(a.HW, 32) * B < min_wgs;
        const dim3 grid((unsigned)ceil_div(a.HW, pl8 ? 8 : 32), (unsigned)B);
#define CDC_LN_LAUNCH(PLV, NPV) hipLaunchKernelGGL((ln_kernel_sliced<PLV, NPV>), grid, dim3(256), 0, st, a)
        if (pl8) {
            switch (a.nparts) {
                case 1: CDC_LN_LAUNCH(8, 1); break;
                case 2: CDC_LN_LAUNCH(8, 2); break;
                case 3: CDC_LN_LAUNCH(8, 3); break;
                case 4: CDC_LN_LAUNCH(8, 4); break;
                default: CDC_LN_LAUNCH(8, 0); break;
            }
        } else {
            switch (a.nparts) {                       // (48 cached values per thread and slice)
                case 1: CDC_LN_LAUNCH(32, 1); break;
                case 2: CDC_LN_LAUNCH(32, 2); break;
                case 3: CDC_LN_LAUNCH(32, 3); break;
                case 4: CDC_LN_LAUNCH(32, 4); break;
                default: CDC_LN_LAUNCH(32, 0); break;
            }
        }
#undef CDC_LN_LAUNCH
        return hipGetLastError();
    }
    const int block = a.HW >= 256 ? 256 : 64;
    dim3 grid((unsigned)ceil_div(a.HW, block), (unsigned)B);
    hipLaunchKernelGGL(ln_kernel, grid, dim3(block), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Time embedding: Unet.time_mlp (unet.py:41) then every ResnetBlock.mlp
// (network_components.py:96-100, LeakyReLU(0.2) -> Linear(dim, cout)).  One workgroup per image.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) temb_kernel(const TembArgs a) {
    extern __shared__ float sm[];
    float *h = sm;                 // [4*dim]
    float *t = sm + 4 * a.dim;     // [dim]  LeakyReLU(temb)
    const int b = blockIdx.x;
    const float tv = a.time[b];
    for (int j = threadIdx.x; j < 4 * a.dim; j += blockDim.x) {
        const float u = a.w0[j] * tv + a.b0[j];
        h[j] = 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));     // nn.GELU() (erf form)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.dim; i += blockDim.x) {
        const float *w = a.w2 + (size_t)i * 4 * a.dim;
        float s = 0.f;
        for (int j = 0; j < 4 * a.dim; ++j) s += w[j] * h[j];
        s += a.b2[i];
        t[i] = s >= 0.f ? s : 0.2f * s;
    }
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l) {
        const TembLayer L = a.layers[l];
        for (int co = threadIdx.x; co < L.cout; co += blockDim.x) {
            const float *w = L.w + (size_t)co * a.dim;
            float s = 0.f;
            for (int i = 0; i < a.dim; ++i) s += w[i] * t[i];
            a.shift[(size_t)b * a.shift_bs + L.out_off + co] = s + L.bias[co];
        }
    }
}

hipError_t temb_launch(const TembArgs &a, int B, hipStream_t st) {
    hipLaunchKernelGGL(temb_kernel, dim3(B), dim3(256), sizeof(float) * 5 * a.dim, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k.softmax(dim=-1) (network_components.py:134), part 1: row maximum over the N pixels.
// (The row sum of exponentials is accumulated by the context kernel, which exponentiates anyway.)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kmax_kernel(const float *k, long long k_bs, int C, int N,
                                                   float *kmax) {
    __shared__ float red[4];
    const int d = blockIdx.x, b = blockIdx.y;
    const float *row = k + (size_t)b * k_bs + (size_t)d * N;
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float m = -INFINITY;
    if ((N & 3) == 0) {
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
        for (int n = threadIdx.x; n < (N >> 2); n += blockDim.x) {
            const float4 v = r4[n];
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    } else {
        for (int n = threadIdx.x; n < N; n += blockDim.x) m = fmaxf(m, row[n]);
    }
    m = wave_max(m);
    if (lane == 0) red[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = red[0];
        for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
        kmax[(size_t)b * C + d] = m;
    }
}

hipError_t kstats_launch(const float *k, long long k_bs, int C, int N, float *kmax, int B,
                         hipStream_t st) {
    const int block = N >= 2048 ? 256 : 64;
    hipLaunchKernelGGL(kmax_kernel, dim3(C, B), dim3(block), 0, st, k, k_bs, C, N, kmax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// context = softmax(k) . v^T  (network_components.py:135), unnormalised partial sums per pixel split:
//   S[b][split][d][e] = sum_{n in split} exp(k[d,n] - kmax[d]) * v[e,n]
//   Zp[b][split][d]   = sum_{n in split} exp(k[d,n] - kmax[d])
// HBM-bound (C/4 FLOP per byte).  64(d) x 64(e) tile per workgroup on v_mfma_f32_32x32x2_f32 with
// A[i=d][k=n], B[k=n][j=e]; k / v tiles of 64 pixels stream global -> LDS by 16-byte LDS-DMA into a
// two-stage ring.  A DMA piece is lane-linear in LDS, so the bank-conflict swizzle is applied on the
// SOURCE address: LDS slot s of row r holds the pixel quad s ^ (r & 15); reads undo it.  exp() is
// applied to the A operand on its way from LDS to the MFMA.  The four waves split each chunk's
// pixels (16 each) and are summed through LDS at the end.
// ---------------------------------------------------------------------------------------------
constexpr int kCtxPch = 64;

__device__ __attribute__((aligned(16))) float g_ctx_zeros[64];

__device__ __forceinline__ void ctx_dma16(const float *g, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
}

// ONE (round 4, few-pixel levels: N <= 1024, one split): the whole of `softmax(k) v^T` in this launch -- the row maxima are taken
// here (kmax_kernel's pass), and the epilogue normalises the tile and writes the per-image weights / planes (ctx_reduce_kernel's
// pass): one launch instead of three on a latency-bound chain, same values as the three with nsplit = 1.
constexpr float kCtxPlaneScale = 256.0f;
template <bool F16, bool ONE = false>      // F16: sum_n p v on the fp16 matrix cores with two-plane operands (as kvctx16_kernel, attn_kernels.hip)
__global__ void __launch_bounds__(256) ctx_partial_kernel(const float *k, const float *v,
                                                          long long kv_bs, int C, int N,
                                                          const float *kmax, float *S, float *Zp,
                                                          int nsplit, int tiles, float scale = 0.f, float *ctxw = nullptr,
                                                          int Cin_pad = 0, int COP = 0, unsigned short *Ws = nullptr, int img_major = 0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // 2 stages x (k 64x64 + v 64x64)
    __shared__ float rows_s[4 * 64];                               // ONE: row maxima, then the waves' row sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // img_major (ctx_one_launch): the image is the fast grid dimension, so that the tiles^2 workgroups of an image -- every k row block is read by
    // `tiles` of them, every v row block too -- run on ONE XCD and meet in its L2 (workgroups go to the XCDs round robin by linear id)
    const int tile = img_major ? blockIdx.z : blockIdx.x;
    const int dt = tile / tiles, et = tile % tiles;
    const int split = blockIdx.y, b = img_major ? blockIdx.x : blockIdx.z;
    const int d0 = dt * 64, e0 = et * 64;
    const int nps = round_up(ceil_div(N, nsplit), kCtxPch);
    const int n_begin = split * nps;
    const int n_end = min(N, n_begin + nps);
    const float *kb = k + (size_t)b * kv_bs;
    const float *vb = v + (size_t)b * kv_bs;
    const unsigned sm_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(const __attribute__((address_space(3))) float *)sm);

    // this lane's share of a stage: 8 pieces (4 rows x 64 pixels each); piece p of wave w covers
    // rows 4*(w*4 + p%4) .. +3 of tensor p/4 (0 = k, 1 = v)
    const int prow = lane >> 4, pslot = lane & 15;
    auto issue = [&](int n0, int stage) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int tensor = p >> 2;
            const int row = 4 * (wave * 4 + (p & 3)) + prow;
            const int n = n0 + 4 * (pslot ^ (row & 15));
            const int ch = (tensor ? e0 : d0) + row;
            const float *src = g_ctx_zeros;
            if (ch < C && n < n_end) src = (tensor ? vb : kb) + (size_t)ch * N + n;
            ctx_dma16(src, sm_lds + (unsigned)(stage * 8192 + tensor * 4096 + (wave * 4 + (p & 3)) * 256) * 4u);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int half = lane >> 5, jj = lane & 31;
    float mrow[2], zrow[2] = {0.f, 0.f};
    if constexpr (ONE) {        // row maxima of this tile's 64 k rows: four threads per row (N % 4 == 0)
        const int r = tid >> 2, part = tid & 3;
        float m = -INFINITY;
        if (d0 + r < C) {
            const float4 *r4 = reinterpret_cast<const float4 *>(kb + (size_t)(d0 + r) * N);
            // eight independent 16-byte loads per round trip (hipcc keeps two in flight on its own: 32 dependent round trips of
            // ~0.8 us per 1024-pixel row, a third of the launch)
            const int n4 = N >> 2;
            int n = part;
            for (; n + 28 < n4; n += 32) {
                float4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = r4[n + 4 * u];
#pragma unroll
                for (int u = 0; u < 8; ++u) m = fmaxf(m, fmaxf(fmaxf(q[u].x, q[u].y), fmaxf(q[u].z, q[u].w)));
            }
            for (; n < n4; n += 4) {
                const float4 q = r4[n];
                m = fmaxf(m, fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
            }
        }
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        if (part == 0) rows_s[r] = m;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) mrow[i] = (d0 + i * 32 + jj < C) ? rows_s[i * 32 + jj] : 0.f;
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) mrow[i] = (d0 + i * 32 + jj < C) ? kmax[(size_t)b * C + d0 + i * 32 + jj] : 0.f;
    }

    if (n_begin < n_end) issue(n_begin, 0);
    int stage = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += kCtxPch, stage ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (n0 + kCtxPch < n_end) issue(n0 + kCtxPch, stage ^ 1);
        const float *kl = sm + stage * 8192, *vl = kl + 4096;
        const int nb = wave * 16;
        if constexpr (F16) {
            // one K = 16 step per wave and chunk: lane (row jj, half) holds pixels nb + 8 half .. + 7 of its row -- two
            // swizzled 4-pixel groups.  p = exp(k - max) in (0, 1] as {PH, PL, PH2 = PH 2^-11}, v as (h, l'): three products.
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const int g0 = (nb + 8 * half) >> 2;
            const int o0 = 4 * (g0 ^ (jj & 15)), o1 = 4 * ((g0 + 1) ^ (jj & 15));
            h8 ph[2], pl[2], ph2[2], vhi[2], vlo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 k0 = *reinterpret_cast<const float4 *>(kl + (i * 32 + jj) * 64 + o0);
                const float4 k1 = *reinterpret_cast<const float4 *>(kl + (i * 32 + jj) * 64 + o1);
                const float kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const bool pv = n0 + nb + 8 * half + t < n_end;
                    // v_exp_f32 (exp2(x log2 e)), as kvctx16_kernel: <= 2 ulp on a weight in (0, 1]
                    const float pp = pv ? __builtin_amdgcn_exp2f((kk[t] - mrow[i]) * 1.44269504088896341f) : 0.f;
                    zrow[i] += pp;
                    const _Float16 hq = (_Float16)pp;
                    ph[i][t] = hq;
                    pl[i][t] = (_Float16)(pp - (float)hq);
                    ph2[i][t] = (_Float16)((float)hq * (1.0f / 2048.0f));
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 v0 = *reinterpret_cast<const float4 *>(vl + (j * 32 + jj) * 64 + o0);
                const float4 v1 = *reinterpret_cast<const float4 *>(vl + (j * 32 + jj) * 64 + o1);
                const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const _Float16 hq = (_Float16)vv[t];
                    vhi[j][t] = hq;
                    vlo[j][t] = (_Float16)((vv[t] - (float)hq) * 2048.0f);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[i], vhi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph2[i], vlo[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[i], vhi[j], acc[i][j], 0, 0, 0);
                }
        } else
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
            const int nl = nb + ks + half;                       // pixel inside the chunk
            const int off = 4 * ((nl >> 2) ^ (jj & 15)) + (nl & 3);   // rows i*32+jj: (row & 15) == jj & 15
            const bool pv = n0 + nl < n_end;
            float a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = pv ? expf(kl[(i * 32 + jj) * 64 + off] - mrow[i]) : 0.f;
                zrow[i] += a[i];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = vl[(j * 32 + jj) * 64 + off];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    }
    // cross-wave reduction through LDS: red[wave][d 64][e 64] (64 KiB = the whole ring)
    __syncthreads();
    float *red = sm;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                red[(wave * 64 + d) * 64 + j * 32 + jj] = acc[i][j][r];
            }
    __syncthreads();
    if constexpr (ONE) {
        // row sums of exp over all pixels (every workgroup of a tile row has them), then ctx_reduce_kernel's work on this tile:
        // ctxw[d][e] = scale * S / z, and the fp16 planes {WH, WL, WH2} of ctxw 2^8 in units of 8 rows d
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float z = zrow[i] + __shfl_xor(zrow[i], 32);
            if (half == 0) rows_s[wave * 64 + i * 32 + jj] = z;
        }
        __syncthreads();
        for (int idx = tid; idx < 8 * 64; idx += 256) {
            const int g = idx >> 6, e = idx & 63;
            if (d0 + g * 8 >= C || e0 + e >= C) continue;
            float vals[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int d = g * 8 + r, q = d * 64 + e;
                float s2 = red[q] + red[4096 + q] + red[8192 + q] + red[12288 + q];
                const float z = rows_s[d] + rows_s[64 + d] + rows_s[128 + d] + rows_s[192 + d];
                s2 = (d0 + d < C) ? s2 / z * scale : 0.f;
                vals[r] = s2;
                if (d0 + d < Cin_pad) ctxw[((size_t)b * Cin_pad + d0 + d) * COP + e0 + e] = s2;
            }
            if (Ws) {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                h8 wh, wl, wh2;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float w = vals[r] * kCtxPlaneScale;
                    const _Float16 hq = (_Float16)w;
                    wh[r] = hq;
                    wl[r] = (_Float16)(w - (float)hq);
                    wh2[r] = (_Float16)((float)hq * (1.0f / 2048.0f));
                }
                const int dd = d0 + g * 8, q = dd >> 4, kh = (dd >> 3) & 1;
                uint4 *dst = reinterpret_cast<uint4 *>(Ws) + (size_t)b * (C / 16) * 6 * C;
                dst[(size_t)((q * 3 + 0) * 2 + kh) * C + e0 + e] = __builtin_bit_cast(uint4, wh);
                dst[(size_t)((q * 3 + 1) * 2 + kh) * C + e0 + e] = __builtin_bit_cast(uint4, wl);
                dst[(size_t)((q * 3 + 2) * 2 + kh) * C + e0 + e] = __builtin_bit_cast(uint4, wh2);
            }
        }
        return;
    }
    float *out = S + (((size_t)b * nsplit + split) * C) * C;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int d = idx >> 6, e = idx & 63;
        if (d0 + d < C && e0 + e < C)
            out[(size_t)(d0 + d) * C + e0 + e] = red[idx] + red[4096 + idx] + red[8192 + idx] + red[12288 + idx];
    }
    if (et == 0) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float z = zrow[i] + __shfl_xor(zrow[i], 32);
            if (half == 0) red[wave * 64 + i * 32 + jj] = z;
        }
        __syncthreads();
        if (tid < 64 && d0 + tid < C)
            Zp[((size_t)b * nsplit + split) * C + d0 + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
    }
}

// Same quantities for pixel counts that are not a multiple of 4 (tiny odd feature maps only): plain
// VALU loops, one workgroup per (d, split, image), threads over e.
__global__ void __launch_bounds__(256) ctx_partial_generic_kernel(const float *k, const float *v,
                                                                  long long kv_bs, int C, int N,
                                                                  const float *kmax, float *S,
                                                                  float *Zp, int nsplit) {
    const int d = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int nps = round_up(ceil_div(N, nsplit), kCtxPch);
    const int n_begin = split * nps, n_end = min(N, n_begin + nps);
    const float *kr = k + (size_t)b * kv_bs + (size_t)d * N;
    const float m = kmax[(size_t)b * C + d];
    for (int e = threadIdx.x; e < C; e += blockDim.x) {
        const float *vr = v + (size_t)b * kv_bs + (size_t)e * N;
        float s = 0.f, z = 0.f;
        for (int n = n_begin; n < n_end; ++n) {
            const float p = expf(kr[n] - m);
            s += p * vr[n];
            z += p;
        }
        S[(((size_t)b * nsplit + split) * C + d) * C + e] = s;
        if (e == 0) Zp[((size_t)b * nsplit + split) * C + d] = z;
    }
}

// ONE launch for kstats + partial + reduce (see ctx_partial_kernel, ONE): N % 4 == 0, C % 64 == 0, Cin_pad == COP == C
hipError_t ctx_one_launch(const float *k, const float *v, long long kv_bs, int C, int N, float scale, float *ctxw, int Cin_pad, int COP,
                          unsigned short *Ws, int B, hipStream_t st, int f16) {
    if ((N & 3) || (C % 64) || Cin_pad != C || COP != C) return hipErrorInvalidValue;
    const int tiles = C / 64;
    const size_t lds = sizeof(float) * 4 * 64 * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)ctx_partial_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)ctx_partial_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int img_major = (tiles > 1 && B % 8 == 0 && !dev_env("CDC_CTX_NO_IMG_MAJOR")) ? 1 : 0;      // (round 6; the switch: A/B)
    const dim3 grid = img_major ? dim3((unsigned)B, 1, (unsigned)(tiles * tiles)) : dim3((unsigned)(tiles * tiles), 1, (unsigned)B);
    if (f16)
        hipLaunchKernelGGL((ctx_partial_kernel<true, true>), grid, dim3(256), lds, st, k, v, kv_bs, C, N, nullptr, nullptr, nullptr, 1, tiles, scale,
                           ctxw, Cin_pad, COP, Ws, img_major);
    else
        hipLaunchKernelGGL((ctx_partial_kernel<false, true>), grid, dim3(256), lds, st, k, v, kv_bs, C, N, nullptr, nullptr, nullptr, 1, tiles, scale,
                           ctxw, Cin_pad, COP, Ws, img_major);
    return hipGetLastError();
}

hipError_t ctx_partial_launch(const float *k, const float *v, long long kv_bs, int C, int N,
                              const float *kmax, float *S, float *Zp, int nsplit, int B,
                              hipStream_t st, int f16) {
    if (N & 3) {                                 // 16-byte DMA pieces need N % 4 == 0
        hipLaunchKernelGGL(ctx_partial_generic_kernel, dim3(C, nsplit, B), dim3(C >= 256 ? 256 : 64), 0, st,
                           k, v, kv_bs, C, N, kmax, S, Zp, nsplit);
        return hipGetLastError();
    }
    const int tiles = ceil_div(C, 64);
    const size_t lds = sizeof(float) * 4 * 64 * 64;      // ring == reduction buffer
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)ctx_partial_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)ctx_partial_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    // (image-major as in ctx_one_launch: the tiles of an image and split share its k / v row blocks)
    const int img_major = (tiles > 1 && B % 8 == 0 && !dev_env("CDC_CTX_NO_IMG_MAJOR")) ? 1 : 0;
    const dim3 grid = img_major ? dim3((unsigned)B, (unsigned)nsplit, (unsigned)(tiles * tiles)) : dim3((unsigned)(tiles * tiles), (unsigned)nsplit, (unsigned)B);
    if (f16)
        hipLaunchKernelGGL(ctx_partial_kernel<true>, grid, dim3(256), lds, st, k, v,
                           kv_bs, C, N, kmax, S, Zp, nsplit, tiles, 0.f, nullptr, 0, 0, nullptr, img_major);
    else
        hipLaunchKernelGGL(ctx_partial_kernel<false>, grid, dim3(256), lds, st, k, v,
                           kv_bs, C, N, kmax, S, Zp, nsplit, tiles, 0.f, nullptr, 0, 0, nullptr, img_major);
    return hipGetLastError();
}

// ctxw[b][d][e] = scale * (sum_split S[b][split][d][e]) / ksum[b][d]   (q*scale folded in, :132)
// One workgroup = 8 rows d (= 8 input channels of the per-image 1x1 convolution out[e] = sum_d ctxw[d][e] q[d]): the
// 8 values of a thread are one 16-byte A-operand unit, so the kernel can also emit the convolution's weight planes
// (Ws: fp16 {WH, WL, WH2} of ctxw 2^8, layout [C/16][3][2][C][8] per image -- conv_split_kernel.h AR = 1).
__global__ void __launch_bounds__(256) ctx_reduce_kernel(const float *S, const float *ksum, int C,
                                                         int nsplit, float scale, float *ctxw,
                                                         int Cin_pad, int COP, unsigned short *Ws) {
    const int d0 = blockIdx.x * 8, b = blockIdx.y;
    float z[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        z[r] = 0.f;
        if (d0 + r < C)
            for (int sp = 0; sp < nsplit; ++sp) z[r] += ksum[((size_t)b * nsplit + sp) * C + d0 + r];
    }
    for (int e = threadIdx.x; e < COP; e += blockDim.x) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float s = 0.f;
            if (e < C && d0 + r < C) {
                for (int sp = 0; sp < nsplit; ++sp)
                    s += S[(((size_t)b * nsplit + sp) * C + d0 + r) * C + e];
                s = s / z[r] * scale;
            }
            v[r] = s;
            if (d0 + r < Cin_pad) ctxw[((size_t)b * Cin_pad + d0 + r) * COP + e] = s;
        }
        if (Ws && e < C && d0 < C) {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 wh, wl, wh2;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float w = v[r] * kCtxPlaneScale;
                const _Float16 hq = (_Float16)w;
                wh[r] = hq;
                wl[r] = (_Float16)(w - (float)hq);
                wh2[r] = (_Float16)((float)hq * (1.0f / 2048.0f));
            }
            const int q = d0 >> 4, kh = (d0 >> 3) & 1;
            uint4 *dst = reinterpret_cast<uint4 *>(Ws) + (size_t)b * (C / 16) * 6 * C;
            dst[(size_t)((q * 3 + 0) * 2 + kh) * C + e] = __builtin_bit_cast(uint4, wh);
            dst[(size_t)((q * 3 + 1) * 2 + kh) * C + e] = __builtin_bit_cast(uint4, wl);
            dst[(size_t)((q * 3 + 2) * 2 + kh) * C + e] = __builtin_bit_cast(uint4, wh2);
        }
    }
}

hipError_t ctx_reduce_launch(const float *S, const float *ksum, int C, int nsplit, float scale,
                             float *ctxw, int Cin_pad, int COP, int B, hipStream_t st, unsigned short *Ws) {
    hipLaunchKernelGGL(ctx_reduce_kernel, dim3(ceil_div(Cin_pad, 8), B), dim3(COP >= 256 ? 256 : 64), 0, st, S,
                       ksum, C, nsplit, scale, ctxw, Cin_pad, COP, Ws);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Attention output folded into per-image 1x1 weights (levels with many pixels):
//   y = to_out(ctx^T (q*scale)) + x ,  q = Wq LN(x)      (network_components.py:128-139)
//     = M' LN(x) + b_out + x ,  M'[c][ci] = scale * sum_d (sum_e Wo[c][e] ctx[d][e]) Wq[d][ci]
// so neither q nor the attention output is ever materialised.  Two tiny per-image GEMMs:
//   R1: T1[b][d][c]  = sum_e (sum_split S[b][split][d][e]) / ksum[b][d] * WoT[e][c]
//   R2: Mt[b][ci][c] = scale * sum_d WqT[ci][d] * T1[b][d][c]     (packed 1x1 weights [Cin_pad][COP])
// ---------------------------------------------------------------------------------------------
static_assert(true, "");
constexpr int kFoldRows = 8;     // rows per workgroup: each WoT / T1 element is reused 8x from registers

// R0: ctxn[b][d][e] = (sum_split S) / (sum_split Zp).  One workgroup per (image, row d): the per-split rescale
// factors exp(M[split][d] - max) are computed once per row (not once per element), the C elements of the row are
// summed by C x SPF threads (SPF interleaved split subsets, combined through LDS) with independent loads.
__global__ void __launch_bounds__(512) ctx_r0_kernel(const float *S, const float *Zp, int C, int nsplit,
                                                     float *ctxn, const float *M, int SPF, int transposed) {
    extern __shared__ float r0s[];                    // [nsplit] factors, [nsplit] Z * factor, [SPF][C] partial sums
    float *fs = r0s, *zf = r0s + nsplit, *red = r0s + 2 * nsplit;
    const int b = blockIdx.y, d = blockIdx.x;
    const int tid = threadIdx.x, e = tid % C, q = tid / C;
    // All three kinds of global loads are issued before the first barrier (round 4): the row maxima, the partial row sums Z and this
    // thread's first kR0Pre partial-context values.  (M -> barrier -> Z -> barrier -> S were three dependent round trips: 25 us per
    // launch, the longest of the fold's three.)
    constexpr int kR0Pre = 8;
    const float *sp0 = S + ((size_t)b * nsplit * C + d) * C + e;
    const size_t ss = (size_t)C * C;
    float sv[kR0Pre];
#pragma unroll
    for (int i = 0; i < kR0Pre; ++i) { const int sp = q + i * SPF; sv[i] = sp < nsplit ? sp0[(size_t)sp * ss] : 0.f; }
    for (int sp = tid; sp < nsplit; sp += blockDim.x) {
        fs[sp] = M ? M[((size_t)b * nsplit + sp) * C + d] : 0.f;
        zf[sp] = Zp[((size_t)b * nsplit + sp) * C + d];
    }
    __syncthreads();
    float mg = -INFINITY;
    if (M)          // every split carries its own row maximum (kvctx kernels): bring them to the common one
        for (int sp = 0; sp < nsplit; ++sp) mg = fmaxf(mg, fs[sp]);
    __syncthreads();
    for (int sp = tid; sp < nsplit; sp += blockDim.x) {
        const float f = M ? expf(fs[sp] - mg) : 1.0f;
        fs[sp] = f;
        zf[sp] = zf[sp] * f;
    }
    __syncthreads();
    float z = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) z += zf[sp];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < kR0Pre; ++i) { const int sp = q + i * SPF; if (sp < nsplit) acc += sv[i] * fs[sp]; }     // same order: sp = q, q + SPF, ...
#pragma unroll 8
    for (int sp = q + kR0Pre * SPF; sp < nsplit; sp += SPF) acc += sp0[(size_t)sp * ss] * fs[sp];
    if (SPF > 1) {
        red[q * C + e] = acc;
        __syncthreads();
        if (q == 0)
            for (int k = 1; k < SPF; ++k) acc += red[k * C + e];
    }
    // transposed: [e][d], the A-operand order of fold_r1_mfma_kernel (its lanes walk d)
    if (q == 0) ctxn[transposed ? ((size_t)b * C + e) * C + d : ((size_t)b * C + d) * C + e] = acc / z;
}

__global__ void __launch_bounds__(256) ctx_r1_kernel(const float *ctxn, int C, const float *WoT, float *T1) {
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [C][kFoldRows]: normalised ctx rows d0..d0+7, e-major
    const int d0 = blockIdx.x * kFoldRows, b = blockIdx.y;
    for (int idx = threadIdx.x; idx < kFoldRows * C; idx += blockDim.x) {
        const int r = idx / C, e = idx - r * C, d = d0 + r;
        rows[e * kFoldRows + r] = d < C ? ctxn[((size_t)b * C + d) * C + e] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc[kFoldRows];
#pragma unroll
        for (int r = 0; r < kFoldRows; ++r) acc[r] = 0.f;
        // C % 8 == 0 on the folded levels; batches of 8 independent loads hide the L2 latency of this
        // otherwise serial (load -> 8 FMAs) chain
#pragma unroll 8
        for (int e = 0; e < C; ++e) {
            const float w = WoT[(size_t)e * C + c];
            const float4 ra = *reinterpret_cast<const float4 *>(rows + e * kFoldRows);
            const float4 rb = *reinterpret_cast<const float4 *>(rows + e * kFoldRows + 4);
            acc[0] += ra.x * w; acc[1] += ra.y * w; acc[2] += ra.z * w; acc[3] += ra.w * w;
            acc[4] += rb.x * w; acc[5] += rb.y * w; acc[6] += rb.z * w; acc[7] += rb.w * w;
        }
#pragma unroll
        for (int r = 0; r < kFoldRows; ++r)
            if (d0 + r < C) T1[((size_t)b * C + d0 + r) * C + c] = acc[r];
    }
}

// ws_f16: 0 three bf16 planes (exact split); 1 fp16 planes {WH, WL, WH2} of M' * 2^8 (conv_split_kernel.h AR = 1; the
// consumer multiplies its accumulators by 2^-8.  |M'| >= 255 overflows to inf -> NaN -> the decode's range guard).
constexpr float kFoldPlaneScale = 256.0f;
__global__ void __launch_bounds__(256) ctx_r2_kernel(const float *T1, const float *WqT, int C,
                                                     float scale, const float *ln_g, float *Mt,
                                                     int Cin_pad, int COP, unsigned short *Ws, int ws_f16) {
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [C][kFoldRows]: WqT rows ci0..ci0+7, d-major
    const int ci0 = blockIdx.x * kFoldRows, b = blockIdx.y;
    for (int idx = threadIdx.x; idx < kFoldRows * C; idx += blockDim.x) {
        const int r = idx / C, d = idx - r * C;
        rows[d * kFoldRows + r] = (ci0 + r < C) ? WqT[(size_t)(ci0 + r) * C + d] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < COP; c += blockDim.x) {
        float acc[kFoldRows];
#pragma unroll
        for (int r = 0; r < kFoldRows; ++r) acc[r] = 0.f;
        if (c < C) {
            const float *t = T1 + (size_t)b * C * C + c;
#pragma unroll 8
            for (int d = 0; d < C; ++d) {
                const float w = t[(size_t)d * C];
                const float4 ra = *reinterpret_cast<const float4 *>(rows + d * kFoldRows);
                const float4 rb = *reinterpret_cast<const float4 *>(rows + d * kFoldRows + 4);
                acc[0] += ra.x * w; acc[1] += ra.y * w; acc[2] += ra.z * w; acc[3] += ra.w * w;
                acc[4] += rb.x * w; acc[5] += rb.y * w; acc[6] += rb.z * w; acc[7] += rb.w * w;
            }
        }
#pragma unroll
        for (int r = 0; r < kFoldRows; ++r) {
            acc[r] = acc[r] * scale * (ci0 + r < C ? ln_g[ci0 + r] : 0.f);   // PreNorm gain folded in (LNMODE 2)
            if (ci0 + r < Cin_pad) Mt[((size_t)b * Cin_pad + ci0 + r) * COP + c] = acc[r];
        }
        if (Ws && c < C && ws_f16) {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 wh, wl, wh2;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float v = acc[r] * kFoldPlaneScale;
                const _Float16 hq = (_Float16)v;
                wh[r] = hq;
                wl[r] = (_Float16)(v - (float)hq);
                wh2[r] = (_Float16)((float)hq * (1.0f / 2048.0f));
            }
            const int q = ci0 >> 4, kh = (ci0 >> 3) & 1;
            uint4 *dst = reinterpret_cast<uint4 *>(Ws) + (size_t)b * (C / 16) * 6 * C;
            dst[(size_t)((q * 3 + 0) * 2 + kh) * C + c] = __builtin_bit_cast(uint4, wh);
            dst[(size_t)((q * 3 + 1) * 2 + kh) * C + c] = __builtin_bit_cast(uint4, wl);
            dst[(size_t)((q * 3 + 2) * 2 + kh) * C + c] = __builtin_bit_cast(uint4, wh2);
        } else if (Ws && c < C) {
            // the same 8 input channels x this output channel as three bf16 planes in the A-operand order of
            // lnconv_kernel: [ci/16][plane][(ci/8)&1][co][8] -- the 8 rows of this workgroup are one 16-byte unit
            unsigned hh[8], mm[8], ll[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                hh[r] = __float_as_uint(acc[r]) & 0xFFFF0000u;
                const float r1 = acc[r] - __uint_as_float(hh[r]);
                mm[r] = __float_as_uint(r1) & 0xFFFF0000u;
                ll[r] = __float_as_uint(r1 - __uint_as_float(mm[r]));
            }
            uint4 v[3];
            v[0] = make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]);
            v[1] = make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]);
            v[2] = make_uint4((ll[0] >> 16) | (ll[1] & 0xFFFF0000u), (ll[2] >> 16) | (ll[3] & 0xFFFF0000u),
                              (ll[4] >> 16) | (ll[5] & 0xFFFF0000u), (ll[6] >> 16) | (ll[7] & 0xFFFF0000u));
            const int q = ci0 >> 4, kh = (ci0 >> 3) & 1;
            uint4 *dst = reinterpret_cast<uint4 *>(Ws) + (size_t)b * (C / 16) * 6 * C;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[(size_t)((q * 3 + pl) * 2 + kh) * C + c] = v[pl];
        }
    }
}

// R3: per-image bias of the folded convolution:  b_out + M' b_ln = b_out + scale * sum_d u[d] T1[b][d][c]
// with u = Wq b_ln (a per-layer constant computed at weight load)
__global__ void __launch_bounds__(256) ctx_r3_kernel(const float *T1, const float *u, const float *b_out,
                                                     float scale, float *biasB, int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float *t = T1 + (size_t)b * C * C + c;
    float acc = 0.f;
#pragma unroll 8
    for (int d = 0; d < C; ++d) acc += u[d] * t[(size_t)d * C];
    biasB[(size_t)b * C + c] = b_out[c] + scale * acc;
}

// ---- round 4: R1 / R2 (+ R3) on the f32 matrix cores ------------------------------------------------------------------
// The two per-image C x C x C products of the fold are plain fp32 GEMMs; as FMA loops they ran at 4 - 10 TFLOP/s and made the fold
// (0.33 ms per DDIM iteration over five attention levels) the slowest "small" step.  v_mfma_f32_32x32x2_f32 multiplies fp32
// operands exactly as an FMA does, so nothing is split: one wave per 32 x 32 block of the result, K walked two at a time, both
// operand blocks staged once through LDS in the order the lanes need them (lane = (k parity, row / column)).
//   R1: T1[b][d][c]  = sum_e ctxnT[b][e][d] * WoT[e][c]                  (ctx_r0_kernel writes the normalised context transposed)
//   R2: Mt[b][ci][c] = scale g[ci] * sum_d Wq[d][ci] * T1[b][d][c]       (+ the planes of M' for the split convolution)
//   R3: bias[b][c]   = b_out[c] + scale * sum_d u[d] T1[b][d][c]         (rides in R2's K loop: the waves of block row 0)
// Both operand blocks (K x 32 floats each) are fetched with 16-byte loads, all in flight at once, into LDS; the K loop then runs
// from LDS.  (Fetching the operands per MFMA step straight from L2 was measured slower than the FMA kernels it replaced: 192 dependent
// 4-byte loads per wave.)
// Staging by 16-byte LDS-DMA (global_load_lds_dwordx4: lane l's 16 bytes land at M0 + 16 l, which is exactly lds[k0 + l / 8][4 (l % 8)]):
// no staging registers, so every load of BOTH operand blocks is in flight at once whatever hipcc schedules.  (As register loads +
// ds_write the staging loop came out as groups of four loads with a full wait each -- six dependent round trips per operand at C = 192 --
// and, fully unrolled, as one load / wait / ds_write at a time: 14 - 30 us per launch for 1 - 3 us of matrix work.)
// M0 handling as in conv_kernel.h dma_b128 (explicit wait state after the M0 write; M0 saved and restored).
__device__ __forceinline__ void fold_dma16(const float *g, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_byte)) : "memory");
}
// CT > 0: the channel count as a compile-time constant (64 / 128 / 192, the folded levels of the full-width models): straight-line code.
template <int CT>
__device__ __forceinline__ void fold_stage2(float *As, const float *a_src, float *Bs, const float *b_src, int ld, int K) {
    const int lane = threadIdx.x, r8 = lane >> 3, c4 = (lane & 7) * 4;
    const unsigned a_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)As;
    const unsigned b_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)Bs;
    const float *ap = a_src + (size_t)r8 * ld + c4, *bp = b_src + (size_t)r8 * ld + c4;
    if constexpr (CT > 0) {
#pragma unroll
        for (int k0 = 0; k0 < CT; k0 += 8) fold_dma16(ap + (size_t)k0 * CT, a_lds + (unsigned)k0 * 128u);
#pragma unroll
        for (int k0 = 0; k0 < CT; k0 += 8) fold_dma16(bp + (size_t)k0 * CT, b_lds + (unsigned)k0 * 128u);
    } else {
        for (int k0 = 0; k0 < K; k0 += 8) fold_dma16(ap + (size_t)k0 * ld, a_lds + (unsigned)k0 * 128u);
        for (int k0 = 0; k0 < K; k0 += 8) fold_dma16(bp + (size_t)k0 * ld, b_lds + (unsigned)k0 * 128u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // hipcc does not count these loads
}
template <int CT>
__global__ void __launch_bounds__(64) fold_r1_mfma_kernel(const float *ctxnT, const float *WoT, float *T1, int Crt) {
    extern __shared__ __attribute__((aligned(16))) float fold_lds[];       // A block [C][32], B block [C][32]
    const int C = CT > 0 ? CT : Crt;
    float *As = fold_lds, *Bs = fold_lds + C * 32;
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, b = blockIdx.z;
    // A[m = d][k = e] = ctxnT[e][d] ; B[k = e][n = c]
    fold_stage2<CT>(As, ctxnT + (size_t)b * C * C + m0, Bs, WoT + n0, C, C);
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (CT > 0) {
#pragma unroll
        for (int k2 = 0; k2 < CT / 2; ++k2) {
            const int k = 2 * k2 + half;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * 32 + j], Bs[k * 32 + j], acc, 0, 0, 0);
        }
    } else {
#pragma unroll 8
        for (int k = half; k < C; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * 32 + j], Bs[k * 32 + j], acc, 0, 0, 0);
    }
    float *o = T1 + (size_t)b * C * C + (size_t)(m0 + 4 * half) * C + n0 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * C] = acc[r];
}

template <int CT>
__global__ void __launch_bounds__(64) fold_r2_mfma_kernel(const float *T1, const float *Wq, int Crt, float scale, const float *ln_g,
                                                          float *Mt, int Cin_pad, int COP, unsigned short *Ws, int ws_f16,
                                                          const float *u, const float *b_out, float *biasB) {
    extern __shared__ __attribute__((aligned(16))) float fold_lds[];
    const int C = CT > 0 ? CT : Crt;
    float *As = fold_lds, *Bs = fold_lds + C * 32;
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, b = blockIdx.z;
    // A[m = ci][k = d] = Wq[d][ci] ; B[k = d][n = c]
    const bool with_bias = blockIdx.y == 0;
    float *us = fold_lds + 2 * C * 32;                       // u [C] (block row 0 only)
    if (with_bias)
        for (int k = lane; k < C; k += 64) us[k] = u[k];     // (u[k] fetched inside the K loop was a dependent global load per step)
    fold_stage2<CT>(As, Wq + m0, Bs, T1 + (size_t)b * C * C + n0, C, C);
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    if constexpr (CT > 0) {
#pragma unroll
        for (int k2 = 0; k2 < CT / 2; ++k2) {
            const int k = 2 * k2 + half;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * 32 + j], Bs[k * 32 + j], acc, 0, 0, 0);
        }
    } else {
#pragma unroll 8
        for (int k = half; k < C; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * 32 + j], Bs[k * 32 + j], acc, 0, 0, 0);
    }
    if (with_bias) {                                         // same summation order as inside the K loop: k = half, half + 2, ...
        if constexpr (CT > 0) {
#pragma unroll
            for (int k2 = 0; k2 < CT / 2; ++k2) { const int k = 2 * k2 + half; bsum += us[k] * Bs[k * 32 + j]; }
        } else {
            for (int k = half; k < C; k += 2) bsum += us[k] * Bs[k * 32 + j];
        }
    }
    if (with_bias) {
        bsum += __shfl_xor(bsum, 32);
        if (half == 0) biasB[(size_t)b * C + n0 + j] = b_out[n0 + j] + scale * bsum;
    }
    const int c = n0 + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ci0 = m0 + 8 * g + 4 * half;               // this lane's 4 consecutive input channels of the 8-channel unit
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = acc[4 * g + i] * scale * ln_g[ci0 + i];   // PreNorm gain folded in (LNMODE 2)
            Mt[((size_t)b * Cin_pad + ci0 + i) * COP + c] = v[i];
        }
        if (Ws) {
            const int q = (m0 + 8 * g) >> 4, kh = ((m0 + 8 * g) >> 3) & 1;
            uint2 *dst = reinterpret_cast<uint2 *>(reinterpret_cast<uint4 *>(Ws) + (size_t)b * (C / 16) * 6 * C) + half;   // 8-byte half of a unit
            if (ws_f16) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 wh, wl, wh2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = v[i] * kFoldPlaneScale;
                    const _Float16 hq = (_Float16)x;
                    wh[i] = hq; wl[i] = (_Float16)(x - (float)hq); wh2[i] = (_Float16)((float)hq * (1.0f / 2048.0f));
                }
                dst[2 * ((size_t)((q * 3 + 0) * 2 + kh) * C + c)] = __builtin_bit_cast(uint2, wh);
                dst[2 * ((size_t)((q * 3 + 1) * 2 + kh) * C + c)] = __builtin_bit_cast(uint2, wl);
                dst[2 * ((size_t)((q * 3 + 2) * 2 + kh) * C + c)] = __builtin_bit_cast(uint2, wh2);
            } else {
                unsigned hh[4], mm[4], ll[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hh[i] = __float_as_uint(v[i]) & 0xFFFF0000u;
                    const float r1 = v[i] - __uint_as_float(hh[i]);
                    mm[i] = __float_as_uint(r1) & 0xFFFF0000u;
                    ll[i] = __float_as_uint(r1 - __uint_as_float(mm[i]));
                }
                dst[2 * ((size_t)((q * 3 + 0) * 2 + kh) * C + c)] = make_uint2((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3]);
                dst[2 * ((size_t)((q * 3 + 1) * 2 + kh) * C + c)] = make_uint2((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3]);
                dst[2 * ((size_t)((q * 3 + 2) * 2 + kh) * C + c)] = make_uint2((ll[0] >> 16) | (ll[1] & 0xFFFF0000u), (ll[2] >> 16) | (ll[3] & 0xFFFF0000u));
            }
        }
    }
}

// R1 and R2 of the fold in ONE launch (round 5; the compile-time channel counts): a workgroup owns a 32-column block c of M' and has one wave
// per 32-row block.  R2's tile (ci block, c block) needs T1[:, c block] -- all d, these 32 columns -- which is exactly what the
// workgroup's waves produce in R1 (wave w: rows d = 32 w ...), so the slab goes through LDS instead of through a launch boundary
// (a dependent launch costs 6.5 us on this part; the two products are 2 - 8 k cycles).  B operands: the WoT column block by LDS-DMA
// (R1), the T1 slab (R2); A operands (ctxnT rows / Wq rows of the wave's block) straight from global memory, all of a product's loads
// in flight at once.  Same products in the same order as fold_r1 / fold_r2_mfma_kernel (k = half, half + 2, ...): the same bits.
template <int CT>
__global__ void __launch_bounds__(64 * (CT / 32)) fold_r12_mfma_kernel(const float *ctxnT, const float *WoT, const float *Wq, float scale,
                                                                       const float *ln_g, float *Mt, int Cin_pad, int COP,
                                                                       unsigned short *Ws, int ws_f16, const float *u,
                                                                       const float *b_out, float *biasB) {
    constexpr int C = CT, NW = CT / 32;
    extern __shared__ __attribute__((aligned(16))) float fold_lds[];       // Bs [C][32] | Ts [C][32] | us [C]
    float *Bs = fold_lds, *Ts = fold_lds + C * 32, *us = Ts + C * 32;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 32, b = blockIdx.y, m0 = wave * 32;
    {
        const int r8 = lane >> 3, c4 = (lane & 7) * 4;
        const unsigned b_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)Bs;
        const float *bp = WoT + n0 + (size_t)r8 * C + c4;
        for (int k0 = 8 * wave; k0 < C; k0 += 8 * NW) fold_dma16(bp + (size_t)k0 * C, b_lds + (unsigned)k0 * 128u);
    }
    for (int k = tid; k < C; k += 64 * NW) us[k] = u[k];
    float av[C / 2];
    {   // R1: A[m = d][k = e] = ctxnT[e][d]
        // (uniform base + 32-bit lane offset: the saddr form -- a 64-bit vector address per load would take 2 x C / 2 registers)
        const char *a1 = reinterpret_cast<const char *>(ctxnT + (size_t)b * C * C + m0);
        const unsigned lo = (unsigned)(half * C + j) * 4u;
#pragma unroll
        for (int k2 = 0; k2 < C / 2; ++k2) av[k2] = *reinterpret_cast<const float *>(a1 + (size_t)k2 * 2 * C * 4 + lo);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // hipcc does not count the DMA loads
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < C / 2; ++k2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k2], Bs[(2 * k2 + half) * 32 + j], acc, 0, 0, 0);
        if ((k2 & 15) == 15) __builtin_amdgcn_sched_barrier(0);          // (the LDS operands 16 at a time, not all C / 2 beside the A operands)
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Ts[(m0 + 4 * half + (r & 3) + 8 * (r >> 2)) * 32 + j] = acc[r];     // T1[d][c]
    {   // R2: A[m = ci][k = d] = Wq[d][ci]
        const char *a2 = reinterpret_cast<const char *>(Wq + m0);
        const unsigned lo = (unsigned)(half * C + j) * 4u;
#pragma unroll
        for (int k2 = 0; k2 < C / 2; ++k2) av[k2] = *reinterpret_cast<const float *>(a2 + (size_t)k2 * 2 * C * 4 + lo);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < C / 2; ++k2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k2], Ts[(2 * k2 + half) * 32 + j], acc, 0, 0, 0);
        if ((k2 & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
    if (wave == 0) {                                          // bias row: same summation order as inside the K loop
        float bsum = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < C / 2; ++k2) { const int k = 2 * k2 + half; bsum += us[k] * Ts[k * 32 + j]; }
        bsum += __shfl_xor(bsum, 32);
        if (half == 0) biasB[(size_t)b * C + n0 + j] = b_out[n0 + j] + scale * bsum;
    }
    const int c = n0 + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ci0 = m0 + 8 * g + 4 * half;               // this lane's 4 consecutive input channels of the 8-channel unit
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = acc[4 * g + i] * scale * ln_g[ci0 + i];   // PreNorm gain folded in (LNMODE 2)
            Mt[((size_t)b * Cin_pad + ci0 + i) * COP + c] = v[i];
        }
        if (Ws) {
            const int q = (m0 + 8 * g) >> 4, kh = ((m0 + 8 * g) >> 3) & 1;
            uint2 *dst = reinterpret_cast<uint2 *>(reinterpret_cast<uint4 *>(Ws) + (size_t)b * (C / 16) * 6 * C) + half;   // 8-byte half of a unit
            if (ws_f16) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 wh, wl, wh2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = v[i] * kFoldPlaneScale;
                    const _Float16 hq = (_Float16)x;
                    wh[i] = hq; wl[i] = (_Float16)(x - (float)hq); wh2[i] = (_Float16)((float)hq * (1.0f / 2048.0f));
                }
                dst[2 * ((size_t)((q * 3 + 0) * 2 + kh) * C + c)] = __builtin_bit_cast(uint2, wh);
                dst[2 * ((size_t)((q * 3 + 1) * 2 + kh) * C + c)] = __builtin_bit_cast(uint2, wl);
                dst[2 * ((size_t)((q * 3 + 2) * 2 + kh) * C + c)] = __builtin_bit_cast(uint2, wh2);
            } else {
                unsigned hh[4], mm[4], ll[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hh[i] = __float_as_uint(v[i]) & 0xFFFF0000u;
                    const float r1 = v[i] - __uint_as_float(hh[i]);
                    mm[i] = __float_as_uint(r1) & 0xFFFF0000u;
                    ll[i] = __float_as_uint(r1 - __uint_as_float(mm[i]));
                }
                dst[2 * ((size_t)((q * 3 + 0) * 2 + kh) * C + c)] = make_uint2((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3]);
                dst[2 * ((size_t)((q * 3 + 1) * 2 + kh) * C + c)] = make_uint2((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3]);
                dst[2 * ((size_t)((q * 3 + 2) * 2 + kh) * C + c)] = make_uint2((ll[0] >> 16) | (ll[1] & 0xFFFF0000u), (ll[2] >> 16) | (ll[3] & 0xFFFF0000u));
            }
        }
    }
}

hipError_t ctx_fold_launch(const float *S, const float *ksum, int C, int nsplit, float scale,
                           const float *WoT, const float *WqT, float *T1, float *Mt, int Cin_pad,
                           int COP, const float *ln_g, const float *u, const float *b_out,
                           float *biasB, int B, hipStream_t st, const float *M, unsigned short *Ws, int ws_f16, const float *Wq) {
    const int blk = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    // Mt doubles as scratch for the normalised context (C*C <= Cin_pad*COP) until R2 overwrites it
    int spf = 1;
    while (2 * spf * C <= 512 && 2 * spf <= nsplit) spf *= 2;
    // whole 32 x 32 blocks (the folded levels of the full-width models: C = 64, 128, 192): R1 / R2 / R3 on the f32 matrix cores
    const bool mfma = Wq && (C % 32) == 0 && Cin_pad == C && COP == C;
    // (the one-launch fold reads the normalised context of ALL column blocks while other workgroups already write M': it lives in T1 there)
    const bool two_launches = dev_env("CDC_FOLD_TWO_LAUNCHES") != nullptr;      // (A/B and tests: R1 and R2 as the two launches of round 4)
    const bool one_launch = mfma && !two_launches && (C == 64 || C == 128 || C == 192);
    hipLaunchKernelGGL(ctx_r0_kernel, dim3(C, B), dim3(C * spf), sizeof(float) * (2 * nsplit + spf * C), st, S, ksum, C, nsplit,
                       one_launch ? T1 : Mt, M, spf, mfma ? 1 : 0);
    if (mfma) {
        const size_t lds = sizeof(float) * (2 * C * 32 + C);
        const dim3 grid(C / 32, C / 32, B);
#define CDC_FOLD_LAUNCH(CTV)                                                                                                        \
        do {                                                                                                                        \
            hipLaunchKernelGGL(fold_r1_mfma_kernel<CTV>, grid, dim3(64), lds, st, Mt, WoT, T1, C);                                      \
            hipLaunchKernelGGL(fold_r2_mfma_kernel<CTV>, grid, dim3(64), lds, st, T1, Wq, C, scale, ln_g, Mt, Cin_pad, COP, Ws, ws_f16, \
                               u, b_out, biasB);                                                                                    \
        } while (0)
#define CDC_FOLD12_LAUNCH(CTV)                                                                                                      \
        hipLaunchKernelGGL(fold_r12_mfma_kernel<CTV>, dim3(CTV / 32, B), dim3(64 * (CTV / 32)), sizeof(float) * (2 * CTV * 32 + CTV), st, T1, WoT, Wq,  \
                           scale, ln_g, Mt, Cin_pad, COP, Ws, ws_f16, u, b_out, biasB)
        if (C == 64 && !two_launches) CDC_FOLD12_LAUNCH(64);
        else if (C == 128 && !two_launches) CDC_FOLD12_LAUNCH(128);
        else if (C == 192 && !two_launches) CDC_FOLD12_LAUNCH(192);
        else if (C == 64) CDC_FOLD_LAUNCH(64);
        else if (C == 128) CDC_FOLD_LAUNCH(128);
        else if (C == 192) CDC_FOLD_LAUNCH(192);
        else CDC_FOLD_LAUNCH(0);                        // (the run-time-count form: the other channel counts)
#undef CDC_FOLD12_LAUNCH
#undef CDC_FOLD_LAUNCH
        return hipGetLastError();
    }
    hipLaunchKernelGGL(ctx_r1_kernel, dim3(ceil_div(C, kFoldRows), B), dim3(blk),
                       sizeof(float) * kFoldRows * C, st, Mt, C, WoT, T1);
    hipLaunchKernelGGL(ctx_r2_kernel, dim3(ceil_div(Cin_pad, kFoldRows), B), dim3(blk),
                       sizeof(float) * kFoldRows * C, st, T1, WqT, C, scale, ln_g, Mt, Cin_pad, COP, Ws, ws_f16);
    hipLaunchKernelGGL(ctx_r3_kernel, dim3(ceil_div(C, 64), B), dim3(64), 0, st, T1, u, b_out, scale, biasB, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Row-folded k x k convolution, second half: the MFMA kernel computed, per INPUT row r, the
// "virtual channels" P[(co,ky)][r][x] = sum_{kx,ci} w[co][ci][ky][kx] in[ci][r][x+kx-pad]
// (a 1 x KW convolution with Cout*KH outputs, which fills an MFMA M-block far better than Cout = 3);
// here out[co][y][x] = bias[co] + sum_ky P[(co,ky)][y+ky-pad][x]   (unet.py:104 final 7x7 conv).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fold_combine_kernel(const float *P, const float *bias,
                                                           float *out, int Cout, int KH, int pad,
                                                           int H, int W) {
    const int b = blockIdx.z, co = blockIdx.y;
    const size_t HW = (size_t)H * W;
    const float *p = P + ((size_t)b * Cout + co) * KH * HW;
    const float bv = bias ? bias[co] : 0.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < HW;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(idx / W);
        float s = bv;
        for (int ky = 0; ky < KH; ++ky) {
            const int r = y + ky - pad;
            if (r >= 0 && r < H) s += p[(size_t)ky * HW + idx + (size_t)(ky - pad) * W];
        }
        out[((size_t)b * Cout + co) * HW + idx] = s;
    }
}

hipError_t fold_combine_launch(const float *P, const float *bias, float *out, int Cout, int KH,
                               int pad, int H, int W, int B, hipStream_t st) {
    const int gx = (int)std::min<size_t>(((size_t)H * W + 255) / 256, 1024);
    hipLaunchKernelGGL(fold_combine_kernel, dim3(gx, Cout, B), dim3(256), 0, st, P, bias, out, Cout, KH,
                       pad, H, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// DDIM update.  x-param: xparam/modules/denoising_diffusion.py:152-174 ; eps-param:
// epsilonparam/modules/denoising_diffusion.py:137-152.  Same operation order as the reference.
// ---------------------------------------------------------------------------------------------
// One element of the update (denoising_diffusion.py ddim(), both trees); shared by the two kernels below.  No fused multiply-adds
// (fp contract off): every product and sum is rounded on its own, as the reference's element-wise tensor operations round them -- and the
// scalar and the four-pixel kernel then hold the same bits whatever hipcc packs (left to the contraction heuristics the two differed in
// the last bit, which a 30-step eps-param chain amplifies to 7e-4).
struct DdimConsts { float c_recip, c_recipm1, c_acp, c_eps, c_sac, c_s1mac, sig; int pred_mode; };
__device__ __forceinline__ float ddim_update(const DdimConsts &k, float fx, float x, float noise, bool clip, bool has_noise) {
#pragma clang fp contract(off)
    float x0, eps;
    if (k.pred_mode == 0 || k.pred_mode == 3) {
        x0 = k.pred_mode == 0 ? fx : k.c_sac * x - k.c_s1mac * fx;
        if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        eps = (k.c_recip * x - x0) / k.c_recipm1;
    } else {
        eps = fx;
        x0 = k.c_recip * x - k.c_recipm1 * eps;
        if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    }
    float xn = k.c_acp * x0 + k.c_eps * eps;
    if (has_noise) xn += k.sig * noise;
    return xn;
}

__global__ void __launch_bounds__(256) ddim_kernel(const DdimArgs a) {
    const int si = a.step_ptr ? *a.step_ptr : a.i;
    const float c_recip = a.tab[0 * a.steps + si];
    const float c_recipm1 = a.tab[1 * a.steps + si];
    const float c_acp = a.tab[2 * a.steps + si];
    const float c_1macp = a.tab[3 * a.steps + si];
    const float sig = a.eta * a.tab[4 * a.steps + si];
    // pred_mode 3 ("v", xparam :128-139,161-162): x0 = sqrt(ac) x - sqrt(1 - ac) v
    const float c_sac = a.tab_v ? a.tab_v[si] : 0.f, c_s1mac = a.tab_v ? a.tab_v[a.steps + si] : 0.f;
    float var = c_1macp - sig * sig;
    // x-tree (pred_mode 0 "x", 2 "noise", 3 "v"): .clamp(min=0) under the square root (xparam :169); eps-tree: none
    if (a.pred_mode != 1) var = fmaxf(var, 0.f);
    const float c_eps = sqrtf(var);
    // clip: 0 none, 1 every image, 2 the first B/2 images only (eps-tree clip_noise "half", eps :142-143)
    const long long clip_n = a.clip == 1 ? a.n : (a.clip == 2 ? a.clip_half_n : 0);
    const DdimConsts kc{c_recip, c_recipm1, c_acp, c_eps, c_sac, c_s1mac, sig, a.pred_mode};
    bool bad = false;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < a.n;
         idx += (long long)gridDim.x * blockDim.x) {
        float fx;
        if (a.P) {
            // the 7-row combine of the row-folded final convolution (fold_combine_kernel) evaluated here: out_fx is never written
            const long long plane = (long long)a.pH * a.pW, img_co = idx / plane;      // = b * Cout + co
            const int pix = (int)(idx - img_co * plane), y = pix / a.pW;
            const float *p = a.P + (size_t)img_co * a.pKH * plane + pix;
            fx = a.P_bias ? a.P_bias[(int)(img_co % a.pC)] : 0.f;
            for (int ky = 0; ky < a.pKH; ++ky) {
                const int r = y + ky - a.pPad;
                if (r >= 0 && r < a.pH) fx += p[(size_t)ky * plane + (long long)(ky - a.pPad) * a.pW];
            }
        } else {
            fx = a.fx[idx];
        }
        const float x = a.x[idx];
        bad |= !(fabsf(fx) <= 3.0e38f);                    // inf / NaN from the U-Net (fp16-plane range overflow)
        a.x_next[idx] = ddim_update(kc, fx, x, a.noise ? a.noise[idx] : 0.f, idx < clip_n, a.noise != nullptr);
    }
    if (bad && a.fault) *a.fault = 1;                      // sticky, read by the host after the decode
}

// The same update with the 7-row combine, four consecutive pixels of one (image, channel) plane per thread (pW % 4 == 0; round 4):
// 16-byte loads / stores, the (image, channel) index is blockIdx.y.  ddim_kernel's element loop derives (plane, pixel) from a 64-bit
// element index -- two 64-bit divisions per element made it VALU-bound (90 us per launch at batch 32 for 0.23 GB of traffic).
// Same sums in the same order, same expressions per component: same bits.
__global__ void __launch_bounds__(256) ddim_rows4_kernel(const DdimArgs a) {
    const int si = a.step_ptr ? *a.step_ptr : a.i;
    const float c_recip = a.tab[0 * a.steps + si];
    const float c_recipm1 = a.tab[1 * a.steps + si];
    const float c_acp = a.tab[2 * a.steps + si];
    const float c_1macp = a.tab[3 * a.steps + si];
    const float sig = a.eta * a.tab[4 * a.steps + si];
    const float c_sac = a.tab_v ? a.tab_v[si] : 0.f, c_s1mac = a.tab_v ? a.tab_v[a.steps + si] : 0.f;
    float var = c_1macp - sig * sig;
    if (a.pred_mode != 1) var = fmaxf(var, 0.f);
    const float c_eps = sqrtf(var);
    const long long clip_n = a.clip == 1 ? a.n : (a.clip == 2 ? a.clip_half_n : 0);
    const int plane = a.pH * a.pW;
    const int pix = 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (pix >= plane) return;
    const int img_co = blockIdx.y, y = pix / a.pW;
    const long long idx = (long long)img_co * plane + pix;
    const float *p = a.P + (size_t)img_co * a.pKH * plane + pix;
    const float b0 = a.P_bias ? a.P_bias[img_co % a.pC] : 0.f;
    float fx[4] = {b0, b0, b0, b0};
    for (int ky = 0; ky < a.pKH; ++ky) {
        const int r = y + ky - a.pPad;
        if (r >= 0 && r < a.pH) {
            const float4 q = *reinterpret_cast<const float4 *>(p + (size_t)ky * plane + (long long)(ky - a.pPad) * a.pW);
            fx[0] += q.x; fx[1] += q.y; fx[2] += q.z; fx[3] += q.w;
        }
    }
    const float4 x4 = *reinterpret_cast<const float4 *>(a.x + idx);
    const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.noise) { const float4 n4 = *reinterpret_cast<const float4 *>(a.noise + idx); nz[0] = n4.x; nz[1] = n4.y; nz[2] = n4.z; nz[3] = n4.w; }
    const bool clip = idx < clip_n;                       // (clip_n is a whole number of images)
    const DdimConsts kc{c_recip, c_recipm1, c_acp, c_eps, c_sac, c_s1mac, sig, a.pred_mode};
    bool bad = false;
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bad |= !(fabsf(fx[c]) <= 3.0e38f);
        out[c] = ddim_update(kc, fx[c], xs[c], nz[c], clip, a.noise != nullptr);
    }
    *reinterpret_cast<float4 *>(a.x_next + idx) = make_float4(out[0], out[1], out[2], out[3]);
    if (bad && a.fault) *a.fault = 1;
}

// end of a graph-replayed DDIM iteration: the next replay works on step - 1
__global__ void step_dec_kernel(int *step) { *step -= 1; }

hipError_t step_dec_launch(int *step, hipStream_t st) {
    hipLaunchKernelGGL(step_dec_kernel, dim3(1), dim3(1), 0, st, step);
    return hipGetLastError();
}

hipError_t ddim_launch(const DdimArgs &a, hipStream_t st) {
    const long long plane = (long long)a.pH * a.pW;
    if (a.P && plane > 0 && (a.pW & 3) == 0 && plane < (1ll << 30) && a.n % plane == 0 && a.n / plane <= 65535 &&
        (((uintptr_t)a.P | (uintptr_t)a.x | (uintptr_t)a.x_next | (uintptr_t)a.noise) & 15) == 0) {
        hipLaunchKernelGGL(ddim_rows4_kernel, dim3((unsigned)ceil_div(plane / 4, 256), (unsigned)(a.n / plane)), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    const int grid = (int)std::min<long long>((a.n + 255) / 256, 4096);
    hipLaunchKernelGGL(ddim_kernel, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

// dst[b][i] = sum_k src[k*part_stride + b*src_bs + i]   (parts = 1: plain channel copy; > 1: split-K slices)
// dst = sum of `PARTS` slices of src (split-K reduce; PARTS = 1: plain copy), float4 per thread; PARTS = 0: run-time
// count.  The slices are summed in slice order (0, 1, 2, ...), all loads of a thread in flight together.
template <int PARTS, class V>
__global__ void __launch_bounds__(256) copy_kernel(const float *src, long long src_bs, float *dst,
                                                   long long dst_bs, long long n, int parts, long long part_stride,
                                                   const int *step_ptr, long long step_stride) {
    const int b = blockIdx.y;
    const V *s = reinterpret_cast<const V *>(src + (size_t)b * src_bs + (step_ptr ? (size_t)*step_ptr * step_stride : 0));
    V *d = reinterpret_cast<V *>(dst + (size_t)b * dst_bs);
    constexpr int VW = sizeof(V) / sizeof(float);
    const long long nv = n / VW, ps = part_stride / VW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
         i += (long long)gridDim.x * blockDim.x) {
        if constexpr (PARTS > 0) {
            V t[PARTS];
#pragma unroll
            for (int k = 0; k < PARTS; ++k) t[k] = s[(size_t)k * ps + i];
            V v = t[0];
#pragma unroll
            for (int k = 1; k < PARTS; ++k) {
                if constexpr (VW == 4) { v.x += t[k].x; v.y += t[k].y; v.z += t[k].z; v.w += t[k].w; }
                else v += t[k];
            }
            d[i] = v;
        } else {
            V v = s[i];
            for (int k = 1; k < parts; ++k) {
                const V u = s[(size_t)k * ps + i];
                if constexpr (VW == 4) { v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
                else v += u;
            }
            d[i] = v;
        }
    }
}

// Rate estimate Compressor.bpp (compress_modules.py:76-90), eval mode:
//   hyper_rate = -log2 FlexiblePrior.likelihood(q_hyper_latent)   (network_components.py:342-378)
//   cond_rate  = -log2 NormalDistribution(mean, scale).likelihood(q_latent)   (utils.py:147-159)
//   bpp[b] = (sum hyper_rate + sum cond_rate) / (H * W)
// One workgroup per image; per-thread partial sums in double, fixed-order tree reduction (deterministic).
__device__ __forceinline__ float prior_logit(const float *p, float x) {
    float h[3], g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { h[k] = x * p[k] + p[3 + k]; h[k] += p[6 + k] * tanhf(h[k]); }
    p += 9;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int j = 0; j < 3; ++j) g[j] = h[0] * p[j] + h[1] * p[3 + j] + h[2] * p[6 + j] + p[9 + j];
#pragma unroll
        for (int j = 0; j < 3; ++j) h[j] = g[j] + p[12 + j] * tanhf(g[j]);
        p += 15;
    }
    return h[0] * p[0] + h[1] * p[1] + h[2] * p[2] + p[3];
}

__global__ void __launch_bounds__(256) bpp_kernel(const float *qh, long long nh, int hw_h, const float *prior,
                                                  const float *ql, const float *mean, const float *scale,
                                                  long long nl, float inv_hw, float *bpp) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    double acc = 0.0;
    const float *xh = qh + (size_t)b * nh;
    for (long long i = threadIdx.x; i < nh; i += blockDim.x) {
        const float *p = prior + (size_t)(i / hw_h) * 44;
        const float x = xh[i];
        const float lower = prior_logit(p, x - 0.5f), upper = prior_logit(p, x + 0.5f);
        const float t = lower + upper;
        const float sgn = t > 0.f ? -1.f : (t < 0.f ? 1.f : 0.f);            // -torch.sign(lower + upper)
        const float su = 1.f / (1.f + expf(-upper * sgn)), sl = 1.f / (1.f + expf(-lower * sgn));
        acc -= (double)log2f(fmaxf(fabsf(su - sl), 1e-9f));
    }
    const float *xl = ql + (size_t)b * nl, *mu = mean + (size_t)b * nl, *sc = scale + (size_t)b * nl;
    for (long long i = threadIdx.x; i < nl; i += blockDim.x) {
        const float d = fabsf(xl[i] - mu[i]), s = sc[i];
        const float c = -0.70710678118654752440f;                            // -(2 ** -0.5)
        const float upper = 0.5f * erfcf(c * ((0.5f - d) / s)), lower = 0.5f * erfcf(c * ((-0.5f - d) / s));
        acc -= (double)log2f(fmaxf(upper - lower, 1e-9f));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) bpp[b] = (float)(red[0] * (double)inv_hw);
}

hipError_t bpp_launch(const float *qh, long long nh, int hw_h, const float *prior, const float *ql, const float *mean,
                      const float *scale, long long nl, float inv_hw, float *bpp, int B, hipStream_t st) {
    hipLaunchKernelGGL(bpp_kernel, dim3(B), dim3(256), 0, st, qh, nh, hw_h, prior, ql, mean, scale, nl, inv_hw, bpp);
    return hipGetLastError();
}

// y = max(x, lo) in place over n floats per image (scale.clamp(min=0.1), compress_modules.py:59)
__global__ void __launch_bounds__(256) clamp_min_kernel(float *x, long long bs, long long n, float lo) {
    float *p = x + (size_t)blockIdx.y * bs;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = fmaxf(p[i], lo);
}

hipError_t clamp_min_launch(float *x, long long bs, long long n, float lo, int B, hipStream_t st) {
    const int gx = (int)std::min<long long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(clamp_min_kernel, dim3(gx, B), dim3(256), 0, st, x, bs, n, lo);
    return hipGetLastError();
}

// sets *flag when any of x[b][0 .. n) (b < B, batch stride bs) is inf or NaN: the fp16-range guard of the entry points
__global__ void __launch_bounds__(256) nonfinite_kernel(const float *x, long long bs, long long n, int *flag) {
    const float *p = x + (size_t)blockIdx.y * bs;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned u = __float_as_uint(p[i]);
        bad |= (u & 0x7f800000u) == 0x7f800000u;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

hipError_t nonfinite_launch(const float *x, long long bs, long long n, int B, int *flag, hipStream_t st) {
    const int gx = (int)std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(nonfinite_kernel, dim3(gx, B), dim3(256), 0, st, x, bs, n, flag);
    return hipGetLastError();
}

// cdc_op_stress (include/cdc_hip.h): does this execution's result equal the first one's, bit for bit?
__global__ void __launch_bounds__(256) bits_differ_kernel(const unsigned *a, const unsigned *b, long long n, long long *c) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) bad |= a[i] != b[i];
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr((unsigned long long *)(c + 2), 1ull);
}
__global__ void bits_differ_tally_kernel(long long *c) { c[0] += 1; if (c[2]) { c[1] += 1; c[2] = 0; } }

hipError_t bits_differ_launch(const float *a, const float *b, long long n, long long *counters, hipStream_t st) {
    const int gx = (int)std::min<long long>((n + 255) / 256, 512);
    hipLaunchKernelGGL(bits_differ_kernel, dim3(gx), dim3(256), 0, st, reinterpret_cast<const unsigned *>(a), reinterpret_cast<const unsigned *>(b), n, counters);
    hipLaunchKernelGGL(bits_differ_tally_kernel, dim3(1), dim3(1), 0, st, counters);
    return hipGetLastError();
}

// round_w_offset (utils.py:72-75): out = round(x - loc) + loc, torch.round = round-half-to-even = rintf
__global__ void __launch_bounds__(256) dequantize_kernel(const float *x, const float *loc, float *out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = rintf(x[i] - loc[i]) + loc[i];
}

hipError_t dequantize_launch(const float *x, const float *loc, float *out, long long n, hipStream_t st) {
    const int gx = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(dequantize_kernel, dim3(gx), dim3(256), 0, st, x, loc, out, n);
    return hipGetLastError();
}

hipError_t copy_channels_launch(const float *src, long long src_bs, float *dst, long long dst_bs,
                                long long n, int B, hipStream_t st, int parts, long long part_stride,
                                const int *step_ptr, long long step_stride) {
    const bool v4 = ((n | src_bs | dst_bs | part_stride | step_stride) & 3) == 0 &&
                    ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0;
    const long long nv = v4 ? n / 4 : n;
    const dim3 grid((unsigned)std::min<long long>((nv + 255) / 256, 2048), (unsigned)B);
#define CDC_COPY(P, V) hipLaunchKernelGGL((copy_kernel<P, V>), grid, dim3(256), 0, st, src, src_bs, dst, dst_bs, n, parts, \
                                          part_stride, step_ptr, step_stride)
    if (v4) {
        switch (parts) {
            case 1: CDC_COPY(1, float4); break;
            case 2: CDC_COPY(2, float4); break;
            case 3: CDC_COPY(3, float4); break;
            case 4: CDC_COPY(4, float4); break;
            default: CDC_COPY(0, float4); break;
        }
    } else {
        CDC_COPY(0, float);
    }
#undef CDC_COPY
    return hipGetLastError();
}

// Column unfold for few-channel k x k convolutions (the first 7x7 layer, unet.py:31 / nc.py:104):
// dst[kx*C + c][y][x] = src[c][y][x + kx - pad] (zero outside), so the layer becomes a KH x 1
// convolution over KW*C channels and runs on the split-bf16 kernel instead of the fp32 MFMA path.
__global__ void __launch_bounds__(256) unfold_x_kernel(const float *src, long long src_bs, float *dst,
                                                       long long dst_bs, int C, int KW, int pad, int H, int W) {
    const int b = blockIdx.z, cc = blockIdx.y;          // cc = kx * C + c
    const int kx = cc / C, c = cc - kx * C;
    const float *s = src + (size_t)b * src_bs + (size_t)c * H * W;
    float *d = dst + (size_t)b * dst_bs + (size_t)cc * H * W;
    const int W4 = W >> 2, n4 = H * W4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const int y = i / W4, x0 = (i - y * W4) * 4 + kx - pad;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (x0 + t >= 0 && x0 + t < W) ? s[y * W + x0 + t] : 0.f;
        *reinterpret_cast<float4 *>(d + (size_t)y * W + (i - y * W4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

hipError_t unfold_x_launch(const float *src, long long src_bs, float *dst, long long dst_bs, int C, int KW,
                           int pad, int H, int W, int B, hipStream_t st) {
    const int gx = std::min((H * (W >> 2) + 255) / 256, 256);
    hipLaunchKernelGGL(unfold_x_kernel, dim3(gx, C * KW, B), dim3(256), 0, st, src, src_bs, dst, dst_bs, C, KW,
                       pad, H, W);
    return hipGetLastError();
}

}  // namespace cdc
