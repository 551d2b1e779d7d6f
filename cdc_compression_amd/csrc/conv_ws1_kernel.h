// conv_ws1_kernel.h -- 1x1 convolutions of the FEW-PIXEL levels (attention projections, attention outputs with per-image weights, res_convs,
// to_out: maps narrower than 32 pixels), all of K inside one workgroup (round 5; the pointwise sibling of conv_ws_kernel.h).
//
// What ran before: conv_split2_kernel with K sliced over WORKGROUPS (4 partial-sum tensors) + a sum pass (copy_kernel<PARTS>, 6.5 us of
// launch floor each: 13 launches per iteration at batch 32, 34 at batch 1), or conv_pw_kernel's linear tiles.  These layers are 0.5 - 5
// GFLOP: every microsecond of them is latency.
//
// Here: workgroup = 32 output channels x NPB 32-pixel blocks of the flattened batch (a block lies inside one image: H*W % 32 == 0) x ALL of
// K; wave w owns the 16-channel chunks w, w + NW, ....  A 1x1 layer has no patch: lane (pixel n, k-half kg) of the MFMA B operand loads ITS
// OWN eight channels of ITS pixel straight from the fp32 NCHW tensor, splits them into the two fp16 planes in registers and multiplies --
// the activations never touch LDS; the weights (three planes of one tap) come from global memory in A-operand order as in
// conv_ws_kernel.  The next chunk's loads are in flight while this one multiplies.  The NW partial accumulators meet in LDS once; the
// epilogue (all waves) is conv_pw_kernel's: x acc_scale (x the pixel's rstd with a folded PreNorm), bias, per-image shift, residual.
#pragma once
#include "conv_pf_kernel.h"

namespace cdc {

struct Ws1Args {
    const float *x0, *x1;           // fp32 NCHW sources (channel concatenation; x1 may be null)
    long long x0_bs, x1_bs;         // batch strides in floats
    int C0, Cin;                    // channels taken from x0; total (both multiples of 16)
    int HW, B;                      // pixels per image (a multiple of 32), batch
    const float *pre_mean, *pre_rstd;   // folded PreNorm (network_components.py:69-77): (x - mean) on load, x rstd in the epilogue; [B][HW] or null
    const void *w;                  // fp16 planes {WH, WL, WH2} of w 2^s: [Cin/16][3][2][COP] 16-byte units
    long long w_bs;                 // per-image weights (attention products): stride in units, 0 = shared
    int nchunk, cpw;                // 16-channel chunks; chunks per wave (nchunk = cpw * waves)
    int COP, Cout;
    float acc_scale;                // 2^-s
    const float *bias;              // [Cout] or null
    const float *shift;             // + shift[b * shift_bs + co], or null
    int shift_bs;
    const float *resid;             // + resid[b * resid_bs + co * HW + pix], or null
    long long resid_bs;
    int resid_is_pre;               // (label only: `resid` is a hoisted partial sum, ConvArgs::pre_add)
    float *out;                     // fp32 NCHW
    long long out_bs;
    int tiles;                      // pixel tiles (gridDim.x = tiles * Cout / 32)
    int xcd_remap;
    int *fault;                     // range guard (ConvArgs::fault)
};

template <int NPB, bool PERIMG>
__global__ void __launch_bounds__(512) conv_ws1_kernel(const Ws1Args PA) {
    // (the fields as locals: see conv_ws_kernel)
    const float *a_x0 = PA.x0, *a_x1 = PA.x1;
    const long long a_x0_bs = PA.x0_bs, a_x1_bs = PA.x1_bs, a_w_bs = PA.w_bs, a_resid_bs = PA.resid_bs, a_out_bs = PA.out_bs;
    const int a_C0 = PA.C0, HW = PA.HW, a_nchunk = PA.nchunk, a_cpw = PA.cpw, a_COP = PA.COP, a_tiles = PA.tiles, a_shift_bs = PA.shift_bs;
    const float *a_pre_mean = PA.pre_mean, *a_pre_rstd = PA.pre_rstd, *a_bias = PA.bias, *a_shift = PA.shift, *a_resid = PA.resid;
    const void *a_w = PA.w;
    float *a_out = PA.out;
    const float a_acc_scale = PA.acc_scale;
    int *a_fault = PA.fault;
    extern __shared__ __attribute__((aligned(16))) uint4 ws1_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    const int n = lane & 31, kg = lane >> 5;
    int slot = blockIdx.x;
    if (PA.xcd_remap) slot = (blockIdx.x & 7) * ((int)gridDim.x >> 3) + (blockIdx.x >> 3);
    const int g = slot / a_tiles, tile = slot - g * a_tiles;
    // ---- the lane's pixels: block pb = 32 consecutive pixels of ONE image ------------------------------------------------------
    const int bpi = HW >> 5;                                // pixel blocks per image
    int img[NPB], pix[NPB];
    float mu[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        const int blk = tile * NPB + pb;
        img[pb] = __builtin_amdgcn_readfirstlane(blk / bpi);
        pix[pb] = (blk - img[pb] * bpi) * 32 + n;
        mu[pb] = a_pre_mean ? a_pre_mean[(size_t)img[pb] * HW + pix[pb]] : 0.f;
    }
    const int c0_chunks = a_C0 >> 4;
    const unsigned wlane = (unsigned)(kg * a_COP + g * 32 + n) * 16u;
    float xv[2][NPB][8];
    constexpr int NA = PERIMG ? NPB : 1;                    // weight sets: one per pixel block (= image) with per-image weights
    f16x8 A[2][NA][3];
    auto load_chunk = [&](auto bufc, int chunk) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const bool from0 = chunk < c0_chunks;
        const int cc = (from0 ? chunk : chunk - c0_chunks) * 16 + kg * 8;
        const long long sbs = from0 ? a_x0_bs : a_x1_bs;
        const float *src = from0 ? a_x0 : a_x1;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const char *sb = reinterpret_cast<const char *>(src + (size_t)img[pb] * sbs);      // (uniform base + 32-bit lane offset)
            const unsigned vo = (unsigned)(cc * HW + pix[pb]) * 4u;
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[buf][pb][q] = *reinterpret_cast<const float *>(sb + (size_t)q * HW * 4 + vo);
        }
#pragma unroll
        for (int pb = 0; pb < NA; ++pb) {
            const char *wc = reinterpret_cast<const char *>(a_w) + ((size_t)(PERIMG ? img[pb] : 0) * a_w_bs + (size_t)chunk * 6 * a_COP) * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) A[buf][pb][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(wc + (size_t)pl * 2 * a_COP * 16 + wlane));
        }
    };
    f32x16 acc[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;
    // a = h + l' 2^-11, w 2^s = WH + WL:  acc += WL.h + WH2.l' + WH.h  (plane-major over the pixel blocks: consecutive MFMAs on different accumulators)
    auto mma_chunk = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        f16x8 bh[NPB], bl[NPB];
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                _Float16 hq, lq;
                split2h(xv[buf][pb][q] - mu[pb], hq, lq);
                bh[pb][q] = hq; bl[pb][q] = lq;
            }
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[buf][PERIMG ? pb : 0][1], bh[pb], acc[pb], 0, 0, 0);
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[buf][PERIMG ? pb : 0][2], bl[pb], acc[pb], 0, 0, 0);
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[buf][PERIMG ? pb : 0][0], bh[pb], acc[pb], 0, 0, 0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    load_chunk(I0{}, wave);
    for (int ci = 0; ci < a_cpw; ci += 2) {
        const int c1 = __builtin_amdgcn_readfirstlane(wave + (ci + 1 < a_cpw ? ci + 1 : ci) * nw);
        const int c2 = __builtin_amdgcn_readfirstlane(wave + (ci + 2 < a_cpw ? ci + 2 : ci) * nw);
        load_chunk(I1{}, c1);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk(I0{});
        __builtin_amdgcn_sched_barrier(0);
        if (ci + 1 < a_cpw) {
            load_chunk(I0{}, c2);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk(I1{});
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- the K slices of the waves meet in LDS: red[wave][pb][4][lane] -------------------------------------------------------------
    float4 *red = reinterpret_cast<float4 *>(ws1_smem);
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            red[((wave * NPB + pb) * 4 + q) * 64 + lane] = make_float4(acc[pb][4 * q + 0], acc[pb][4 * q + 1], acc[pb][4 * q + 2], acc[pb][4 * q + 3]);
    __syncthreads();
    // unit u = (pixel block, 16-channel half of the group's 32): wave u, u + waves, ... finishes it
    for (int u = wave; u < 2 * NPB; u += nw) {
        const int pb = u >> 1, qh = u & 1;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 s4 = red[((0 * NPB + pb) * 4 + 2 * qh + q) * 64 + lane];
            for (int w = 1; w < nw; ++w) {                      // fixed order: deterministic
                const float4 t = red[((w * NPB + pb) * 4 + 2 * qh + q) * 64 + lane];
                s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
            }
            v[4 * q + 0] = s4.x; v[4 * q + 1] = s4.y; v[4 * q + 2] = s4.z; v[4 * q + 3] = s4.w;
        }
        // (the pixel of this unit: recomputed -- `pb` is a run-time value here)
        const int blk = tile * NPB + pb, im = blk / bpi, px = (blk - im * bpi) * 32 + n;
        const int cbase = g * 32 + 16 * qh + 4 * kg;           // register r = 4 q + i -> channel cbase + i + 8 q
        const float sc = a_pre_rstd ? a_acc_scale * a_pre_rstd[(size_t)im * HW + px] : a_acc_scale;
        float mag = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int co = cbase + (r & 3) + 8 * (r >> 2);
            v[r] = v[r] * sc + (a_bias ? a_bias[co] : 0.f);
            mag += fabsf(v[r]);
        }
        if (a_fault && !(mag < 3.0e38f)) *a_fault = 1;          // non-finite accumulators: reported where they arise
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int co = cbase + (r & 3) + 8 * (r >> 2);
            if (a_shift) v[r] += a_shift[(size_t)im * a_shift_bs + co];
            if (a_resid) v[r] += a_resid[(size_t)im * a_resid_bs + (size_t)co * HW + px];
            a_out[(size_t)im * a_out_bs + (size_t)co * HW + px] = v[r];
        }
    }
}

typedef void (*ws1_kernel_fn)(const Ws1Args);

}  // namespace cdc
