// attn_kernels.hip -- fused front half of the folded linear attention (network_components.py:128-136):
//
//     k, v = W_kv LN(x)          (1x1 projection, PreNorm LayerNorm folded into the weights)
//     ctx  = softmax_N(k) v^T    (C x C per image)
//
// in ONE pass over x: k and v (2C x N floats per image, 2.1 GB at 256x256 / batch 32) never reach HBM and
// the separate row-maximum pass disappears.  Per workgroup (image b, pixel split s), per 32-pixel tile:
//   phase 1  kv[2C][32] = W'[2C][C] (x - mean)    v_mfma_f32_32x32x2_f32, the wave's rows of W' live in
//            registers for the whole kernel; each wave owns 2C/128 row blocks; scaled by rstd, + W b_ln
//   LDS      the tile is written [channel][pixel] so that phase 2 reads it transposed (lane = channel)
//   phase 2  S[d][e] += sum_n exp(k[d][n] - m[d]) v[e][n]   with a running row maximum m (online softmax:
//            when a tile raises m the accumulated rows are rescaled by exp(m_old - m_new))
// Outputs per (b, s): S [C][C], Z[d] = sum_n exp(k - m), M[d] = m; ctx_r0 combines the splits.
// All products and sums are IEEE fp32 (the f32 MFMA is an fmaf chain), exp is expf.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdlib.h>

#include "cdc_internal.h"
#include "conv_kernel.h"     // pf_store_block (PF copy of the attention output)

namespace cdc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// exact three-way bf16 split of an fp32 value (truncation; see conv_split_kernel.h)
__device__ __forceinline__ void split3(float a, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(a) & 0xFFFF0000u;
    const float r = a - __uint_as_float(h);
    m = __float_as_uint(r) & 0xFFFF0000u;
    l = __float_as_uint(r - __uint_as_float(m));
}

// F16 = true (CDC_ARITH_F16X2): both contractions on the fp16 matrix cores with two-plane operands (three products):
//   projection:  W' 2^s as {WH, WL, WH2} (static planes), x - mean as (h, l') -- as in conv_split_kernel.h AR = 1;
//   sum_n p v:   p = exp(k - max) in (0, 1] as {PH, PL = fp16(p - PH), PH2 = PH 2^-11} (absolute error <= 3e-8, i.e. fp32
//                resolution of the largest weight p = 1), v as (h, l'); K = 16 pixels per instruction: 6 fp16 MFMAs per
//                32-pixel tile and S block instead of 16 v_mfma_f32_32x32x2_f32 (32 vs 64 cycles each).
template <int CB, int NW, bool F16 = false>      // C = 32 * CB channels, NW waves per workgroup
__global__ void __launch_bounds__(64 * NW, CB == 2 ? 3 : 1) kvctx_kernel(const KvCtxArgs a) {
    constexpr int C = 32 * CB, NBLK = 2 * CB;        // kv row blocks of 32 channels
    constexpr int BPW = NBLK / NW;                    // row blocks per wave in phase 1
    constexpr int SPW = CB * CB / NW;                 // S blocks per wave in phase 2
    constexpr int WPD = NW / CB;                      // waves sharing one d block (each takes SPW e blocks)
    constexpr int LDK = F16 ? 36 : 33;                // pixel stride of the LDS tile (F16: 16-byte aligned rows, conflict-free b128)
    static_assert(!F16 || BPW == 1, "the fp16 projection keeps one row block of split weights per wave");
    static_assert(NBLK % NW == 0 && (CB * CB) % NW == 0 && NW % CB == 0, "the waves share the blocks evenly");
    __shared__ float kvbuf[2][2 * C * LDK];         // double-buffered tile: one barrier per tile
    __shared__ __attribute__((aligned(16))) float fac[NW][32];

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, sp = blockIdx.x;
    const int N = a.N;
    const int per = N / a.nsplit;                     // pixels of this split (multiple of 32, host-enforced)
    const int p0 = sp * per;
    const float *x = a.x + (size_t)b * a.x_bs;
    const float *mean = a.mean + (size_t)b * N, *rstd = a.rstd + (size_t)b * N;

    // ---- the wave's rows of W' ---------------------------------------------------------------------
    // C = 64: the projection runs fp32-exact on the bf16 cores (three-way split operands, six products);
    // A operand of v_mfma_f32_32x32x16_bf16: lane (i = j, kh) holds W'[row i][16q + 8kh .. +7] per plane.
    // With two row blocks per wave (<4,4>) the split planes would not fit the register file: f32 MFMA there.
    constexpr bool kSplitP1 = BPW == 1;               // <2,4> and <4,8>: 12 * C/16 registers of split weights
    float wr[kSplitP1 ? 1 : BPW][kSplitP1 ? 1 : C / 2], bias[BPW][16];
    bf16x8 ws[kSplitP1 ? BPW : 1][kSplitP1 ? C / 16 : 1][3];
#pragma unroll
    for (int q = 0; q < BPW; ++q) {
        const int blk = wave * BPW + q;
        if constexpr (kSplitP1) {
            const uint4 *wp = reinterpret_cast<const uint4 *>(a.Ws);
#pragma unroll
            for (int c = 0; c < C / 16; ++c)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    ws[q][c][pl] = __builtin_bit_cast(bf16x8, wp[(size_t)((c * 3 + pl) * 2 + kh) * (2 * C) + blk * 32 + j]);
        } else {
#pragma unroll
            for (int s = 0; s < C / 2; ++s) wr[q][s] = a.Wt[(size_t)(2 * s + kh) * (2 * C) + blk * 32 + j];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[q][r] = a.bias[blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
    }
    // ---- phase-2 ownership: wave -> d block wave / WPD, e blocks (wave % WPD) * SPW .. + SPW
    const int db = wave / WPD;
    const int eb0 = (wave % WPD) * SPW;
    f32x16 S[SPW];
#pragma unroll
    for (int q = 0; q < SPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[q][r] = 0.f;
    float m_run = -INFINITY, zsum = 0.f;             // row d = db*32 + j (both halves of the wave hold it)

    // The tile's B operand (lane = pixel j, channel 2s + kh) is fetched one tile ahead: the loads of tile
    // t+1 are in flight during the LDS / softmax / phase-2 part of tile t.
    float xn[C / 2], mu_n, rs_n;
    const char *xb = reinterpret_cast<const char *>(x);             // wave-uniform base; lane part = voff
    {
        const int px = p0 + j;
        mu_n = mean[px]; rs_n = rstd[px];
        // f32 MFMA: register s <-> channel 2s + kh;  bf16 MFMA: register s <-> channel 16(s/8) + 8kh + s%8
        const unsigned voff = (unsigned)((kSplitP1 ? 8 * kh : kh) * N + px) * 4u;
#pragma unroll
        for (int s = 0; s < C / 2; ++s)
            xn[s] = *reinterpret_cast<const float *>(xb + (size_t)(kSplitP1 ? 16 * (s >> 3) + (s & 7) : 2 * s) * N * 4 + voff);
    }
    for (int t0 = 0; t0 < per; t0 += 32) {
        const float mu = mu_n, rs = rs_n;
        // ---- phase 1 -------------------------------------------------------------------------------
        f32x16 acc[BPW];
#pragma unroll
        for (int q = 0; q < BPW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        if constexpr (F16) {
#pragma unroll
            for (int c = 0; c < C / 16; ++c) {
                f16x8 xh, xl;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    _Float16 hq, lq;
                    split2h(xn[8 * c + i] - mu, hq, lq);
                    xh[i] = hq; xl[i] = lq;
                }
                // planes {WH, WL, WH2} sit where the bf16 planes {w1, w2, w3} sit; smallest terms first
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ws[0][c][1]), xh, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ws[0][c][2]), xl, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ws[0][c][0]), xh, acc[0], 0, 0, 0);
            }
        } else if constexpr (kSplitP1) {
#pragma unroll
            for (int c = 0; c < C / 16; ++c) {
                unsigned hh[8], mm[8], ll[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) split3(xn[8 * c + i] - mu, hh[i], mm[i], ll[i]);
                uint4 vh, vm, vl;
                vh.x = (hh[0] >> 16) | hh[1]; vh.y = (hh[2] >> 16) | hh[3];
                vh.z = (hh[4] >> 16) | hh[5]; vh.w = (hh[6] >> 16) | hh[7];
                vm.x = (mm[0] >> 16) | mm[1]; vm.y = (mm[2] >> 16) | mm[3];
                vm.z = (mm[4] >> 16) | mm[5]; vm.w = (mm[6] >> 16) | mm[7];
                vl.x = (ll[0] >> 16) | (ll[1] & 0xFFFF0000u); vl.y = (ll[2] >> 16) | (ll[3] & 0xFFFF0000u);
                vl.z = (ll[4] >> 16) | (ll[5] & 0xFFFF0000u); vl.w = (ll[6] >> 16) | (ll[7] & 0xFFFF0000u);
                const bf16x8 B[3] = {__builtin_bit_cast(bf16x8, vh), __builtin_bit_cast(bf16x8, vm),
                                     __builtin_bit_cast(bf16x8, vl)};
#pragma unroll
                for (int q = 0; q < BPW; ++q)
#pragma unroll
                    for (int pa = 2; pa >= 0; --pa)          // smallest terms first, six products in all
#pragma unroll
                        for (int pb = 2 - pa; pb >= 0; --pb)
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws[q][c][pa], B[pb], acc[q], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < C / 2; ++s) {
                const float xv = xn[s] - mu;
#pragma unroll
                for (int q = 0; q < BPW; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[q][s], xv, acc[q], 0, 0, 0);
            }
        }
        if (t0 + 32 < per) {
            const int px = p0 + t0 + 32 + j;
            mu_n = mean[px]; rs_n = rstd[px];
            const unsigned voff = (unsigned)((kSplitP1 ? 8 * kh : kh) * N + px) * 4u;
#pragma unroll
            for (int s = 0; s < C / 2; ++s)
                xn[s] = *reinterpret_cast<const float *>(xb + (size_t)(kSplitP1 ? 16 * (s >> 3) + (s & 7) : 2 * s) * N * 4 + voff);
        }
        float *kv = kvbuf[(t0 >> 5) & 1];             // the barrier of tile t orders it after every read of tile t-2
#pragma unroll
        for (int q = 0; q < BPW; ++q) {
            const int blk = wave * BPW + q;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                kv[row * LDK + j] = acc[q][r] * (F16 ? rs * a.wscale_inv : rs) + bias[q][r];
            }
        }
        __syncthreads();
        // ---- phase 2 -------------------------------------------------------------------------------
        // f32 MFMA: register s <-> pixel 2s + kh;  fp16 MFMA: register s <-> pixel 16(s/8) + 8kh + s%8
        const float *krow = kv + (db * 32 + j) * LDK + (F16 ? 8 * kh : kh);
        float kk[16];
        float tmax = -INFINITY;
        if constexpr (F16) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const float4 k0 = *reinterpret_cast<const float4 *>(krow + 16 * st), k1 = *reinterpret_cast<const float4 *>(krow + 16 * st + 4);
                kk[8 * st + 0] = k0.x; kk[8 * st + 1] = k0.y; kk[8 * st + 2] = k0.z; kk[8 * st + 3] = k0.w;
                kk[8 * st + 4] = k1.x; kk[8 * st + 5] = k1.y; kk[8 * st + 6] = k1.z; kk[8 * st + 7] = k1.w;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) tmax = fmaxf(tmax, kk[s]);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) { kk[s] = krow[2 * s]; tmax = fmaxf(tmax, kk[s]); }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        if (__any(tmax > m_run)) {                   // a new row maximum somewhere in the block: rescale
            const float mn = fmaxf(m_run, tmax);
            const float f = expf(m_run - mn);         // exp(-inf) = 0 on the first tile
            m_run = mn;
            zsum *= f;
            if (kh == 0) fac[wave][j] = f;
            __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the wave's own LDS writes are visible
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 f4 = *reinterpret_cast<const float4 *>(&fac[wave][8 * r4 + 4 * kh]);
#pragma unroll
                for (int q = 0; q < SPW; ++q) {
                    S[q][4 * r4 + 0] *= f4.x; S[q][4 * r4 + 1] *= f4.y;
                    S[q][4 * r4 + 2] *= f4.z; S[q][4 * r4 + 3] *= f4.w;
                }
            }
        }
        float pv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) { pv[s] = expf(kk[s] - m_run); zsum += pv[s]; }
        if constexpr (F16) {
            f16x8 ph[2], pl[2], ph2[2];
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float p = pv[8 * st + i];
                    const _Float16 h = (_Float16)p;
                    ph[st][i] = h;
                    pl[st][i] = (_Float16)(p - (float)h);
                    ph2[st][i] = (_Float16)((float)h * (1.0f / 2048.0f));
                }
#pragma unroll
            for (int q = 0; q < SPW; ++q) {
                const float *vrow = kv + (C + (eb0 + q) * 32 + j) * LDK + 8 * kh;   // v[e][16 st + 8kh + i]
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(vrow + 16 * st), v1 = *reinterpret_cast<const float4 *>(vrow + 16 * st + 4);
                    const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    f16x8 vh, vl;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        _Float16 hq, lq;
                        split2h(vv[i], hq, lq);
                        vh[i] = hq; vl[i] = lq;
                    }
                    S[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[st], vh, S[q], 0, 0, 0);
                    S[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph2[st], vl, S[q], 0, 0, 0);
                    S[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[st], vh, S[q], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            const float *vrow = kv + (C + (eb0 + q) * 32 + j) * LDK + kh;   // v[e][2s + kh]
#pragma unroll
            for (int s = 0; s < 16; ++s)
                S[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[s], vrow[2 * s], S[q], 0, 0, 0);
        }
        }
    }
    // ---- partial results of this split ---------------------------------------------------------------
    const size_t slot = (size_t)b * a.nsplit + sp;
#pragma unroll
    for (int q = 0; q < SPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            a.S[(slot * C + d) * C + (eb0 + q) * 32 + j] = S[q][r];
        }
    zsum += __shfl_xor(zsum, 32);
    if (kh == 0 && eb0 == 0) {
        a.Z[slot * C + db * 32 + j] = zsum;
        a.M[slot * C + db * 32 + j] = m_run;
    }
}

// ------------------------------------------------------------------------------------------------
// kvctx16_kernel: the CDC_ARITH_F16X2 form of the kernel above, organised around its real bound -- the VALU, not
// the matrix cores (per 32-pixel tile and wave the three-product contractions are 18 MFMAs, while splitting the
// operands and the exponentials were ~600 VALU instructions):
//   * the tile of x is split into fp16 planes ONCE per workgroup: wave w converts channel chunk w (16 channels x
//     32 pixels = 8 values per lane) and publishes the two planes in B-operand order in LDS; every wave then reads
//     the C/16 chunks as ds_read_b128 (was: every wave loaded and split the whole tile, NW-fold redundant);
//   * VP: the v rows of the projection leave phase 1 already split (h, l' as fp16 [row][pixel] arrays): split once by
//     the wave that produced them instead of once per S block that consumes them;
//   * exp(k - max) and the rescale factor use v_exp_f32 (exp2(x log2 e)): <= 2 ulp of relative error on a weight in
//     (0, 1], against the ~10-instruction expf.
// One barrier per tile as before: x planes of tile t+1, k/v of tile t are written before barrier t (double buffers).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

template <int CB, int NW, bool VP>
__global__ void __launch_bounds__(64 * NW, CB == 2 ? (VP ? 2 : 3) : 1) kvctx16_kernel(const KvCtxArgs a) {
    constexpr int C = 32 * CB, NBLK = 2 * CB;
    static_assert(NBLK == NW && C / 16 == NW, "one projection row block and one channel chunk per wave");
    constexpr int SPW = CB * CB / NW;                 // S blocks per wave in phase 2
    constexpr int WPD = NW / CB;                      // waves sharing one d block (each takes SPW e blocks)
    constexpr int LDK = 36;                           // fp32 pixel stride of a k (v) row: 16-byte rows, conflict-free b128
    constexpr int LDH = 40;                           // fp16 pixel stride of a v plane row
    constexpr int KROWS = VP ? C : 2 * C;
    __shared__ __attribute__((aligned(16))) float kbuf[2][KROWS * LDK];
    __shared__ __attribute__((aligned(16))) _Float16 vbuf[2][2][VP ? C * LDH : 8];
    __shared__ __attribute__((aligned(16))) uint4 xp[2][NW][2][64];       // [buffer][chunk][plane h / l'][lane]
    __shared__ __attribute__((aligned(16))) float fac[NW][32];

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, sp = blockIdx.x;
    const int N = a.N;
    const int per = N / a.nsplit;                     // pixels of this split (multiple of 32, host-enforced)
    const int p0 = sp * per;
    const float *mean = a.mean + (size_t)b * N, *rstd = a.rstd + (size_t)b * N;

    // the wave's row block of W' 2^s as fp16 planes {WH, WL, WH2}: lane (i = j, kh) holds W'[row i][16c + 8kh .. +7]
    f16x8 ws[C / 16][3];
    float bias[16];
    {
        const uint4 *wp = reinterpret_cast<const uint4 *>(a.Ws);
#pragma unroll
        for (int c = 0; c < C / 16; ++c)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                ws[c][pl] = __builtin_bit_cast(f16x8, wp[(size_t)((c * 3 + pl) * 2 + kh) * (2 * C) + wave * 32 + j]);
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = a.bias[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
    }
    // The k rows (waves 0 .. CB-1) live in the log2 domain: k log2(e) -- folded into the scale and the bias of the
    // projection's write-out -- so that exp(k - max) is one subtraction and one v_exp_f32; M is converted back at the end.
    constexpr float kLog2e = 1.44269504088896341f;
    const float kdom = wave < CB ? kLog2e : 1.0f;
    if (wave < CB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] *= kLog2e;
    }
    const int db = wave / WPD;
    const int eb0 = (wave % WPD) * SPW;
    f32x16 S[SPW];
#pragma unroll
    for (int q = 0; q < SPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[q][r] = 0.f;
    float m_run = -INFINITY, zsum = 0.f;             // row d = db*32 + j (both halves of the wave hold it)

    // this wave's chunk of the x tile: channels 16 wave + 8 kh + i at pixel j (wave-uniform base + lane offset)
    const char *xb = reinterpret_cast<const char *>(a.x + (size_t)b * a.x_bs + (size_t)(16 * wave) * N);
    float xr[8], mu_r, rs_r;                          // raw values / statistics of the tile loaded last
    auto load_x = [&](int t0) {
        const int px = p0 + t0 + j;
        mu_r = mean[px]; rs_r = rstd[px];
        const unsigned voff = (unsigned)(8 * kh * N + px) * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = *reinterpret_cast<const float *>(xb + (size_t)i * N * 4 + voff);
    };
    auto publish_x = [&](int buf) {                   // split (x - mean) and store the two planes of this chunk
        f16x8 xh, xl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            _Float16 hq, lq;
            split2h(xr[i] - mu_r, hq, lq);
            xh[i] = hq; xl[i] = lq;
        }
        xp[buf][wave][0][lane] = __builtin_bit_cast(uint4, xh);
        xp[buf][wave][1][lane] = __builtin_bit_cast(uint4, xl);
    };
    load_x(0);
    publish_x(0);
    float rs = rs_r;                                  // rstd of the tile in phase 1
    if (32 < per) load_x(32);
    __syncthreads();
    for (int t0 = 0; t0 < per; t0 += 32) {
        const int buf = (t0 >> 5) & 1;
        // ---- phase 1: kv rows of this wave = W' planes x (x - mean) planes, three products ----------------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < C / 16; ++c) {
            const f16x8 xh = __builtin_bit_cast(f16x8, xp[buf][c][0][lane]), xl = __builtin_bit_cast(f16x8, xp[buf][c][1][lane]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ws[c][1], xh, acc, 0, 0, 0);        // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ws[c][2], xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ws[c][0], xh, acc, 0, 0, 0);
        }
        // the next tile's planes (its values arrived during the previous tile), then the loads of the one after
        const float rs_cur = rs;
        if (t0 + 32 < per) {
            publish_x(buf ^ 1);
            rs = rs_r;
            if (t0 + 64 < per) load_x(t0 + 64);
        }
        // k rows as fp32 [row][pixel]; v rows as fp32 or (VP) as the planes phase 2 multiplies with
        {
            const float sc = rs_cur * a.wscale_inv * kdom;
            if (!VP || wave < CB) {
                float *kv = kbuf[buf];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    kv[row * LDK + j] = acc[r] * sc + bias[r];
                }
            } else {
                _Float16 *vh = vbuf[buf][0], *vl = vbuf[buf][1];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wave - CB) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    _Float16 hq, lq;
                    split2h(acc[r] * sc + bias[r], hq, lq);
                    vh[row * LDH + j] = hq;
                    vl[row * LDH + j] = lq;
                }
            }
        }
        __syncthreads();
        // ---- phase 2: S[d][e] += sum_n exp(k[d][n] - m[d]) v[e][n]; register s <-> pixel 16(s/8) + 8kh + s%8 -------
        const float *krow = kbuf[buf] + (db * 32 + j) * LDK + 8 * kh;
        float kk[16];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const float4 k0 = *reinterpret_cast<const float4 *>(krow + 16 * st), k1 = *reinterpret_cast<const float4 *>(krow + 16 * st + 4);
            kk[8 * st + 0] = k0.x; kk[8 * st + 1] = k0.y; kk[8 * st + 2] = k0.z; kk[8 * st + 3] = k0.w;
            kk[8 * st + 4] = k1.x; kk[8 * st + 5] = k1.y; kk[8 * st + 6] = k1.z; kk[8 * st + 7] = k1.w;
        }
        float tmax = kk[0];
#pragma unroll
        for (int s = 1; s < 16; ++s) tmax = fmaxf(tmax, kk[s]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        if (__any(tmax > m_run)) {                   // a new row maximum somewhere in the block: rescale
            const float mn = fmaxf(m_run, tmax);
            const float f = __builtin_amdgcn_exp2f(m_run - mn);     // exp2(-inf) = 0 on the first tile
            m_run = mn;
            zsum *= f;
            if (kh == 0) fac[wave][j] = f;
            __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the wave's own LDS writes are visible
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 f4 = *reinterpret_cast<const float4 *>(&fac[wave][8 * r4 + 4 * kh]);
#pragma unroll
                for (int q = 0; q < SPW; ++q) {
                    S[q][4 * r4 + 0] *= f4.x; S[q][4 * r4 + 1] *= f4.y;
                    S[q][4 * r4 + 2] *= f4.z; S[q][4 * r4 + 3] *= f4.w;
                }
            }
        }
        // p in (0, 1] as {PH, PL = fp16(p - PH), PH2 = PH 2^-11} (absolute error <= 3e-8)
        f16x8 ph[2], pl[2], ph2[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float p = __builtin_amdgcn_exp2f(kk[8 * st + i] - m_run);
                zsum += p;
                const _Float16 h = (_Float16)p;
                ph[st][i] = h;
                pl[st][i] = (_Float16)(p - (float)h);
            }
#pragma unroll
        for (int st = 0; st < 2; ++st) ph2[st] = ph[st] * (_Float16)(1.0f / 2048.0f);     // packed fp16 multiplies (same rounding)
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f16x8 vh, vl;
                if constexpr (VP) {
                    const int o = ((eb0 + q) * 32 + j) * LDH + 16 * st + 8 * kh;     // v[e][16 st + 8kh + i]
                    vh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(&vbuf[buf][0][o]));
                    vl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(&vbuf[buf][1][o]));
                } else {
                    const float *vrow = kbuf[buf] + (C + (eb0 + q) * 32 + j) * LDK + 8 * kh + 16 * st;
                    const float4 v0 = *reinterpret_cast<const float4 *>(vrow), v1 = *reinterpret_cast<const float4 *>(vrow + 4);
                    const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        _Float16 hq, lq;
                        split2h(vv[i], hq, lq);
                        vh[i] = hq; vl[i] = lq;
                    }
                }
                S[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[st], vh, S[q], 0, 0, 0);
                S[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph2[st], vl, S[q], 0, 0, 0);
                S[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[st], vh, S[q], 0, 0, 0);
            }
        }
    }
    // ---- partial results of this split ---------------------------------------------------------------
    const size_t slot = (size_t)b * a.nsplit + sp;
#pragma unroll
    for (int q = 0; q < SPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            a.S[(slot * C + d) * C + (eb0 + q) * 32 + j] = S[q][r];
        }
    zsum += __shfl_xor(zsum, 32);
    if (kh == 0 && eb0 == 0) {
        a.Z[slot * C + db * 32 + j] = zsum;
        a.M[slot * C + db * 32 + j] = m_run * (1.0f / kLog2e);     // back to the natural-log domain (ctx_r0_kernel)
    }
}

hipError_t kvctx_launch(const KvCtxArgs &a, int B, hipStream_t st) {
    if (a.N % (32 * a.nsplit)) return hipErrorInvalidValue;
    dim3 grid((unsigned)a.nsplit, (unsigned)B);
    // fp16 arithmetic: kvctx16_kernel (shared x split).  Plane-form v tile (VP) only at C = 128: at C = 64 it costs
    // the third workgroup per CU (LDS) and the barrier-synchronised waves gain nothing from moving the split
    // (measured, batch 32: C = 64 / 256^2 0.417 -> 0.280 ms without VP, 0.350 with; C = 128 / 128^2 0.416 -> 0.253 / 0.245).
    if (a.C == 64 && a.f16) hipLaunchKernelGGL((kvctx16_kernel<2, 4, false>), grid, dim3(256), 0, st, a);
    else if (a.C == 128 && a.f16) hipLaunchKernelGGL((kvctx16_kernel<4, 8, true>), grid, dim3(512), 0, st, a);
    else if (a.C == 64) hipLaunchKernelGGL((kvctx_kernel<2, 4>), grid, dim3(256), 0, st, a);
    else if (a.C == 128) hipLaunchKernelGGL((kvctx_kernel<4, 8>), grid, dim3(512), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Folded attention output (network_components.py:136-139 after the fold of DESIGN.md section 4):
//     y[c][n] = rstd[n] * sum_ci M'[b][c][ci] (x[ci][n] - mean[n]) + bias[b][c] + x[c][n]
// A bandwidth-bound C x C pointwise product with per-image weights: no LDS, no barrier.  Wave w owns output
// channels [32w, 32w+32) with its three bf16 weight planes in registers for the whole pixel range; the pixel
// operand is loaded per lane (lane = pixel: 128-byte rows), split in registers, six bf16 MFMAs per 16 input
// channels; the accumulator layout (lane = pixel, 16 channels) stores 128-byte row segments directly.
// ------------------------------------------------------------------------------------------------
template <int CB>
__global__ void __launch_bounds__(64 * CB, 2) lnconv_kernel(const LnConvArgs a) {
    constexpr int C = 32 * CB;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, sp = blockIdx.x;
    const int N = a.N, per = N / a.nsplit, p0 = sp * per;
    const float *x = a.x + (size_t)b * a.x_bs;
    float *y = a.y + (size_t)b * a.y_bs;
    const float *mean = a.mean + (size_t)b * N, *rstd = a.rstd + (size_t)b * N;
    bf16x8 ws[C / 16][3];
    {
        const uint4 *wp = reinterpret_cast<const uint4 *>(a.Ws) + (size_t)b * (C / 16) * 6 * C;
#pragma unroll
        for (int c = 0; c < C / 16; ++c)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                ws[c][pl] = __builtin_bit_cast(bf16x8, wp[(size_t)((c * 3 + pl) * 2 + kh) * C + wave * 32 + j]);
    }
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = a.bias[(size_t)b * C + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
    const char *xb = reinterpret_cast<const char *>(x);
    char *yb = reinterpret_cast<char *>(y);
    for (int t0 = 0; t0 < per; t0 += 32) {
        const int px = p0 + t0 + j;
        const float mu = mean[px], rs = rstd[px];
        const unsigned voff = (unsigned)(8 * kh * N + px) * 4u;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < C / 16; ++c) {
            unsigned hh[8], mm[8], ll[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                split3(*reinterpret_cast<const float *>(xb + (size_t)(16 * c + i) * N * 4 + voff) - mu, hh[i], mm[i], ll[i]);
            uint4 vh, vm, vl;
            vh.x = (hh[0] >> 16) | hh[1]; vh.y = (hh[2] >> 16) | hh[3];
            vh.z = (hh[4] >> 16) | hh[5]; vh.w = (hh[6] >> 16) | hh[7];
            vm.x = (mm[0] >> 16) | mm[1]; vm.y = (mm[2] >> 16) | mm[3];
            vm.z = (mm[4] >> 16) | mm[5]; vm.w = (mm[6] >> 16) | mm[7];
            vl.x = (ll[0] >> 16) | (ll[1] & 0xFFFF0000u); vl.y = (ll[2] >> 16) | (ll[3] & 0xFFFF0000u);
            vl.z = (ll[4] >> 16) | (ll[5] & 0xFFFF0000u); vl.w = (ll[6] >> 16) | (ll[7] & 0xFFFF0000u);
            const bf16x8 B[3] = {__builtin_bit_cast(bf16x8, vh), __builtin_bit_cast(bf16x8, vm),
                                 __builtin_bit_cast(bf16x8, vl)};
#pragma unroll
            for (int pa = 2; pa >= 0; --pa)
#pragma unroll
                for (int pb = 2 - pa; pb >= 0; --pb)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws[c][pa], B[pb], acc, 0, 0, 0);
        }
        // epilogue: lane = pixel, register r = channel 32w + (r&3) + 8(r>>2) + 4kh
        const unsigned eoff = (unsigned)((wave * 32 + 4 * kh) * N + px) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t row = (size_t)((r & 3) + 8 * (r >> 2)) * N * 4;
            const float res = *reinterpret_cast<const float *>(xb + row + eoff);
            acc[r] = acc[r] * rs + bias[r] + res;
            *reinterpret_cast<float *>(yb + row + eoff) = acc[r];
        }
        if (a.y_pf) {
            const int yy = px / a.W, xx = px - yy * a.W;
            const long long u0 = (long long)b * a.pf_bs + (long long)(yy + 1) * (a.W + 2) + xx + 1 +
                                 (long long)(wave * 4) * 2 * a.pf_ps;
            pf_store_block(reinterpret_cast<uint4 *>(a.y_pf), u0, a.pf_ps, kh, acc);
        }
    }
}

hipError_t lnconv_launch(const LnConvArgs &a, int B, hipStream_t st) {
    if (a.N % (32 * a.nsplit)) return hipErrorInvalidValue;
    dim3 grid((unsigned)a.nsplit, (unsigned)B);
    if (a.C == 64) hipLaunchKernelGGL(lnconv_kernel<2>, grid, dim3(128), 0, st, a);
    else if (a.C == 128) hipLaunchKernelGGL(lnconv_kernel<4>, grid, dim3(256), 0, st, a);
    else if (a.C == 192) hipLaunchKernelGGL(lnconv_kernel<6>, grid, dim3(384), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace cdc
