// conv_kernel.h -- implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 with fused epilogue.
// See conv_args.h for the orientation.  Included by the conv_inst_*.hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_args.h"

namespace cdc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Two-plane fp16 split of an activation (see conv_split_kernel.h, AR = 1): a = h + l * 2^-11.
__device__ __forceinline__ void split2h(float a, _Float16 &h, _Float16 &l) {
    h = (_Float16)a;
    l = (_Float16)((a - (float)h) * 2048.0f);
}

// Stores the 16 accumulator values of one 32x32 block (lane = pixel, r -> channel (r&3) + 8(r>>2) + 4*half)
// as PF units (conv_pf_kernel.h): per 8-channel group the lane owns 4 consecutive channels = 8 bytes of each
// plane's 16-byte unit; lanes l and l+32 complete the unit, a wave writes 512-byte runs.
__device__ __forceinline__ void pf_store_block(uint4 *pf, long long unit0, long long pf_ps, int half, const f32x16 &v) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        _Float16 h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split2h(v[g * 4 + i], h[i], l[i]);
        f16x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
        uint2 *p = reinterpret_cast<uint2 *>(pf + unit0 + (long long)g * 2 * pf_ps) + half;
        *p = __builtin_bit_cast(uint2, hv);
        *(p + 2 * pf_ps) = __builtin_bit_cast(uint2, lv);
    }
}

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned magic) {
    return magic ? __umulhi(n, magic) : n;
}

__device__ __attribute__((aligned(16))) float g_zeros[64];   // DMA source of zero padding

// LDS-DMA (global_load_lds): each lane supplies a global address; the data lands at
// LDS[M0 + lane*size].  Hand-written so that the wait state between the M0 write and the DMA is
// explicit: with the compiler builtin the M0 update is emitted back-to-back with the DMA and whole
// workgroup tiles came out wrong about once per 500 workgroups (stale-M0 signature: a piece lands at
// the previous piece's base).  M0 is compiler-reserved, hence saved and restored.  hipcc does not
// count these loads: dma_wait() must precede the barrier that publishes the buffer.
__device__ __forceinline__ void dma_b128(const float *g, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_byte)) : "memory");
}
__device__ __forceinline__ void dma_b32(const float *g, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_byte)) : "memory");
}
// same, source address = scalar base (SGPR pair) + per-lane unsigned 32-bit byte offset
__device__ __forceinline__ void dma_b128_s(unsigned voff, const float *sbase, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_byte)) : "memory");
}
__device__ __forceinline__ void dma_b32_s(unsigned voff, const float *sbase, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_byte)) : "memory");
}
__device__ __forceinline__ const float *uniform_ptr(const float *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const float *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}

constexpr int conv_min_waves(int MB, int NPW) { return MB * NPW <= 8 ? 2 : 1; }

struct TileGeom {      // where this workgroup / lane sits (shared by the kernel variants' epilogue)
    int tid, nthr, wave, half, pr, pc, b, z, cog, oy0, ox0, NBH;
    // small feature maps: a workgroup covers ipw images, wpi waves each (b = first image; default 1 image)
    int ipw = 1, wpi = 4, nimg = 1 << 30;
    long long out_off = 0;      // split-K: this slice's partial-sum plane
    bool no_bias = false;       // split-K: slices > 0
};

// Fused epilogue on the MFMA accumulators (C/D layout of the 32x32 forms: lane = pixel l&31, register
// r = channel (r&3) + 8(r>>2) + 4(l>>5)): bias, hoisted partial sums, channel LayerNorm (+ReLU),
// time-embedding shift, residual add, LN statistics of the result, NCHW store.
template <int MB, int NPW, int LNMODE, int ABL>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &P, const TileGeom &g, f32x16 (&acc)[MB][NPW],
                                              float *smem, const float *prstd) {
    constexpr int COPT = MB * 32;
    const int tid = g.tid, nthr = g.nthr, half = g.half, pr = g.pr, pc = g.pc;
    const int z = g.z, cog = g.cog, oy0 = g.oy0, ox0 = g.ox0, NBH = g.NBH;
    const int img_l = g.ipw > 1 ? g.wave / g.wpi : 0;          // image of this wave inside the workgroup
    const int wave = g.ipw > 1 ? g.wave % g.wpi : g.wave;      // row-block index inside its image
    const int b = g.b + img_l;
    const bool img_ok = b < g.nimg;
    // ---- epilogue -------------------------------------------------------------------------------
    __syncthreads();
    float *ep = smem;   // [3 + ipw (+3)][COPT]: bias, ln g, ln b, shift (one row per image), res3 weights
    for (int i = tid; i < COPT; i += nthr) {
        const int co = cog * COPT + i;
        const bool ok = co < P.Cout;
        ep[i] = (ok && P.bias && !g.no_bias) ? P.bias[co] : 0.f;
        ep[COPT + i] = (ok && P.ep_g) ? P.ep_g[co] : 0.f;
        ep[2 * COPT + i] = (ok && P.ep_b) ? P.ep_b[co] : 0.f;
        for (int q = 0; q < g.ipw; ++q)
            ep[(3 + q) * COPT + i] =
                (ok && P.shift && g.b + q < g.nimg) ? P.shift[(size_t)(g.b + q) * P.shift_bs + co] : 0.f;
        if (P.res3_w)
            for (int c = 0; c < 3; ++c) ep[(3 + g.ipw + c) * COPT + i] = ok ? P.res3_w[(size_t)c * P.COP + co] : 0.f;
    }
    __syncthreads();

    // Channel masks are only needed in the last, partially filled 32-channel block of a cout
    // group; the LayerNorm / statistics paths require Cout % 32 == 0 (host-enforced), so they
    // carry no masks at all.
    const float inv_c = 1.0f / (float)P.Cout;
    const int cobase = cog * COPT;
    const int nvalid = P.Cout - cobase;     // channels of this group that exist
    const float *epl = ep + 4 * half;       // lane's channel = const + 4*half
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const int oy = oy0 + (wave * NPW + n) * NBH + pr;
        const int ox = ox0 + pc;
        const bool valid = (oy < P.Ho) && (ox < P.Wo) && img_ok;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = LNMODE == 2 ? prstd[n] * P.acc_scale : P.acc_scale;
                acc[m][n][r] = acc[m][n][r] * sc + epl[m * 32 + (r & 3) + 8 * (r >> 2)];
            }
        const size_t pix = (size_t)oy * P.out_ys + (size_t)ox * P.out_xs + P.out_zoff[z];
        if (P.pre_add) {
            // hoisted partial sums (context half of a concatenated input), same addressing as out
            // (loading them BEFORE the K loop as the accumulators' start value was measured in round 4: slower, 0.40 -> 0.43 ms on
            // the first 7x1 layer -- the 64 loads per lane delay the prologue, while here the other workgroups of the CU cover them)
            const float *pp = P.pre_add + (size_t)b * P.out_bs + pix +
                              (size_t)(cobase + 4 * half) * P.out_cs;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m * 32 + 32 <= nvalid) {
                    if (valid) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[m][n][r] += pp[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.out_cs];
                    }
                } else if (m * 32 < nvalid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ci = m * 32 + (r & 3) + 8 * (r >> 2);
                        if (valid && ci + 4 * half < nvalid) acc[m][n][r] += pp[(size_t)ci * P.out_cs];
                    }
                }
            }
        }
        if (P.ep_g) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[m][n][r];
            s += __shfl_xor(s, 32);
            const float mean = s * inv_c;
            float q = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[m][n][r] - mean;
                    q += d * d;
                }
            q += __shfl_xor(q, 32);
            // range guard: an inf / NaN accumulator (an operand left the fp16 range) makes q non-finite; the LayerNorm
            // + ReLU below could turn it into a finite, wrong value, so it is reported here
            if (P.fault && !(q < 3.0e38f)) *P.fault = 1;
            const float rinv = 1.0f / sqrtf(q * inv_c + P.eps);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = m * 32 + (r & 3) + 8 * (r >> 2);
                    acc[m][n][r] = (acc[m][n][r] - mean) * rinv * epl[COPT + ci] + epl[2 * COPT + ci];
                }
        } else if (P.fault) {
            float s = 0.f;              // any inf / NaN among the pixel's values makes the sum of |v| non-finite
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += fabsf(acc[m][n][r]);
            if (!(s < 3.0e38f)) *P.fault = 1;
        }
        if (P.relu) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = fmaxf(acc[m][n][r], P.relu_slope * acc[m][n][r]);
        }
        if (P.shift) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[m][n][r] += epl[(3 + img_l) * COPT + m * 32 + (r & 3) + 8 * (r >> 2)];
        }
        if (P.resid && !g.no_bias) {       // (split-K: the residual enters through slice 0 only)
            const float *rp = P.resid + (size_t)b * P.resid_bs + pix +
                              (size_t)(cobase + 4 * half) * P.resid_cs;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m * 32 + 32 <= nvalid) {
                    if (valid) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[m][n][r] += rp[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.resid_cs];
                    }
                } else if (m * 32 < nvalid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ci = m * 32 + (r & 3) + 8 * (r >> 2);
                        if (valid && ci + 4 * half < nvalid) acc[m][n][r] += rp[(size_t)ci * P.resid_cs];
                    }
                }
            }
        }
        if (P.res3_w) {
            const size_t hw = (size_t)P.Ho * P.Wo;
            const float *xp = P.res3_x + (size_t)b * P.res3_bs + (size_t)oy * P.Wo + ox;
            const float x0 = valid ? xp[0] : 0.f, x1 = valid ? xp[hw] : 0.f, x2 = valid ? xp[2 * hw] : 0.f;
            const float *w3 = epl + (3 + g.ipw) * COPT;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = m * 32 + (r & 3) + 8 * (r >> 2);
                    acc[m][n][r] += w3[ci] * x0 + w3[COPT + ci] * x1 + w3[2 * COPT + ci] * x2;
                }
        }
        if (P.stat_mean) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[m][n][r];
            s += __shfl_xor(s, 32);
            const float mean = s * inv_c;
            float q = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[m][n][r] - mean;
                    q += d * d;
                }
            q += __shfl_xor(q, 32);
            if (valid && half == 0) {     // one plane of the output tensor (phase-aware addressing)
                P.stat_mean[(size_t)b * P.out_cs + pix] = mean;
                P.stat_rstd[(size_t)b * P.out_cs + pix] = 1.0f / sqrtf(q * inv_c + P.eps);
            }
        }
        if constexpr ((ABL & 8) != 0) {          // timing aid: one store per lane instead of MB*16
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[m][n][r];
            if (valid) P.out[(size_t)b * P.out_bs + pix + (size_t)(cobase + 4 * half) * P.out_cs] = t;
            continue;
        }
        if (P.out_pf && valid) {           // (host: Cout % 32 == 0, no split-K slices)
            const long long u0 = (long long)b * P.pf_bs + (long long)oy * P.pf_ys + (long long)ox * P.pf_xs + P.pf_zoff[z];
#pragma unroll
            for (int m = 0; m < MB; ++m)
                pf_store_block(reinterpret_cast<uint4 *>(P.out_pf), u0 + (long long)((cobase >> 3) + m * 4) * 2 * P.pf_ps,
                               P.pf_ps, half, acc[m][n]);
        }
        if (P.pf_only) continue;
        float *op = P.out + g.out_off + (size_t)b * P.out_bs + pix + (size_t)(cobase + 4 * half) * P.out_cs;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m * 32 + 32 <= nvalid) {
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        op[(size_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * P.out_cs] = acc[m][n][r];
                }
            } else if (m * 32 < nvalid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = m * 32 + (r & 3) + 8 * (r >> 2);
                    if (valid && ci + 4 * half < nvalid) op[(size_t)ci * P.out_cs] = acc[m][n][r];
                }
            }
        }
    }
}

// Workgroup = WN waves (blockDim.x = 64*WN).  Wave w owns NPW N-blocks (32 pixels each) stacked
// vertically and all MB*32 output channels of the workgroup's cout group.
//
// Staging: both operands go global -> LDS by LDS-DMA (global_load_lds, no staging registers): the
// weight slab [taps][KC][MB*32] in 16-byte pieces, the haloed input patch [KC][PH][PW] in 4-byte
// pieces whose per-lane source address is either the tensor element or a zero word (padding,
// channel tail).  Two LDS buffer sets; chunk c+1 streams in while chunk c feeds the MFMAs; one
// barrier per chunk (hipcc drains vmcnt(0) in front of it).
//
// LNMODE: PreNorm LayerNorm of the input (network_components.py:69-77) fused into the convolution.
//   0  none
//   1  general: the landed patch is normalised in place in LDS (any k x k, zero padding preserved)
//   2  1x1 only, algebraically folded: LN(x).W = rstd[p] * ((x - mean[p]) . (g*W)) + (W.b); the host
//      packs g*W and W.b, the kernel subtracts the pixel mean while fetching the B operand and
//      scales the accumulators by rstd[p] in the epilogue -- no extra LDS pass, no extra barrier.
// ABL (compile-time, tuning aid only): 1 no DMA after chunk 0, 2 no operand fetch, 4 no per-chunk barrier,
// 8 one store per lane, 16 no prologue descriptors/DMA at all -- timing experiments, wrong results.
template <int MB, int NPW, int LNMODE, int ABL = 0>
__global__ void __launch_bounds__(256, conv_min_waves(MB, NPW)) conv_mfma_kernel(const ConvArgs P) {
    constexpr bool LNLOAD = LNMODE == 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int COPT = MB * 32;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WN = nthr >> 6;
    const int z = blockIdx.z;
    const int cog = blockIdx.y;

    int bid = blockIdx.x;
    if (P.xcd_remap) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);   // one contiguous tile band per XCD
    const int tx = bid % P.tiles_x;
    bid /= P.tiles_x;
    const int ty = bid % P.tiles_y;
    const int b = bid / P.tiles_y;

    const int NBW = 1 << P.lognbw;
    const int NBH = 32 >> P.lognbw;
    const int TH = WN * NPW * NBH;
    const int oy0 = ty * TH, ox0 = tx * NBW;
    const int iy0 = oy0 * P.stride - P.pad_y[z];
    const int ix0 = ox0 * P.stride - P.pad_x[z];
    const int PH = P.PH, PW = P.PW;
    const int taps = P.KH * P.KW;
    const int KC = P.KC;
    const int plane = PH * PW;

    // Input patch pieces: 4-byte (any geometry) or, when rows can be 16-byte aligned in global memory
    // (W % 4 == 0: the patch is widened to start at a multiple of 4, P.xshift[z] columns early), 16-byte.
    const int xv = P.xvec ? 4 : 1;
    const int n_x = KC * plane / xv;                  // staged input pieces per chunk
    const int n_w4 = taps * KC * (COPT / 4);          // staged weight float4 per chunk
    const int xs = (n_x + nthr - 1) / nthr;           // DMA slots per thread
    const int ws = (n_w4 + nthr - 1) / nthr;
    const int w_floats = ws * nthr * 4;
    const int buf_floats = w_floats + xs * nthr * xv;

    // ---- chunk-invariant descriptors of this thread's input-patch slots --------------------------
    // xo = (c_local << 27) | (iy*W + ix) for an element inside the image, -1 for zero padding and for
    // slots beyond the patch.  Padding positions are the same in every chunk: they are zeroed once in
    // both LDS buffers and never written again (their lanes are masked out of the DMA), so the
    // per-chunk issue needs no bounds logic at all.
    int xo[kXS];
    float xmean[LNLOAD ? kXS : 1], xrstd[LNLOAD ? kXS : 1];
#pragma unroll
    for (int i = 0; i < kXS; ++i) {
        const unsigned e = tid + i * nthr;
        xo[i] = -1;
        if constexpr (LNLOAD) { xmean[i] = 0.f; xrstd[i] = 0.f; }
        if (i < xs) {
            if (e < (unsigned)n_x) {
                const unsigned c = fdiv(e, P.magic_hw);               // / (PH * PW / xv)
                const unsigned rem = e - c * (unsigned)(plane / xv);
                const unsigned r = fdiv(rem, P.magic_w);              // / (PW / xv)
                const unsigned col = (rem - r * (unsigned)(PW / xv)) * xv;
                const int iy = iy0 + (int)r, ix = ix0 - P.xshift[z] + (int)col;
                if (iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) {     // a 16-byte piece is all in or all out
                    xo[i] = (int)(c << 27) | (iy * P.W + ix);
                    if constexpr (LNLOAD) {
                        xmean[i] = P.ln_mean[(size_t)b * P.H * P.W + iy * P.W + ix];
                        xrstd[i] = P.ln_rstd[(size_t)b * P.H * P.W + iy * P.W + ix];
                    }
                }
            }
            if (xo[i] < 0) {
                for (int t = 0; t < xv; ++t) {
                    smem[w_floats + e * xv + t] = 0.f;
                    smem[buf_floats + w_floats + e * xv + t] = 0.f;
                }
            }
        }
    }
    const unsigned HW = (unsigned)(P.H * P.W);
    const float *s0 = P.src0 + (size_t)b * P.src0_bs;
    const float *s1 = P.src1 ? P.src1 + (size_t)b * P.src1_bs : nullptr;
    const float *wsrc = P.wp + (size_t)b * P.w_bs + (size_t)z * P.w_zs + (size_t)cog * COPT;

    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // A chunk lies entirely in one concat source (the host guarantees C0 % KC == 0), so the source
    // address is a scalar base per chunk plus a 32-bit per-lane offset.
    auto issue = [&](int chunk, int which) {
        const int cbase = chunk * KC;
        const unsigned buf_lds = smem_lds + (unsigned)(which * buf_floats) * 4u;
        const float *wbase = uniform_ptr(wsrc + (size_t)cbase * P.COP);
        for (int i = 0; i < ws; ++i) {
            const int e4 = tid + i * nthr;
            if (e4 < n_w4) {
                const int row = e4 / (COPT / 4);
                const int c4 = e4 - row * (COPT / 4);
                const int tap = row >> P.logKC, kcl = row & (KC - 1);
                const unsigned voff = (unsigned)((tap * P.Cin_pad + kcl) * P.COP + c4 * 4) * 4u;
                dma_b128_s(voff, wbase, buf_lds + (unsigned)(i * nthr + wave * 64) * 16u);
            }
        }
        const float *xbase = uniform_ptr(cbase < P.C0 ? s0 + (size_t)cbase * HW
                                                      : s1 + (size_t)(cbase - P.C0) * HW);
        const int ncm1 = min(KC, P.Cin - cbase) - 1;       // channel tail: re-read the last valid
        const unsigned xb_lds = buf_lds + (unsigned)w_floats * 4u;   // channel (its weights are zero)
#pragma unroll
        for (int i = 0; i < kXS; ++i) {
            if (i < xs && xo[i] >= 0) {
                const unsigned c = (unsigned)min(xo[i] >> 27, ncm1);
                const unsigned voff = (c * HW + (unsigned)(xo[i] & 0x7FFFFFF)) * 4u;
                if (P.xvec) dma_b128_s(voff, xbase, xb_lds + (unsigned)(i * nthr + wave * 64) * 16u);
                else dma_b32_s(voff, xbase, xb_lds + (unsigned)(i * nthr + wave * 64) * 4u);
            }
        }
    };

    f32x16 acc[MB][NPW];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NPW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int half = lane >> 5;
    const int j = lane & 31;
    const int pr = j >> P.lognbw, pc = j & (NBW - 1);
    const int a_lane = half * COPT + j;
    const int b_lane = half * plane + (wave * NPW * NBH + pr) * P.stride * PW + pc * P.stride + P.xshift[z];
    const int nb_stride = NBH * P.stride * PW;

    float pmean[LNMODE == 2 ? NPW : 1], prstd[LNMODE == 2 ? NPW : 1];
    if constexpr (LNMODE == 2) {
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int oy = oy0 + (wave * NPW + n) * NBH + pr, ox = ox0 + pc;
            const bool ok = oy < P.Ho && ox < P.Wo;
            pmean[n] = ok ? P.ln_mean[(size_t)b * P.H * P.W + oy * P.W + ox] : 0.f;
            prstd[n] = ok ? P.ln_rstd[(size_t)b * P.H * P.W + oy * P.W + ox] : 0.f;
        }
    }

    issue(0, 0);
    for (int chunk = 0; chunk < P.nchunk; ++chunk) {
        float *buf = smem + (chunk & 1) * buf_floats;
        dma_wait();               // this wave's pieces of `chunk` have landed ...
        if (!(ABL & 4) || chunk == 0)
            __syncthreads();      // ... and so have everyone's; the other buffer is free again
        if (chunk + 1 < P.nchunk && !(ABL & 1)) issue(chunk + 1, (chunk + 1) & 1);
        float *w_lds = buf;
        float *x_lds = buf + w_floats;
        if constexpr (LNLOAD) {
            // PreNorm LayerNorm applied in place on the landed patch (network_components.py:69-77)
            const int cbase = chunk * KC;
#pragma unroll
            for (int i = 0; i < kXS; ++i) {
                if (i < xs && xo[i] >= 0) {
                    const int c = cbase + (xo[i] >> 27);
                    if (c < P.Cin) {
                        const int e = tid + i * nthr;
                        x_lds[e] = (x_lds[e] - xmean[i]) * xrstd[i] * P.ln_g[c] + P.ln_b[c];
                    }
                }
            }
            __syncthreads();
        }
        // Flattened (tap, channel-pair) sequence, software-pipelined by hand: the A/B operands of
        // step s+1 are fetched from LDS before the MFMAs of step s issue, so one wave alone keeps
        // the matrix pipe busy (the LDS latency hides behind MB*NPW MFMAs of 64 cycles each).
        {
            int nky = 0, nkx = 0, nkc = 0;           // (tap, channel) of the NEXT step to fetch
            auto fetch = [&](float (&a)[MB], float (&bv)[NPW]) {
                const float *wl = w_lds + ((nky * P.KW + nkx) * KC + nkc) * COPT + a_lane;
                const float *xl = x_lds + b_lane + nky * PW + nkx + nkc * plane;
#pragma unroll
                for (int m = 0; m < MB; ++m) a[m] = wl[m * 32];
#pragma unroll
                for (int n = 0; n < NPW; ++n) bv[n] = xl[n * nb_stride];
                nkc += 2;
                if (nkc == KC) { nkc = 0; if (++nkx == P.KW) { nkx = 0; ++nky; } }
            };
            auto fma = [&](const float (&a)[MB], const float (&bv)[NPW]) {
#pragma unroll
                for (int n = 0; n < NPW; ++n) {
                    const float bn = LNMODE == 2 ? bv[n] - pmean[n] : bv[n];
#pragma unroll
                    for (int m = 0; m < MB; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bn, acc[m][n], 0, 0, 0);
                }
            };
            const int nsteps = taps * (KC >> 1);     // even: KC is a multiple of 4
            float a0[MB], b0[NPW], a1[MB], b1[NPW];
            fetch(a0, b0);
            if constexpr ((ABL & 2) != 0) {
                fetch(a1, b1);
                for (int st = 0; st < nsteps; st += 2) { fma(a0, b0); fma(a1, b1); }
            } else {
                for (int st = 0; st < nsteps; st += 2) {
                    fetch(a1, b1);
                    fma(a0, b0);
                    if (st + 2 < nsteps) fetch(a0, b0);
                    fma(a1, b1);
                }
            }
        }
    }

    // ---- epilogue -------------------------------------------------------------------------------
    const TileGeom geom{tid, nthr, wave, half, pr, pc, b, z, cog, oy0, ox0, NBH};
    conv_epilogue<MB, NPW, LNMODE, ABL>(P, geom, acc, smem, prstd);
}

typedef void (*conv_kernel_fn)(const ConvArgs);

}  // namespace cdc
