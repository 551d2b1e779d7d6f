// conv_split_kernel.h -- k x k convolution with fp32-exact products on the bf16 matrix cores.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  Every fp32 number is EXACTLY the sum of
// three bf16 numbers (a = a1 + a2 + a3: top / middle / bottom 8 significant bits, obtained by
// truncation, so no rounding is involved), and a bf16 x bf16 product is exact in fp32.  Hence
//      a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1) + O(2^-24 |ab|)
// is six v_mfma_f32_32x32x16_bf16 per 16-deep K step instead of eight 32x32x2_f32: 2.67x the matrix
// rate at fp32-class accuracy (measured end to end against a float64 convolution: 1.9e-6 vs 8e-6 for plain
// fp32; the dropped terms a2b3 + a3b2 + a3b3 are below the fp32 rounding of the running sum).
//
// Tensors stay float32 NCHW in HBM.  Per 16-channel chunk the haloed input patch lands in LDS as fp32
// (16-byte LDS-DMA, same scheme as conv_kernel.h), is split by the workgroup into three bf16 planes in
// MFMA B-operand order [plane][k-half][row][col][8 channels] (one ds_read_b128 per operand), and the
// pre-split weights [tap][chunk][plane][k-half][cout][8 cin] stream in one kernel row (KW taps) at a
// time through a two-stage LDS ring.  Accumulators, tiling, epilogue: as in conv_kernel.h.
#pragma once
#include "conv_kernel.h"

namespace cdc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- AR = 1: two-plane fp16 operands -----------------------------------------------------------------
// a = h + l' * 2^-11 with h = fp16(a) (round to nearest) and l' = fp16((a - h) * 2^11): the residual is
// scaled back into h's binade, so it is a NORMAL fp16 number wherever h is (no subnormal loss for small
// activations) and the pair carries 22-23 significant bits for 6e-5 <= |a| < 65504 (absolute error
// <= 1.5e-11 below that).  The weights come pre-scaled by a per-layer power of two 2^s (max |w| 2^s in
// [2^13, 2^14)) as three planes WH = fp16(w 2^s), WL = fp16(w 2^s - WH), WH2 = WH 2^-11 (exact), and
//      a * w 2^s = h WH + h WL + l' WH2  + O(2^-22 |a w 2^s|)
// is three v_mfma_f32_32x32x16_f16 per 16-deep K step (fp16 x fp16 products are exact in the fp32
// accumulator); the epilogue multiplies the accumulators by 2^-s (exact).  |a| >= 65504 becomes inf / NaN
// and propagates to the output, where the sampler kernels flag it (the caller then re-runs the exact
// three-plane bf16 arithmetic, AR = 0).
// (split2h lives in conv_kernel.h: the shared epilogue uses it to emit PF tensors)

// a -> (hi, mid, lo) as fp32 bit patterns whose low 16 bits are zero; exact: a == hi + mid + lo
__device__ __forceinline__ void split3(float a, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(a) & 0xFFFF0000u;
    const float r = a - __uint_as_float(h);
    m = __float_as_uint(r) & 0xFFFF0000u;
    l = __float_as_uint(r - __uint_as_float(m));          // <= 8 significant bits: already a bf16
}

// ABL (tuning aid, wrong results): 1 convert only chunk 0, 2 no input DMA after chunk 0, 4 no weight DMA after
// the first row, 8 no per-row barrier, 16 skip the MFMAs
template <int MB, int NPW, int ABL = 0>
__global__ void __launch_bounds__(256, 1) conv_split_kernel(const ConvArgs P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int COPT = MB * 32;
    constexpr int KC = 16;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WN = nthr >> 6;
    const int z = blockIdx.z;
    const int cog = blockIdx.y;

    int bid = blockIdx.x;
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
    const int PH = P.PH, PW = P.PW;                   // PW: 16-byte aligned row stride (floats)
    const int plane = PH * PW;
    const int taps = P.KH * P.KW;
    const int nc16 = P.Cin_pad >> 4;

    // LDS carve (float units): xc = split patch (3 planes x 2 k-halves x plane x 16 B),
    // xs = fp32 landing area (16 channels x plane), w = 2 stages x KW taps x 6 x COPT x 16 B
    const int xc_floats = 24 * plane, xs_floats = 16 * plane, wst_floats = P.KW * 24 * COPT;
    float *xc = smem;
    float *xs = smem + xc_floats;
    float *wl = xs + xs_floats;
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned xs_lds = smem_lds + (unsigned)xc_floats * 4u;
    const unsigned wl_lds = xs_lds + (unsigned)xs_floats * 4u;

    // ---- input patch descriptors: 16-byte pieces (4 columns of one channel row) --------------------
    const int n_x = 4 * plane;                        // pieces per chunk (16 ch * plane / 4)
    const int xsl = (n_x + nthr - 1) / nthr;
    int xo[kXS];
#pragma unroll
    for (int i = 0; i < kXS; ++i) {
        const unsigned e = tid + i * nthr;
        xo[i] = -1;
        if (i < xsl) {
            if (e < (unsigned)n_x) {
                const unsigned c = fdiv(e, P.magic_hw);               // / (plane / 4)
                const unsigned rem = e - c * (unsigned)(plane / 4);
                const unsigned r = fdiv(rem, P.magic_w);              // / (PW / 4)
                const unsigned col = (rem - r * (unsigned)(PW / 4)) * 4;
                const int iy = iy0 + (int)r, ix = ix0 - P.xshift[z] + (int)col;
                if (iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) xo[i] = (int)(c << 27) | (iy * P.W + ix);
            }
            if (xo[i] < 0 && e < (unsigned)n_x)
                *reinterpret_cast<float4 *>(xs + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const unsigned HW = (unsigned)(P.H * P.W);
    const float *s0 = P.src0 + (size_t)b * P.src0_bs;
    const float *s1 = P.src1 ? P.src1 + (size_t)b * P.src1_bs : nullptr;

    auto issue_x = [&](int chunk) {
        const int cbase = chunk * KC;
        const float *xbase = uniform_ptr(cbase < P.C0 ? s0 + (size_t)cbase * HW
                                                      : s1 + (size_t)(cbase - P.C0) * HW);
        const int ncm1 = min(KC, P.Cin - cbase) - 1;
#pragma unroll
        for (int i = 0; i < kXS; ++i) {
            if (i < xsl && xo[i] >= 0) {
                const unsigned c = (unsigned)min(xo[i] >> 27, ncm1);
                const unsigned voff = (c * HW + (unsigned)(xo[i] & 0x7FFFFFF)) * 4u;
                dma_b128_s(voff, xbase, xs_lds + (unsigned)(i * nthr + wave * 64) * 16u);
            }
        }
    };
    // weights of kernel row `ky`, chunk `chunk`: [KW taps][3 planes][2 k-halves][COPT] 16-byte pieces
    const int n_w = P.KW * 6 * COPT;
    const int wsl = (n_w + nthr - 1) / nthr;
    const unsigned short *wsrc = P.wsp + (size_t)z * P.wsp_zs + (size_t)cog * COPT * 8;
    auto issue_w = [&](int ky, int chunk, int stage) {
        const unsigned short *base = wsrc + ((size_t)(ky * P.KW) * nc16 + chunk) * 6 * P.COP * 8;
        const float *wbase = uniform_ptr(reinterpret_cast<const float *>(base));
        for (int i = 0; i < wsl; ++i) {
            const int e = tid + i * nthr;
            if (e < n_w) {
                const int t = e / (6 * COPT);
                const int rem = e - t * 6 * COPT;
                const int pk = rem / COPT, co = rem - pk * COPT;
                const unsigned voff = (unsigned)(((t * nc16) * 6 + pk) * P.COP + co) * 16u;
                dma_b128_s(voff, wbase, wl_lds + (unsigned)(stage * wst_floats) * 4u +
                                            (unsigned)(i * nthr + wave * 64) * 16u);
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
    // B operand (16 B units): ((plane_p*2 + half) * PH*PW + row*PW + col); A: ((tap*6 + p*2 + half)*COPT + cout)
    const int b_lane = half * plane + (wave * NPW * NBH + pr) * P.stride * PW + pc * P.stride + P.xshift[z];
    const int nb_stride = NBH * P.stride * PW;
    const int a_lane = half * COPT + j;

    issue_x(0);
    issue_w(0, 0, 0);
    int wstage = 0;
    for (int chunk = 0; chunk < nc16; ++chunk) {
        dma_wait();
        __syncthreads();                    // fp32 patch of `chunk` (and the first weight row) landed
        // ---- split the patch into three bf16 planes, 8 channels per 16-byte unit --------------------
        if (!(ABL & 1) || chunk == 0)
        for (int u = tid; u < 2 * plane; u += nthr) {
            const int kg = u >= plane ? 1 : 0;
            const int rc = u - kg * plane;
            const float *src = xs + (kg * 8) * plane + rc;
            unsigned hh[8], mm[8], ll[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) split3(src[q * plane], hh[q], mm[q], ll[q]);
            uint4 vh, vm, vl;
            vh.x = (hh[0] >> 16) | hh[1]; vh.y = (hh[2] >> 16) | hh[3];
            vh.z = (hh[4] >> 16) | hh[5]; vh.w = (hh[6] >> 16) | hh[7];
            vm.x = (mm[0] >> 16) | mm[1]; vm.y = (mm[2] >> 16) | mm[3];
            vm.z = (mm[4] >> 16) | mm[5]; vm.w = (mm[6] >> 16) | mm[7];
            vl.x = (ll[0] >> 16) | (ll[1] & 0xFFFF0000u); vl.y = (ll[2] >> 16) | (ll[3] & 0xFFFF0000u);
            vl.z = (ll[4] >> 16) | (ll[5] & 0xFFFF0000u); vl.w = (ll[6] >> 16) | (ll[7] & 0xFFFF0000u);
            uint4 *dst = reinterpret_cast<uint4 *>(xc);
            dst[(0 * 2 + kg) * plane + rc] = vh;
            dst[(1 * 2 + kg) * plane + rc] = vm;
            dst[(2 * 2 + kg) * plane + rc] = vl;
        }
        __syncthreads();                    // split patch ready; landing area free again
        if (chunk + 1 < nc16 && !(ABL & 2)) issue_x(chunk + 1);
        for (int ky = 0; ky < P.KH; ++ky) {
            // prefetch the next weight row (next ky, or row 0 of the next chunk) into the other stage
            if (!(ABL & 4)) {
                if (ky + 1 < P.KH) issue_w(ky + 1, chunk, wstage ^ 1);
                else if (chunk + 1 < nc16) issue_w(0, chunk + 1, wstage ^ 1);
            }
            const uint4 *wa = reinterpret_cast<const uint4 *>(wl + wstage * wst_floats);
            const uint4 *xb = reinterpret_cast<const uint4 *>(xc);
            for (int kx = 0; kx < P.KW; ++kx) {
                bf16x8 A[3][MB], Bv[3][NPW];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const uint4 t = wa[(kx * 6 + p * 2) * COPT + a_lane + m * 32];
                        A[p][m] = __builtin_bit_cast(bf16x8, t);
                    }
#pragma unroll
                    for (int n = 0; n < NPW; ++n) {
                        const uint4 t = xb[(p * 2) * plane + b_lane + ky * PW + kx + n * nb_stride];
                        Bv[p][n] = __builtin_bit_cast(bf16x8, t);
                    }
                }
                // six product terms, smallest first
                if constexpr ((ABL & 16) != 0) {
                    asm volatile("" ::"v"(A[0][0]), "v"(Bv[0][0]), "v"(A[2][MB - 1]), "v"(Bv[2][NPW - 1]));
                } else
#pragma unroll
                for (int term = 0; term < 6; ++term) {
                    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
                    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int n = 0; n < NPW; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[PA[term]][m], Bv[PB[term]][n],
                                                                                acc[m][n], 0, 0, 0);
                }
            }
            dma_wait();
            if (!(ABL & 8)) __syncthreads();   // next weight row landed; this stage may be overwritten
            wstage ^= 1;
        }
    }

    const TileGeom geom{tid, nthr, wave, half, pr, pc, b, z, cog, oy0, ox0, NBH};
    conv_epilogue<MB, NPW, 0, 0>(P, geom, acc, smem, nullptr);
}


// ------------------------------------------------------------------------------------------------
// Variant for small channel groups (MB*NPW <= 4): two workgroups per CU.  The ablation of the kernel
// above (round 2, profiles/HISTORY_r01_r04.md) shows MFMA time 0.56 ms and everything else 0.61 ms with NO
// overlap at one wave per SIMD, so this variant shrinks the LDS footprint below 80 KiB: the fp32 patch
// never touches LDS -- it is fetched global -> registers one chunk ahead (16-byte loads of the aligned
// patch rows), split in registers and written straight into the three bf16 planes.  While one
// workgroup converts / waits, the other one owns the matrix pipe.
// ------------------------------------------------------------------------------------------------
constexpr int kXR = 8;      // float4 registers per thread for the in-flight patch

// LNMODE 2 (1x1 only): PreNorm LayerNorm folded as in conv_kernel.h -- the pixel mean is subtracted before
// the split, the accumulators are scaled by rstd in the epilogue, g and W.b live in the packed weights.
// three workgroups per CU wherever the accumulators leave room (<= 168 VGPRs)
constexpr int split2_min_wgs(int MB, int NPW, int LNMODE, int XU = 1, int AR = 0, int PIPE = 0) {
    if (PIPE && XU == 1 && MB * NPW <= 2) return 3;     // PIPE: two operand register sets; the small tiles still fit three workgroups per CU
    if (XU > 1 || PIPE) return MB * NPW > 4 ? 1 : 2;
    return (LNMODE == 0 ? MB * NPW <= 4 : (MB * NPW <= 2 || (LNMODE == 2 && (MB * NPW == 3 || (MB == 4 && NPW == 1))))) ? 3 : 2;
}

// XU = patch units per thread (2 for the large stride-2 patches).  Stride 2 keeps the even and the odd
// patch columns in separate half-rows of the LDS planes, so that the B-operand reads of a tap (every
// other column) are contiguous 16-byte units instead of a 2-way bank conflict.
// (An all-phase ConvTranspose2d variant -- four phases per workgroup, one shared patch -- was measured slower than four
// phase launches folded onto one XCD, 0.54 vs 0.46 ms at 128^2 -> 256^2, and was removed in round 2.)
// PIPE = 1 (AR = 1 only; round 4): software-pipelined tap loop.  The round-3 loop read the operands of a tap, waited for LDS,
// multiplied, read the next term's weights, waited, ... -- three to four dependent LDS round trips per tap with two to four MFMAs
// between them (cycle timeline of wave 0, stride-2 64->64 @256^2: 1570 cycles per tap for 192 cycles of matrix work).  PIPE keeps two
// operand register sets: all ds_reads of tap t+1 (activation planes always; weight planes unless the tap opens a new weight stage,
// which is read right after the stage's barrier) are issued BEFORE the MFMAs of tap t.
// UF = 1 (round 4): unfold on load (ConvArgs::uf_c) -- the first 7x7 layer as a 7x1 convolution whose kx-unfolded input channels are
// gathered from the image while the patch is loaded (a compile-time variant: the same code behind a run-time branch in the common
// loader made every other layer of this kernel 5 - 25 % slower).
template <int MB, int NPW, int LNMODE = 0, int XU = 1, int AR = 0, int PIPE = 0, int UF = 0>
__global__ void __launch_bounds__(256, split2_min_wgs(MB, NPW, LNMODE, XU, AR, PIPE)) conv_split2_kernel(const ConvArgs P) {
    static_assert(PIPE == 0 || AR == 1, "the pipelined tap loop exists for the fp16 arithmetic only");
    static_assert(UF == 0 || (XU == 1 && LNMODE == 0), "unfold on load: stride 1, no LayerNorm on load");
    constexpr int NP = AR == 1 ? 2 : 3;               // B-operand (activation) planes
    static_assert(XU == 1 || LNMODE == 0, "two-unit variant carries no LayerNorm-on-load");
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef CDC_TIMELINE
    // development build (tools/build_variant.sh timeline -DCDC_TIMELINE): cycles of wave 0 per category, summed over the tile:
    // 0 prologue (descriptors, first loads), 1 convert + ds_write (store_x), 2 chunk-head wait + barrier, 3 tap loops (ds_read + MFMA),
    // 4 group-end DMA wait, 5 group-end barrier, 6 epilogue; slot 7 = start stamp, 8 = end stamp
    unsigned long long tl_acc[7] = {0, 0, 0, 0, 0, 0, 0}, tl_last = __builtin_readcyclecounter();
    const unsigned long long tl_start = tl_last;
#define TLC(c) do { const unsigned long long n_ = __builtin_readcyclecounter(); tl_acc[c] += n_ - tl_last; tl_last = n_; } while (0)
#else
#define TLC(c) do { } while (0)
#endif
    constexpr int COPT = MB * 32;
    constexpr int KC = 16;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WN = nthr >> 6;
    int z = blockIdx.z, ks = 0;
    unsigned bid0 = blockIdx.x;
    if (P.ksplit > 1) { ks = z / P.nzz; z -= ks * P.nzz; }
    if (P.zfold) { const unsigned g8 = bid0 >> 5, r = bid0 & 31; z = (int)(r >> 3); bid0 = g8 * 8 + (r & 7); }
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs; give every XCD a contiguous band
    // of tiles so that neighbouring tiles (shared halo rows, same weights) meet in ONE L2
    else if (P.xcd_remap) bid0 = (bid0 & 7) * (gridDim.x >> 3) + (bid0 >> 3);
    const int cog = blockIdx.y;

    const int NBW = 1 << P.lognbw;
    const int NBH = 32 >> P.lognbw;
    // Small feature maps: the workgroup covers ipw whole images, wpi waves each, so that the weight
    // stages are shared by four waves even when an image has only one or two 32-pixel blocks.
    const int ipw = P.ipw, wpi = WN / ipw;
    int bid = (int)bid0, tx = 0, ty = 0, b;
    if (ipw > 1) {
        b = bid * ipw + wave / wpi;
    } else {
        tx = bid % P.tiles_x;
        bid /= P.tiles_x;
        ty = bid % P.tiles_y;
        b = bid / P.tiles_y;
    }
    const bool img_ok = b < P.B;
    if (!img_ok) b = P.B - 1;                       // keep addresses valid; results are masked out
    const int rb = ipw > 1 ? wave % wpi : wave;     // this wave's row-block inside its image / tile
    const int team = wpi * 64, uid = (ipw > 1 ? (wave % wpi) * 64 : wave * 64) + lane;
    const int TH = wpi * NPW * NBH;
    const int oy0 = ty * TH, ox0 = tx * NBW;
    const int iy0 = oy0 * P.stride - P.pad_y[z];
    const int ix0 = ox0 * P.stride - P.pad_x[z];
    const int PH = P.PH, PW = P.PW;
    const int plane = PH * PW;
    const int nc16 = P.Cin_pad >> 4;

    const int TG = P.tg;                              // taps per weight stage (a kernel row, or 1)
    const int ntg1 = (P.KH * P.KW) / TG;              // stages per phase
    const int ntg = ntg1;                             // stages per chunk
    const int xc_floats = NP * 8 * plane, wst_floats = TG * 24 * COPT;
    float *xc = smem + (ipw > 1 ? (wave / wpi) * xc_floats : 0);     // one split patch per image
    float *wl = smem + ipw * xc_floats;
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned wl_lds = smem_lds + (unsigned)(ipw * xc_floats) * 4u;

    // ---- the patch unit owned by this thread: (k-half kg, row r, 4-column group q4) = 8 channels x 4
    // pixels = 8 float4 registers; after the split it becomes 4 pixels x 3 planes of 16-byte units
    // (host guarantees 2 * PH * PW/4 <= blockDim.x: one unit per thread)
    const int units = plane / 2;                      // 2 * plane / 4
    constexpr bool s2 = XU == 2;                      // the two-unit variant IS the stride-2 variant (host-enforced)
    const int ustep1 = s2 ? PW / 2 : 1, ustep2 = s2 ? 1 : 2;   // LDS position of unit pixel t: base + (t&1)*ustep1 + (t>>1)*ustep2
    int xsp[XU], ukg[XU], urc[XU];                    // xsp -2: no unit, -1: zero padding, else iy*W+ix
    int uix[UF ? XU : 1];                             // the unit's first image column (unfold on load)
    uix[0] = 0;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        xsp[u] = -2; ukg[u] = 0; urc[u] = 0;
        const int id = uid + u * team;
        if (id < units) {
            ukg[u] = id >= plane / 4 ? 1 : 0;
            const unsigned rem = id - ukg[u] * (plane / 4);
            const unsigned r = fdiv(rem, P.magic_w);      // / (PW / 4)
            const unsigned col = (rem - r * (unsigned)(PW / 4)) * 4;
            urc[u] = (int)(r * PW + (s2 ? col >> 1 : col));
            const int iy = iy0 + (int)r, ix = ix0 - P.xshift[z] + (int)col;
            xsp[u] = (iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) ? iy * P.W + ix : -1;
            if constexpr (UF != 0) uix[u] = ix;
        }
    }
    const unsigned HW = (unsigned)(P.H * P.W);
    const float *s0 = P.src0 + (size_t)b * P.src0_bs;
    const float *s1 = P.src1 ? P.src1 + (size_t)b * P.src1_bs : nullptr;

    float umean[4] = {0.f, 0.f, 0.f, 0.f};          // LNMODE 1/2: statistics of the unit's 4 pixels
    float urstd[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LNMODE != 0) {
        if (xsp[0] >= 0) {
            const float4 m4 = *reinterpret_cast<const float4 *>(P.ln_mean + (size_t)b * HW + (unsigned)xsp[0]);
            umean[0] = m4.x; umean[1] = m4.y; umean[2] = m4.z; umean[3] = m4.w;
            if constexpr (LNMODE == 1) {
                const float4 r4 = *reinterpret_cast<const float4 *>(P.ln_rstd + (size_t)b * HW + (unsigned)xsp[0]);
                urstd[0] = r4.x; urstd[1] = r4.y; urstd[2] = r4.z; urstd[3] = r4.w;
            }
        }
    }
    float4 xr[XU][kXR];
    // byte offset of the unit's first channel row inside a 16-channel chunk: the chunk / channel part of
    // the address is wave-uniform and goes into the scalar base (global_load ... v, s[base])
    unsigned xvo[XU];
#pragma unroll
    for (int u = 0; u < XU; ++u) xvo[u] = xsp[u] >= 0 ? ((unsigned)(ukg[u] * 8) * HW + (unsigned)xsp[u]) * 4u : 0u;
    auto load_x = [&](int chunk) {
        const int cbase = chunk * KC;
        if constexpr (UF != 0) {                      // unfold on load (ConvArgs::uf_c): 4-byte loads of the shifted image row
#pragma unroll
            for (int u = 0; u < XU; ++u)
#pragma unroll
                for (int i = 0; i < kXR; ++i) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int cc = cbase + ukg[u] * 8 + i;
                    if (xsp[u] >= 0 && cc < P.Cin) {
                        const int kx = (int)fdiv((unsigned)cc, P.uf_magic), c = cc - kx * P.uf_c;
                        const float *row = s0 + (size_t)c * HW + (unsigned)(xsp[u] - uix[u]);
                        const int sx = uix[u] + kx - P.uf_pad;
                        if (sx >= 0 && sx < P.W) v.x = row[sx];
                        if (sx + 1 >= 0 && sx + 1 < P.W) v.y = row[sx + 1];
                        if (sx + 2 >= 0 && sx + 2 < P.W) v.z = row[sx + 2];
                        if (sx + 3 >= 0 && sx + 3 < P.W) v.w = row[sx + 3];
                    }
                    xr[u][i] = v;
                }
            return;
        }
        const float *xbase = cbase < P.C0 ? s0 + (size_t)cbase * HW : s1 + (size_t)(cbase - P.C0) * HW;
        if (cbase + KC <= P.Cin) {
#pragma unroll
            for (int u = 0; u < XU; ++u)
#pragma unroll
                for (int i = 0; i < kXR; ++i) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    const char *rowb = reinterpret_cast<const char *>(xbase + (size_t)i * HW);
                    if (xsp[u] >= 0) v = *reinterpret_cast<const float4 *>(rowb + xvo[u]);
                    xr[u][i] = v;
                }
            return;
        }
        const int ncm1 = min(KC, P.Cin - cbase) - 1;  // channel tail: re-read the last valid (weights 0)
#pragma unroll
        for (int u = 0; u < XU; ++u)
#pragma unroll
            for (int i = 0; i < kXR; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xsp[u] >= 0) {
                    const unsigned c = (unsigned)min(ukg[u] * 8 + i, ncm1);
                    v = *reinterpret_cast<const float4 *>(xbase + (size_t)c * HW + (unsigned)xsp[u]);
                }
                xr[u][i] = v;
            }
    };
    auto store_x = [&](int chunk) {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if (xsp[u] > -2) {
                uint4 *dst = reinterpret_cast<uint4 *>(xc);
                float lg[8], lb[8];
                if constexpr (LNMODE == 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c = min(chunk * KC + ukg[u] * 8 + q, P.Cin - 1);
                        lg[q] = P.ln_g[c]; lb[q] = P.ln_b[c];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int pos = urc[u] + (t & 1) * ustep1 + (t >> 1) * ustep2;
                    if constexpr (AR == 1) {
                        f16x8 vh, vl;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            float v = t == 0 ? xr[u][q].x : (t == 1 ? xr[u][q].y : (t == 2 ? xr[u][q].z : xr[u][q].w));
                            if constexpr (LNMODE == 2) v -= umean[t];
                            if constexpr (LNMODE == 1) v = xsp[u] >= 0 ? (v - umean[t]) * urstd[t] * lg[q] + lb[q] : 0.f;
                            _Float16 hq, lq;
                            split2h(v, hq, lq);
                            vh[q] = hq; vl[q] = lq;
                        }
                        dst[(0 * 2 + ukg[u]) * plane + pos] = __builtin_bit_cast(uint4, vh);
                        dst[(1 * 2 + ukg[u]) * plane + pos] = __builtin_bit_cast(uint4, vl);
                        continue;
                    }
                    unsigned hh[8], mm[8], ll[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float v = t == 0 ? xr[u][q].x : (t == 1 ? xr[u][q].y : (t == 2 ? xr[u][q].z : xr[u][q].w));
                        if constexpr (LNMODE == 2) v -= umean[t];
                        if constexpr (LNMODE == 1) v = xsp[u] >= 0 ? (v - umean[t]) * urstd[t] * lg[q] + lb[q] : 0.f;
                        split3(v, hh[q], mm[q], ll[q]);
                    }
                    uint4 vh, vm, vl;
                    vh.x = (hh[0] >> 16) | hh[1]; vh.y = (hh[2] >> 16) | hh[3];
                    vh.z = (hh[4] >> 16) | hh[5]; vh.w = (hh[6] >> 16) | hh[7];
                    vm.x = (mm[0] >> 16) | mm[1]; vm.y = (mm[2] >> 16) | mm[3];
                    vm.z = (mm[4] >> 16) | mm[5]; vm.w = (mm[6] >> 16) | mm[7];
                    vl.x = (ll[0] >> 16) | (ll[1] & 0xFFFF0000u); vl.y = (ll[2] >> 16) | (ll[3] & 0xFFFF0000u);
                    vl.z = (ll[4] >> 16) | (ll[5] & 0xFFFF0000u); vl.w = (ll[6] >> 16) | (ll[7] & 0xFFFF0000u);
                    dst[(0 * 2 + ukg[u]) * plane + pos] = vh;
                    dst[(1 * 2 + ukg[u]) * plane + pos] = vm;
                    dst[(2 * 2 + ukg[u]) * plane + pos] = vl;
                }
            }
        }
    };
    const int n_w = TG * 6 * COPT;
    const int wsl = (n_w + nthr - 1) / nthr;
    const unsigned short *wsrc = P.wsp + (size_t)z * P.wsp_zs + (size_t)cog * COPT * 8 + (size_t)b * P.wsp_bs;
    // Per-thread byte offsets of its units inside one (group, chunk) weight stage: they do not depend on
    // the group or the chunk (those move the scalar base), so the in-loop DMA issue is address-free.
    // A stage has TG*6*COPT units = a whole number of waves, so the tail test is wave-uniform.
    constexpr int kWS = 7;
    constexpr bool kWaveRows = MB % 2 == 0;           // a wave's 64 units never straddle a (tap, plane) row
    unsigned wvo[kWaveRows ? 1 : kWS];                // per-lane part
    unsigned wso[kWS];                                // wave-uniform part (SGPRs)
    if constexpr (kWaveRows) {
        wvo[0] = (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < kWS; ++i) {
            const int u = min(wave + i * WN, n_w / 64 - 1);
            const int row = u / (COPT / 64), seg = u - row * (COPT / 64);
            const int t = row / 6, pk = row - t * 6;
            wso[i] = (unsigned)(((t * nc16) * 6 + pk) * P.COP + seg * 64) * 16u;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kWS; ++i) {
            const int e = min(tid + i * nthr, n_w - 1);
            const int t = e / (6 * COPT);
            const int rem = e - t * 6 * COPT;
            const int pk = rem / COPT, co = rem - pk * COPT;
            wvo[i] = (unsigned)(((t * nc16) * 6 + pk) * P.COP + co) * 16u;
            wso[i] = 0;
        }
    }
    const bool w_fast = wsl <= kWS;
    auto issue_w = [&](int grp, int chunk, int stage) {
        const int g1 = grp;
        const unsigned short *base = wsrc + ((size_t)(g1 * TG) * nc16 + chunk) * 6 * P.COP * 8;
        const float *wbase = uniform_ptr(reinterpret_cast<const float *>(base));
        const unsigned dst = wl_lds + (unsigned)(stage * wst_floats) * 4u + (unsigned)(wave * 64) * 16u;
        if (w_fast) {
#pragma unroll
            for (int i = 0; i < kWS; ++i)
                if (i * nthr + wave * 64 < n_w)
                    dma_b128_s(wvo[kWaveRows ? 0 : i], reinterpret_cast<const float *>(
                                   reinterpret_cast<const char *>(wbase) + wso[i]), dst + (unsigned)(i * nthr) * 16u);
            return;
        }
        for (int i = 0; i < wsl; ++i) {
            const int e = tid + i * nthr;
            if (e < n_w) {
                const int t = e / (6 * COPT);
                const int rem = e - t * 6 * COPT;
                const int pk = rem / COPT, co = rem - pk * COPT;
                const unsigned voff = (unsigned)(((t * nc16) * 6 + pk) * P.COP + co) * 16u;
                dma_b128_s(voff, wbase, dst + (unsigned)(i * nthr) * 16u);
            }
        }
    };

    f32x16 acc[1][MB][NPW];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NPW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][m][n][r] = 0.f;

    const int half = lane >> 5;
    const int j = lane & 31;
    const int pr = j >> P.lognbw, pc = j & (NBW - 1);
    const int xs = P.xshift[z];
    const int b_lane = half * plane + (rb * NPW * NBH + pr) * P.stride * PW + (s2 ? pc : pc + xs);
    const int nb_stride = NBH * P.stride * PW;
    const int a_lane = half * COPT + j;
    int bidx[NP][NPW];                                // B-operand unit index of tap (0,0), per plane / block
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            bidx[p][n] = (p * 2) * plane + b_lane + n * nb_stride;
            asm volatile("" : "+v"(bidx[p][n]));      // keep it a plain per-lane index (+ scalar tap offset)
        }

    const int c_lo = P.ksplit > 1 ? ks * nc16 / P.ksplit : 0;
    const int c_hi = P.ksplit > 1 ? (ks + 1) * nc16 / P.ksplit : nc16;
    load_x(c_lo);
    issue_w(0, c_lo, 0);
    int wstage = 0;
    TLC(0);
    if constexpr (PIPE != 0) {
        typedef f16x8 OpA[3][MB];
        typedef f16x8 OpB[2][NPW];
        OpA A0, A1;
        OpB B0, B1;
        const int T = P.KH * P.KW;
        const uint4 *xcu = reinterpret_cast<const uint4 *>(xc);
        const uint4 *wlu = reinterpret_cast<const uint4 *>(wl) + a_lane;
        const int wst_u = wst_floats / 4;
        // fetch cursor = the next tap to read: (fky, fkx), its index ftt inside its weight stage, the stage buffer fst
        int fky = 0, fkx = 0, ftt = 0, fst = 0;
        auto fetchB = [&](OpB &Bv) {
            const uint4 *xb = xcu + (fky * PW + (s2 ? ((fkx + xs) & 1) * (PW / 2) + ((fkx + xs) >> 1) : fkx));
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int n = 0; n < NPW; ++n) Bv[p][n] = __builtin_bit_cast(f16x8, xb[bidx[p][n]]);
        };
        auto fetchA = [&](OpA &A) {
            const uint4 *wa = wlu + fst * wst_u + ftt * 6 * COPT;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int m = 0; m < MB; ++m) A[pl][m] = __builtin_bit_cast(f16x8, wa[(pl * 2) * COPT + m * 32]);
        };
        auto advance = [&]() {
            if (++fkx == P.KW) { fkx = 0; ++fky; }
            if (++ftt == TG) { ftt = 0; fst ^= 1; }
        };
        // planes: B = {h, l'}, A = {WH, WL, WH2}; terms smallest first: WL.h, WH2.l', WH.h
        auto mma = [&](const OpA &A, const OpB &Bv) {
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                constexpr int PA[3] = {1, 2, 0};
                constexpr int PB[3] = {0, 1, 0};
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NPW; ++n)
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PA[term]][m], Bv[PB[term]][n], acc[0][m][n], 0, 0, 0);
            }
        };
        for (int chunk = c_lo; chunk < c_hi; ++chunk) {
            store_x(chunk);
            TLC(1);
            dma_wait();                         // first weight stage landed (nothing else is in flight)
            __syncthreads();                    // planes of `chunk` + first weight stage visible
            TLC(2);
            if (chunk + 1 < c_hi) load_x(chunk + 1);
            int grp = 0, ctt = 0;               // weight stage (tap group) of the current tap, the tap's index inside it
            if (1 < ntg) issue_w(1, chunk, wstage ^ 1);
            else if (chunk + 1 < c_hi) issue_w(0, chunk + 1, wstage ^ 1);
            fky = 0; fkx = 0; ftt = 0; fst = wstage;
            fetchA(A0);
            fetchB(B0);
            advance();
            __builtin_amdgcn_s_setprio(3);
            // The fetches of a step are unconditional: a fetch under a condition makes hipcc's s_waitcnt insertion merge the two
            // paths conservatively (the MFMAs then wait for the operands just requested), and MFMAs on two branches double the
            // accumulator registers.  At the last tap of a weight stage the next tap's weights are not visible yet: that read
            // returns stale data and is repeated right after the stage's barrier.
            auto group_end_sync = [&]() {
                __builtin_amdgcn_s_setprio(0);
                TLC(3);
                dma_wait();                     // the next stage (this wave's pieces) landed ...
                TLC(4);
                __syncthreads();                // ... everyone's did, and everyone is done reading this one
                TLC(5);
                wstage ^= 1;
                ++grp;
                ctt = 0;
            };
            // one tap that has a successor: operands (Ax, Bx) are in registers, the cursor points at the next tap -> (Ay, By)
            auto step = [&](const OpA &Ax, const OpB &Bx, OpA &Ay, OpB &By) {
                fetchB(By);
                fetchA(Ay);
                __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks the reads below the first MFMAs)
                mma(Ax, Bx);
                __builtin_amdgcn_sched_barrier(0);
                if (++ctt == TG) {
                    group_end_sync();
                    if (grp + 1 < ntg) issue_w(grp + 1, chunk, wstage ^ 1);
                    else if (chunk + 1 < c_hi) issue_w(0, chunk + 1, wstage ^ 1);
                    fetchA(Ay);
                    __builtin_amdgcn_s_setprio(3);
                }
                advance();
            };
            auto last = [&](const OpA &Ax, const OpB &Bx) {
                mma(Ax, Bx);
                group_end_sync();
            };
            int cur = 0;
            for (; cur + 2 < T; cur += 2) {
                step(A0, B0, A1, B1);
                step(A1, B1, A0, B0);
            }
            if (T - cur == 2) {
                step(A0, B0, A1, B1);
                last(A1, B1);
            } else {
                last(A0, B0);
            }
        }
    } else
    for (int chunk = c_lo; chunk < c_hi; ++chunk) {
        // (everyone finished reading the previous chunk's planes: the barrier that ends its last tap group)
#ifndef CDC_AB_NOSTOREX
        store_x(chunk);
#else
        if (chunk == c_lo) store_x(chunk);
#endif
        TLC(1);
        dma_wait();                         // first weight stage landed (nothing else is in flight)
        __syncthreads();                    // planes of `chunk` + first weight row visible
        TLC(2);
#ifndef CDC_AB_NOLOADX
        if (chunk + 1 < c_hi) load_x(chunk + 1);     // in flight during the tap loop; the group-end
                                                     // waits cover it (issued >= one group earlier)
#endif
        constexpr int zz = 0;
        for (int g1 = 0; g1 < ntg1; ++g1) {
            const int grp = g1;
#ifndef CDC_AB_NOW
            if (grp + 1 < ntg) issue_w(grp + 1, chunk, wstage ^ 1);
            else if (chunk + 1 < c_hi) issue_w(0, chunk + 1, wstage ^ 1);
#endif
            const uint4 *wa = reinterpret_cast<const uint4 *>(wl + wstage * wst_floats) + a_lane;
            int ky = (g1 * TG) / P.KW, kx = g1 * TG - ky * P.KW;        // uniform tap walk (SALU)
            const int zoff = 0;
            __builtin_amdgcn_s_setprio(3);      // waves inside the tap loop win issue arbitration over waves that are
                                                // converting / in an epilogue (+1 % measured, whole model)
            for (int t = 0; t < TG; ++t) {
                const uint4 *xb = reinterpret_cast<const uint4 *>(xc) + zoff +
                                  (ky * PW + (s2 ? ((kx + xs) & 1) * (PW / 2) + ((kx + xs) >> 1) : kx));
                if (++kx == P.KW) { kx = 0; ++ky; }
                // B planes stay live for the tap; the A planes are fetched one at a time, smallest
                // first: plane 2 feeds one product term, plane 1 two, plane 0 three (six in all)
                if constexpr (AR == 1) {
                    // planes: B = {h, l'}, A = {WH, WL, WH2}; terms smallest first: WL.h, WH2.l', WH.h
                    f16x8 Bh[2][NPW];
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int n = 0; n < NPW; ++n) Bh[p][n] = __builtin_bit_cast(f16x8, xb[bidx[p][n]]);
#pragma unroll
                    for (int term = 0; term < 3; ++term) {
                        constexpr int PA[3] = {1, 2, 0};
                        constexpr int PB[3] = {0, 1, 0};
                        f16x8 Ah[MB];
#pragma unroll
                        for (int m = 0; m < MB; ++m)
                            Ah[m] = __builtin_bit_cast(f16x8, wa[(t * 6 + PA[term] * 2) * COPT + m * 32]);
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int n = 0; n < NPW; ++n)
                                acc[zz][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m], Bh[PB[term]][n], acc[zz][m][n],
                                                                                       0, 0, 0);
                    }
                } else {
                bf16x8 Bv[3][NPW];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int n = 0; n < NPW; ++n) Bv[p][n] = __builtin_bit_cast(bf16x8, xb[bidx[p][n]]);
#pragma unroll
                for (int pa = 2; pa >= 0; --pa) {
                    bf16x8 A[MB];
#pragma unroll
                    for (int m = 0; m < MB; ++m)
                        A[m] = __builtin_bit_cast(bf16x8, wa[(t * 6 + pa * 2) * COPT + m * 32]);
#pragma unroll
                    for (int pb = 2 - pa; pb >= 0; --pb)
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int n = 0; n < NPW; ++n)
#ifndef CDC_AB_NOMFMA
                                acc[zz][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], Bv[pb][n], acc[zz][m][n],
                                                                                        0, 0, 0);
#else
                                acc[zz][m][n][0] += __builtin_bit_cast(float4, A[m]).x * __builtin_bit_cast(float4, Bv[pb][n]).y;
#endif
                }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            TLC(3);
            dma_wait();
            TLC(4);
            __syncthreads();
            TLC(5);
            wstage ^= 1;
        }
    }

    float prstd[LNMODE == 2 ? NPW : 1];
    if constexpr (LNMODE == 2) {
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int oy = oy0 + (rb * NPW + n) * NBH + pr, ox = ox0 + pc;
            prstd[n] = (oy < P.Ho && ox < P.Wo) ? P.ln_rstd[(size_t)b * HW + oy * P.W + ox] : 0.f;
        }
    }
    TileGeom geom{tid, nthr, wave, half, pr, pc, ipw > 1 ? (int)blockIdx.x * ipw : b, z, cog, oy0, ox0, NBH};
    geom.ipw = ipw; geom.wpi = wpi; geom.nimg = P.B;
    geom.out_off = (long long)ks * P.out_ks; geom.no_bias = ks > 0;
    {
#ifdef CDC_AB_NOEPI
    { float sacc = 0.f;
      for (int m = 0; m < MB; ++m) for (int n = 0; n < NPW; ++n) for (int r = 0; r < 16; ++r) sacc += acc[0][m][n][r];
      if (sacc == 12345.678f) P.out[tid] = sacc; }
#else
    conv_epilogue<MB, NPW, LNMODE, 0>(P, geom, acc[0], smem, prstd);
#endif
    }
    TLC(6);
#ifdef CDC_TIMELINE
    if (P.tl && threadIdx.x == 0) {
        unsigned long long *o = P.tl + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16;
        for (int c = 0; c < 7; ++c) o[c] = tl_acc[c];
        o[7] = tl_start; o[8] = tl_last;
    }
#endif
#undef TLC
}

}  // namespace cdc
