// conv_args.h -- kernel argument block + launch plan of the implicit-GEMM convolution kernel.
//
// One kernel family implements every dense contraction of the U-Net
// (reference call sites: Block conv k x k  network_components.py:87, res_conv 1x1 :105,
//  to_qkv / to_out 1x1 :125-126, Downsample 3x3 s2 :50, Upsample ConvTranspose 4x4 s2 :39 as four
//  2x2 phase convolutions, final 7x7 unet.py:104, and the per-image ctx^T.q product :137).
//
// GEMM orientation: D[cout][pixel] = sum_k Wp[k][cout] * X[k][pixel], k = (tap, cin), computed
// with v_mfma_f32_32x32x2_f32 (A = weights, B = pixels).  Each lane then owns ONE pixel and 16
// output channels per 32x32 block, so the channel-LayerNorm reduction is in-lane adds plus one
// cross-half exchange, and NCHW rows are written as 128-byte segments.
#pragma once
#include <stdint.h>

namespace cdc {

constexpr int kWaveSize = 64;
constexpr int kXS = 20;   // max LDS-DMA slots per thread for the input patch (4 B each)

struct ConvArgs {
    // input: channel-concatenation of up to two NCHW sources (torch.cat sites unet.py:109,124)
    const float *src0, *src1;
    long long src0_bs, src1_bs;     // batch strides in floats (0 = broadcast over batch)
    // conv_split2_kernel, round 4 -- unfold on load (first 7x7 layer as a 7x1 convolution over kx-unfolded channels): uf_c > 0: src0 is the
    // [uf_c][H][W] image itself and logical input channel cc = kx * uf_c + c reads src0[c][y][x + kx - uf_pad] (zero outside the row);
    // the unfolded [KW * uf_c][H][W] tensor (unfold_x_kernel) is never written
    int uf_c, uf_pad;
    unsigned uf_magic;              // fast division by uf_c
    int C0, Cin;                    // channels taken from src0; total input channels
    int H, W;                       // input spatial size
    // optional LayerNorm applied while staging the input (PreNorm, network_components.py:69-77)
    const float *ln_mean, *ln_rstd; // [B][H*W] statistics of the (single) source
    const float *ln_g, *ln_b;       // [Cin]
    // packed weights [z][taps][Cin_pad][COP] (zero padded); w_bs = per-image stride (0 = shared)
    const float *wp;
    long long w_bs, w_zs;
    // split-bf16 form (conv_split_kernel.h): [z][tap][Cin_pad/16][plane 3][k-half 2][COP][8] bf16
    const unsigned short *wsp;
    long long wsp_zs;
    long long wsp_bs;               // per-image split weights (folded attention output): stride per image, 0 = shared
    // arith 1 (conv_split2_kernel AR = 1): wsp holds the fp16 planes {WH, WL, WH2} of w * 2^s in the same
    // layout and the epilogue multiplies the accumulators by acc_scale = 2^-s (1 otherwise)
    float acc_scale;
    int tg;                         // taps per LDS weight stage (conv_split2_kernel)
    int ipw, B;                     // images per workgroup (small feature maps), batch size
    int KH, KW, stride;
    int pad_y[4], pad_x[4];         // per blockIdx.z (ConvTranspose phases); z = 0 otherwise
    int KC, logKC, nchunk, Cin_pad, COP, Cout;
    // output tensor addressing (floats): b*out_bs + co*out_cs + oy*out_ys + ox*out_xs + out_zoff[z]
    float *out;
    long long out_bs, out_cs;
    int out_ys, out_xs, out_zoff[4];
    int Ho, Wo;                     // logical output extent of one phase
    // tiling
    int lognbw;                     // a 32-pixel N-block is (32>>lognbw) rows x (1<<lognbw) cols
    int tiles_x, tiles_y;
    int PH, PW;                     // staged input patch: rows, row stride (floats)
    int xvec, xshift[4];            // 16-byte input pieces: patch starts xshift[z] columns early
    unsigned magic_hw, magic_w;     // fast division by PH*PW and by PW (0 => divisor is 1)
    // epilogue
    const float *bias;              // [Cout] or null
    const float *pre_add;           // hoisted partial sums, added before LN (addressing = out)
    const float *ep_g, *ep_b;       // channel LayerNorm after bias (needs gridDim.y == 1)
    float eps;
    int relu;
    float relu_slope;               // 0: ReLU; 0.2: LeakyReLU(0.2) of the hyper decoder (max(v, slope*v))
    const float *shift;             // [B][shift_bs] added after ReLU (time-embedding add)
    int shift_bs;
    const float *resid;             // same addressing as out, added last (ResnetBlock / Residual)
    long long resid_bs, resid_cs;
    // residual branch of the first ResnetBlock: res_conv over the 3 image channels, evaluated in the
    // epilogue (3 FMAs per value) instead of a separate 1x1 launch: += sum_c res3_w[c][co] * res3_x[b][c][pix]
    const float *res3_w, *res3_x;   // packed 1x1 weights [>=3][COP] / stride-1 input of the same H x W
    long long res3_bs;
    float *stat_mean, *stat_rstd;   // [B][Ho*Wo]: LN statistics of the final values (for the next
                                    // PreNorm); needs gridDim.y == 1
    // split-K (conv_split2_kernel, plain bias-only epilogue): gridDim.z = nzz * ksplit; slice ks handles chunks
    // [ks*nchunk/ksplit, (ks+1)*nchunk/ksplit) and stores its partial sums at out + ks*out_ks (bias in
    // slice 0); the consumer (ln_kernel_sliced) adds the slices.
    int ksplit, nzz;
    int xcd_remap;  // gridDim.x % 8 == 0: tile = (id % 8) * (gridDim.x / 8) + id / 8
    int zfold;      // transposed conv: the 4 phases of a tile are consecutive-by-8 workgroup ids in gridDim.x, so
                    // they run on ONE XCD at about the same time and the L2 merges their interleaved stores
    long long out_ks;
    // optional second copy of the result as a PF tensor (two fp16 planes in B-operand order, conv_pf_kernel.h):
    // unit index b*pf_bs + ((co/8)*2 + plane)*pf_ps + oy*pf_ys + ox*pf_xs + pf_zoff[z]; needs Cout % 32 == 0
    void *out_pf;
    long long pf_bs, pf_ps;
    int pf_ys, pf_xs, pf_zoff[4];
    int pf_only;    // the planes are the ONLY copy of the result (its single consumer reads planes): skip the fp32 store
    // range guard of the fp16 arithmetic (cdc_api.hip): set to 1 when an accumulator of this launch is inf / NaN -- checked
    // BEFORE the LayerNorm / ReLU of the epilogue can turn it into a finite value (may be null)
    int *fault;
#ifdef CDC_TIMELINE
    unsigned long long *tl;         // tools/build_variant.sh timeline -DCDC_TIMELINE: 64 cycle stamps per workgroup
#endif
};

// Arguments of conv_pf_kernel (conv_pf_kernel.h): activations arrive as PF tensors (two fp16 planes, 16-byte units
// of 8 channels, one-pixel zero halo).
struct PfArgs {
    const void *src0, *src1;        // PF sources (channel concatenation); src1 may be null
    long long src0_bs, src1_bs;     // batch strides in units
    int C0, Cin;                    // channels taken from src0; total (both multiples of 16)
    int H, W;                       // input extent without the halo
    const void *w;                  // fp16 planes {WH, WL, WH2}: [z][tap][Cin/16][3][2][COP] units
    long long w_zs;
    int KH, KW, nz;
    int stride;                     // 0 / 1, or 2: conv_pf_kernel<..., STR = 2> (3x3 / pad 1; H, W stay the INPUT extent)
    int tz;                         // 4: conv_pf_kernel<..., TZ = 4>: the four phases z of a transposed convolution inside one workgroup
    int pad_y[4], pad_x[4];
    int nchunk, COP, Cout;
    float acc_scale;
    int ring;                       // weight ring slots (3..6)
    int n_iter;                     // conv_pf3_kernel: tile iterations per group = ceil(max tiles per workgroup / 2)
    // fp32 NCHW output (may be null): b*out_bs + co*out_cs + oy*out_ys + ox*out_xs + out_zoff[z]
    float *out;
    long long out_bs, out_cs;
    int out_ys, out_xs, out_zoff[4];
    // PF output (may be null), unit index: b*pf_bs + ((co/8)*2 + plane)*pf_ps + oy*pf_ys + ox*pf_xs + pf_zoff[z]
    // (pf_zoff includes the +1,+1 halo origin)
    void *out_pf;
    long long pf_bs, pf_ps;
    int pf_ys, pf_xs, pf_zoff[4];
    int Ho, Wo;
    int lognbw, tiles_x, tiles_y, B;
    int xcd_remap;
    // epilogue (same meaning as ConvArgs)
    const float *bias, *pre_add, *ep_g, *ep_b;
    int pre_c4;                     // conv_pf_kernel: pre_add is stored in accumulator order, [B][Cout / 4][Ho * Wo][4] (16-byte loads; c4_pack_kernel)
    float eps;
    int relu;
    float relu_slope;
    const float *shift;
    int shift_bs;
    const float *resid;
    long long resid_bs, resid_cs;
    // a residual over a channel concatenation (round 4: the identity residual of downs.1.0 over cat[x, context]): channels >= resid_c0
    // come from resid1 (same channel stride as resid); both boundaries fall on a wave's channel part (host-checked)
    const float *resid1;
    long long resid1_bs;
    int resid_c0;
    // the residual as a PF tensor (round 4: a ResnetBlock-chain output that exists as planes only; replaces `resid`):
    // unit index b*rpf_bs + ((co/8)*2 + plane)*rpf_ps + oy*rpf_ys + ox + rpf_zoff (the zoff includes the +1,+1 halo origin)
    const void *resid_pf;
    long long rpf_bs, rpf_ps;
    int rpf_ys, rpf_zoff;
    float *stat_mean, *stat_rstd;
    const float *res3_w, *res3_x;   // 3-channel res_conv in the epilogue (see ConvArgs)
    long long res3_bs;
    // conv_pw_kernel (conv_pw_kernel.h): the activation operand is read from fp32 NCHW sources by the lanes themselves
    const float *x0, *x1;           // fp32 sources (channel concatenation), x1 may be null
    long long x0_bs, x1_bs;         // batch strides in floats
    const float *pre_mean, *pre_rstd;   // [B][H*W] PreNorm statistics: (x - mean) on load, rstd in the epilogue (or null)
    long long w_bs;                 // per-image weight planes (folded attention output): stride in units, 0 = shared
    int lin;                        // conv_pw_kernel on maps narrower than 32 pixels: a 32-pixel block = 32 consecutive pixels
                                    // of one image's flattened H*W (H*W % 32 == 0), a workgroup may span images
    int dbg;                        // CDC_PF_DBG (timing experiments, wrong results): 1 no weight DMA in the loop, 2 no patch
                                    // DMA in the loop, 4 no barrier in the loop, 8 no DMA waits, 16 no epilogue stores
    int *fault;                     // range guard: set to 1 when an accumulator is inf / NaN (see ConvArgs::fault; may be null)
#ifdef CDC_TIMELINE
    unsigned long long *tl;         // tools/build_variant.sh timeline -DCDC_TIMELINE: 16 values per workgroup (conv_pf_kernel)
#endif
};

constexpr int kPfXS = 12;   // patch DMA instructions per chunk and patch wave (two patch waves: <= 24 per chunk)

// Host-side launch plan for one convolution.
struct ConvPlan {
    int MB, NPW, WN;        // cout blocks per wave, pixel blocks per wave, waves per workgroup
    int groups;             // cout groups (gridDim.y)
    int KC, nchunk;
    int lognbw, tiles_x, tiles_y, PH, PW;
    int xvec, xshift[4];
    size_t lds_bytes;
    int lnmode;
    int tg;                 // taps per weight stage of the split kernels
    int ipw;                // images per workgroup (1: tiles inside one image)
    int ksplit = 1;         // K slices (split2 only)
    int xu = 1;             // patch units per thread (split2 only)
    int arith = 0;          // split2 only: 0 three bf16 planes (6 MFMA products), 1 two fp16 planes (3 products)
    int split;              // 1: conv_split_kernel (three-plane bf16 operands on the bf16 MFMA)
};

}  // namespace cdc
