/*
 * cdc_hip.h -- C-ABI of libcdc_hip.so: the MI355X (gfx950) decode hot path of CDC
 * (conditional-diffusion image compression), i.e. the N-step DDIM loop over the denoising U-Net.
 *
 * The reference (buggyyang/CDC_compression) has no FFI: the path sits behind Python methods.
 * Each entry point below names the reference interface it replaces (paths relative to the
 * reference root).  A drop-in binding is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; tensors are dense float32, NCHW, exactly as the reference passes them;
 *   - every function returns CDC_OK (0) or a negative cdc_status; it never throws;
 *     cdc_last_error(h) returns a human-readable message for the last failure on that handle;
 *   - a handle is bound to one HIP device and is NOT thread-safe (one handle per GPU per thread);
 *   - pointers tagged `mem` are host pointers (CDC_MEM_HOST: the library stages them through
 *     its own device buffers) or device pointers on the handle's device (CDC_MEM_DEVICE: no host
 *     round trip; outputs are written in place, the inputs x / init / context are copied once per
 *     call, device to device on `stream`, into the buffers the launch program was built on --
 *     2.1 GB for a batch-32 context pyramid at 256x256, < 1 ms against a 500-iteration decode).
 *     With device pointers the work is enqueued asynchronously on
 *     `stream` (a hipStream_t passed as void*; NULL = the HIP null stream, i.e. torch's default
 *     stream); with host pointers `stream` is ignored, the library uses its own stream and the
 *     call is synchronous (result valid on return).
 */
#ifndef CDC_HIP_H
#define CDC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cdc_handle cdc_handle;

typedef enum {
    CDC_OK = 0,
    CDC_ERR_INVALID = -1,      /* bad argument / shape / name */
    CDC_ERR_STATE = -2,        /* call order (weights not finalized, schedule not set, ...) */
    CDC_ERR_HIP = -3,          /* HIP runtime error (message has the hipError string) */
    CDC_ERR_UNSUPPORTED = -4,  /* configuration outside what the kernels implement */
    CDC_ERR_NOMEM = -5
} cdc_status;

enum { CDC_MEM_HOST = 0, CDC_MEM_DEVICE = 1 };
/* pred_mode of the sampler entry points: which tree's ddim() rules apply and what the U-Net predicts.
 *   CDC_PRED_X            x-tree, pred_mode "x"     (xparam/modules/denoising_diffusion.py:157-158)
 *   CDC_PRED_NOISE        eps-tree, pred_mode "noise" (epsilonparam/...:137-152: no clamp under the square root)
 *   CDC_PRED_NOISE_XTREE  x-tree, pred_mode "noise" (xparam :155-156,165: x0 = predict_start_from_noise, .clamp(min=0))
 *   CDC_PRED_V            x-tree, pred_mode "v"     (xparam :161-162: x0 = predict_start_from_v, :128-139; needs cdc_set_schedule_v)
 * clip: CDC_CLIP_NONE, CDC_CLIP_ALL (x-tree clip_denoised=True; eps-tree clip_noise "full"),
 *       CDC_CLIP_HALF (eps-tree clip_noise "half": only the first B/2 images, eps :142-143). */
enum { CDC_PRED_X = 0, CDC_PRED_NOISE = 1, CDC_PRED_NOISE_XTREE = 2, CDC_PRED_V = 3 };
enum { CDC_CLIP_NONE = 0, CDC_CLIP_ALL = 1, CDC_CLIP_HALF = 2 };
enum { CDC_MAX_LEVELS = 8 };

/* Unet.__init__ arguments: xparam/modules/unet.py:19-29 (epsilonparam/modules/unet.py:18-27).
 * with_time_emb is always true on the tested path; embd_type "01" only (unet.py:39-41). */
typedef struct {
    int32_t dim;
    int32_t channels;
    int32_t context_channels;
    int32_t out_dim;                                /* 0 -> channels (unet.py:103) */
    int32_t n_dim_mults;
    int32_t dim_mults[CDC_MAX_LEVELS];
    int32_t n_context_dim_mults;
    int32_t context_dim_mults[CDC_MAX_LEVELS];
} cdc_unet_config;

/* ---- lifetime ------------------------------------------------------------------------------ */

/* Builds the layer graph of Unet.__init__ (unet.py:19-104) for `device`. */
int cdc_create(const cdc_unet_config *cfg, int device, cdc_handle **out);
void cdc_destroy(cdc_handle *h);
const char *cdc_last_error(const cdc_handle *h);   /* h may be NULL: last cdc_create error */
const char *cdc_version(void);

/* ---- arithmetic of the dense contractions ------------------------------------------------------
 * Tensors are float32 everywhere; the k x k (and wide 1x1) convolutions form their fp32 products on the
 * 16-bit matrix cores from split operands (no reference counterpart -- torch delegates to oneDNN / cuDNN fp32):
 *   CDC_ARITH_F16X2 (default)  a = h + l*2^-11 as two fp16 numbers, w*2^s as {WH, WL}: three
 *                              v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulation; operands carry 22-23 significant
 *                              bits for 6e-5 <= |a| < 65504.  RANGE GUARD: |a| >= 65504 makes the accumulators of its
 *                              convolution inf / NaN.  Every convolution / LayerNorm launch reports that itself, BEFORE a
 *                              fused LayerNorm + ReLU can turn it into a finite wrong value (round 4), and EVERY entry point
 *                              that runs the arithmetic (cdc_unet_forward, cdc_ddim_step, cdc_decode, cdc_ctxdec_decode,
 *                              cdc_hyperdec_decode, cdc_encoder_encode, cdc_entropy_encode, the cdc_op_* operators) also
 *                              checks its results (one small kernel + a 4-byte read-back: the call synchronises its
 *                              stream) and repeats a faulting call ONCE in CDC_ARITH_BF16X3.  The handle then STAYS in that mode (a warning is printed once;
 *                              cdc_get_arith / cdc_get_range_faults tell).  Results that are non-finite in the full-range
 *                              arithmetic too (non-finite inputs, parameters beyond fp32) come back as they are, as the
 *                              reference's would (counted by cdc_get_nonfinite_results); the range was not their cause,
 *                              so such a call leaves the handle in CDC_ARITH_F16X2 and is not counted as a range fault.
 *                              CDC_NO_RANGE_GUARD=1 in the environment switches the check off.
 *   CDC_ARITH_BF16X3           a = a1 + a2 + a3 exactly as three bf16 numbers: six v_mfma_f32_32x32x16_bf16,
 *                              full fp32 range.
 * Changing the mode drops the handle's launch program (rebuilt on the next call).  New handles take
 * CDC_ARITH_F16X2 unless the environment says CDC_ARITH=0. */
enum { CDC_ARITH_BF16X3 = 0, CDC_ARITH_F16X2 = 1 };
int cdc_set_arith(cdc_handle *h, int mode);
int cdc_get_arith(const cdc_handle *h);
int cdc_get_range_faults(const cdc_handle *h);        /* calls repeated in CDC_ARITH_BF16X3 by the range guard */
int cdc_get_nonfinite_results(const cdc_handle *h);   /* calls whose results are non-finite in the full-range arithmetic as well */

/* ---- parameters: replaces nn.Module.load_state_dict on the Unet (test_xparam.py:62-68) ------- */

/* Manifest of the reference Unet.state_dict(): count, then (name, shape) per index. */
int cdc_num_tensors(const cdc_handle *h);
int cdc_tensor_info(const cdc_handle *h, int index, const char **name, int64_t shape[4],
                    int *ndim);
/* `name` = reference state_dict key (e.g. "downs.0.0.block1.block.0.weight"), data in the
 * reference's layout (Conv2d OIHW, ConvTranspose2d IOHW, Linear [out][in], LayerNorm [1,C,1,1]);
 * the library repacks into its MFMA operand layout.  Host pointer only. */
int cdc_load_tensor(cdc_handle *h, const char *name, const float *data, const int64_t *shape,
                    int ndim);
/* Fails (CDC_ERR_STATE) listing the first missing tensor if any manifest entry was not loaded. */
int cdc_finalize_weights(cdc_handle *h);

/* ---- Unet.forward(x, time, context): xparam/modules/unet.py:131-135 -------------------------- */

/* x [B,channels,H,W]; time [B] (the reference passes [B,1]); ctx[l] [B,C_l,H>>l,W>>l] for
 * l < n_ctx (C_l = context_dims[l], unet.py:34,109); out [B,out_dim,H,W]. */
int cdc_unet_forward(cdc_handle *h, const float *x, const float *time, const float *const *ctx,
                     int n_ctx, float *out, int B, int H, int W, int mem, void *stream);

/* Intermediate activation of the LAST cdc_unet_forward / DDIM iteration, by the reference's module path: the output
 * of "downs.<i>.<0|1|2|3>" (ResnetBlock, ResnetBlock, attention, Downsample: unet.py:110-116), "mid_block1",
 * "mid_attn", "mid_block2", "ups.<i>" (output of the stage's Upsample, unet.py:125-128).  What a
 * register_forward_hook on that module records in the reference; used by the per-stage parity tests.
 * out may be NULL (shape query); host pointer, synchronous. */
int cdc_unet_tap(cdc_handle *h, const char *name, float *out, int64_t shape[4]);

/* ---- sampler: GaussianDiffusion.set_sample_schedule / ddim / p_sample_loop ------------------- */

/* Per-sample-step scalars, index i = 0..steps-1 in the reference's `t` indexing
 * (xparam/modules/denoising_diffusion.py:89-108 ; epsilonparam/...:81-97).  The host computes
 * them (float64 beta schedule -> float32 tables) and hands them over:
 *   time_in[i]         value fed to the U-Net:  x-param index[i]/num_timesteps (:154),
 *                                               eps-param i/sample_steps (eps :138)
 *   sqrt_recip[i], sqrt_recipm1[i]   sqrt(1/ac), sqrt(1/ac-1)
 *   sqrt_ac_prev[i], one_minus_ac_prev[i], sigma[i]                                        */
int cdc_set_schedule(cdc_handle *h, int steps, const float *time_in, const float *sqrt_recip,
                     const float *sqrt_recipm1, const float *sqrt_ac_prev,
                     const float *one_minus_ac_prev, const float *sigma);
/* The two extra tables pred_mode "v" reads (xparam :99,:103 sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod; used by
 * predict_start_from_v, :128-139).  Call after cdc_set_schedule with the same number of steps. */
int cdc_set_schedule_v(cdc_handle *h, int steps, const float *sqrt_ac, const float *sqrt_one_minus_ac);

/* One DDIM update x_t -> x_{t-1} at sample index i (x: :152-174 ; eps: :137-152).
 * clip: x-param clamp of x0 to [-1,1] (clip_denoised=True in compress, :223);
 *       eps-param clip_noise "full" -> 1, anything else -> 0.
 * noise: the torch.randn_like draw of that step (used only when eta != 0; may be NULL). */
int cdc_ddim_step(cdc_handle *h, const float *x_in, int i, const float *const *ctx, int n_ctx,
                  const float *noise, float eta, float *x_out, int B, int H, int W,
                  int pred_mode, int clip, int mem, void *stream);

/* p_sample_loop with eta = 0 (x: :179-205 ; eps: :166-192): for i = steps-1 .. 0: ddim.
 * init may be NULL (zeros, :183).  This is the timed hot path of bench.py. */
int cdc_decode(cdc_handle *h, const float *init, const float *const *ctx, int n_ctx, float *out,
               int B, int H, int W, int pred_mode, int clip, int mem, void *stream);

/* ---- measurement --------------------------------------------------------------------------- */

/* ---- context decoder: Compressor.decode ---------------------------- */

/* The synthesis transform of the context model: `for resnet, up in dec: x = up(resnet(x))`, outputs
 * returned finest first (xparam/modules/compress_modules.py:68-74, epsilonparam/...:74-82).  Level i =
 * ResnetBlock(rev[i] -> rev[i+1], no time embedding; the last level keeps rev[i]) + ConvTranspose2d(4,2,1)
 * to rev[i+1]; rev = [dim*m for m in rev_mults] + [out_channels] (compress_modules.py:21-22,147-156).
 * up_index = position of the Upsample in each ModuleList: 1 for xparam's ResnetCompressor
 * ("dec.0.1.conv.weight"), 2 for epsilonparam's BigCompressor (an nn.Identity sits at index 1). */
typedef struct {
    int32_t dim;
    int32_t n_rev_mults;
    int32_t rev_mults[CDC_MAX_LEVELS];    /* xparam: reverse_dim_mults (4,3,2,1); eps: reversed(dim_mults) */
    int32_t out_channels;                 /* = the U-Net's context_channels */
    int32_t up_index;
} cdc_ctxdec_config;

/* Same handle type and the same parameter entry points (cdc_num_tensors / cdc_tensor_info /
 * cdc_load_tensor with the reference's "dec.*" keys / cdc_finalize_weights / cdc_destroy). */
int cdc_ctxdec_create(const cdc_ctxdec_config *cfg, int device, cdc_handle **out);

/* q_latent [B][rev[0]][h][w] -> n_outs = n_rev_mults tensors, outs[0] = finest
 * ([B][out_channels][h*2^n][w*2^n]) ... outs[n-1] = [B][rev[1]][2h][2w]: exactly the `context` list
 * cdc_unet_forward / cdc_decode take.  mem_kind / stream as for cdc_unet_forward. */
int cdc_ctxdec_decode(cdc_handle *h, const float *q_latent, float *const *outs, int n_outs, int B,
                      int h_latent, int w_latent, int mem_kind, void *stream);

/* ---- encoder: analysis transform + hyper encoder --------------------- */

/* Compressor.encode up to the quantisers (compress_modules.py:43-51,131-165): `enc` = n_dim_mults x
 * [ResnetBlock(dims[i] -> dims[i+1], no time embedding, 7x7 first block on level 0), Downsample (Conv2d 3x3 s2)],
 * then `hyper_enc` = Conv2d(3x3) + n_hyper-1 x Conv2d(5x5, stride 2, padding 2), LeakyReLU(0.2) between.
 * dims = [channels] + [dim*m for m in dim_mults]; hyper dims = [dims[-1]] + [dim*m for m in hyper_mults].
 * down_index: position of the Downsample in each `enc` ModuleList (1 xparam ResnetCompressor, 2 epsilonparam
 * BigCompressor).  Parameters: the reference's "enc.*" / "hyper_enc.*" keys through cdc_load_tensor. */
typedef struct {
    int32_t dim, channels;
    int32_t n_dim_mults;
    int32_t dim_mults[CDC_MAX_LEVELS];
    int32_t n_hyper_mults;
    int32_t hyper_mults[CDC_MAX_LEVELS];
    int32_t down_index;
} cdc_encoder_config;

int cdc_encoder_create(const cdc_encoder_config *cfg, int device, cdc_handle **out);

/* images [B][channels][H][W] (H, W multiples of 2^(n_dim_mults + n_hyper_mults - 1)) ->
 * latent [B][dims[-1]][H/2^n][W/2^n], hyper_latent [B][hyper_dims[-1]][...]: the UNquantised tensors
 * (`latent`, `hyper_latent` of state4bpp); quantisation is cdc_dequantize with the prior medians / the mean. */
int cdc_encoder_encode(cdc_handle *h, const float *images, float *latent, float *hyper_latent, int B, int H,
                       int W, int mem_kind, void *stream);

/* ---- hyperprior decoder (decode side) ------------------------------------ */

/* Compressor.hyper_dec (xparam/modules/compress_modules.py:54-60,166-177; epsilonparam/...:58-66,171-185):
 * n_layers-1 x [ConvTranspose2d(dims[i] -> dims[i+1], 5, stride 2, padding 2, output_padding 1), LeakyReLU(0.2)]
 * then Conv2d(dims[n-1] -> dims[n], 3, padding 1).  dims = reversed_hyper_dims, e.g. {256,256,256,512}.
 * Parameters: the reference's "hyper_dec.<i>.0.weight" / ".bias" through cdc_load_tensor. */
typedef struct {
    int32_t n_layers;
    int32_t dims[CDC_MAX_LEVELS + 1];
} cdc_hyperdec_config;

int cdc_hyperdec_create(const cdc_hyperdec_config *cfg, int device, cdc_handle **out);

/* q_hyper_latent [B][dims[0]][h][w] -> mean, scale [B][dims[n]/2][4h][4w] each:
 * `mean, scale = hyper_dec(q_hyper_latent).chunk(2, 1)`, `scale.clamp(min=scale_min)` (compress_modules.py:58-59). */
int cdc_hyperdec_decode(cdc_handle *h, const float *q_hyper_latent, float *mean, float *scale, int B,
                        int h_hyper, int w_hyper, float scale_min, int mem_kind, void *stream);

/* Compressor.bpp, eval mode (compress_modules.py:76-90): bpp[b] = (sum -log2 FlexiblePrior.likelihood(q_hyper_latent)
 * + sum -log2 NormalDistribution(mean, scale).likelihood(q_latent)) / (H_img * W_img).  Needs the FlexiblePrior
 * tensors, loaded (optionally) through cdc_load_tensor under the reference's keys with the singleton axes squeezed:
 * "prior.affine.<i>.weight" [C][in][out], "prior.affine.<i>.bias" [C][out], "prior.a.<i>" [C][out]
 * (network_components.py:316-336; dims 1-3-3-3-1).  q_latent / mean / scale: [B][dims[n]/2][4h][4w]. */
int cdc_bpp(cdc_handle *h, const float *q_hyper_latent, const float *q_latent, const float *mean,
            const float *scale, float *bpp, int B, int h_hyper, int w_hyper, int H_img, int W_img, int mem_kind,
            void *stream);

/* ---- entropy coder -- no reference counterpart: the reference only estimates the rate ------
 * Codes exactly the symbols Compressor.bpp prices (compress_modules.py:76-90) with exactly its two models:
 *   q_hyper_latent - medians   under FlexiblePrior.likelihood (network_components.py:372-378), one table per channel;
 *   q_latent - mean            under NormalDistribution(mean, scale).likelihood (utils.py:147-159), tables by scale,
 * 64-lane interleaved byte-wise range-ANS, 16-bit probabilities, coded and decoded ON THE GPU -- one wave per section, the
 * 2 B sections of a batch in two launches (format and table specification: csrc/entropy.hip; restated in
 * oracle/entropy_oracle.c and, as a second structurally different decoder, in tests/test_entropy.py; streams agree byte for
 * byte).  Entry points of a hyper-decoder handle (cdc_hyperdec_create) that has the prior.* tensors.
 * latent / hyper_latent are the UNquantised encoder outputs (cdc_encoder_encode); medians [dims[0]] is a host array.
 * out receives B concatenated streams, image b at [offsets[b], offsets[b+1]) (offsets has B+1 entries).  Each stream:
 *   'C' 'D' 'C' 3 | arith u8 | 0 | h_hyper u16 | w_hyper u16 | n_hyper u32 | n_latent u32 | model u32 | symbols u32 |
 *   esc_hyper u32 | esc_latent u32 | hyper section | latent section                   (version 3, 34-byte header, little endian)
 * section = 64 x u32 final lane states | renormalisation bytes | escape payloads (u32 each, symbol order); n_* = section bytes.
 * model   = FNV-1a over every integer of the probability tables (per table K, then its 2K+2 frequencies; the per-channel
 *           hyper tables, then the 128 scale tables): a decoder whose tables differ (other prior.* parameters or medians,
 *           another build, another libm) refuses the stream instead of decoding garbage;
 * symbols = sum over both sections of mix(section, index, symbol) mod 2^32 (csrc/entropy.hip: sym_mix; order-independent,
 *           so the lanes accumulate it in parallel): the decoder checks it after decoding.  The coder's own end
 *           conditions (every lane back at 2^23, every byte and payload consumed) catch most corruption before that.
 * Encoder and decoder run hyper_dec for the whole batch through the launch plan of ONE image (no kernel mixes images,
 * so image b of a batch holds the bits of a batch-1 call; a batch encode returns the batch-1 streams byte for byte) in
 * the arithmetic the header records, so that the decoder reproduces the encoder's scale bins bit for bit -- the contract
 * every learned codec has.  One cdc_entropy_decode call takes streams of one image size; their arithmetics may differ.
 * LIMITS: same library build and same GPU architecture on both sides (the CDC_* development switches, honoured only under
 * CDC_DEV=1, change launch plans and void this); a decoder whose hyper_dec output lands in other scale bins decodes other
 * symbols and fails the end conditions or the `symbols` check loudly.
 * cdc_entropy_decode leaves the handle's own arithmetic as it found it.  hh, wh >= 1 and hh * wh <= 2^22 are enforced
 * before anything is sized by them; section sizes and escape counts are checked against the stream length; no C++ exception
 * crosses this boundary (CDC_ERR_NOMEM / CDC_ERR_INVALID instead).
 * cdc_entropy_encode refuses non-finite latents / means / scales (CDC_ERR_INVALID).
 * Synchronous; latent / hyper_latent / q_latent / q_hyper_latent follow `mem`, in / out / offsets / medians are host. */
int cdc_entropy_encode(cdc_handle *h, const float *latent, const float *hyper_latent, const float *medians, int B,
                       int h_hyper, int w_hyper, unsigned char *out, size_t cap, size_t *offsets, int mem_kind, void *stream);
int cdc_entropy_peek(const unsigned char *in, size_t n, int *h_hyper, int *w_hyper, int *arith);
/* A stream header sizes the decoder's allocations and its hyper_dec launch program (up to 2^22 positions = a 131072 x 131072
 * image: tens of GB on a 288 GB part).  A caller that knows what it expects bounds that BEFORE decoding untrusted bytes: streams whose
 * header asks for more than max_hyper_positions = h_hyper * w_hyper are refused (CDC_ERR_INVALID) before anything is allocated. */
int cdc_entropy_set_limit(cdc_handle *h, int max_hyper_positions);
/* -> q_latent [B][dims[n]/2][up*h][up*w] (exactly the encoder's dequantised latent) and, optionally, q_hyper_latent. */
int cdc_entropy_decode(cdc_handle *h, const unsigned char *in, const size_t *offsets, const float *medians, int B,
                       float *q_latent, float *q_hyper_latent, int mem_kind, void *stream);

/* quantize(x, "dequantize", offset) = round(x - offset) + offset, round = half-to-even (utils.py:72-85). */
int cdc_dequantize(cdc_handle *h, const float *x, const float *offset, float *out, long long n, int mem_kind,
                   void *stream);

/* Measurement aids of bench.py (csrc/probe.hip; no handle, no model): what THIS part sustains, measured on the spot, to print
 * beside the nominal peaks a roofline is quoted against.
 * cdc_probe_mfma_f16: a register-only loop of v_mfma_f32_32x32x16_f16 (two waves per SIMD, four independent accumulators, `iters`
 *   x 16 instructions per wave) -> TFLOP/s.  random_operands = 1 cycles eight pseudo-random operand pairs (the multiplier inputs
 *   toggle as on real data; the power management gives back clock), 0 multiplies the same registers every time.
 * cdc_probe_hbm_copy: a float4 copy of `bytes` (>= 1 MiB) from one device buffer to another, best of `reps` -> GB/s counting
 *   bytes read + bytes written.  Both run on the null stream of `device` and synchronise. */
int cdc_probe_mfma_f16(int device, int random_operands, int iters, double *tflops);
int cdc_probe_hbm_copy(int device, size_t bytes, int reps, double *gbytes_per_s);

/* Kernel-class timing: hipEvent pairs recorded (without host synchronisation) on the launch stream
 * around every kernel of a forward / DDIM iteration and resolved at cdc_prof_get.
 * on = 0: off (default); on = 1: every launch; on = n > 1: inside cdc_decode only the DDIM
 * iterations with i % n == 0 are instrumented (sampling inside the timed region). */
int cdc_prof_enable(cdc_handle *h, int on);
int cdc_prof_num_classes(void);
const char *cdc_prof_name(int cls);
/* ms = accumulated GPU milliseconds, launches = kernel launches, flops = executed MFMA flops,
 * bytes = algorithmic global bytes (inputs read once + outputs written once per launch). */
int cdc_prof_get(cdc_handle *h, int cls, double *ms, int64_t *launches, double *flops,
                 double *bytes);
int cdc_prof_reset(cdc_handle *h);
/* Per-launch view of the same events: one entry per kernel launch of the current program (label = kernel kind,
 * shape and launch plan; ms = accumulated GPU time over `launches` instrumented executions; flops per execution). */
int cdc_prof_num_ops(cdc_handle *h);
int cdc_prof_op(cdc_handle *h, int idx, const char **label, double *ms, int64_t *launches, double *flops);

/* ---- single operators (used by the parity tests; same kernels the U-Net graph launches) ------ */

/* F.conv2d(x, w[Cout,Cin,KH,KW], bias, stride, padding) with the fused epilogue options of
 * Block/ResnetBlock (network_components.py:83-114):
 *   ln_g/ln_b != NULL: channel LayerNorm (eps 1e-5) after bias;  relu: ReLU after LN;
 *   shift [B,Cout] != NULL: + shift[b,co] after ReLU (the time-embedding add, :110-111);
 *   resid [B,Cout,Ho,Wo] != NULL: + resid at the end (:114).  Host pointers. */
int cdc_op_conv2d(cdc_handle *h, const float *x, const float *w, const float *bias, float *y,
                  int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                  const float *ln_g, const float *ln_b, int relu, const float *shift,
                  const float *resid);
/* F.conv_transpose2d(x, w[Cin,Cout,4,4], bias, stride=2, padding=1) (Upsample, :34-42). */
int cdc_op_conv_transpose2d(cdc_handle *h, const float *x, const float *w, const float *bias,
                            float *y, int B, int Cin, int H, int W, int Cout);
/* LayerNorm.forward (:56-66). */
int cdc_op_chan_layernorm(cdc_handle *h, const float *x, const float *g, const float *b, float *y,
                          int B, int C, int HW);
/* Residual(PreNorm(LinearAttention)) (:10-16,69-77,117-139): y = to_out(attn(to_qkv(LN(x)))) + x. */
int cdc_op_linear_attention(cdc_handle *h, const float *x, const float *norm_g,
                            const float *norm_b, const float *w_qkv, const float *w_out,
                            const float *b_out, float *y, int B, int C, int H, int W);

/* Determinism stress of the single-operator entry points (test infrastructure of the model path: it has no atomics and every
 * summation order is fixed by the launch geometry, so a launch program must reproduce its own bits).  After
 * cdc_op_stress(h, n), every cdc_op_* call launches its program n more times behind the first execution and counts the
 * executions whose result differs bitwise from the first one -- all on the device, no host round trip per launch.
 * cdc_op_stress_result returns the counts of the LAST cdc_op_* call.  n = 0 (default) turns it off. */
int cdc_op_stress(cdc_handle *h, int repeats);
int cdc_op_stress_result(cdc_handle *h, int64_t *launches, int64_t *differing);

#ifdef __cplusplus
}
#endif
#endif /* CDC_HIP_H */
