// cdc_api.hip -- C-ABI of libcdc_hip.so (include/cdc_hip.h): parameter repacking, the per-shape
// launch program of Unet.forward (xparam/modules/unet.py:106-135), and the DDIM sampler loop
// (xparam/modules/denoising_diffusion.py:152-205, epsilonparam/...:137-192).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/cdc_hip.h"
#include "cdc_internal.h"
#include "conv_ws_kernel.h"
#include "conv_ws1_kernel.h"
#include "entropy.h"

using namespace cdc;

namespace {

enum ProfClass { PC_CONV3 = 0, PC_CONV7, PC_CONV1, PC_DOWN, PC_UP, PC_ATTN_CTX, PC_LN, PC_SMALL,
                 PC_COUNT };
const char *kProfNames[PC_COUNT] = {"conv3x3", "conv7x7", "conv1x1", "downsample", "upsample",
                                    "attn_ctx", "layernorm", "small"};

struct Param {                    // one state_dict entry
    std::string name;
    std::vector<int64_t> shape;
    std::vector<float> host;
    bool loaded = false;
    bool optional = false;        // not part of the enumerated manifest; may stay unloaded (prior of the rate estimate)
    size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};

struct ConvW {                    // packed convolution weights (device)
    int Cin = 0, Cout = 0, KH = 1, KW = 1, stride = 1, pad = 0;
    int pad_y = -1, pad_x = -1;   // override `pad` per axis when >= 0 (row-folded convolutions)
    int tk = 4;                   // transposed: kernel size of the reference layer (4: (4,2,1); 5: (5,2,2,op 1))
    bool transposed = false;      // ConvTranspose2d 4x4 s2 p1 as four 2x2 phase convolutions
    int Cin_pad = 0, COP = 0, nz = 1;
    float *wp = nullptr, *bias = nullptr;
    long long w_zs = 0, w_bs = 0;
    unsigned short *wsp = nullptr;   // three-plane bf16 form for conv_split_kernel (k x k, Cin >= 16)
    long long wsp_zs = 0;
    unsigned short *wsh = nullptr;   // fp16 planes {WH, WL, WH2} of w * 2^s (conv_split2_kernel AR = 1), same layout
    float wscale_inv = 1.f;          // 2^-s
};

struct Act { float *p = nullptr; int C = 0, H = 0, W = 0;
             const void *pf = nullptr; long long pf_bs = 0;   // set for a planes-only tensor: its PF copy (cdc_unet_tap unpacks it into p)
             long long bs() const { return (long long)C * H * W; } };

struct ResBlockW { std::string prefix; int cin, cout, k; bool has_res; int shift_off;
                   ConvW c1, c2, cres; float *g1, *b1, *g2, *b2, *mlp_w = nullptr, *mlp_b = nullptr;
                   // context hoisting: input = cat(x [hoist_cx ch], ctx); the ctx halves of block1 and
                   // res_conv are step-invariant, so they are split off and evaluated once per decode
                   int hoist_cx = 0; ConvW c1x, c1c, cresx, cresc;
                  ConvW c1u; bool has_unfold = false; bool has_mlp = true; };   // c1x as a KH x 1 conv over KW*cx unfolded channels
struct AttnW { std::string prefix; int C; ConvW qkv, out; float *ng, *nb;
               // qkv / kv: to_qkv (all rows / k,v rows) with the PreNorm affine folded in (g*W, W.b)
               ConvW kv; float *WoT = nullptr, *WqT = nullptr, *uq = nullptr, *Wq = nullptr;   // Wq [d][ci]: the A operand of fold_r2_mfma_kernel
               float *kvWt = nullptr, *kvb = nullptr; unsigned short *kvWs = nullptr;
               unsigned short *kvWh = nullptr; float kv_scale_inv = 1.f; };   // fp16 planes {WH, WL, WH2} of W' 2^s   // fused front half: (W_kv diag(g))^T [C][2C], W_kv b_ln [2C]   // folded output; uq = Wq b_ln

struct Op {
    enum Kind { CONV, LN, TEMB, KSTATS, CTXP, CTXR, CTXF, COMBINE, DDIM, COPY, UNFOLD, KVCTX, LNCONV, CONVPF, PFPACK, CONVWS, CONVWS1 } kind;
    int prof = PC_SMALL;
    int id = -1;                  // index into cdc_handle::op_ms (per-op timing table, debug aid)
    char label[96] = {0};
    double flops = 0, bytes = 0;
    ConvArgs conv; ConvPlan plan; int nz = 1;
    PfArgs pf; PfPlan pfplan;     // CONVPF: pre-split fp16 operands by LDS-DMA (conv_pf_kernel.h)
    WsArgs ws; WsPlan wsplan;     // CONVWS: weight-stationary 3x3 convolution of the few-pixel levels (conv_ws_kernel.h)
    Ws1Args ws1; Ws1Plan ws1plan; // CONVWS1: its 1x1 sibling (conv_ws1_kernel.h)
    bool pw = false;              // CONVPF on conv_pw_kernel (pointwise, activations from the fp32 tensor)
    LnArgs ln;
    TembArgs temb;
    struct { const float *k, *v; long long bs; int C, N; float *kmax, *ksum, *S, *ctxw;
             int nsplit, Cin_pad, COP; float scale; const float *WoT, *WqT; float *T1;
             const float *ln_g, *ln_b, *b_out; float *biasB; } at;
    struct { const float *P, *bias; float *out; int Cout, KH, pad, H, W; } cb;
    DdimArgs ddim;
    struct { const float *src; long long src_bs; float *dst; long long dst_bs, n; } cp;
    int cp_parts = 1; long long cp_part_stride = 0;
    const int *cp_step = nullptr; long long cp_step_stride = 0;   // COPY: source row selected by a device step index
    KvCtxArgs kvc;
    LnConvArgs lnc;
    int at_ws_f16 = 0;                 // CTXF: the planes are fp16 {WH, WL, WH2} of M' 2^8 (split convolution) instead of bf16
    unsigned short *at_Ws = nullptr;   // CTXF: also emit M' as bf16 planes for lnconv_kernel
    const float *at_Wq = nullptr;      // CTXF: Wq [d][ci] (fold_r2_mfma_kernel)
    const float *at_M = nullptr;  // CTXF after KVCTX: per-split row maxima   // COPY: dst = sum of cp_parts planes of src
    int at_one = 0;                    // CTXP: row maxima, partial context and reduction in this ONE launch (ctx_one_launch)
    struct { const float *src; long long src_bs; float *dst; long long dst_bs; int C, KW, pad, H, W; } uf;
    struct { const float *src; long long src_bs; void *dst; long long dst_bs; int C, H, W; int c4; } pk;   // PFPACK (c4: fp32 -> accumulator order)
};

}  // namespace

static int default_arith() {      // CDC_ARITH=0 selects the three-plane bf16 arithmetic for new handles
    const char *e = getenv("CDC_ARITH");
    return e ? (atoi(e) ? 1 : 0) : 1;
}

struct cdc_handle {
    cdc_unet_config cfg;
    int kind = 0;                 // 0: denoising U-Net, 1: context decoder (Compressor.decode), 2: hyper decoder,
                                  // 3: encoder (enc + hyper_enc)
    std::vector<int> enc_dims, henc_dims;     // kind 3
    int down_index = 1;
    std::vector<int> hyper_dims;  // kind 2: reversed_hyper_dims
    std::vector<ConvW> hconvs;    // kind 2: packed layers
    float *d_prior = nullptr;     // kind 2: FlexiblePrior per channel, 44 floats (softplus / tanh applied), or null
    std::vector<double> h_prior;  // kind 2: the same in float64 (probability tables of the entropy coder)
    std::unique_ptr<cdc::EntropyModel> ent;   // kind 2: entropy coder tables (built on first use)
    uint32_t ent_model_hash = 0;
    int ent_max_positions = 1 << 22;          // kind 2: largest hh * wh cdc_entropy_decode accepts from a stream header (cdc_entropy_set_limit)
    std::vector<int> rev_dims;    // kind 1: [dim*m for m in rev_mults] + [out_channels]
    int up_index = 1;
    std::vector<Act> dec_outs;    // kind 1: outputs of the program, coarsest first
    int device = 0;
    int arith = default_arith();  // k x k / wide 1x1 convolutions: 1 two fp16 planes (3 MFMA products), 0 three bf16 planes (6)
    std::string err;
    hipStream_t own_stream = nullptr;
    // architecture (unet.py:33-35)
    std::vector<int> dims, context_dims;
    int n_res = 0, out_dim = 0;
    std::vector<Param> params;
    std::map<std::string, int> pindex;
    bool finalized = false;
    std::vector<void *> weight_allocs;
    // weights
    float *tm_w0 = nullptr, *tm_b0 = nullptr, *tm_w2 = nullptr, *tm_b2 = nullptr;
    std::vector<ResBlockW> rbs;          // in forward order
    std::vector<AttnW> attns;
    std::vector<ConvW> downs, ups;
    float *fin_g = nullptr, *fin_b = nullptr;
    ConvW fin_conv;               // row-folded: 1 x 7 taps, out_dim*7 virtual channels
    float *fin_bias = nullptr;
    float *fin_P = nullptr;
    TembLayer *d_temb_layers = nullptr;
    int shift_bs = 0;
    // program
    int pB = 0, pH = 0, pW = 0;
    bool retry_futile = false;    // range guard: the BF16X3 repetition was non-finite too
    bool p_batch1_plan = false;   // the program was planned as for one image (entropy coder contract, entropy.hip)
    std::vector<Op> ops;          // per DDIM iteration (depends on x_t and t)
    std::vector<Op> pre_ops;      // depends on the context pyramid only: once per decode / forward
    std::vector<void *> act_allocs;
    size_t act_bytes = 0;
    float *in_x = nullptr, *in_time = nullptr, *out_fx = nullptr, *shift = nullptr;
    std::vector<Act> in_ctx;
    std::map<std::string, Act> taps;     // named intermediate activations of the last forward (cdc_unet_tap)
    float *xa = nullptr, *xb = nullptr, *noise_buf = nullptr;     // decode ping-pong
    // schedule
    int steps = 0;
    float *d_tab = nullptr;              // [5][steps]
    float *d_tab_v = nullptr;            // [2][steps] (cdc_set_schedule_v), valid for schedule generation tab_v_gen
    int tab_v_gen = -1, tab_v_steps = 0;
    std::vector<float> h_tab;            // host copy of d_tab: an unchanged schedule is not uploaded again
    size_t tab_cap = 0, trows_cap = 0;   // capacities (floats) of d_tab / d_shift_tab: buffers are reused, not leaked
    int sched_gen = 0;                   // bumped whenever the device tables change (invalidates the captured graph)
    std::vector<float> h_time_in;
    float *d_time_steps = nullptr;       // [steps] U-Net time input per sample step
    float *d_shift_tab = nullptr;        // [steps][shift_bs]: time-embedding shifts of every step
    // hipGraph replay of one DDIM iteration (launch-bound small batches): the step index lives on the device
    int *d_step = nullptr;
    int range_faults = 0;                // calls repeated in bf16x3 arithmetic after an fp16 range overflow
    int nonfinite_results = 0;           // results that are non-finite in the full-range arithmetic too (as the reference's would be)
    bool in_retry = false;               // the current call is the bf16x3 repetition of a faulted one
    int *d_fault = nullptr;              // sticky "non-finite U-Net output" flag written by the sampler kernel
    hipGraphExec_t graph_exec = nullptr;
    hipEvent_t gev_in = nullptr, gev_out = nullptr;   // order the caller's stream around the graph stream
    int graph_key[4] = {0, 0, 0, 0};      // steps, pred_mode, clip, stream-independent program generation
    int time_steps_B = 0;
    // profiling
    bool prof = false;
    double prof_ms[PC_COUNT] = {0}, prof_flops[PC_COUNT] = {0}, prof_bytes[PC_COUNT] = {0};
    int64_t prof_launches[PC_COUNT] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // deferred (non-blocking) event timing: pairs recorded on the launch stream, resolved at
    // cdc_prof_get.  prof_every = n profiles only the DDIM iterations with i % n == 0.
    struct Pending { hipEvent_t a, b; int cls; double flops, bytes; int id; };
    std::vector<double> op_ms;
    std::vector<long> op_n;
    std::vector<std::string> op_label;
    std::vector<double> op_flops;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_free;
    int prof_every = 1;
    bool prof_now = false;
};

namespace {

std::string g_create_err;

int fail(cdc_handle *h, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(h, CDC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                    \
    } while (0)

// ------------------------------------------------------------------------------------------------
// manifest (reference state_dict order: time_mlp, downs, ups, mid_*, final_conv)
// ------------------------------------------------------------------------------------------------
void add_param(cdc_handle *h, const std::string &name, std::vector<int64_t> shape, bool optional = false) {
    Param p;
    p.name = name;
    p.optional = optional;
    p.shape = std::move(shape);
    h->pindex[name] = (int)h->params.size();
    h->params.push_back(std::move(p));
}

void add_resblock_params(cdc_handle *h, const std::string &p, int cin, int cout, int k, bool with_mlp = true) {
    const int d = h->cfg.dim;
    if (with_mlp) {
        add_param(h, p + ".mlp.1.weight", {cout, d});
        add_param(h, p + ".mlp.1.bias", {cout});
    }
    add_param(h, p + ".block1.block.0.weight", {cout, cin, k, k});
    add_param(h, p + ".block1.block.0.bias", {cout});
    add_param(h, p + ".block1.block.1.g", {1, cout, 1, 1});
    add_param(h, p + ".block1.block.1.b", {1, cout, 1, 1});
    add_param(h, p + ".block2.block.0.weight", {cout, cout, 3, 3});
    add_param(h, p + ".block2.block.0.bias", {cout});
    add_param(h, p + ".block2.block.1.g", {1, cout, 1, 1});
    add_param(h, p + ".block2.block.1.b", {1, cout, 1, 1});
    if (cin != cout) {
        add_param(h, p + ".res_conv.weight", {cout, cin, 1, 1});
        add_param(h, p + ".res_conv.bias", {cout});
    }
}

void add_attn_params(cdc_handle *h, const std::string &p, int c) {
    add_param(h, p + ".fn.fn.to_qkv.weight", {3 * c, c, 1, 1});
    add_param(h, p + ".fn.fn.to_out.weight", {c, c, 1, 1});
    add_param(h, p + ".fn.fn.to_out.bias", {c});
    add_param(h, p + ".fn.norm.g", {1, c, 1, 1});
    add_param(h, p + ".fn.norm.b", {1, c, 1, 1});
}

int down_in_channels(const cdc_handle *h, int ind) {     // unet.py:65-68
    const int dim_in = h->dims[ind];
    const bool is_last = ind >= h->n_res - 1;
    if (!is_last && ind < (int)h->context_dims.size() - 1) return dim_in + h->context_dims[ind];
    return dim_in;
}

void build_manifest(cdc_handle *h) {
    const int d = h->cfg.dim;
    add_param(h, "time_mlp.0.weight", {4 * d, 1});
    add_param(h, "time_mlp.0.bias", {4 * d});
    add_param(h, "time_mlp.2.weight", {d, 4 * d});
    add_param(h, "time_mlp.2.bias", {d});
    const int n = h->n_res;
    for (int i = 0; i < n; ++i) {
        const std::string p = "downs." + std::to_string(i);
        const int dout = h->dims[i + 1];
        add_resblock_params(h, p + ".0", down_in_channels(h, i), dout, i == 0 ? 7 : 3);
        add_resblock_params(h, p + ".1", dout, dout, 3);
        add_attn_params(h, p + ".2", dout);
        if (i < n - 1) {
            add_param(h, p + ".3.conv.weight", {dout, dout, 3, 3});
            add_param(h, p + ".3.conv.bias", {dout});
        }
    }
    for (int i = 0; i < n - 1; ++i) {          // reversed(in_out[1:]), unet.py:88
        const int lvl = n - 1 - i;             // in_out[lvl] = (dims[lvl], dims[lvl+1])
        const int din = h->dims[lvl], dout = h->dims[lvl + 1];
        const std::string p = "ups." + std::to_string(i);
        add_resblock_params(h, p + ".0", dout * 2, din, 3);
        add_resblock_params(h, p + ".1", din, din, 3);
        add_attn_params(h, p + ".2", din);
        add_param(h, p + ".3.conv.weight", {din, din, 4, 4});
        add_param(h, p + ".3.conv.bias", {din});
    }
    const int mid = h->dims[n];
    add_resblock_params(h, "mid_block1", mid, mid, 3);
    add_attn_params(h, "mid_attn", mid);
    add_resblock_params(h, "mid_block2", mid, mid, 3);
    add_param(h, "final_conv.0.g", {1, d, 1, 1});
    add_param(h, "final_conv.0.b", {1, d, 1, 1});
    add_param(h, "final_conv.1.weight", {h->out_dim, d, 7, 7});
    add_param(h, "final_conv.1.bias", {h->out_dim});
}

// ------------------------------------------------------------------------------------------------
// weight upload / repacking
// ------------------------------------------------------------------------------------------------
int upload(cdc_handle *h, const float *src, size_t n, float **dst, std::vector<void *> *pool) {
    void *p = nullptr;
    HIP_TRY(h, hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)));
    pool->push_back(p);
    if (src && n) HIP_TRY(h, hipMemcpy(p, src, n * sizeof(float), hipMemcpyHostToDevice));
    *dst = (float *)p;
    return CDC_OK;
}

const std::vector<float> &hostp(cdc_handle *h, const std::string &name) {
    return h->params[h->pindex.at(name)].host;
}

int upload_param(cdc_handle *h, const std::string &name, float **dst) {
    const auto &v = hostp(h, name);
    return upload(h, v.data(), v.size(), dst, &h->weight_allocs);
}

// Conv2d OIHW -> [tap][Cin_pad][COP]; ConvTranspose2d IOHW(4x4,s2,p1) -> [phase][2x2 tap][Cin_pad][COP].
// (ci0, ncin) / (co0, ncout) select an input / output channel slice of the full weight (hoisted
// context halves, the k,v rows of to_qkv); ncin/ncout = 0 take everything.
int pack_conv(cdc_handle *h, const float *w, const float *bias, int CoutF, int CinF, int KH, int KW,
              int stride, int pad, bool transposed, ConvW *cw, std::vector<void *> *pool, int ci0 = 0,
              int ncin = 0, int co0 = 0, int ncout = 0) {
    const int Cin = ncin ? ncin : CinF, Cout = ncout ? ncout : CoutF;
    cw->Cin = Cin; cw->Cout = Cout; cw->stride = stride; cw->pad = pad;
    cw->transposed = transposed;
    cw->Cin_pad = round_up(Cin, 16);
    cw->COP = round_up(Cout, 32);
    std::vector<float> packed;
    if (!transposed) {
        cw->KH = KH; cw->KW = KW; cw->nz = 1;
        const int taps = KH * KW;
        packed.assign((size_t)taps * cw->Cin_pad * cw->COP, 0.f);
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci)
                for (int t = 0; t < taps; ++t)
                    packed[((size_t)t * cw->Cin_pad + ci) * cw->COP + co] =
                        w[((size_t)(co0 + co) * CinF + ci0 + ci) * taps + t];
        cw->w_zs = 0;
    } else if (KH == 5) {
        // ConvTranspose2d(5, stride 2, padding 2, output_padding 1) (hyper decoder, compress_modules.py:166-177):
        // out[2m+py] takes ky = 2d + py + 2 from x[m-d]: phase 0 rows m-1, m, m+1 (ky 4, 2, 0), phase 1 rows m, m+1
        // (ky 3, 1).  Every phase becomes a 3x3 / pad-1 convolution; the taps a phase lacks stay zero.
        cw->KH = 3; cw->KW = 3; cw->nz = 4; cw->stride = 1; cw->tk = 5;
        cw->w_zs = (long long)9 * cw->Cin_pad * cw->COP;
        packed.assign((size_t)4 * cw->w_zs, 0.f);
        for (int z = 0; z < 4; ++z) {
            const int py = z >> 1, px = z & 1;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    const int ky = py == 0 ? 4 - 2 * a : 5 - 2 * a, kx = px == 0 ? 4 - 2 * b : 5 - 2 * b;
                    if (ky > 4 || kx > 4) continue;          // (py = 1, a = 0): no such tap
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int co = 0; co < Cout; ++co)
                            packed[(size_t)z * cw->w_zs +
                                   ((size_t)(a * 3 + b) * cw->Cin_pad + ci) * cw->COP + co] =
                                w[(((size_t)(ci0 + ci) * CoutF + co0 + co) * 5 + ky) * 5 + kx];
                }
        }
    } else {
        // out[2m+py][2n+px] = sum_{a,b in {0,1}} x[m+a-(1-py)][n+b-(1-px)] * w[ci][co][3-py-2a][3-px-2b]
        cw->KH = 2; cw->KW = 2; cw->nz = 4; cw->stride = 1;
        cw->w_zs = (long long)4 * cw->Cin_pad * cw->COP;
        packed.assign((size_t)4 * cw->w_zs, 0.f);
        for (int z = 0; z < 4; ++z) {
            const int py = z >> 1, px = z & 1;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const int ky = 3 - py - 2 * a, kx = 3 - px - 2 * b;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int co = 0; co < Cout; ++co)
                            packed[(size_t)z * cw->w_zs +
                                   ((size_t)(a * 2 + b) * cw->Cin_pad + ci) * cw->COP + co] =
                                w[(((size_t)(ci0 + ci) * CoutF + co0 + co) * 4 + ky) * 4 + kx];
                }
        }
    }
    int rc = upload(h, packed.data(), packed.size(), &cw->wp, pool);
    if (rc) return rc;
    cw->wsp = nullptr;
    if (((cw->KH * cw->KW > 1 && Cin >= 16) || Cin >= 32) && !dev_env("CDC_NO_SPLIT")) {
        // exact three-way bf16 split (truncation): w = w1 + w2 + w3, laid out in MFMA A-operand order
        // [z][tap][Cin_pad/16][plane][k-half][COP][8 cin]
        const int taps = cw->KH * cw->KW, nc16 = cw->Cin_pad / 16;
        const size_t per_z = (size_t)taps * nc16 * 6 * cw->COP * 8;
        std::vector<unsigned short> sp(per_z * cw->nz, 0);
        for (int z = 0; z < cw->nz; ++z)
            for (int t = 0; t < taps; ++t)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int co = 0; co < Cout; ++co) {
                        const float v = packed[(size_t)z * (size_t)taps * cw->Cin_pad * cw->COP +
                                               ((size_t)t * cw->Cin_pad + ci) * cw->COP + co];
                        uint32_t u; memcpy(&u, &v, 4);
                        const uint32_t h1 = u & 0xFFFF0000u; float f1; memcpy(&f1, &h1, 4);
                        const float r = v - f1; uint32_t ur; memcpy(&ur, &r, 4);
                        const uint32_t h2 = ur & 0xFFFF0000u; float f2; memcpy(&f2, &h2, 4);
                        const float r2 = r - f2; uint32_t h3; memcpy(&h3, &r2, 4);
                        const uint32_t parts[3] = {h1, h2, h3};
                        const int c16 = ci >> 4, kg = (ci >> 3) & 1, q = ci & 7;
                        for (int pl = 0; pl < 3; ++pl)
                            sp[(size_t)z * per_z +
                               ((((size_t)t * nc16 + c16) * 6 + pl * 2 + kg) * cw->COP + co) * 8 + q] =
                                (unsigned short)(parts[pl] >> 16);
                    }
        float *dsp = nullptr;
        if ((rc = upload(h, reinterpret_cast<const float *>(sp.data()), (sp.size() + 1) / 2, &dsp, pool)))
            return rc;
        cw->wsp = reinterpret_cast<unsigned short *>(dsp);
        cw->wsp_zs = (long long)per_z;
        // fp16 planes of w * 2^s, max |w| 2^s in [2^13, 2^14): WH = fp16(w 2^s), WL = fp16(w 2^s - WH), WH2 = WH 2^-11
        // (exact: a power-of-two scale of a normal fp16; |WH| < 2^-3 may round -- 17 binades below the layer's
        // largest weight).  See conv_split_kernel.h (AR = 1).
        float wmax = 0.f;
        for (float v : packed) wmax = std::max(wmax, fabsf(v));
        int sexp = 0;
        if (wmax > 0.f && std::isfinite(wmax)) { int e; frexpf(wmax, &e); sexp = 14 - e; }   // wmax = m 2^e, m in [.5, 1)
        sexp = std::max(-100, std::min(100, sexp));
        const float scl = ldexpf(1.f, sexp);
        std::vector<unsigned short> sh(per_z * cw->nz, 0);
        auto f16bits = [](float f) { const _Float16 hf = (_Float16)f; unsigned short u; memcpy(&u, &hf, 2); return u; };
        for (int z = 0; z < cw->nz; ++z)
            for (int t = 0; t < taps; ++t)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int co = 0; co < Cout; ++co) {
                        const float v = packed[(size_t)z * (size_t)taps * cw->Cin_pad * cw->COP +
                                               ((size_t)t * cw->Cin_pad + ci) * cw->COP + co] * scl;
                        const _Float16 wh = (_Float16)v;
                        const float wl = v - (float)wh;
                        const unsigned short parts[3] = {f16bits((float)wh), f16bits(wl), f16bits((float)wh * (1.0f / 2048.0f))};
                        const int c16 = ci >> 4, kg = (ci >> 3) & 1, q = ci & 7;
                        for (int pl = 0; pl < 3; ++pl)
                            sh[(size_t)z * per_z +
                               ((((size_t)t * nc16 + c16) * 6 + pl * 2 + kg) * cw->COP + co) * 8 + q] = parts[pl];
                    }
        float *dsh = nullptr;
        if ((rc = upload(h, reinterpret_cast<const float *>(sh.data()), (sh.size() + 1) / 2, &dsh, pool)))
            return rc;
        cw->wsh = reinterpret_cast<unsigned short *>(dsh);
        cw->wscale_inv = ldexpf(1.f, -sexp);
    }
    cw->bias = nullptr;
    if (bias) rc = upload(h, bias + co0, Cout, &cw->bias, pool);
    return rc;
}

int pack_named_conv(cdc_handle *h, const std::string &wname, const std::string &bname, int stride,
                    int pad, bool transposed, ConvW *cw, int ci0 = 0, int ncin = 0, int co0 = 0,
                    int ncout = 0) {
    const Param &p = h->params[h->pindex.at(wname)];
    const float *bias = bname.empty() ? nullptr : hostp(h, bname).data();
    const int d0 = (int)p.shape[0], d1 = (int)p.shape[1];
    const int KH = (int)p.shape[2], KW = (int)p.shape[3];
    if (!transposed)
        return pack_conv(h, p.host.data(), bias, d0, d1, KH, KW, stride, pad, false, cw,
                         &h->weight_allocs, ci0, ncin, co0, ncout);
    return pack_conv(h, p.host.data(), bias, d1, d0, KH, KW, stride, pad, true, cw,
                     &h->weight_allocs, ci0, ncin, co0, ncout);
}

int pack_resblock(cdc_handle *h, const std::string &p, int cin, int cout, int k, int *shift_off,
                  int hoist_cx = 0, bool with_mlp = true) {
    ResBlockW rb;
    rb.has_mlp = with_mlp;
    rb.prefix = p; rb.cin = cin; rb.cout = cout; rb.k = k; rb.has_res = cin != cout;
    rb.hoist_cx = hoist_cx;
    rb.shift_off = *shift_off;
    *shift_off += round_up(cout, 32);
    int rc;
    if ((rc = pack_named_conv(h, p + ".block1.block.0.weight", p + ".block1.block.0.bias", 1, k / 2,
                              false, &rb.c1))) return rc;
    if ((rc = pack_named_conv(h, p + ".block2.block.0.weight", p + ".block2.block.0.bias", 1, 1,
                              false, &rb.c2))) return rc;
    if (rb.has_res &&
        (rc = pack_named_conv(h, p + ".res_conv.weight", p + ".res_conv.bias", 1, 0, false, &rb.cres)))
        return rc;
    if (hoist_cx > 0) {
        const std::string w1 = p + ".block1.block.0.weight", b1 = p + ".block1.block.0.bias";
        if ((rc = pack_named_conv(h, w1, "", 1, k / 2, false, &rb.c1x, 0, hoist_cx))) return rc;
        if (k > 3 && hoist_cx * k <= 32) {
            // column-unfolded form of the few-channel k x k layer: w'[co][kx*cx + c][ky][0] = w[co][c][ky][kx]
            const Param &pw = h->params[h->pindex.at(w1)];
            const int co_n = (int)pw.shape[0], ci_n = (int)pw.shape[1], cu = hoist_cx * k;
            std::vector<float> wu((size_t)co_n * cu * k);
            for (int co = 0; co < co_n; ++co)
                for (int c = 0; c < hoist_cx; ++c)
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx)
                            wu[((size_t)co * cu + kx * hoist_cx + c) * k + ky] =
                                pw.host[(((size_t)co * ci_n + c) * k + ky) * k + kx];
            if ((rc = pack_conv(h, wu.data(), nullptr, co_n, cu, k, 1, 1, 0, false, &rb.c1u, &h->weight_allocs)))
                return rc;
            rb.c1u.pad_y = k / 2; rb.c1u.pad_x = 0;
            rb.has_unfold = rb.c1u.wsp != nullptr;
        }
        if ((rc = pack_named_conv(h, w1, b1, 1, k / 2, false, &rb.c1c, hoist_cx, cin - hoist_cx)))
            return rc;
        if (rb.has_res) {
            const std::string wr = p + ".res_conv.weight", br = p + ".res_conv.bias";
            if ((rc = pack_named_conv(h, wr, "", 1, 0, false, &rb.cresx, 0, hoist_cx))) return rc;
            if ((rc = pack_named_conv(h, wr, br, 1, 0, false, &rb.cresc, hoist_cx, cin - hoist_cx)))
                return rc;
        }
    }
    if ((rc = upload_param(h, p + ".block1.block.1.g", &rb.g1))) return rc;
    if ((rc = upload_param(h, p + ".block1.block.1.b", &rb.b1))) return rc;
    if ((rc = upload_param(h, p + ".block2.block.1.g", &rb.g2))) return rc;
    if ((rc = upload_param(h, p + ".block2.block.1.b", &rb.b2))) return rc;
    if (with_mlp) {
        if ((rc = upload_param(h, p + ".mlp.1.weight", &rb.mlp_w))) return rc;
        if ((rc = upload_param(h, p + ".mlp.1.bias", &rb.mlp_b))) return rc;
    }
    h->rbs.push_back(rb);
    return CDC_OK;
}

// to_qkv rows [co0, co0+nco) with the PreNorm affine folded in: W' = W diag(g), bias' = W b
// (LNMODE 2 of the conv kernel).
int pack_qkv_folded(cdc_handle *h, const float *wq, const float *g, const float *bln, int C, int co0,
                    int nco, ConvW *cw, std::vector<void *> *pool) {
    std::vector<float> w((size_t)nco * C), bias(nco);
    for (int co = 0; co < nco; ++co) {
        double acc = 0;
        for (int ci = 0; ci < C; ++ci) {
            const float v = wq[(size_t)(co0 + co) * C + ci];
            w[(size_t)co * C + ci] = v * g[ci];
            acc += (double)v * bln[ci];
        }
        bias[co] = (float)acc;
    }
    return pack_conv(h, w.data(), bias.data(), nco, C, 1, 1, 1, 0, false, cw, pool);
}

int pack_attn(cdc_handle *h, const std::string &p, int c) {
    AttnW a;
    a.prefix = p; a.C = c;
    int rc;
    {
        const auto &wq = hostp(h, p + ".fn.fn.to_qkv.weight");
        const auto &g = hostp(h, p + ".fn.norm.g");
        const auto &bn = hostp(h, p + ".fn.norm.b");
        if ((rc = pack_qkv_folded(h, wq.data(), g.data(), bn.data(), c, 0, 3 * c, &a.qkv, &h->weight_allocs)))
            return rc;
        if ((rc = pack_qkv_folded(h, wq.data(), g.data(), bn.data(), c, c, 2 * c, &a.kv, &h->weight_allocs)))
            return rc;
    }
    if ((rc = pack_named_conv(h, p + ".fn.fn.to_out.weight", p + ".fn.fn.to_out.bias", 1, 0, false,
                              &a.out))) return rc;
    if ((rc = upload_param(h, p + ".fn.norm.g", &a.ng))) return rc;
    if ((rc = upload_param(h, p + ".fn.norm.b", &a.nb))) return rc;
    // folded output: Wo^T [e][c] and Wq^T [ci][d]
    {
        const auto &wo = hostp(h, p + ".fn.fn.to_out.weight");   // [c][e]
        const auto &wq = hostp(h, p + ".fn.fn.to_qkv.weight");   // rows 0..C-1 = Wq [d][ci]
        std::vector<float> woT((size_t)c * c), wqT((size_t)c * c);
        for (int i = 0; i < c; ++i)
            for (int j = 0; j < c; ++j) {
                woT[(size_t)j * c + i] = wo[(size_t)i * c + j];
                wqT[(size_t)j * c + i] = wq[(size_t)i * c + j];
            }
        if ((rc = upload(h, woT.data(), woT.size(), &a.WoT, &h->weight_allocs))) return rc;
        if ((rc = upload(h, wqT.data(), wqT.size(), &a.WqT, &h->weight_allocs))) return rc;
        if ((rc = upload(h, wq.data(), (size_t)c * c, &a.Wq, &h->weight_allocs))) return rc;       // rows 0..C-1 of to_qkv
        const auto &bn = hostp(h, p + ".fn.norm.b");
        std::vector<float> uq(c);
        for (int d = 0; d < c; ++d) {
            double acc = 0;
            for (int ci = 0; ci < c; ++ci) acc += (double)wq[(size_t)d * c + ci] * bn[ci];
            uq[d] = (float)acc;
        }
        if ((rc = upload(h, uq.data(), uq.size(), &a.uq, &h->weight_allocs))) return rc;
        // fused kv-projection + context kernel (attn_kernels.hip): W' = W_kv diag(g) transposed, bias' = W_kv b_ln
        const auto &g = hostp(h, p + ".fn.norm.g");
        std::vector<float> wt((size_t)c * 2 * c), kb(2 * c);
        for (int co = 0; co < 2 * c; ++co) {
            double acc = 0;
            for (int ci = 0; ci < c; ++ci) {
                const float v = wq[(size_t)(c + co) * c + ci];          // k rows then v rows of to_qkv
                wt[(size_t)ci * 2 * c + co] = v * g[ci];
                acc += (double)v * bn[ci];
            }
            kb[co] = (float)acc;
        }
        if ((rc = upload(h, wt.data(), wt.size(), &a.kvWt, &h->weight_allocs))) return rc;
        if (c % 16 == 0) {      // three bf16 planes of W' in A-operand order (kvctx_kernel, C = 64)
            std::vector<unsigned short> sp((size_t)(c / 16) * 3 * 2 * 2 * c * 8);
            for (int co = 0; co < 2 * c; ++co)
                for (int ci = 0; ci < c; ++ci) {
                    const float v = wt[(size_t)ci * 2 * c + co];
                    uint32_t u; memcpy(&u, &v, 4);
                    const uint32_t h1 = u & 0xFFFF0000u; float f1; memcpy(&f1, &h1, 4);
                    const float r = v - f1; uint32_t ur; memcpy(&ur, &r, 4);
                    const uint32_t h2 = ur & 0xFFFF0000u; float f2; memcpy(&f2, &h2, 4);
                    const float r2 = r - f2; uint32_t h3; memcpy(&h3, &r2, 4);
                    const uint32_t parts[3] = {h1, h2, h3};
                    const int q = ci >> 4, kh = (ci >> 3) & 1, i = ci & 7;
                    for (int pl = 0; pl < 3; ++pl)
                        sp[((size_t)((q * 3 + pl) * 2 + kh) * 2 * c + co) * 8 + i] = (unsigned short)(parts[pl] >> 16);
                }
            float *dsp = nullptr;
            if ((rc = upload(h, reinterpret_cast<const float *>(sp.data()), (sp.size() + 1) / 2, &dsp, &h->weight_allocs)))
                return rc;
            a.kvWs = reinterpret_cast<unsigned short *>(dsp);
            // the same in two-plane fp16 arithmetic: {WH, WL, WH2 = WH 2^-11} of W' 2^s (see conv_split_kernel.h AR = 1)
            float wmax = 0.f;
            for (float v : wt) wmax = std::max(wmax, fabsf(v));
            int sexp = 0;
            if (wmax > 0.f && std::isfinite(wmax)) { int e; frexpf(wmax, &e); sexp = 14 - e; }
            sexp = std::max(-100, std::min(100, sexp));
            const float scl = ldexpf(1.f, sexp);
            auto f16bits = [](float f) { const _Float16 hf = (_Float16)f; unsigned short u; memcpy(&u, &hf, 2); return u; };
            std::vector<unsigned short> sh(sp.size(), 0);
            for (int co = 0; co < 2 * c; ++co)
                for (int ci = 0; ci < c; ++ci) {
                    const float v = wt[(size_t)ci * 2 * c + co] * scl;
                    const _Float16 wh = (_Float16)v;
                    const unsigned short parts[3] = {f16bits((float)wh), f16bits(v - (float)wh), f16bits((float)wh * (1.0f / 2048.0f))};
                    const int q = ci >> 4, kh = (ci >> 3) & 1, i = ci & 7;
                    for (int pl = 0; pl < 3; ++pl) sh[((size_t)((q * 3 + pl) * 2 + kh) * 2 * c + co) * 8 + i] = parts[pl];
                }
            float *dsh = nullptr;
            if ((rc = upload(h, reinterpret_cast<const float *>(sh.data()), (sh.size() + 1) / 2, &dsh, &h->weight_allocs)))
                return rc;
            a.kvWh = reinterpret_cast<unsigned short *>(dsh);
            a.kv_scale_inv = ldexpf(1.f, -sexp);
        }
        if ((rc = upload(h, kb.data(), kb.size(), &a.kvb, &h->weight_allocs))) return rc;
    }
    h->attns.push_back(a);
    return CDC_OK;
}

void free_pool(std::vector<void *> *pool) {
    for (void *p : *pool) (void)hipFree(p);
    pool->clear();
}

// ------------------------------------------------------------------------------------------------
// program construction
// ------------------------------------------------------------------------------------------------
constexpr int kKsTarget = 1024;     // split-K: workgroups a few-pixel launch is sliced up to

struct Builder {
    cdc_handle *h;
    int B;
    std::vector<void *> *pool;      // where device allocations are recorded
    int rc = CDC_OK;
    int planB = 0;                  // > 0: choose every kernel variant / K split as for this batch (buffers and grids still use B)
    int pb() const { return planB > 0 ? planB : B; }
    std::vector<Op> *cur = nullptr; // op list being emitted to (h->ops unless set)

    int ws1_h = 0;                  // (label only: rows of the map of the CONVWS1 op being emitted)
    void emit(Op op) {
        op.id = (int)h->op_ms.size();
        h->op_ms.push_back(0); h->op_n.push_back(0); h->op_flops.push_back(op.flops);
        char buf[160];
        const char *kinds[] = {"conv", "ln", "temb", "kstats", "ctxp", "ctxr", "ctxf", "combine", "ddim", "copy", "unfold", "kvctx", "lnconv", "convpf", "pfpack", "convws", "convws1"};
        if (op.kind == Op::PFPACK && op.pk.c4 == 2) kinds[Op::PFPACK] = "pfunpack";
        if (op.kind == Op::CONV)
            snprintf(buf, sizeof buf, "conv %dx%d s%d %4d->%-4d out %3dx%-3d MB%d NPW%d WN%d g%d tg%d ipw%d ks%d%s%s%s%s%s", op.conv.KH,
                     op.conv.KW, op.conv.stride, op.conv.Cin, op.conv.Cout, op.conv.Ho, op.conv.Wo, op.plan.MB,
                     op.plan.NPW, op.plan.WN, op.plan.groups, op.plan.tg, op.plan.ipw, op.plan.ksplit, op.plan.split == 2 ? (op.plan.arith ? " SPLIT2H" : " SPLIT2") : (op.plan.split ? " SPLIT" : ""),
                     op.conv.ep_g ? " LN" : "",
                     op.conv.ln_mean ? " pre" : "", cur == &h->pre_ops ? " HOIST" : "", op.conv.resid ? " +res" : "");
        else if (op.kind == Op::CONVPF)
            snprintf(buf, sizeof buf, "conv %dx%d s%d %4d->%-4d out %3dx%-3d MB%d NPW%d WM%d WP%d g%d R%d %s%s%s%s%s%s", op.pf.KH, op.pf.KW,
                     op.pf.stride == 2 ? 2 : 1, op.pf.Cin, op.pf.Cout, op.pf.Ho, op.pf.Wo, op.pfplan.MB, op.pfplan.NPW, op.pfplan.WM, op.pfplan.WP,
                     op.pfplan.groups, op.pfplan.ring, op.pw ? (op.pf.pre_mean ? "PW pre" : "PW") : (op.pfplan.pf3_epv ? (op.pf.ep_g ? "PF3 LN" : "PF3") : (op.pf.ep_g ? "PF LN" : "PF")), op.pf.out ? "" : " nof32", op.pf.out_pf ? " +pf" : "",
                     op.pf.resid ? " +res" : (op.pf.resid_pf ? " +resP" : ""), cur == &h->pre_ops ? " HOIST" : "", op.pf.tz == 4 ? " TZ4" : "");
        else if (op.kind == Op::CONVWS)
            snprintf(buf, sizeof buf, "conv 3x3 s1 %4d->%-4d out %3dx%-3d NPB%d waves%d tiles%d g%d WS%s", op.ws.Cin, op.ws.Cout, op.ws.H, op.wsplan.W,
                     op.wsplan.NPB, op.wsplan.waves, op.wsplan.tiles, op.wsplan.groups, "");
        else if (op.kind == Op::CONVWS1)
            snprintf(buf, sizeof buf, "conv 1x1 s1 %4d->%-4d out %3dx%-3d NPB%d waves%d tiles%d g%d WS1%s%s%s", op.ws1.Cin, op.ws1.Cout, ws1_h, op.ws1.HW / std::max(ws1_h, 1),
                     op.ws1plan.NPB, op.ws1plan.waves, op.ws1plan.tiles, op.ws1plan.groups, op.ws1.pre_mean ? " pre" : "", op.ws1.w_bs ? " perimg" : "",
                     op.ws1.resid ? " +res" : "");
        else if (op.kind == Op::LN)
            snprintf(buf, sizeof buf, "ln C=%d HW=%d%s", op.ln.C, op.ln.HW, op.ln.out ? "" : " stats");
        else if (op.kind == Op::KVCTX)
            snprintf(buf, sizeof buf, "kvctx C=%d N=%d nsplit=%d", op.kvc.C, op.kvc.N, op.kvc.nsplit);
        else if (op.kind == Op::LNCONV)
            snprintf(buf, sizeof buf, "lnconv C=%d N=%d nsplit=%d", op.lnc.C, op.lnc.N, op.lnc.nsplit);
        else if (op.kind == Op::KSTATS || op.kind == Op::CTXP || op.kind == Op::CTXR || op.kind == Op::CTXF)
            snprintf(buf, sizeof buf, "%s C=%d N=%d nsplit=%d", op.at_one ? "ctx1" : kinds[op.kind], op.at.C, op.at.N, op.at_one ? 1 : op.at.nsplit);
        else
            snprintf(buf, sizeof buf, "%s", kinds[op.kind]);
        h->op_label.push_back(buf);
        (cur ? cur : &h->ops)->push_back(op);
    }

    // Range-guard flag of the handle (conv_args.h: ConvArgs::fault): every convolution / LayerNorm launch of a program reports
    // non-finite accumulators there; the entry points clear it before a call and read it back after (guard_check).
    int *fault_flag() {
        if (!h->d_fault && !rc) {
            void *p = nullptr;
            hipError_t e = hipMalloc(&p, sizeof(int));
            if (e == hipSuccess) e = hipMemset(p, 0, sizeof(int));
            if (e != hipSuccess) { rc = fail(h, CDC_ERR_NOMEM, "range-guard flag: %s", hipGetErrorString(e)); return nullptr; }
            h->d_fault = (int *)p;
            h->weight_allocs.push_back(p);
        }
        return h->d_fault;
    }

    float *dalloc(size_t nfloats) {
        if (rc) return nullptr;
        void *p = nullptr;
        const size_t bytes = std::max<size_t>(nfloats, 1) * sizeof(float);
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            rc = fail(h, CDC_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            return nullptr;
        }
        pool->push_back(p);
        h->act_bytes += bytes;
        return (float *)p;
    }
    // ---- PF twins: a second copy of an activation as two fp16 planes with a zero halo (conv_pf_kernel.h),
    // keyed by the fp32 tensor's address; `valid` once a producer of this program has emitted it.
    struct PfTwin { void *p = nullptr; int C = 0, H = 0, W = 0; bool valid = false;
                    bool only = false;          // the planes are the ONLY copy: the producer wrote no fp32 (readers must take planes)
                    long long ps() const { return (long long)(H + 2) * (W + 2); }
                    long long bs() const { return (long long)(C / 8) * 2 * ps(); } };
    std::map<const float *, PfTwin> pfmap;
    // Opt-in (CDC_PF=1).  The kernel's main loop is 20-30 % faster than the register-staged split kernel (more on
    // long-K layers: 256->64 @128^2 0.56 -> 0.42 ms, 1x1 384->128 @64^2 0.131 -> 0.078 ms), but every tensor that is also
    // needed in fp32 (residual stream, attention input, stride-2 / transposed convolutions) is then written twice, and
    // at batch 32 those extra HBM writes in the producers cost as much as the consumers gain (round 2: 3.64 images/s
    // with planes everywhere, 3.68 with planes up to 128 x 128 (CDC_PF_MAXPIX), 3.75 without).  It pays once the
    // remaining fp32 consumers read planes too.
    // CDC_PF: 0 off; 1 planes for every activation (see above); 2: planes ONLY on the block1 -> block2 edge of a
    // ResnetBlock -- h1 has a single consumer, so it is written as planes INSTEAD of fp32 (same bytes) and block2, half
    // of all 3x3 convolutions, runs on the DMA-fed kernel at no extra traffic; default 3: 2 + a second copy where the
    // per-op table says the consumer gains more than the producer loses (batch 32, ms per launch, producer / consumer):
    //   ResnetBlock output feeding the next ResnetBlock of the level   +0.08 / -0.12 @256^2 ... +0.01 / -0.09 @32^2
    //   Downsample output (next level's first ResnetBlock)             +0.03 / -0.07
    //   attention and Upsample outputs up to 64 x 64 (the two halves of a decoder concat: 384 -> 128 @64^2 -0.19 for
    //   +0.025); at 128^2 the two costs (+0.13) eat the gain (-0.13), the 256^2 skip has no reader at all.
    enum Site { SITE_NONE, SITE_ALWAYS, SITE_RB_CHAIN, SITE_DOWN, SITE_JOIN, SITE_ALWAYS_PLANES };
    int pf_mode() const { const char *e = dev_env("CDC_PF"); return h->arith != 1 ? 0 : (e ? atoi(e) : 3); }
    bool pf_site(Site s, int H, int W) const {
        const int m = pf_mode();
        if (m == 1) return s != SITE_NONE;
        if (m != 3) return false;
        // (round 4: up to 128 x 128 -- the Upsample half of a join is written as planes INSTEAD of fp32 when both of its readers take
        //  planes, join_reads_planes, so only the skip half costs a second copy: 256->64 @128^2 0.52 -> 0.37 ms on conv_pf3_kernel)
        const long long join_max = dev_env("CDC_PF_JOIN_MAXPIX") ? atoll(dev_env("CDC_PF_JOIN_MAXPIX")) : 16384;
        return s == SITE_RB_CHAIN || s == SITE_DOWN || s == SITE_ALWAYS_PLANES || (s == SITE_JOIN && (long long)H * W <= join_max);
    }
    bool pf_on() const { return pf_mode() != 0; }
    static long long pf_maxpix() { const char *e = dev_env("CDC_PF_MAXPIX"); const long long v = e ? atoll(e) : 0; return v > 0 ? v : (1LL << 40); }
    PfTwin *twin(const float *p) { auto it = pfmap.find(p); return it == pfmap.end() ? nullptr : &it->second; }
    bool still_planes_only(const float *p) { PfTwin *t = twin(p); return t && t->only; }    // (false once ensure_f32 unpacked it)
    void add_twin(const float *p, int C, int H, int W) {
        if (rc || !p || !pf_on() || (C % 16) || W < 32 || H < 2 || (long long)H * W > pf_maxpix()) return;
        PfTwin t; t.C = C; t.H = H; t.W = W;
        const size_t bytes = (size_t)B * t.bs() * 16;
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, bytes);
        if (e == hipSuccess) e = hipMemset(q, 0, bytes);           // the halo is written here and never again
        if (e != hipSuccess) { rc = fail(h, CDC_ERR_NOMEM, "PF tensor (%zu bytes): %s", bytes, hipGetErrorString(e)); return; }
        pool->push_back(q);
        h->act_bytes += bytes;
        t.p = q;
        pfmap[p] = t;
    }
    // fp32 -> PF for a tensor whose producer cannot emit planes (a caller-supplied input)
    void pack(const float *p, long long bs) {
        PfTwin *t = twin(p);
        if (rc || !t) return;
        Op op; op.kind = Op::PFPACK; op.prof = PC_SMALL;
        op.pk = {p, bs, t->p, t->bs(), t->C, t->H, t->W, 0};
        op.bytes = 8.0 * B * t->C * t->H * t->W;
        emit(op);
        t->valid = true;
    }
    // PF -> fp32 for a planes-only tensor that reaches a reader of fp32 after all (the planes-only decision is taken by shape
    // predicates BEFORE the readers are planned -- join_would_read_planes, pf_s2_would_plan, ... -- and a predicate can miss: an odd
    // channel split of a join, a reader that needs split-K, a shape below the plane kernels' minimum grid).  The fp32 buffer of
    // every activation is allocated anyway, so the program unpacks h + l * 2^-11 into it once, ahead of that reader (the values the
    // plane readers see), instead of failing the build.  Returns true when `q` is readable as fp32 afterwards.
    bool ensure_f32(const float *q, long long bs) {
        PfTwin *t = q ? twin(q) : nullptr;
        if (!t || !t->only) return true;
        if (rc || !t->valid || bs != (long long)t->C * t->H * t->W) return false;
        Op op; op.kind = Op::PFPACK; op.prof = PC_SMALL;
        op.pk = {q, bs, t->p, t->bs(), t->C, t->H, t->W, 2};      // c4 == 2: unpack (dst = the planes, src = the fp32 buffer to fill)
        op.bytes = 8.0 * B * t->C * t->H * t->W;
        emit(op);
        t->only = false;
        ++n_unpacked;
        if (getenv("CDC_DEBUG_PLAN")) fprintf(stderr, "[plan] planes-only tensor %dx%dx%d unpacked for an fp32 reader\n", t->C, t->H, t->W);
        return true;
    }
    int n_unpacked = 0;

    // A hoisted (step-invariant) partial-sum tensor in accumulator order for conv_pf_kernel's epilogue (PfArgs::pre_c4): packed once per
    // decode, in the context-only part of the program.
    std::map<const float *, float *> c4map;
    const float *pre_add_c4(const float *p, int C, int H, int W, long long bs) {
        if (rc || (C % 4) || bs != (long long)C * H * W || dev_env("CDC_NO_PRE_C4")) return nullptr;
        auto it = c4map.find(p);
        if (it != c4map.end()) return it->second;
        float *q = dalloc((size_t)B * C * H * W);
        if (rc) return nullptr;
        std::vector<Op> *saved = cur;
        cur = &h->pre_ops;
        Op op; op.kind = Op::PFPACK; op.prof = PC_SMALL;
        op.pk = {p, bs, q, 0, C, H, W, 1};
        op.bytes = 8.0 * B * C * H * W;
        emit(op);
        cur = saved;
        c4map[p] = q;
        return q;
    }
    Act new_act(int C, int H, int W, bool want_twin = true, Site site = SITE_ALWAYS) {
        Act a; a.C = C; a.H = H; a.W = W;
        a.p = dalloc((size_t)B * C * H * W);
        if (want_twin && pf_site(site, H, W)) add_twin(a.p, C, H, W);
        return a;
    }

    struct ConvOpts {
        const float *ln_g = nullptr, *ln_b = nullptr;  // fused LN after bias
        int relu = 0;
        float relu_slope = 0.f;                        // LeakyReLU slope (0 = ReLU)
        const float *shift = nullptr;                  // + shift[b][co]
        const float *resid = nullptr; long long resid_bs = 0, resid_cs = 0;
        const float *resid1 = nullptr; long long resid1_bs = 0; int resid_c0 = 0;   // residual over cat[resid (resid_c0 channels), resid1]: plane-operand kernels only
        const float *pre_add = nullptr;                // hoisted partial sums (same layout as out)
        float *stat_mean = nullptr, *stat_rstd = nullptr;
        const float *pre_mean = nullptr, *pre_rstd = nullptr, *pre_g = nullptr, *pre_b = nullptr;
        int pre_mode = 1;                              // 1 in-LDS LN, 2 folded (1x1, weights carry g / W.b)
        int shift_bs = -1;                             // row stride of `shift` (-1: the U-Net's table)
        long long w_bs = 0;
        long long wsp_bs = 0;                          // per-image split planes (elements of 16 bits)
        bool no_bias = false;
        const float *res3_w = nullptr, *res3_x = nullptr; long long res3_bs = 0;   // 3-channel res_conv in the epilogue
        int max_ksplit = 1;                            // > 1: `out` has room for that many partial-sum planes
        bool emit_pf = false;                          // `out` holds final values: also write its PF twin (if it has one)
        bool pf_only = false;                          // plan with conv_pf_kernel or return false
        bool no_f32 = false;                           // PF path only: nobody reads the fp32 copy of `out`
        int uf_c = 0, uf_pad = 0;                      // unfold on load (ConvArgs::uf_c): s0 is the uf_c-channel image, w a KH x 1 layer over KW*uf_c channels
    };
    int last_ksplit = 1;                               // slices the last conv() call really used
    bool last_pf_only = false;                         // the last conv() call wrote its result as planes only (no fp32 copy exists)
    const float *next_res3_w = nullptr, *next_res3_x = nullptr; long long next_res3_bs = 0;   // for the next block()

    // Would BOTH readers of a decoder join cat[a0, a1] -- block1 (3x3, fused LayerNorm) and res_conv (1x1) of the ResnetBlock -- run
    // on the plane-operand kernels, given the twins of the two halves?  Then a0 (an Upsample output, read by nothing else) needs no
    // fp32 copy at all.  Mirrors the conditions of try_pf.
    bool join_reads_planes(const ResBlockW &rb, const float *p0, int C0, const float *p1, int H, int W) {
        PfTwin *t0 = twin(p0), *t1 = twin(p1);
        if (!t0 || !t1 || !t1->valid || t0->H != H || t0->W != W || t1->H != H || t1->W != W) return false;
        if (t0->C != C0 || t0->C + t1->C != rb.c1.Cin) return false;
        return join_would_read_planes(rb, C0, H, W);
    }
    // ... the same question by shapes alone (asked in the encoder path, before the decoder half of the join exists)
    bool join_would_read_planes(const ResBlockW &rb, int C0, int H, int W) {
        if (!pf_on() || !rb.has_res || rb.hoist_cx) return false;
        if (rb.cres.Cin != rb.c1.Cin || (C0 % 16) || C0 <= 0 || C0 >= rb.c1.Cin || !pf_site(SITE_JOIN, H, W)) return false;
        for (const ConvW *w : {&rb.c1, &rb.cres}) {
            const bool k3 = w->KH == 3 && w->KW == 3, k1 = w->KH == 1 && w->KW == 1;
            if (!w->wsh || w->stride != 1 || w->transposed || (w->Cin % 16) || !(k3 || k1)) return false;
            if ((w->pad_y >= 0 ? w->pad_y : w->pad) != w->KH / 2 || (w->pad_x >= 0 ? w->pad_x : w->pad) != w->KW / 2) return false;
            PfShape ps;
            ps.Cin = w->Cin; ps.Cout = w->Cout; ps.C0 = C0; ps.KH = w->KH; ps.KW = w->KW; ps.nz = 1; ps.Ho = H; ps.Wo = W; ps.B = pb();
            ps.need_all_cout = k3;
            PfPlan plan;
            if (!pf_make_plan(ps, &plan)) return false;
        }
        return true;
    }

    // Would a Downsample convolution (3x3 / stride 2 / pad 1) run on conv_pf_kernel<..., STR = 2> given a PF input of H x W?
    bool pf_s2_would_plan(const ConvW &w, int H, int W) {
        if (!pf_on() || !w.wsh || w.stride != 2 || w.transposed || w.KH != 3 || w.KW != 3 || (H & 1) || (W & 1) || (w.Cin % 16)) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != 1 || (w.pad_x >= 0 ? w.pad_x : w.pad) != 1) return false;
        PfShape ps;
        ps.Cin = w.Cin; ps.Cout = w.Cout; ps.KH = 3; ps.KW = 3; ps.Ho = H / 2; ps.Wo = W / 2; ps.B = pb(); ps.stride = 2;
        PfPlan plan;
        return pf_make_plan(ps, &plan);
    }

    // Would an Upsample (ConvTranspose2d 4x4 / stride 2 / pad 1) run on conv_pf_kernel<..., TZ = 4> given a PF input of H x W?
    bool pf_tz_would_plan(const ConvW &w, int H, int W) {
        if (!pf_on() || !w.wsh || !w.transposed || w.tk != 4 || w.KH != 2 || w.KW != 2 || (w.Cin % 16)) return false;
        PfShape ps;
        ps.Cin = w.Cin; ps.Cout = w.Cout; ps.KH = 2; ps.KW = 2; ps.nz = w.nz; ps.Ho = H; ps.Wo = W; ps.B = pb(); ps.tz = 4;
        PfPlan plan;
        return pf_make_plan(ps, &plan);
    }

    // Would the row-folded final convolution (1x7) run on conv_pf_kernel given a PF input of H x W?
    bool pf_17_would_plan(const ConvW &w, int H, int W) {
        if (!pf_on() || !w.wsh || w.KH != 1 || w.KW != 7 || w.stride != 1 || w.transposed || (w.Cin % 16)) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != 0 || (w.pad_x >= 0 ? w.pad_x : w.pad) != 3) return false;
        PfShape ps;
        ps.Cin = w.Cin; ps.Cout = w.Cout; ps.KH = 1; ps.KW = 7; ps.Ho = H; ps.Wo = W; ps.B = pb(); ps.cop = w.COP;
        PfPlan plan;
        return pf_make_plan(ps, &plan);
    }

    // Would a single-source 3x3 / 1x1 layer with fused LayerNorm run on conv_pf_kernel (given a PF input)?
    bool pf_would_plan(const ConvW &w, int H, int W) {
        if (!pf_on() || !w.wsh || w.stride != 1 || w.transposed) return false;
        if (!((w.KH == 3 && w.KW == 3) || (w.KH == 1 && w.KW == 1))) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != w.KH / 2 || (w.pad_x >= 0 ? w.pad_x : w.pad) != w.KW / 2) return false;
        PfShape ps;
        ps.Cin = w.Cin; ps.Cout = w.Cout; ps.KH = w.KH; ps.KW = w.KW; ps.Ho = H; ps.Wo = W; ps.B = pb(); ps.need_all_cout = true;
        PfPlan plan;
        return (w.Cin % 16) == 0 && pf_make_plan(ps, &plan);
    }

    // Plans the convolution on conv_pf_kernel when every source has a valid PF twin and the layer is a stride-1
    // k x k / 1x1 / phase-decomposed transposed convolution with "same" geometry.
    bool try_pf(const ConvW &w, const float *s0, int C0, const float *s1, int H, int W, float *out, long long out_bs,
                const ConvOpts &o, bool need_all, int prof, const ConvShape &s) {
        if (!pf_on() || !w.wsh || (w.stride != 1 && w.stride != 2) || o.pre_mean || o.w_bs || o.wsp_bs || o.max_ksplit > 1) return false;
        // stride 2: the 3x3 / pad 1 Downsample form on even extents, single source (conv_pf_kernel, STR = 2)
        if (w.stride == 2 && (w.transposed || w.KH != 3 || w.KW != 3 || s1 || (H & 1) || (W & 1) || o.pre_add || o.res3_w)) return false;
        if (s1 && dev_env("CDC_TEST_JOIN_MISS")) return false;   // test hook: the joins miss the plane kernels AFTER their halves were made planes-only (ensure_f32)
        PfTwin *t0 = twin(s0), *t1 = s1 ? twin(s1) : nullptr;
        if (!t0 || !t0->valid || (s1 && (!t1 || !t1->valid))) return false;
        if (t0->H != H || t0->W != W || (t1 && (t1->H != H || t1->W != W))) return false;
        if (s1 ? (t0->C != C0 || t0->C + t1->C != w.Cin) : t0->C != w.Cin) return false;
        // transposed 4x4: the four 2x2 phases fused in one workgroup (TZ = 4)
        const bool k3 = w.KH == 3 && w.KW == 3 && !w.transposed, k1 = w.KH == 1 && w.KW == 1, k2 = w.transposed && w.tk == 4 && !s1;
        const bool k17 = w.KH == 1 && w.KW == 7 && !w.transposed && w.stride == 1 && !s1 && !o.ln_g && !o.emit_pf;   // row-folded final convolution
        if (!(k3 || k1 || k2 || k17)) return false;
        if (!w.transposed && ((w.pad_y >= 0 ? w.pad_y : w.pad) != w.KH / 2 || (w.pad_x >= 0 ? w.pad_x : w.pad) != w.KW / 2)) return false;
        PfShape ps;
        ps.Cin = w.Cin; ps.Cout = w.Cout; ps.C0 = s1 ? C0 : 0; ps.KH = w.KH; ps.KW = w.KW; ps.nz = w.nz;
        ps.Ho = s.Ho; ps.Wo = s.Wo; ps.B = pb(); ps.need_all_cout = need_all; ps.stride = w.stride;
        ps.tz = k2 ? 4 : 1;
        ps.cop = w.COP;
        PfPlan plan;
        if (!pf_make_plan(ps, &plan)) return false;
        Op op;
        op.kind = Op::CONVPF; op.prof = prof; op.pfplan = plan; op.nz = w.nz;
        PfArgs &a = op.pf;
        memset(&a, 0, sizeof a);
        a.src0 = t0->p; a.src0_bs = t0->bs();
        if (t1) { a.src1 = t1->p; a.src1_bs = t1->bs(); }
        a.C0 = s1 ? C0 : w.Cin; a.Cin = w.Cin; a.H = H; a.W = W;
        a.w = w.wsh; a.w_zs = w.wsp_zs / 8;            // planes of 8 halfs = one unit
        a.KH = w.KH; a.KW = w.KW; a.nz = w.nz; a.stride = w.stride; a.tz = ps.tz;
        a.nchunk = w.Cin / 16; a.COP = w.COP; a.Cout = w.Cout;
        a.acc_scale = w.wscale_inv;
        a.out = o.no_f32 ? nullptr : out; a.out_bs = out_bs;
        const int Ht = w.transposed ? 2 * H : s.Ho, Wt = w.transposed ? 2 * W : s.Wo;
        if (w.transposed) {
            for (int z = 0; z < 4; ++z) {
                const int py = z >> 1, px = z & 1;
                a.pad_y[z] = w.tk == 5 ? 1 : 1 - py; a.pad_x[z] = w.tk == 5 ? 1 : 1 - px;
                a.out_zoff[z] = py * 2 * W + px;
            }
            a.out_cs = (long long)4 * H * W; a.out_ys = 4 * W; a.out_xs = 2;
        } else {
            a.pad_y[0] = w.KH / 2; a.pad_x[0] = w.KW / 2;
            a.out_cs = (long long)s.Ho * s.Wo; a.out_ys = s.Wo; a.out_xs = 1;
        }
        a.Ho = s.Ho; a.Wo = s.Wo;
        PfTwin *to = o.emit_pf ? twin(out) : nullptr;
        if (to && to->C == w.Cout && to->H == Ht && to->W == Wt && out_bs == (long long)w.Cout * Ht * Wt) {
            a.out_pf = to->p; a.pf_bs = to->bs(); a.pf_ps = to->ps();
            if (w.transposed) {
                a.pf_ys = 2 * (Wt + 2); a.pf_xs = 2;
                for (int z = 0; z < 4; ++z) a.pf_zoff[z] = ((z >> 1) + 1) * (Wt + 2) + (z & 1) + 1;
            } else {
                a.pf_ys = Wt + 2; a.pf_xs = 1; a.pf_zoff[0] = (Wt + 2) + 1;
            }
            to->valid = true;
        } else if (o.no_f32) {
            a.out = out;                                // nothing else would hold the result
        }
        a.bias = o.no_bias ? nullptr : w.bias;
        a.pre_add = o.pre_add;
        a.ep_g = o.ln_g; a.ep_b = o.ln_b; a.eps = 1e-5f; a.relu = o.relu; a.relu_slope = o.relu_slope;
        a.shift = o.shift; a.shift_bs = o.shift_bs >= 0 ? o.shift_bs : h->shift_bs;
        a.resid = o.resid; a.resid_bs = o.resid_bs; a.resid_cs = o.resid_cs;
        if (o.resid1) {         // residual over a channel concatenation: every wave's channel part inside one source (64 / 96 / 128-channel parts)
            if (!o.resid || o.resid_c0 <= 0 || o.resid_c0 >= w.Cout || (o.resid_c0 % 64) || (o.resid_c0 % (plan.MB * 32))) return false;
            a.resid1 = o.resid1; a.resid1_bs = o.resid1_bs; a.resid_c0 = o.resid_c0;
        }
        if (PfTwin *tr = o.resid ? twin(o.resid) : nullptr)
            if (tr->only) {     // the residual exists as planes only (a ResnetBlock-chain output): read it from there
                if (!tr->valid || tr->C != w.Cout || tr->H != s.Ho || tr->W != s.Wo || w.transposed || w.stride != 1 ||
                    o.resid_bs != (long long)w.Cout * s.Ho * s.Wo || o.resid_cs != (long long)s.Ho * s.Wo) {
                    if (!ensure_f32(o.resid, o.resid_bs)) {
                        if (!rc) rc = fail(h, CDC_ERR_UNSUPPORTED, "planes-only residual of a shape the plane-operand kernels do not read");
                        return true;
                    }
                } else {
                    a.resid = nullptr;
                    a.resid_pf = tr->p; a.rpf_bs = tr->bs(); a.rpf_ps = tr->ps(); a.rpf_ys = s.Wo + 2; a.rpf_zoff = (s.Wo + 2) + 1;
                }
            }
        a.stat_mean = o.stat_mean; a.stat_rstd = o.stat_rstd;
        a.res3_w = o.res3_w; a.res3_x = o.res3_x; a.res3_bs = o.res3_bs;
        a.fault = fault_flag();
        // large 3x3 layers: the persistent ping-ponged kernel.  Its chunk summation order depends on the launch geometry
        // (batch size, CU count), so a program planned "as for one image" (planB: the entropy coder's bit-exactness
        // contract between batch sizes) never uses it.
        if (planB == 0) pf3_make_plan(a, B, w.nz, &op.pfplan);
        // (hoisted partial sums in accumulator order, pre_add_c4: measured only on the first layer's form, try_pf_uf -- 0.364 -> 0.354 ms;
        //  the 192 / 256-channel layers did not move, their 1x1 res_convs lost 5 %)
        if (getenv("CDC_DEBUG_PLAN"))
            fprintf(stderr, "[plan] conv %dx%d %d->%d out %dx%d on conv_pf%s_kernel (epv %d, %d workgroups x %d tiles per group)\n", w.KH, w.KW, w.Cin, w.Cout,
                    s.Ho, s.Wo, op.pfplan.pf3_epv ? "3" : "", op.pfplan.pf3_epv, op.pfplan.pf3_G, op.pfplan.pf3_iters);
        const double px = (double)B * s.Ho * s.Wo * w.nz;
        op.flops = 2.0 * px * w.Cout * w.Cin * w.KH * w.KW;
        op.bytes = 4.0 * ((double)B * w.Cin * H * W + px * w.Cout);
        last_ksplit = 1;
        last_pf_only = a.out == nullptr;
        emit(op);
        return true;
    }

    // The first layer (7x1 over the kx-unfolded 3-channel image, ConvOpts::uf_c) on conv_pf_kernel's UF form: the kernel builds its patch
    // buffers from the image itself, everything else (weight ring, tap loop, epilogue with hoisted partial sums, LayerNorm, planes out)
    // is the plane-operand kernel.
    bool try_pf_uf(const ConvW &w, const float *s0, long long bs0, int H, int W, float *out, long long out_bs, const ConvOpts &o,
                   bool need_all, int prof, const ConvShape &s) {
        if (!pf_on() || !w.wsh || o.uf_c != 3 || o.uf_pad != 3 || w.KH != 7 || w.KW != 1 || w.stride != 1 || w.transposed || w.nz != 1) return false;
        if (o.pre_mean || o.w_bs || o.wsp_bs || o.max_ksplit > 1 || o.resid || o.res3_w || o.stat_mean) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != 3 || (w.pad_x >= 0 ? w.pad_x : w.pad) != 0 || s.Ho != H || s.Wo != W) return false;
        PfShape ps;
        ps.Cin = w.Cin; ps.Cout = w.Cout; ps.KH = 7; ps.KW = 1; ps.Ho = H; ps.Wo = W; ps.B = pb(); ps.need_all_cout = need_all;
        ps.uf = 3; ps.cop = w.COP;
        PfPlan plan;
        if (!pf_make_plan(ps, &plan)) return false;
        Op op;
        op.kind = Op::CONVPF; op.prof = prof; op.pfplan = plan; op.nz = 1;
        PfArgs &a = op.pf;
        memset(&a, 0, sizeof a);
        a.x0 = s0; a.x0_bs = bs0;
        a.C0 = a.Cin = 32; a.H = H; a.W = W;
        a.w = w.wsh; a.w_zs = w.wsp_zs / 8;
        a.KH = 7; a.KW = 1; a.nz = 1;
        a.nchunk = 2; a.COP = w.COP; a.Cout = w.Cout;
        a.acc_scale = w.wscale_inv;
        a.pad_y[0] = 3; a.pad_x[0] = 0;
        a.out = o.no_f32 ? nullptr : out; a.out_bs = out_bs;
        a.out_cs = (long long)H * W; a.out_ys = W; a.out_xs = 1;
        a.Ho = H; a.Wo = W;
        PfTwin *to = o.emit_pf ? twin(out) : nullptr;
        if (to && to->C == w.Cout && to->H == H && to->W == W && out_bs == (long long)w.Cout * H * W) {
            a.out_pf = to->p; a.pf_bs = to->bs(); a.pf_ps = to->ps();
            a.pf_ys = W + 2; a.pf_xs = 1; a.pf_zoff[0] = (W + 2) + 1;
            to->valid = true;
        } else if (o.no_f32) {
            a.out = out;
        }
        a.bias = o.no_bias ? nullptr : w.bias;
        a.pre_add = o.pre_add;
        if (a.pre_add && cur != &h->pre_ops)
            if (const float *q = pre_add_c4(a.pre_add, w.Cout, H, W, out_bs)) { a.pre_add = q; a.pre_c4 = 1; }
        a.ep_g = o.ln_g; a.ep_b = o.ln_b; a.eps = 1e-5f; a.relu = o.relu; a.relu_slope = o.relu_slope;
        a.shift = o.shift; a.shift_bs = o.shift_bs >= 0 ? o.shift_bs : h->shift_bs;
        a.fault = fault_flag();
        const double px = (double)B * H * W;
        op.flops = 2.0 * px * w.Cout * w.Cin * 7;
        op.bytes = 4.0 * ((double)B * 3 * H * W + px * w.Cout);
        last_ksplit = 1;
        last_pf_only = a.out == nullptr;
        emit(op);
        return true;
    }

    // Pointwise convolutions at the >= 32-pixel-wide levels on conv_pw_kernel (fp16 arithmetic): activations staged
    // per wave straight from the fp32 sources, PreNorm folded as (x - mean) on load / rstd in the epilogue.
    bool try_pw(const ConvW &w, const float *s0, int C0, long long bs0, const float *s1, long long bs1, int H, int W,
                float *out, long long out_bs, const ConvOpts &o, bool need_all, int prof, const ConvShape &s) {
        if (h->arith != 1 || !w.wsh || w.KH != 1 || w.KW != 1 || w.stride != 1 || w.transposed || w.nz != 1) return false;
        if (dev_env("CDC_NO_PW")) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != 0 || (w.pad_x >= 0 ? w.pad_x : w.pad) != 0) return false;
        if (need_all || o.ln_g || o.stat_mean || o.res3_w || o.pf_only) return false;
        if (o.pre_mean && o.pre_mode != 2) return false;
        if (o.w_bs && !o.wsp_bs) return false;              // per-image weights without planes
        PfShape ps;
        ps.Cin = w.Cin; ps.Cout = w.Cout; ps.C0 = s1 ? C0 : 0; ps.KH = 1; ps.KW = 1; ps.nz = 1;
        ps.Ho = s.Ho; ps.Wo = s.Wo; ps.B = pb(); ps.need_all_cout = false;
        PfPlan plan;
        if (!pw_make_plan(ps, &plan)) return false;
        if (plan.lin && (o.shift || o.wsp_bs || (long long)w.Cout * s.Ho * s.Wo != out_bs)) return false;   // tiles span images
        Op op;
        if (getenv("CDC_DEBUG_PLAN"))
            fprintf(stderr, "[plan] conv1x1 PW Cin=%4d Cout=%4d out=%3dx%-3d %s%s| MB=%d NPW=%d WM=%d WP=%d groups=%d R=%d wgs=%d lds=%zu\n", w.Cin, w.Cout,
                    s.Ho, s.Wo, o.pre_mean ? "pre2 " : "", o.wsp_bs ? "per-image " : "", plan.MB, plan.NPW, plan.WM, plan.WP, plan.groups, plan.ring,
                    plan.tiles_x * plan.tiles_y * B * plan.groups, plan.lds_bytes);
        op.kind = Op::CONVPF; op.prof = prof; op.pfplan = plan; op.nz = 1; op.pw = true;
        PfArgs &a = op.pf;
        memset(&a, 0, sizeof a);
        a.x0 = s0; a.x0_bs = bs0; a.x1 = s1; a.x1_bs = bs1;
        a.C0 = s1 ? C0 : w.Cin; a.Cin = w.Cin; a.H = H; a.W = W;
        a.pre_mean = o.pre_mean; a.pre_rstd = o.pre_mean ? o.pre_rstd : nullptr;
        a.w = w.wsh; a.w_bs = o.wsp_bs / 8;             // units of 8 halfs
        a.KH = 1; a.KW = 1; a.nz = 1;
        a.nchunk = w.Cin / 16; a.COP = w.COP; a.Cout = w.Cout;
        a.acc_scale = w.wscale_inv;
        a.out = out; a.out_bs = out_bs;
        a.out_cs = (long long)s.Ho * s.Wo; a.out_ys = s.Wo; a.out_xs = 1;
        a.Ho = s.Ho; a.Wo = s.Wo;
        if (PfTwin *to = (o.emit_pf && !plan.lin) ? twin(out) : nullptr)
            if (to->C == w.Cout && to->H == s.Ho && to->W == s.Wo && out_bs == (long long)w.Cout * s.Ho * s.Wo && (w.Cout % 32) == 0) {
                a.out_pf = to->p; a.pf_bs = to->bs(); a.pf_ps = to->ps();
                a.pf_ys = s.Wo + 2; a.pf_xs = 1; a.pf_zoff[0] = (s.Wo + 2) + 1;
                to->valid = true;
                if (o.no_f32) a.out = nullptr;          // the planes are the only copy (their reader is a plane-operand kernel)
            }
        a.bias = o.no_bias ? nullptr : w.bias;
        a.pre_add = o.pre_add;
        a.relu = o.relu; a.relu_slope = o.relu_slope; a.eps = 1e-5f;
        a.shift = o.shift; a.shift_bs = o.shift_bs >= 0 ? o.shift_bs : h->shift_bs;
        a.resid = o.resid; a.resid_bs = o.resid_bs; a.resid_cs = o.resid_cs;
        a.fault = fault_flag();
        const double px = (double)B * s.Ho * s.Wo;
        op.flops = 2.0 * px * w.Cout * w.Cin;
        op.bytes = 4.0 * ((double)B * w.Cin * H * W + px * w.Cout);
        last_ksplit = 1;
        last_pf_only = a.out == nullptr;
        emit(op);
        return true;
    }

    // 3x3 / stride-1 / pad-1 layer of a few-pixel level on conv_ws_kernel (conv_ws_kernel.h): the RAW result (bias added, no LayerNorm)
    // goes to `raw`; a Block's LayerNorm / ReLU / shift / residual is the in-place pass its caller emits behind it.
    bool ws_would_plan(const ConvW &w, int C0, bool two_src, int H, int W) {
        if (h->arith != 1 || !w.wsh || planB > 0 || w.KH != 3 || w.KW != 3 || w.stride != 1 || w.transposed || w.nz != 1) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != 1 || (w.pad_x >= 0 ? w.pad_x : w.pad) != 1) return false;
        if (w.COP != w.Cout || w.Cin_pad != w.Cin) return false;
        WsPlan plan;
        return ws_make_plan(w.Cin, two_src ? C0 : w.Cin, w.Cout, H, W, pb(), &plan);
    }
    bool try_ws(const ConvW &w, const float *s0, int C0, long long bs0, const float *s1, long long bs1, int H, int W, float *raw,
                long long raw_bs, int prof) {
        if (rc || !ws_would_plan(w, C0, s1 != nullptr, H, W)) return false;
        if (!ensure_f32(s0, bs0) || (s1 && !ensure_f32(s1, bs1))) return false;
        Op op;
        op.kind = Op::CONVWS; op.prof = prof;
        if (!ws_make_plan(w.Cin, s1 ? C0 : w.Cin, w.Cout, H, W, pb(), &op.wsplan)) return false;
        WsArgs &a = op.ws;
        memset(&a, 0, sizeof a);
        a.x0 = s0; a.x0_bs = bs0; a.x1 = s1; a.x1_bs = bs1;
        a.C0 = s1 ? C0 : w.Cin; a.Cin = w.Cin; a.H = H; a.B = B;
        a.w = w.wsh; a.nchunk = w.Cin / 16; a.COP = w.COP; a.Cout = w.Cout; a.acc_scale = w.wscale_inv;
        a.bias = w.bias;
        a.out = raw; a.out_bs = raw_bs;
        a.fault = fault_flag();
        const double px = (double)B * H * W;
        op.flops = 2.0 * px * w.Cout * w.Cin * 9;
        op.bytes = 4.0 * px * (w.Cin + w.Cout);
        if (getenv("CDC_DEBUG_PLAN"))
            fprintf(stderr, "[plan] conv 3x3 %d->%d out %dx%d on conv_ws_kernel: %d tiles of %d pixels x %d groups, %d waves, %zu bytes of LDS\n", w.Cin, w.Cout, H, W,
                    op.wsplan.tiles, op.wsplan.NPB * 32, op.wsplan.groups, op.wsplan.waves, op.wsplan.lds_bytes);
        last_ksplit = 1;
        last_pf_only = false;
        emit(op);
        return true;
    }

    // 1x1 layer of a few-pixel level (maps narrower than 32 pixels) on conv_ws1_kernel (conv_ws1_kernel.h): all of K inside the
    // workgroup -- no partial-sum tensors, no sum pass.  Epilogue: bias, folded PreNorm (mean on load, rstd after), per-image shift,
    // residual; shared or per-image weight planes.
    bool try_ws1(const ConvW &w, const float *s0, int C0, long long bs0, const float *s1, long long bs1, int H, int W, float *out, long long out_bs,
                 const ConvOpts &o, bool need_all, int prof) {
        if (rc || h->arith != 1 || !w.wsh || planB > 0 || W >= 32 || w.KH != 1 || w.KW != 1 || w.stride != 1 || w.transposed || w.nz != 1) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != 0 || (w.pad_x >= 0 ? w.pad_x : w.pad) != 0) return false;
        if (need_all || o.ln_g || o.stat_mean || o.res3_w || o.pf_only || o.pre_add || o.relu || o.resid1 || o.uf_c) return false;
        if (o.pre_mean && o.pre_mode != 2) return false;
        if (o.w_bs && !o.wsp_bs) return false;               // per-image weights without planes
        if (w.COP != w.Cout || w.Cin_pad != w.Cin || out_bs != (long long)w.Cout * H * W) return false;
        if (o.resid && (o.resid_cs != (long long)H * W)) return false;
        // Measured (batch 32, profiles/per_op_r05*.txt): it wins wherever the alternative is a split-K launch + sum pass (the 8x8 level, the
        // per-image attention products everywhere); at 16x16 and batch 32 the wide folded-PreNorm projections (24 - 36 channel groups, each
        // converting the same activations again) and the res_convs are faster on conv_pw_kernel's 64 - 96-channel workgroups.
        const long long blocks = (long long)pb() * (H * W / 32);
        if (blocks > 128 && !o.wsp_bs) return false;
        Op op;
        op.kind = Op::CONVWS1; op.prof = prof;
        if (!ws1_make_plan(w.Cin, s1 ? C0 : w.Cin, w.Cout, H * W, pb(), o.wsp_bs != 0, &op.ws1plan)) return false;
        if (!ensure_f32(s0, bs0) || (s1 && !ensure_f32(s1, bs1)) || (o.resid && !ensure_f32(o.resid, o.resid_bs))) return false;
        Ws1Args &a = op.ws1;
        memset(&a, 0, sizeof a);
        a.x0 = s0; a.x0_bs = bs0; a.x1 = s1; a.x1_bs = bs1;
        a.C0 = s1 ? C0 : w.Cin; a.Cin = w.Cin; a.HW = H * W; a.B = B;
        a.pre_mean = o.pre_mean; a.pre_rstd = o.pre_mean ? o.pre_rstd : nullptr;
        a.w = w.wsh; a.w_bs = o.wsp_bs / 8;                 // units of 8 halfs
        a.nchunk = w.Cin / 16; a.COP = w.COP; a.Cout = w.Cout; a.acc_scale = w.wscale_inv;
        a.bias = o.no_bias ? nullptr : w.bias;
        a.shift = o.shift; a.shift_bs = o.shift_bs >= 0 ? o.shift_bs : h->shift_bs;
        a.resid = o.resid; a.resid_bs = o.resid_bs;
        a.out = out; a.out_bs = out_bs;
        a.fault = fault_flag();
        const double px = (double)B * H * W;
        op.flops = 2.0 * px * w.Cout * w.Cin;
        op.bytes = 4.0 * px * (w.Cin + w.Cout);
        if (getenv("CDC_DEBUG_PLAN"))
            fprintf(stderr, "[plan] conv 1x1 %d->%d out %dx%d on conv_ws1_kernel: %d tiles of %d pixels x %d groups, %d waves%s%s\n", w.Cin, w.Cout, H, W,
                    op.ws1plan.tiles, op.ws1plan.NPB * 32, op.ws1plan.groups, op.ws1plan.waves, o.pre_mean ? ", folded PreNorm" : "", o.wsp_bs ? ", per-image weights" : "");
        last_ksplit = 1;
        last_pf_only = false;
        ws1_h = H;
        emit(op);
        return true;
    }

    // Emits one convolution.  s1 (optional) is the second concat source.  Returns false when
    // `need_all` (fused LN / statistics) cannot be planned; the caller then emits the unfused form.
    bool conv(const ConvW &w, const float *s0, int C0, long long bs0, const float *s1,
              long long bs1, int H, int W, float *out, long long out_bs, const ConvOpts &o,
              bool need_all, int prof) {
        if (rc) return true;
        const int pad_y = w.pad_y >= 0 ? w.pad_y : w.pad, pad_x = w.pad_x >= 0 ? w.pad_x : w.pad;
        if (s1 && (C0 % 4)) {
            // the kernel wants every K-chunk inside one concat source: materialise odd seams
            const long long n0 = (long long)C0 * H * W, n1 = (long long)(w.Cin - C0) * H * W;
            float *cat = dalloc((size_t)B * (n0 + n1));
            copy(s0, bs0, cat, n0 + n1, n0);
            copy(s1, bs1, cat + n0, n0 + n1, n1);
            s0 = cat; bs0 = n0 + n1; s1 = nullptr; bs1 = 0;
        }
        ConvShape s;
        s.Cin = w.Cin; s.Cout = w.Cout; s.KH = w.KH; s.KW = w.KW; s.stride = w.stride;
        s.C0 = s1 ? C0 : 0;
        s.Win = W; s.nz = w.nz;
        s.allow_split = w.wsp != nullptr;
        s.arith = (h->arith == 1 && w.wsh) ? 1 : 0;
        s.per_image_w = o.wsp_bs != 0;
        for (int z = 0; z < 4; ++z) s.pad_x[z] = w.transposed ? (w.tk == 5 ? 1 : 1 - (z & 1)) : pad_x;
        if (w.transposed) { s.Ho = H; s.Wo = W; }
        else {
            s.Ho = (H + 2 * pad_y - w.KH) / w.stride + 1;
            s.Wo = (W + 2 * pad_x - w.KW) / w.stride + 1;
        }
        s.B = pb(); s.need_all_cout = need_all; s.lnmode = o.pre_mean ? o.pre_mode : 0;
        if (!o.uf_c && try_pf(w, s0, C0, s1, H, W, out, out_bs, o, need_all, prof, s)) return true;
        if (o.uf_c && !s1 && try_pf_uf(w, s0, bs0, H, W, out, out_bs, o, need_all, prof, s)) return true;
        if (o.pf_only) return false;
        if (o.resid1 && !rc) { rc = fail(h, CDC_ERR_UNSUPPORTED, "a two-source residual reached a kernel without it"); return true; }
        {   // a planes-only source / residual and a reader of fp32: unpack it once (ensure_f32)
            const std::pair<const float *, long long> rd[3] = {{s0, bs0}, {s1, bs1}, {o.resid, o.resid_bs}};
            for (const auto &q : rd)
                if (!ensure_f32(q.first, q.second) && !rc) {
                    rc = fail(h, CDC_ERR_UNSUPPORTED, "a planes-only tensor reached a kernel that reads fp32 and cannot be unpacked");
                    return true;
                }
        }
        if (!o.uf_c && try_ws1(w, s0, C0, bs0, s1, bs1, H, W, out, out_bs, o, need_all, prof)) return true;
        if (!o.uf_c && try_pw(w, s0, C0, bs0, s1, bs1, H, W, out, out_bs, o, need_all, prof, s)) return true;
        const bool linear_ep = !need_all && !o.ln_g && !o.relu && !o.shift && !o.stat_mean && !o.pre_add && !o.res3_w &&
                               w.nz == 1 && !w.transposed;
        s.max_ksplit = o.max_ksplit > 1 ? o.max_ksplit : (linear_ep ? 4 : 1);
        if (need_all && (w.Cout % 32)) return false;
        ConvPlan plan;
        if (!conv_make_plan(s, &plan)) {
            if (need_all) return false;
            rc = fail(h, CDC_ERR_UNSUPPORTED, "no launch plan for conv Cin=%d Cout=%d k=%dx%d out=%dx%d",
                      w.Cin, w.Cout, w.KH, w.KW, s.Ho, s.Wo);
            return true;
        }
        if (o.uf_c && !(plan.split == 2 && plan.arith == 1 && plan.xu == 1 && plan.lnmode == 0 &&
                        conv_lookup_split2hu(plan.MB, plan.NPW))) return false;                       // only that kernel unfolds on load
        last_ksplit = 1;
        last_pf_only = false;
        if (o.max_ksplit > 1 && plan.split == 2 && !need_all) {
            // few workgroups and a long K loop (low-resolution levels): slice K so that the chip holds
            // >= 4 workgroups per CU; the LayerNorm kernel that follows adds the slices
            const long long wgs = (long long)(plan.ipw > 1 ? ceil_div(pb(), plan.ipw) : plan.tiles_x * plan.tiles_y * pb()) *
                                  plan.groups * w.nz;
            int ks = (int)std::min<long long>(ceil_div(kKsTarget, wgs), std::min(o.max_ksplit, plan.nchunk / 4));
            if (ks > 1) { plan.ksplit = ks; last_ksplit = ks; }
        }
        // Plain (linear) epilogues at the few-workgroup levels -- the attention projections and res_convs
        // at 8x8 / 16x16: slice K as well, slice 0 carries bias + residual, a sum pass follows.
        float *ks_scratch = nullptr;
        const long long dense_bs = (long long)w.Cout * s.Ho * s.Wo;
        if (o.max_ksplit <= 1 && plan.split == 2 && !need_all && !o.ln_g && !o.relu && !o.shift && !o.stat_mean &&
            !o.pre_add && !o.res3_w && w.nz == 1 && !w.transposed) {
            const long long wgs = (long long)(plan.ipw > 1 ? ceil_div(pb(), plan.ipw) : plan.tiles_x * plan.tiles_y * pb()) *
                                  plan.groups;
            const int ks = (int)std::min<long long>(ceil_div(kKsTarget, wgs), std::min(4, plan.nchunk / 4));
            if (ks > 1 && (planB > 0 || (size_t)B * dense_bs * 4 * ks <= (64u << 20))) {
                plan.ksplit = ks;
                ks_scratch = dalloc((size_t)ks * B * dense_bs);
            }
        }
        if (getenv("CDC_DEBUG_PLAN"))
            fprintf(stderr, "[plan] %-10s Cin=%4d Cout=%4d k=%dx%d s=%d in=%3dx%-3d out=%3dx%-3d %s%s%s| MB=%2d NPW=%d "
                    "WN=%d groups=%2d KC=%2d nchunk=%3d tiles=%dx%d wgs=%6d lds=%6zu\n", kProfNames[prof], w.Cin,
                    w.Cout, w.KH, w.KW, w.stride, H, W, s.Ho, s.Wo, need_all ? "LN " : "   ",
                    s.lnmode ? (s.lnmode == 2 ? "pre2 " : "pre1 ") : "     ", cur == &h->pre_ops ? "HOIST " : "", plan.MB, plan.NPW, plan.WN,
                    plan.groups, plan.KC, plan.nchunk, plan.tiles_x, plan.tiles_y,
                    plan.tiles_x * plan.tiles_y * B * plan.groups * w.nz, plan.lds_bytes);
        Op op;
        op.kind = Op::CONV; op.prof = prof; op.plan = plan; op.nz = w.nz;
        ConvArgs &a = op.conv;
        memset(&a, 0, sizeof a);
        a.src0 = s0; a.src1 = s1; a.src0_bs = bs0; a.src1_bs = bs1;
        a.C0 = s1 ? C0 : w.Cin; a.Cin = w.Cin; a.H = H; a.W = W;
        a.ln_mean = o.pre_mean; a.ln_rstd = o.pre_rstd; a.ln_g = o.pre_g; a.ln_b = o.pre_b;
        a.wp = w.wp; a.w_bs = o.w_bs; a.w_zs = w.w_zs;
        a.wsp = w.wsp; a.wsp_zs = w.wsp_zs; a.wsp_bs = o.wsp_bs;
        a.acc_scale = 1.f;
        if (plan.split == 2 && plan.arith == 1) { a.wsp = w.wsh; a.acc_scale = w.wscale_inv; }
        a.KH = w.KH; a.KW = w.KW; a.stride = w.stride;
        a.Cin_pad = w.Cin_pad; a.COP = w.COP; a.Cout = w.Cout;
        a.out = out; a.out_bs = out_bs;
        if (w.transposed) {
            for (int z = 0; z < 4; ++z) {
                const int py = z >> 1, px = z & 1;
                a.pad_y[z] = w.tk == 5 ? 1 : 1 - py; a.pad_x[z] = w.tk == 5 ? 1 : 1 - px;
                a.out_zoff[z] = py * 2 * W + px;
            }
            a.out_cs = (long long)4 * H * W; a.out_ys = 4 * W; a.out_xs = 2;
        } else {
            a.pad_y[0] = pad_y; a.pad_x[0] = pad_x;
            a.out_cs = (long long)s.Ho * s.Wo; a.out_ys = s.Wo; a.out_xs = 1;
        }
        a.Ho = s.Ho; a.Wo = s.Wo;
        a.bias = o.no_bias ? nullptr : w.bias;
        a.pre_add = o.pre_add;
        a.ep_g = o.ln_g; a.ep_b = o.ln_b; a.eps = 1e-5f; a.relu = o.relu; a.relu_slope = o.relu_slope;
        a.shift = o.shift; a.shift_bs = o.shift_bs >= 0 ? o.shift_bs : h->shift_bs;
        a.resid = o.resid; a.resid_bs = o.resid_bs; a.resid_cs = o.resid_cs;
        a.stat_mean = o.stat_mean; a.stat_rstd = o.stat_rstd;
        a.out_ks = (long long)B * out_bs;
        a.res3_w = o.res3_w; a.res3_x = o.res3_x; a.res3_bs = o.res3_bs;
        a.fault = fault_flag();
        if (o.uf_c) {
            a.uf_c = o.uf_c; a.uf_pad = o.uf_pad;
            a.uf_magic = o.uf_c > 1 ? (unsigned)(((1ull << 32) + o.uf_c - 1) / o.uf_c) : 0u;
        }
        if (o.emit_pf && !ks_scratch && plan.ksplit <= 1 && (w.Cout % 32) == 0)
            if (PfTwin *to = twin(out)) {
                const int Ht = w.transposed ? 2 * H : s.Ho, Wt = w.transposed ? 2 * W : s.Wo;
                if (to->C == w.Cout && to->H == Ht && to->W == Wt && out_bs == (long long)w.Cout * Ht * Wt) {
                    a.out_pf = to->p; a.pf_bs = to->bs(); a.pf_ps = to->ps();
                    if (w.transposed) {
                        a.pf_ys = 2 * (Wt + 2); a.pf_xs = 2;
                        for (int z = 0; z < 4; ++z) a.pf_zoff[z] = ((z >> 1) + 1) * (Wt + 2) + (z & 1) + 1;
                    } else {
                        a.pf_ys = Wt + 2; a.pf_xs = 1; a.pf_zoff[0] = (Wt + 2) + 1;
                    }
                    a.pf_only = o.no_f32 ? 1 : 0;
                    last_pf_only = a.pf_only != 0;
                    to->valid = true;
                }
            }
        const double px = (double)B * s.Ho * s.Wo * w.nz;
        op.flops = 2.0 * px * w.Cout * w.Cin * w.KH * w.KW;
        op.bytes = 4.0 * ((double)B * w.Cin * H * W + px * w.Cout);
        if (ks_scratch) {
            a.out = ks_scratch; a.out_bs = dense_bs; a.out_ks = (long long)B * dense_bs;
        }
        emit(op);
        if (ks_scratch) {
            Op c; c.kind = Op::COPY; c.prof = prof;
            c.cp = {ks_scratch, dense_bs, out, out_bs, dense_bs};
            c.cp_parts = plan.ksplit; c.cp_part_stride = (long long)B * dense_bs;
            c.bytes = 4.0 * B * dense_bs * (plan.ksplit + 1);
            emit(c);
            // split-K epilogues cannot emit planes (partial sums): pack the reduced tensor where a reader wants them
            if (o.emit_pf && !w.transposed)
                if (PfTwin *to = twin(out))
                    if (to->C == w.Cout && to->H == s.Ho && to->W == s.Wo && out_bs == (long long)w.Cout * s.Ho * s.Wo) pack(out, out_bs);
        }
        return true;
    }

    void ln(const float *in, float *out, int C, int HW, const float *g, const float *b, int relu,
            const float *shift, const float *resid, float *sm, float *sr, int nparts = 1) {
        if (rc) return;
        Op op;
        op.kind = Op::LN; op.prof = PC_LN;
        LnArgs &a = op.ln;
        a.nparts = nparts; a.part_stride = (long long)B * C * HW;
        a.in = in; a.out = out; a.C = C; a.HW = HW; a.g = g; a.b = b; a.eps = 1e-5f; a.relu = relu;
        a.shift = shift; a.shift_bs = h->shift_bs; a.resid = resid; a.stat_mean = sm; a.stat_rstd = sr;
        a.fault = fault_flag();
        op.bytes = 4.0 * B * C * HW * (out ? 2 : 1);
        emit(op);
    }

    void copy(const float *src, long long src_bs, float *dst, long long dst_bs, long long n) {
        Op c; c.kind = Op::COPY; c.prof = PC_SMALL;
        c.cp = {src, src_bs, dst, dst_bs, n};
        c.bytes = 8.0 * B * n;
        emit(c);
    }

    // Fused LayerNorm epilogue needs every output channel in one workgroup; at few-pixel levels that
    // leaves most CUs idle, so split channels over workgroups and run the standalone LN instead.
    bool prefer_fused(const ConvW &w, int H, int W) {
        if (w.Cout % 32 || w.Cout > 384) return false;
        // the split-bf16 kernels hold at most 6 channel blocks per workgroup: wider layers run them over
        // channel groups (2x the matrix rate) and normalise in a separate pass
        // (round 2: eight blocks = 256 channels with NPW = 1, two workgroups per CU)
        if (w.wsp && w.Cout > 256 && (W & 3) == 0 && !dev_env("CDC_NO_SPLIT")) return false;
        ConvShape s;
        s.Cin = w.Cin; s.Cout = w.Cout; s.KH = w.KH; s.KW = w.KW; s.stride = w.stride;
        s.Ho = H; s.Wo = W; s.B = pb(); s.lnmode = 0;
        s.Win = W; s.pad_x[0] = w.pad;
        ConvPlan pf, pu;
        s.need_all_cout = true;
        if (!conv_make_plan(s, &pf)) return false;
        s.need_all_cout = false;
        if (!conv_make_plan(s, &pu)) return true;
        const double wf = (double)pf.tiles_x * pf.tiles_y * B * pf.WN;
        const double wu = (double)pu.tiles_x * pu.tiles_y * B * pu.groups * pu.WN;
        static const double thr = 512;
        if (wf >= thr) return true;              // >= half of the chip's 1024 SIMDs busy
        return wu < 1.5 * wf;
    }

    // conv -> channel LN -> ReLU (+shift) (+resid) (+stats), fused when possible (Block.forward,
    // network_components.py:83-91, plus the adds of ResnetBlock.forward :107-114)
    // uf_c > 0 (unfold on load, ConvArgs::uf_c): s0 is the uf_c-channel image and w the KH x 1 layer over its kx-unfolded channels; only
    // the fused register-staged plan can do that -- returns false, with nothing emitted, when it is not available.
    bool block(const ConvW &w, const float *s0, int C0, long long bs0, const float *s1, long long bs1,
               int H, int W, Act out, const float *g, const float *b, const float *shift,
               const float *pre_add, const float *resid, long long resid_bs, float *sm, float *sr,
               int prof, bool pf_only_out = false, int uf_c = 0, int uf_pad = 0, const float *resid1 = nullptr, long long resid1_bs = 0,
               int resid_c0 = 0) {
        ConvOpts o;
        o.resid1 = resid1; o.resid1_bs = resid1_bs; o.resid_c0 = resid_c0;
        o.uf_c = uf_c; o.uf_pad = uf_pad;
        o.no_f32 = pf_only_out;
        o.ln_g = g; o.ln_b = b; o.relu = 1; o.shift = shift; o.pre_add = pre_add;
        o.no_bias = pre_add != nullptr;          // the hoisted partial already carries the bias
        o.resid = resid; o.resid_bs = resid_bs; o.resid_cs = (long long)H * W;
        o.stat_mean = sm; o.stat_rstd = sr;
        o.res3_w = next_res3_w; o.res3_x = next_res3_x; o.res3_bs = next_res3_bs;
        const bool want_res3 = next_res3_w != nullptr;
        next_res3_w = next_res3_x = nullptr;
        o.emit_pf = true;
        {   // pre-split operands first: there the fused LayerNorm reduces across waves (up to 256 channels)
            ConvOpts op = o;
            op.pf_only = true;
            if (conv(w, s0, C0, bs0, s1, bs1, H, W, out.p, out.bs(), op, true, prof)) return true;
        }
        // few-pixel levels: the weight-stationary kernel (all of K in one workgroup, no partial-sum tensors) + an in-place LayerNorm pass
        if (!pre_add && !want_res3 && !uf_c && !resid1 && out.bs() == (long long)w.Cout * H * W &&
            try_ws(w, s0, C0, bs0, s1, bs1, H, W, out.p, out.bs(), prof)) {
            ln(out.p, out.p, w.Cout, H * W, g, b, 1, shift, resid, sm, sr);
            return true;
        }
        if (prefer_fused(w, H, W) && conv(w, s0, C0, bs0, s1, bs1, H, W, out.p, out.bs(), o, true, prof))
            return true;
        if (uf_c) return false;
        if (want_res3) { rc = fail(h, CDC_ERR_UNSUPPORTED, "epilogue res_conv needs the fused LayerNorm plan"); return true; }
        ConvOpts u;
        u.pre_add = pre_add; u.no_bias = o.no_bias;
        const size_t plane_f = (size_t)B * w.Cout * H * W;
        if (!pre_add && w.wsp && w.Cout <= 8 * 48 && plane_f * 4 * 4 <= (160u << 20)) {
            // low-resolution levels: split-K partial sums into scratch, summed by the LayerNorm kernel
            // small batches: up to six slices (the 8 x 8 level: 24 chunks -> 6 slices of 4; measured at batch 1: its 3 x 3 layers
            // 0.53 -> 0.46 ms per iteration, LayerNorm passes unchanged with ln_kernel_vec<2, 6>; whole model -1.4 % at batch 1 - 4,
            // -0.4 % at 8, +0.5 % at 16, +2.2 % at 32: larger batches fill the chip with four)
            const int kmax = (pb() <= 8 ? 6 : 4);
            float *part = dalloc(plane_f * kmax);
            u.max_ksplit = kmax;
            conv(w, s0, C0, bs0, s1, bs1, H, W, part, out.bs(), u, false, prof);
            ln(part, out.p, w.Cout, H * W, g, b, 1, shift, resid, sm, sr, last_ksplit);
            return true;
        }
        conv(w, s0, C0, bs0, s1, bs1, H, W, out.p, out.bs(), u, false, prof);
        ln(out.p, out.p, w.Cout, H * W, g, b, 1, shift, resid, sm, sr);
        return true;
    }

    // ResnetBlock.forward (network_components.py:107-114).  a1 = second concat source.  If
    // `a1_is_context` the block was packed with split weights: the context halves are evaluated into
    // h->pre_ops (once per decode) and enter the per-step convolutions as `pre_add`.
    // Would a ResnetBlock read its input x ONLY as planes -- block1 on a plane-operand kernel, identity residual read from planes in
    // block2's epilogue (round 4: PfArgs::resid_pf)?  Then the block that produces x writes no fp32 copy (out_planes_only below).
    bool rb_reads_planes_only(const ResBlockW &rb, int C, int H, int W) {
        if (!pf_on() || rb.has_res || rb.hoist_cx || rb.cin != C || rb.cout != C || dev_env("CDC_NO_RESID_PF")) return false;
        return pf_would_plan(rb.c1, H, W) && pf_would_plan(rb.c2, H, W);
    }

    Act resblock(const ResBlockW &rb, Act a0, const Act *a1, bool a1_is_context, float *sm, float *sr,
                 Site out_site = SITE_ALWAYS, bool out_planes_only = false) {
        if (rc) return Act();
        const int H = a0.H, W = a0.W, HW = H * W;
        const int prof1 = rb.k == 7 ? PC_CONV7 : PC_CONV3;
        const float *shift = rb.has_mlp ? h->shift + rb.shift_off : nullptr;   // Compressor blocks: no time embedding
        Act h1 = new_act(rb.cout, H, W), out = new_act(rb.cout, H, W, true, out_site);
        if (pf_mode() >= 2 && pf_would_plan(rb.c2, H, W)) add_twin(h1.p, rb.cout, H, W);
        // h1 feeds block2 only: when block2 runs on the pre-split operand kernel the fp32 copy is never read
        const bool h1_pf_only = twin(h1.p) && pf_would_plan(rb.c2, H, W);
        if (a1 && a1_is_context && rb.hoist_cx == a0.C) {
            // identity residual over cat[x, context] (downs.1.0): read from its two sources in block2's epilogue where that runs on a
            // plane-operand kernel (PfArgs::resid1) instead of materialising the concatenation every iteration
            const bool res2 = !rb.has_res && (a0.C % 64) == 0 && a0.C + a1->C == rb.cout && pf_would_plan(rb.c2, H, W) && !dev_env("CDC_NO_RESID2");
            std::vector<Op> *saved = cur;
            Act p1 = new_act(rb.cout, H, W, false);
            cur = &h->pre_ops;
            conv(rb.c1c, a1->p, a1->C, a1->bs(), nullptr, 0, H, W, p1.p, p1.bs(), ConvOpts(), false, prof1);
            const float *res = nullptr;
            long long res_bs = 0;
            Act pr, cat;
            if (rb.has_res) {
                pr = new_act(rb.cout, H, W, false);
                conv(rb.cresc, a1->p, a1->C, a1->bs(), nullptr, 0, H, W, pr.p, pr.bs(), ConvOpts(), false,
                     PC_CONV1);
            } else if (!res2) {
                // identity residual over the concatenation (downs.1.0): context half copied once
                cat = new_act(a0.C + a1->C, H, W, false);
                copy(a1->p, a1->bs(), cat.p + a0.bs(), cat.bs(), a1->bs());
            }
            cur = saved;
            // first 7x7 layer = 7x1 convolution over the kx-unfolded image: the unfolding happens while the patch is loaded
            // (round 4); the explicit unfold pass + its 21-channel tensor remain the fall-back
            if (rb.has_unfold && (W & 3) == 0 &&
                block(rb.c1u, a0.p, a0.C * rb.k, a0.bs(), nullptr, 0, H, W, h1, rb.g1, rb.b1, shift, p1.p, nullptr, 0,
                      nullptr, nullptr, prof1, h1_pf_only, a0.C, rb.k / 2)) {
            } else if (rb.has_unfold && (W & 3) == 0) {
                Act u = new_act(a0.C * rb.k, H, W, false);
                Op uo; uo.kind = Op::UNFOLD; uo.prof = prof1;
                uo.uf = {a0.p, a0.bs(), u.p, u.bs(), a0.C, rb.k, rb.k / 2, H, W};
                uo.bytes = 4.0 * B * (a0.C + u.C) * HW;
                emit(uo);
                block(rb.c1u, u.p, u.C, u.bs(), nullptr, 0, H, W, h1, rb.g1, rb.b1, shift, p1.p, nullptr, 0,
                      nullptr, nullptr, prof1, h1_pf_only);
            } else
            block(rb.c1x, a0.p, a0.C, a0.bs(), nullptr, 0, H, W, h1, rb.g1, rb.b1, shift, p1.p, nullptr, 0,
                  nullptr, nullptr, prof1, h1_pf_only);
            if (rb.has_res && a0.C == 3 && rb.cresx.COP == round_up(rb.cout, 32) && prefer_fused(rb.c2, H, W)) {
                // res_conv over the 3 image channels rides in block2's epilogue; its context half (with
                // the bias) is the hoisted tensor
                res = pr.p; res_bs = pr.bs();
                next_res3_w = rb.cresx.wp; next_res3_x = a0.p; next_res3_bs = a0.bs();
            } else if (rb.has_res) {
                Act r = new_act(rb.cout, H, W, false);
                ConvOpts orr; orr.pre_add = pr.p; orr.no_bias = true;
                conv(rb.cresx, a0.p, a0.C, a0.bs(), nullptr, 0, H, W, r.p, r.bs(), orr, false, PC_CONV1);
                res = r.p; res_bs = r.bs();
            } else if (res2) {
                res = a0.p; res_bs = a0.bs();
            } else {
                copy(a0.p, a0.bs(), cat.p, cat.bs(), a0.bs());
                res = cat.p; res_bs = cat.bs();
            }
            block(rb.c2, h1.p, rb.cout, h1.bs(), nullptr, 0, H, W, out, rb.g2, rb.b2, nullptr, nullptr, res,
                  res_bs, sm, sr, PC_CONV3, out_planes_only && twin(out.p), 0, 0, res2 ? a1->p : nullptr, res2 ? a1->bs() : 0, res2 ? a0.C : 0);
            mark_planes_only(out);
            return out;
        }
        const float *s0 = a0.p, *s1 = a1 ? a1->p : nullptr;
        int C0 = a0.C;
        long long bs0 = a0.bs(), bs1 = a1 ? a1->bs() : 0;
        if (!rb.has_res && a1) {
            Act cat = new_act(a0.C + a1->C, H, W, false);
            copy(a0.p, a0.bs(), cat.p, cat.bs(), a0.bs());
            copy(a1->p, a1->bs(), cat.p + a0.bs(), cat.bs(), a1->bs());
            s0 = cat.p; s1 = nullptr; C0 = cat.C; bs0 = cat.bs(); bs1 = 0;
        }
        block(rb.c1, s0, C0, bs0, s1, bs1, H, W, h1, rb.g1, rb.b1, shift, nullptr, nullptr, 0, nullptr,
              nullptr, prof1, h1_pf_only);
        const float *res = s0;
        long long res_bs = bs0;
        if (rb.has_res) {
            Act r = new_act(rb.cout, H, W, false);
            conv(rb.cres, s0, C0, bs0, s1, bs1, H, W, r.p, r.bs(), ConvOpts(), false, PC_CONV1);
            res = r.p; res_bs = r.bs();
        }
        block(rb.c2, h1.p, rb.cout, h1.bs(), nullptr, 0, H, W, out, rb.g2, rb.b2, nullptr, nullptr, res,
              res_bs, sm, sr, PC_CONV3, out_planes_only && twin(out.p));
        mark_planes_only(out);
        return out;
    }
    // after the block() call that produced `a`: if it wrote planes only, say so on the twin (its readers must take planes) and on the Act
    void mark_planes_only(Act &a) {
        if (!last_pf_only) return;
        PfTwin *t = twin(a.p);
        t->only = true;
        a.pf = t->p; a.pf_bs = t->bs();
    }

    // Residual(PreNorm(LinearAttention)) (network_components.py:10-16,69-77,117-139)
    // out_planes_only: the output's only reader takes planes (the level-0 Downsample) -- no fp32 copy is written where the folded
    // output runs on the pointwise kernel; Act::pf of the result says whether that happened
    Act attention(const AttnW &at, Act x, float *sm, float *sr, Site out_site = SITE_JOIN, bool out_planes_only = false) {
        if (rc) return Act();
        const int C = at.C, H = x.H, W = x.W, N = H * W;
        const bool fold = at.WoT && N >= 16 * C && !dev_env("CDC_NO_ATTN_FOLD");   // 2 C^3 extra vs 2 C^2 N saved
        // C = 64 levels: k/v projection, row maxima and softmax(k) v^T in ONE pass over x (attn_kernels.hip)
        const bool fused = fold && (C == 64 || C == 128) &&
                           at.kvWt && N % 2048 == 0 && !dev_env("CDC_NO_KVCTX");
        const int kvc = fold ? 2 * C : 3 * C;          // channels of the staged projection
        Act qkv = fused ? Act() : new_act(kvc, H, W, false);
        ConvOpts oq;                                   // LN(x) folded into the projection (LNMODE 2)
        oq.pre_mean = sm; oq.pre_rstd = sr; oq.pre_mode = 2;
        if (!fused)
            conv(fold ? at.kv : at.qkv, x.p, C, x.bs(), nullptr, 0, H, W, qkv.p, qkv.bs(), oq, false, PC_CONV1);
        const float *kp = fused ? nullptr : qkv.p + (size_t)(fold ? 0 : C) * N, *vp = fused ? nullptr : kp + (size_t)C * N;
        float *kmax = dalloc((size_t)B * C);
        const int tiles = ceil_div(C, 64);
        int nsplit = std::max(1, ceil_div(1024, tiles * tiles * B));
        nsplit = std::min(nsplit, std::max(1, N / 64));
        if (fused) {        // >= 2.6 rounds of 3 workgroups per CU (C = 64) / 4 rounds of one (C = 128)
            static const int kv64 = 1024;   // (round 4: 2048 -> 1024: half the partial sums for the fold to add, 13.92 -> 13.89 ms per iteration)
            static const int kv128 = 1024;
            nsplit = std::min(128, C == 64 ? ceil_div(kv64, B) : ceil_div(kv128, B));   // (the fold sums the splits serially)
            while (nsplit > 1 && N % (32 * nsplit)) --nsplit;
        }
        float *kmaxs = fused ? dalloc((size_t)B * nsplit * C) : nullptr;   // per-split row maxima
        float *S = dalloc((size_t)B * nsplit * C * C);
        float *ksum = dalloc((size_t)B * nsplit * C);      // per-split partial sums of exp(k - max)
        const int Cin_pad = round_up(C, 16), COP = round_up(C, 32);
        float *ctxw = dalloc((size_t)B * Cin_pad * COP);
        float *T1 = fold ? dalloc((size_t)B * C * C) : nullptr;
        float *biasB = fold ? dalloc((size_t)B * C) : nullptr;
        if (rc) return Act();
        Op k; k.kind = Op::KSTATS; k.prof = PC_SMALL;
        k.at = {kp, vp, fused ? 0 : qkv.bs(), C, N, kmax, ksum, S, ctxw, nsplit, Cin_pad, COP,
                1.0f / sqrtf((float)C), at.WoT, at.WqT, T1, at.ng, at.uq, at.out.bias, biasB};
        k.bytes = 8.0 * B * C * N;
        if (fused) {
            Op f; f.kind = Op::KVCTX; f.prof = PC_ATTN_CTX;
            f.kvc = {x.p, x.bs(), sm, sr, at.kvWt, at.kvb, at.kvWs, C, N, nsplit, S, ksum, kmaxs};
            if (h->arith == 1 && at.kvWh) { f.kvc.Ws = at.kvWh; f.kvc.f16 = 1; f.kvc.wscale_inv = at.kv_scale_inv; }
            f.flops = 6.0 * B * (double)C * C * N; f.bytes = 4.0 * B * C * N;
            emit(f);
        }
        // few-pixel levels (not folded): kstats + partial context + reduction as ONE launch (round 4; the chain is latency-bound)
        const bool ctx_one = !fused && !fold && (N & 3) == 0 && N <= 1024 && (C % 64) == 0 && Cin_pad == C && COP == C && !dev_env("CDC_NO_CTX_ONE");
        if (!fused && !ctx_one) {
            emit(k);
            Op p = k; p.kind = Op::CTXP; p.prof = PC_ATTN_CTX;
            p.flops = 2.0 * B * (double)C * C * N; p.bytes = 8.0 * B * C * N;
            emit(p);
        }
        // folded output as one streaming pass (lnconv_kernel) where the level is wide enough to be bandwidth-bound
        const bool stream_out = fold && (C == 64 || C == 192) && N >= 4096 && N % 1024 == 0;
        // folded output as a 1x1 split convolution with per-image planes (C % 16 == 0, planes layout = the A-operand
        // layout of conv_split2_kernel with COP == C): replaces the f32-MFMA kernel and, where faster, lnconv_kernel
        const bool no_pic = dev_env("CDC_NO_PERIMAGE_SPLIT") != nullptr;
        // (measured, batch 32: 0.41 -> 0.31 ms at C = 64 / 256^2, 0.30 -> 0.20 at C = 128 / 128^2, 0.18 -> 0.10 at C = 192 / 64^2:
        //  faster than the streaming lnconv_kernel everywhere, which stays as the CDC_NO_PERIMAGE_SPLIT fallback)
        const bool split_out = fold && !no_pic && (C % 32) == 0 && (W & 3) == 0;
        // few-pixel levels (not folded): the per-image product out = ctx^T q as a split convolution as well (planes from
        // ctx_reduce_kernel) -- it was the last user of the fp32 -> bf16x3 register-staged kernel on the decode path
        const bool split_ctxq = !fold && !no_pic && h->arith == 1 && (C % 32) == 0 && (W & 3) == 0 && !dev_env("CDC_NO_CTXQ_SPLIT");
        const bool planes_f16 = (split_out && h->arith == 1) || split_ctxq;
        unsigned short *Ws = (stream_out || split_out || split_ctxq) ? reinterpret_cast<unsigned short *>(dalloc((size_t)B * C * C * 3 / 2 + 8)) : nullptr;
        Op r = k; r.kind = fold ? Op::CTXF : Op::CTXR; r.prof = PC_SMALL;
        r.at_M = kmaxs; r.at_Ws = Ws; r.at_ws_f16 = planes_f16 ? 1 : 0; r.at_Wq = at.Wq;
        r.bytes = 4.0 * B * nsplit * C * C;
        r.flops = fold ? 4.0 * B * (double)C * C * C : 0.0;
        if (ctx_one) {
            r.kind = Op::CTXP; r.prof = PC_ATTN_CTX; r.at_one = 1;
            r.flops = 2.0 * B * (double)C * C * N; r.bytes = 8.0 * B * C * N;
        }
        emit(r);
        ConvW cw;    // per-image weights produced above
        cw.Cin = C; cw.Cout = C; cw.KH = cw.KW = 1; cw.stride = 1; cw.pad = 0;
        cw.Cin_pad = Cin_pad; cw.COP = COP; cw.wp = ctxw; cw.nz = 1; cw.bias = nullptr;
        Act y = new_act(C, H, W, true, out_site);       // a skip tensor is a decoder concat half; an Upsample input has no plane reader
        if (split_out || split_ctxq) {
            cw.wsp = Ws;                                   // (bf16 planes unless planes_f16)
            if (planes_f16) { cw.wsh = Ws; cw.wscale_inv = 1.0f / 256.0f; }
        }
        if (stream_out && !split_out) {
            Op f; f.kind = Op::LNCONV; f.prof = PC_CONV1;
            int ns = std::max(1, ceil_div(2048, B));
            while (ns > 1 && N % (32 * ns)) --ns;
            f.lnc = {x.p, x.bs(), sm, sr, Ws, biasB, y.p, y.bs(), C, N, ns};
            if (PfTwin *ty = twin(y.p))
                if ((W % 32) == 0) { f.lnc.y_pf = ty->p; f.lnc.pf_bs = ty->bs(); f.lnc.pf_ps = ty->ps(); f.lnc.W = W; ty->valid = true; }
            f.flops = 2.0 * B * (double)C * C * N; f.bytes = 12.0 * B * C * N;
            emit(f);
            return y;
        }
        if (fold) {
            // y = M' LN(x) + b_out + x with g folded into M' and (M' b_ln + b_out) as per-image shift
            ConvOpts oy;
            oy.pre_mean = sm; oy.pre_rstd = sr; oy.pre_mode = 2;
            oy.w_bs = (long long)Cin_pad * COP;
            if (split_out) oy.wsp_bs = (long long)(C / 16) * 6 * C * 8;
            oy.shift = biasB; oy.shift_bs = C;
            oy.resid = x.p; oy.resid_bs = x.bs(); oy.resid_cs = N;
            oy.emit_pf = true;
            oy.no_f32 = out_planes_only && split_out && twin(y.p) != nullptr;
            conv(cw, x.p, C, x.bs(), nullptr, 0, H, W, y.p, y.bs(), oy, false, PC_CONV1);
            if (last_pf_only) { PfTwin *ty = twin(y.p); ty->only = true; y.pf = ty->p; y.pf_bs = ty->bs(); }
            return y;
        }
        // out[e,n] = sum_d ctx[d,e] q[d,n]  as a 1x1 convolution with per-image weights (:137)
        Act o = new_act(C, H, W, false);
        ConvOpts oo; oo.w_bs = (long long)Cin_pad * COP; oo.no_bias = true;
        if (split_ctxq) oo.wsp_bs = (long long)(C / 16) * 6 * C * 8;
        conv(cw, qkv.p, C, qkv.bs(), nullptr, 0, H, W, o.p, o.bs(), oo, false, PC_CONV1);
        ConvOpts oy; oy.resid = x.p; oy.resid_bs = x.bs(); oy.resid_cs = N;
        oy.emit_pf = true;
        conv(at.out, o.p, C, o.bs(), nullptr, 0, H, W, y.p, y.bs(), oy, false, PC_CONV1);
        return y;
    }
};

void free_program(cdc_handle *h) {
    free_pool(&h->act_allocs);
    h->ops.clear();
    h->pre_ops.clear();
    h->op_ms.clear(); h->op_n.clear(); h->op_label.clear(); h->op_flops.clear();
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    h->in_ctx.clear();
    h->taps.clear();
    h->dec_outs.clear();
    h->act_bytes = 0;
    h->pB = h->pH = h->pW = 0;
    h->p_batch1_plan = false;
    h->time_steps_B = 0;
}

// Builds the launch program of Unet.forward for batch B at H x W (unet.py:106-135).
int build_program(cdc_handle *h, int B, int H, int W) {
    if (h->pB == B && h->pH == H && h->pW == W) return CDC_OK;
    free_program(h);
    const int n = h->n_res;
    const int down = 1 << (n - 1);
    if (H % down || W % down)
        return fail(h, CDC_ERR_INVALID, "H=%d, W=%d must be multiples of %d (%d downsamples)", H, W,
                    down, n - 1);
    Builder bd{h, B, &h->act_allocs};
    h->in_x = bd.dalloc((size_t)B * h->cfg.channels * H * W);
    h->in_time = bd.dalloc(B);
    h->shift = bd.dalloc((size_t)B * h->shift_bs);
    h->xa = bd.dalloc((size_t)B * h->cfg.channels * H * W);
    h->xb = bd.dalloc((size_t)B * h->cfg.channels * H * W);
    h->noise_buf = bd.dalloc((size_t)B * h->cfg.channels * H * W);
    const int n_ctx = std::min(n - 1, (int)h->context_dims.size() - 1);   // unet.py:65-68,109
    for (int l = 0; l < n_ctx; ++l) h->in_ctx.push_back(bd.new_act(h->context_dims[l], H >> l, W >> l, false));
    if (bd.rc) return bd.rc;

    Op t; t.kind = Op::TEMB; t.prof = PC_SMALL;
    t.temb.time = h->in_time; t.temb.w0 = h->tm_w0; t.temb.b0 = h->tm_b0; t.temb.w2 = h->tm_w2;
    t.temb.b2 = h->tm_b2; t.temb.dim = h->cfg.dim; t.temb.layers = h->d_temb_layers;
    t.temb.n_layers = (int)h->rbs.size(); t.temb.shift = h->shift; t.temb.shift_bs = h->shift_bs;
    bd.emit(t);

    Act x; x.p = h->in_x; x.C = h->cfg.channels; x.H = H; x.W = W;
    std::vector<Act> skips;
    size_t rbi = 0, ati = 0;
    for (int i = 0; i < n; ++i) {
        const int HWl = x.H * x.W;
        float *sm = bd.dalloc((size_t)B * HWl), *sr = bd.dalloc((size_t)B * HWl);
        const bool has_ctx = i < n_ctx;
        const std::string dn = "downs." + std::to_string(i);
        // (its output goes to the second ResnetBlock only: planes INSTEAD of fp32 where that block reads nothing else)
        x = bd.resblock(h->rbs[rbi], x, has_ctx ? &h->in_ctx[i] : nullptr, true, nullptr, nullptr, Builder::SITE_RB_CHAIN,
                        bd.rb_reads_planes_only(h->rbs[rbi + 1], h->rbs[rbi].cout, x.H, x.W));
        ++rbi;
        h->taps[dn + ".0"] = x;
        x = bd.resblock(h->rbs[rbi++], x, nullptr, false, sm, sr);
        h->taps[dn + ".1"] = x;
        // (the level-0 skip is never popped -- unet.py:113 pushes six, :123 pops five: its only reader is the Downsample)
        // Its output goes to the Downsample as planes INSTEAD of fp32 where that convolution runs on the plane-operand kernel.
        const bool l0_planes = i == 0 && n > 1 && bd.pf_s2_would_plan(h->downs[0], x.H, x.W);
        // A skip (levels >= 1) has two readers, the Downsample and the decoder join of its level (ResnetBlock 2 n + 2 + 2 (n - 1 - i): block1
        // and res_conv over cat[upsampled, skip]): planes only where all of them take planes.
        bool skip_planes = false;
        if (i >= 1 && i < n - 1 && !dev_env("CDC_NO_PF_SKIP_PLANES")) {
            const ResBlockW &jrb = h->rbs[(size_t)2 * n + 2 + 2 * (n - 1 - i)];
            skip_planes = bd.pf_s2_would_plan(h->downs[i], x.H, x.W) && bd.join_would_read_planes(jrb, jrb.c1.Cin - x.C, x.H, x.W);
        }
        x = bd.attention(h->attns[ati++], x, sm, sr, i >= 1 ? Builder::SITE_JOIN : (l0_planes ? Builder::SITE_ALWAYS_PLANES : Builder::SITE_NONE),
                         l0_planes || skip_planes);
        h->taps[dn + ".2"] = x;
        skips.push_back(x);
        if (i < n - 1) {
            const ConvW &dw = h->downs[i];
            Act y = bd.new_act(dw.Cout, x.H / 2, x.W / 2, true, Builder::SITE_DOWN);
            // planes of the Downsample output only where its reader -- block1 of the next level's first ResnetBlock -- takes planes (small
            // batches: it does not, and a split-K Downsample would need a pack launch to make them)
            const ResBlockW &nrb = h->rbs[rbi];
            Builder::ConvOpts od; od.emit_pf = bd.pf_would_plan(nrb.hoist_cx ? nrb.c1x : nrb.c1, y.H, y.W);
            bd.conv(dw, x.p, x.C, x.bs(), nullptr, 0, x.H, x.W, y.p, y.bs(), od, false, PC_DOWN);
            if (x.pf && bd.still_planes_only(x.p) && (h->ops.empty() || h->ops.back().kind != Op::CONVPF || h->ops.back().pw))
                return fail(h, CDC_ERR_UNSUPPORTED, "planes-only Downsample input without a plane-operand kernel");
            x = y;
            h->taps[dn + ".3"] = x;
        }
        if (bd.rc) return bd.rc;
    }
    {
        const int HWl = x.H * x.W;
        float *sm = bd.dalloc((size_t)B * HWl), *sr = bd.dalloc((size_t)B * HWl);
        x = bd.resblock(h->rbs[rbi++], x, nullptr, false, sm, sr);          // mid_block1
        h->taps["mid_block1"] = x;
        x = bd.attention(h->attns[ati++], x, sm, sr);                       // mid_attn
        h->taps["mid_attn"] = x;
        x = bd.resblock(h->rbs[rbi++], x, nullptr, false, nullptr, nullptr, Builder::SITE_JOIN); // mid_block2 (decoder concat half)
        h->taps["mid_block2"] = x;
    }
    float *fsm = nullptr, *fsr = nullptr;       // LN statistics of the last Upsample output
    bool final_ln_done = false;                 // ... unless the Upsample epilogue already normalised it
    for (int i = 0; i < n - 1; ++i) {
        Act skip = skips.back();
        skips.pop_back();
        const int HWl = x.H * x.W;
        float *sm = bd.dalloc((size_t)B * HWl), *sr = bd.dalloc((size_t)B * HWl);
        x = bd.resblock(h->rbs[rbi], x, &skip, false, nullptr, nullptr, Builder::SITE_RB_CHAIN,
                        bd.rb_reads_planes_only(h->rbs[rbi + 1], h->rbs[rbi].cout, x.H, x.W));
        ++rbi;
        x = bd.resblock(h->rbs[rbi++], x, nullptr, false, sm, sr);
        // (an Upsample reads fp32: planes of its input only with the development switch that runs it on conv_pf_kernel)
        // ... or, where the fused-phase plane-operand kernel takes it, planes INSTEAD of fp32 (the Upsample is the only reader)
        const bool up_planes = bd.pf_tz_would_plan(h->ups[i], x.H, x.W);
        x = bd.attention(h->attns[ati++], x, sm, sr, up_planes ? Builder::SITE_ALWAYS_PLANES : Builder::SITE_NONE, up_planes);
        const bool x_planes_only = x.pf != nullptr;
        const size_t ops_before = h->ops.size();
        const ConvW &uw = h->ups[i];
        // the last Upsample feeds the final convolution only: planes INSTEAD of fp32 when that runs on the plane-operand kernel
        // (which needs the final LayerNorm applied here, in this epilogue)
        const bool fin_planes = i == n - 2 && bd.pf_17_would_plan(h->fin_conv, x.H * 2, x.W * 2);
        Act y = bd.new_act(uw.Cout, x.H * 2, x.W * 2, true, fin_planes ? Builder::SITE_ALWAYS_PLANES : Builder::SITE_JOIN);
        Builder::ConvOpts ou;
        ou.emit_pf = i < n - 2;
        // the next level's join is this tensor's only reader: planes INSTEAD of fp32 when block1 and res_conv both take planes
        if (i < n - 2 && !skips.empty())
            ou.no_f32 = bd.join_reads_planes(h->rbs[rbi], y.p, y.C, skips.back().p, y.H, y.W);
        bool up_pf_only = false;
        bool done = false;
        if (i == n - 2) {
            // the final LayerNorm (unet.py:104) needs per-pixel statistics of this output: emit them
            // from the epilogue when one workgroup owns all channels
            // ... or, better, apply that LayerNorm right there (every phase workgroup owns all channels of
            // its pixels): the final convolution then reads an already normalised tensor
            Builder::ConvOpts ol;
            ol.ln_g = h->fin_g; ol.ln_b = h->fin_b;
            ol.emit_pf = ol.no_f32 = fin_planes;
            done = bd.conv(uw, x.p, x.C, x.bs(), nullptr, 0, x.H, x.W, y.p, y.bs(), ol, true, PC_UP);
            final_ln_done = done;
            if (done && bd.last_pf_only) { Builder::PfTwin *ty = bd.twin(y.p); y.pf = ty->p; y.pf_bs = ty->bs(); }
            if (!done) {
                fsm = bd.dalloc((size_t)B * 4 * HWl); fsr = bd.dalloc((size_t)B * 4 * HWl);
                ou.stat_mean = fsm; ou.stat_rstd = fsr;
                done = bd.conv(uw, x.p, x.C, x.bs(), nullptr, 0, x.H, x.W, y.p, y.bs(), ou, true, PC_UP);
                if (!done) { ou.stat_mean = ou.stat_rstd = nullptr; }
            }
        }
        if (!done) {
            bd.conv(uw, x.p, x.C, x.bs(), nullptr, 0, x.H, x.W, y.p, y.bs(), ou, false, PC_UP);
            up_pf_only = bd.last_pf_only;
            if (up_pf_only) { Builder::PfTwin *ty = bd.twin(y.p); ty->only = true; y.pf = ty->p; y.pf_bs = ty->bs(); }
            if (i == n - 2) {
                if (!fsm) { fsm = bd.dalloc((size_t)B * 4 * HWl); fsr = bd.dalloc((size_t)B * 4 * HWl); }
                bd.ln(y.p, nullptr, y.C, y.H * y.W, nullptr, nullptr, 0, nullptr, nullptr, fsm, fsr);
            }
        }
        if (x_planes_only && bd.still_planes_only(x.p)) {
            bool on_pf = false;
            for (size_t q = ops_before; q < h->ops.size(); ++q) on_pf = on_pf || (h->ops[q].kind == Op::CONVPF && !h->ops[q].pw);
            if (!on_pf) return fail(h, CDC_ERR_UNSUPPORTED, "planes-only Upsample input without a plane-operand kernel");
        }
        x = y;
        if (!final_ln_done) h->taps["ups." + std::to_string(i)] = x;   // (the last one may hold LN(up(x)) instead; a planes-only tensor is unpacked on demand: Act::pf)
        if (bd.rc) return bd.rc;
    }
    if (n == 1) {       // no Upsample stage: statistics of the last attention output
        fsm = bd.dalloc((size_t)B * H * W); fsr = bd.dalloc((size_t)B * H * W);
        bd.ln(x.p, nullptr, x.C, x.H * x.W, nullptr, nullptr, 0, nullptr, nullptr, fsm, fsr);
    }
    // final_conv = Sequential(LayerNorm(dim), Conv2d(dim, out_dim, 7, padding=3))  (unet.py:104):
    // LN applied while staging; the 7x7 conv runs row-folded (1x7 taps, out_dim*7 virtual channels)
    // followed by the 7-row combine.
    const int KHf = 7;
    h->fin_P = bd.dalloc((size_t)B * h->out_dim * KHf * H * W);
    h->out_fx = bd.dalloc((size_t)B * h->out_dim * H * W);
    Builder::ConvOpts of;
    if (!final_ln_done) { of.pre_mean = fsm; of.pre_rstd = fsr; of.pre_g = h->fin_g; of.pre_b = h->fin_b; }
    of.no_bias = true;
    bd.conv(h->fin_conv, x.p, x.C, x.bs(), nullptr, 0, H, W, h->fin_P, (long long)h->out_dim * KHf * H * W,
            of, false, PC_CONV7);
    if (x.pf && !bd.rc && bd.still_planes_only(x.p) && (h->ops.empty() || h->ops.back().kind != Op::CONVPF || h->ops.back().pw))
        return fail(h, CDC_ERR_UNSUPPORTED, "planes-only final-convolution input without a plane-operand kernel");
    Op cb; cb.kind = Op::COMBINE; cb.prof = PC_SMALL;
    cb.cb = {h->fin_P, h->fin_bias, h->out_fx, h->out_dim, KHf, 3, H, W};
    cb.bytes = 4.0 * B * h->out_dim * (KHf + 1) * H * W;
    bd.emit(cb);
    if (bd.rc) return bd.rc;
    h->pB = B; h->pH = H; h->pW = W;
    return CDC_OK;
}

// Launch program of Compressor.encode up to the quantisers (compress_modules.py:43-51) for images [B][C][H][W].
int build_encoder_program(cdc_handle *h, int B, int H, int W) {
    if (h->pB == B && h->pH == H && h->pW == W) return CDC_OK;
    free_program(h);
    const int n = (int)h->enc_dims.size() - 1, nh = (int)h->henc_dims.size() - 1;
    const int down = 1 << (n + nh - 1);
    if (H % down || W % down)
        return fail(h, CDC_ERR_INVALID, "H=%d, W=%d must be multiples of %d", H, W, down);
    Builder bd{h, B, &h->act_allocs};
    h->in_x = bd.dalloc((size_t)B * h->enc_dims[0] * H * W);
    if (bd.rc) return bd.rc;
    Act x; x.p = h->in_x; x.C = h->enc_dims[0]; x.H = H; x.W = W;
    for (int i = 0; i < n; ++i) {
        x = bd.resblock(h->rbs[i], x, nullptr, false, nullptr, nullptr);
        const ConvW &dw = h->downs[i];
        Act y = bd.new_act(dw.Cout, x.H / 2, x.W / 2);
        Builder::ConvOpts od; od.emit_pf = true;
        bd.conv(dw, x.p, x.C, x.bs(), nullptr, 0, x.H, x.W, y.p, y.bs(), od, false, PC_DOWN);
        x = y;
        if (bd.rc) return bd.rc;
    }
    h->dec_outs.clear();
    h->dec_outs.push_back(x);                       // latent
    for (int i = 0; i < nh; ++i) {
        const ConvW &cw = h->hconvs[i];
        const int s = i == 0 ? 1 : 2;
        Act y = bd.new_act(cw.Cout, x.H / s, x.W / s);
        Builder::ConvOpts o;
        if (i < nh - 1) { o.relu = 1; o.relu_slope = 0.2f; }
        bd.conv(cw, x.p, x.C, x.bs(), nullptr, 0, x.H, x.W, y.p, y.bs(), o, false, i == 0 ? PC_CONV3 : PC_DOWN);
        x = y;
        if (bd.rc) return bd.rc;
    }
    h->dec_outs.push_back(x);                       // hyper_latent
    h->pB = B; h->pH = H; h->pW = W;
    return CDC_OK;
}

// Launch program of Compressor.hyper_dec (compress_modules.py:54-60) for q_hyper_latent [B][dims[0]][hh][wh].
// batch1_plan: every image runs the kernels a batch-1 call would run (the entropy coder's contract, entropy.hip)
int build_hyperdec_program(cdc_handle *h, int B, int hh, int wh, bool batch1_plan = false) {
    if (h->pB == B && h->pH == hh && h->pW == wh && h->p_batch1_plan == batch1_plan) return CDC_OK;
    free_program(h);
    Builder bd{h, B, &h->act_allocs};
    if (batch1_plan) bd.planB = 1;
    h->in_x = bd.dalloc((size_t)B * h->hyper_dims[0] * hh * wh);
    if (bd.rc) return bd.rc;
    Act x; x.p = h->in_x; x.C = h->hyper_dims[0]; x.H = hh; x.W = wh;
    const int n = (int)h->hconvs.size();
    for (int i = 0; i < n; ++i) {
        const ConvW &cw = h->hconvs[i];
        const bool last = i == n - 1;
        Act y = bd.new_act(cw.Cout, last ? x.H : x.H * 2, last ? x.W : x.W * 2);
        Builder::ConvOpts o;
        if (!last) { o.relu = 1; o.relu_slope = 0.2f; }           // nn.LeakyReLU(0.2)
        bd.conv(cw, x.p, x.C, x.bs(), nullptr, 0, x.H, x.W, y.p, y.bs(), o, false, last ? PC_CONV3 : PC_UP);
        x = y;
        if (bd.rc) return bd.rc;
    }
    h->dec_outs.clear();
    h->dec_outs.push_back(x);
    h->pB = B; h->pH = hh; h->pW = wh;
    h->p_batch1_plan = batch1_plan;
    return CDC_OK;
}

// Launch program of Compressor.decode (compress_modules.py:68-74) for q_latent [B][rev[0]][hl][wl].
int build_ctxdec_program(cdc_handle *h, int B, int hl, int wl) {
    if (h->pB == B && h->pH == hl && h->pW == wl) return CDC_OK;
    free_program(h);
    h->dec_outs.clear();
    Builder bd{h, B, &h->act_allocs};
    h->in_x = bd.dalloc((size_t)B * h->rev_dims[0] * hl * wl);
    if (bd.rc) return bd.rc;
    Act x; x.p = h->in_x; x.C = h->rev_dims[0]; x.H = hl; x.W = wl;
    const int n = (int)h->rev_dims.size() - 1;
    for (int i = 0; i < n; ++i) {
        x = bd.resblock(h->rbs[i], x, nullptr, false, nullptr, nullptr);
        const ConvW &uw = h->ups[i];
        Act y = bd.new_act(uw.Cout, x.H * 2, x.W * 2);
        Builder::ConvOpts ouu; ouu.emit_pf = true;
        bd.conv(uw, x.p, x.C, x.bs(), nullptr, 0, x.H, x.W, y.p, y.bs(), ouu, false, PC_UP);
        x = y;
        h->dec_outs.push_back(y);
        if (bd.rc) return bd.rc;
    }
    h->pB = B; h->pH = hl; h->pW = wl;
    return CDC_OK;
}

hipEvent_t get_event(cdc_handle *h) {
    if (!h->ev_free.empty()) { hipEvent_t e = h->ev_free.back(); h->ev_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int resolve_pending(cdc_handle *h) {
    for (auto &p : h->pending) {
        HIP_TRY(h, hipEventSynchronize(p.b));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, p.a, p.b));
        h->prof_ms[p.cls] += ms;
        h->prof_launches[p.cls] += 1;
        h->prof_flops[p.cls] += p.flops;
        h->prof_bytes[p.cls] += p.bytes;
        if (p.id >= 0 && p.id < (int)h->op_ms.size()) { h->op_ms[p.id] += ms; h->op_n[p.id] += 1; }
        h->ev_free.push_back(p.a);
        h->ev_free.push_back(p.b);
    }
    h->pending.clear();
    return CDC_OK;
}

int run_op(cdc_handle *h, const Op &op, int B, hipStream_t st) {
    const bool prof = h->prof && h->prof_now;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (prof) {
        ea = get_event(h); eb = get_event(h);
        HIP_TRY(h, hipEventRecord(ea, st));
    }
    switch (op.kind) {
        case Op::CONV: HIP_TRY(h, conv_launch(op.conv, op.plan, B, op.nz, st)); break;
        case Op::CONVPF:
            if (op.pw) HIP_TRY(h, pw_launch(op.pf, op.pfplan, B, st));
            else HIP_TRY(h, pf_launch(op.pf, op.pfplan, B, op.nz, st));
            break;
        case Op::CONVWS: HIP_TRY(h, ws_launch(op.ws, op.wsplan, st)); break;
        case Op::CONVWS1: HIP_TRY(h, ws1_launch(op.ws1, op.ws1plan, st)); break;
        case Op::PFPACK:
            if (op.pk.c4 == 2) {       // unpack: planes (pk.dst) -> the tensor's fp32 buffer (pk.src)
                HIP_TRY(h, pf_unpack_launch(op.pk.dst, op.pk.dst_bs, const_cast<float *>(op.pk.src), op.pk.src_bs, op.pk.C, op.pk.H, op.pk.W, B, st));
                break;
            }
            if (op.pk.c4) { HIP_TRY(h, c4_pack_launch(op.pk.src, op.pk.src_bs, (float *)op.pk.dst, op.pk.C, (long long)op.pk.H * op.pk.W, B, st)); break; }
            HIP_TRY(h, pf_pack_launch(op.pk.src, op.pk.src_bs, op.pk.dst, op.pk.dst_bs, op.pk.C, op.pk.H, op.pk.W, B, st));
            break;
        case Op::LN: HIP_TRY(h, ln_launch(op.ln, B, st)); break;
        case Op::TEMB: HIP_TRY(h, temb_launch(op.temb, B, st)); break;
        case Op::KSTATS:
            HIP_TRY(h, kstats_launch(op.at.k, op.at.bs, op.at.C, op.at.N, op.at.kmax, B, st));
            break;
        case Op::CTXP: {
            if (op.at_one) {
                HIP_TRY(h, ctx_one_launch(op.at.k, op.at.v, op.at.bs, op.at.C, op.at.N, op.at.scale, op.at.ctxw, op.at.Cin_pad, op.at.COP,
                                          op.at_ws_f16 ? op.at_Ws : nullptr, B, st, h->arith == 1));
                break;
            }
            HIP_TRY(h, ctx_partial_launch(op.at.k, op.at.v, op.at.bs, op.at.C, op.at.N, op.at.kmax,
                                          op.at.S, op.at.ksum, op.at.nsplit, B, st, h->arith == 1));
            break;
        }
        case Op::CTXR:
            HIP_TRY(h, ctx_reduce_launch(op.at.S, op.at.ksum, op.at.C, op.at.nsplit, op.at.scale,
                                         op.at.ctxw, op.at.Cin_pad, op.at.COP, B, st, op.at_ws_f16 ? op.at_Ws : nullptr));
            break;
        case Op::KVCTX: HIP_TRY(h, kvctx_launch(op.kvc, B, st)); break;
        case Op::LNCONV: HIP_TRY(h, lnconv_launch(op.lnc, B, st)); break;
        case Op::CTXF:
            HIP_TRY(h, ctx_fold_launch(op.at.S, op.at.ksum, op.at.C, op.at.nsplit, op.at.scale, op.at.WoT,
                                       op.at.WqT, op.at.T1, op.at.ctxw, op.at.Cin_pad, op.at.COP, op.at.ln_g,
                                       op.at.ln_b, op.at.b_out, op.at.biasB, B, st, op.at_M, op.at_Ws, op.at_ws_f16, op.at_Wq));
            break;
        case Op::COMBINE:
            HIP_TRY(h, fold_combine_launch(op.cb.P, op.cb.bias, op.cb.out, op.cb.Cout, op.cb.KH, op.cb.pad,
                                           op.cb.H, op.cb.W, B, st));
            break;
        case Op::DDIM: HIP_TRY(h, ddim_launch(op.ddim, st)); break;
        case Op::UNFOLD:
            HIP_TRY(h, unfold_x_launch(op.uf.src, op.uf.src_bs, op.uf.dst, op.uf.dst_bs, op.uf.C, op.uf.KW,
                                       op.uf.pad, op.uf.H, op.uf.W, B, st));
            break;
        case Op::COPY:
            HIP_TRY(h, copy_channels_launch(op.cp.src, op.cp.src_bs, op.cp.dst, op.cp.dst_bs, op.cp.n,
                                            B, st, op.cp_parts, op.cp_part_stride, op.cp_step, op.cp_step_stride));
            break;
    }
    if (prof) {
        HIP_TRY(h, hipEventRecord(eb, st));
        h->pending.push_back({ea, eb, op.prof, op.flops, op.bytes, op.id});
    }
    static const bool sync_each = getenv("CDC_SYNC_EACH_OP") != nullptr;     // debugging aid
    if (sync_each) HIP_TRY(h, hipStreamSynchronize(st));
    return CDC_OK;
}

// Context-only part of the program (hoisted context halves): once per decode / forward.
int run_pre(cdc_handle *h, hipStream_t st) {
    for (const Op &op : h->pre_ops) {
        int rc = run_op(h, op, h->pB, st);
        if (rc) return rc;
    }
    return CDC_OK;
}

// step >= 0: sampler iteration `step` (time-embedding shifts come from the per-decode table);
// step < 0: plain Unet.forward with the caller's per-image time values.
int run_unet(cdc_handle *h, hipStream_t st, int step, bool skip_combine = false) {
    for (const Op &op : h->ops) {
        if (skip_combine && op.kind == Op::COMBINE) continue;      // (the sampler kernel evaluates it: ddim_on_device)
        if (op.kind == Op::TEMB && (step >= 0 || step == -2)) {
            Op c = op;
            c.kind = Op::COPY;
            c.cp = {h->d_shift_tab + (step >= 0 ? (size_t)step * h->shift_bs : 0), 0, h->shift, h->shift_bs, h->shift_bs};
            if (step == -2) { c.cp_step = h->d_step; c.cp_step_stride = h->shift_bs; }
            int rc = run_op(h, c, h->pB, st);
            if (rc) return rc;
            continue;
        }
        int rc = run_op(h, op, h->pB, st);
        if (rc) return rc;
    }
    return CDC_OK;
}

int copy_in(cdc_handle *h, float *dst, const float *src, size_t n, int mem, hipStream_t st) {
    if (mem == CDC_MEM_HOST)
        HIP_TRY(h, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyHostToDevice, st));
    else
        HIP_TRY(h, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return CDC_OK;
}

int copy_out(cdc_handle *h, float *dst, const float *src, size_t n, int mem, hipStream_t st) {
    if (mem == CDC_MEM_HOST) {
        HIP_TRY(h, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
    } else {
        HIP_TRY(h, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return CDC_OK;
}

int stage_ctx(cdc_handle *h, const float *const *ctx, int n_ctx, int B, int mem, hipStream_t st) {
    if (n_ctx != (int)h->in_ctx.size())
        return fail(h, CDC_ERR_INVALID, "expected %d context tensors, got %d", (int)h->in_ctx.size(),
                    n_ctx);
    for (int l = 0; l < n_ctx; ++l) {
        int rc = copy_in(h, h->in_ctx[l].p, ctx[l], (size_t)B * h->in_ctx[l].bs(), mem, st);
        if (rc) return rc;
    }
    return CDC_OK;
}

int ensure_device(cdc_handle *h) {
    if (!h) return CDC_ERR_INVALID;
    if (!h->own_stream) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0)
            return fail(h, CDC_ERR_HIP, "no HIP device available: %s (there is no CPU fallback)",
                        hipGetErrorString(e));
        if (h->device >= ndev)
            return fail(h, CDC_ERR_INVALID, "device %d out of range (%d devices)", h->device, ndev);
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreate(&h->ev0));
        HIP_TRY(h, hipEventCreate(&h->ev1));
    }
    HIP_TRY(h, hipSetDevice(h->device));
    return CDC_OK;
}

// Host pointers: the library's own stream (the call synchronises before returning).  Device pointers:
// the caller's stream, where NULL is the HIP null stream (torch's default stream), so that work
// is ordered with the caller's own kernels and copies.
hipStream_t pick_stream(cdc_handle *h, void *stream, int mem) {
    return mem == CDC_MEM_DEVICE ? (hipStream_t)stream : h->own_stream;
}

int check_ready(cdc_handle *h) {
    if (!h) return CDC_ERR_INVALID;
    if (!h->finalized) return fail(h, CDC_ERR_STATE, "weights not finalized (cdc_finalize_weights)");
    HIP_TRY(h, hipSetDevice(h->device));
    return CDC_OK;
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
namespace {
// no C++ exception crosses the C boundary (std::bad_alloc from a vector sized by a hostile header, ...)
template <class F> int no_throw(cdc_handle *h, F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { return h ? fail(h, CDC_ERR_NOMEM, "out of host memory") : CDC_ERR_NOMEM; }
    catch (const std::exception &e) { return h ? fail(h, CDC_ERR_INVALID, "internal error: %s", e.what()) : CDC_ERR_INVALID; }
    catch (...) { return h ? fail(h, CDC_ERR_INVALID, "internal error") : CDC_ERR_INVALID; }
}
}  // namespace

// ---- range guard of the two-plane fp16 arithmetic ------------------------------------------------------------------
// |activation| >= 65504 becomes inf / NaN in CDC_ARITH_F16X2 and propagates to the results of the call.  Every entry point
// that runs the arithmetic checks its results (one small kernel + one 4-byte read-back, i.e. a stream synchronisation);
// a call whose results are not finite is repeated ONCE in the full-range three-plane bf16 arithmetic, and the handle stays
// in that mode (cdc_get_arith / cdc_get_range_faults tell).  Results that are non-finite there too -- a non-finite input,
// parameters that overflow fp32 -- are returned as they are, as the reference would (cdc_get_nonfinite_results counts them).
namespace {
bool guard_enabled(const cdc_handle *h) {
    static const bool no_guard = getenv("CDC_NO_RANGE_GUARD") != nullptr;
    return !no_guard && (h->arith == CDC_ARITH_F16X2 || h->in_retry);
}
int ensure_fault_flag(cdc_handle *h) {
    if (!h->d_fault) { void *p = nullptr; HIP_TRY(h, hipMalloc(&p, sizeof(int))); h->d_fault = (int *)p; h->weight_allocs.push_back(p); }
    return CDC_OK;
}
struct GuardBuf { const float *p; long long bs, n; };
// OR of "not finite" over device tensors (and of whatever a kernel already left in d_fault) -> *fault
int guard_check(cdc_handle *h, std::initializer_list<GuardBuf> bufs, int B, hipStream_t st, int *fault) {
    for (const GuardBuf &g : bufs)
        if (g.p && g.n > 0) HIP_TRY(h, cdc::nonfinite_launch(g.p, g.bs, g.n, B, h->d_fault, st));
    *fault = 0;
    HIP_TRY(h, hipMemcpyAsync(fault, h->d_fault, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return CDC_OK;
}
// a fault in F16X2: switch to the full-range arithmetic (returns true: repeat the call); a fault in the repetition: count it
bool guard_escalate(cdc_handle *h, int *rc) {
    *rc = CDC_OK;
    if (h->in_retry || h->arith != CDC_ARITH_F16X2) {
        ++h->nonfinite_results;
        if (h->in_retry) h->retry_futile = true;            // non-finite in the full-range arithmetic too: it was not the fp16 range
        return false;
    }
    ++h->range_faults;
    static bool warned = false;
    if (!warned) {
        warned = true;
        fprintf(stderr, "cdc_hip: a value left the fp16 range of CDC_ARITH_F16X2; the call is repeated in CDC_ARITH_BF16X3 and the handle stays "
                        "in that (slower, full-range) arithmetic -- see cdc_get_range_faults()\n");
    }
    *rc = cdc_set_arith(h, CDC_ARITH_BF16X3);
    return *rc == CDC_OK;
}
// The repetition of a call in BF16X3.  When that result is non-finite as well (a NaN / inf in the inputs or the parameters), the
// range was not the cause: the handle goes back to F16X2 and the fault is counted in nonfinite_results only.
struct RetryScope {
    cdc_handle *h;
    explicit RetryScope(cdc_handle *h_) : h(h_) { h->in_retry = true; h->retry_futile = false; }
    ~RetryScope() {
        h->in_retry = false;
        if (h->retry_futile) {
            h->retry_futile = false;
            if (h->range_faults > 0) --h->range_faults;
            (void)cdc_set_arith(h, CDC_ARITH_F16X2);
        }
    }
};
}  // namespace

namespace {
// device scratch of one entropy call, released on every exit path
struct DevPool {
    std::vector<void *> v;
    ~DevPool() { for (void *p : v) (void)hipFree(p); }
    template <class T> hipError_t get(T **p, size_t n) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) { v.push_back(q); *p = (T *)q; }
        return e;
    }
};
}  // namespace

extern "C" {

const char *cdc_version(void) { return "cdc_hip 0.9 (gfx950; fp32-class convolutions from split fp16 / bf16 operands on the matrix cores)"; }

const char *cdc_last_error(const cdc_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int cdc_create(const cdc_unet_config *cfg, int device, cdc_handle **out) {
    if (!cfg || !out) return fail(nullptr, CDC_ERR_INVALID, "null argument");
    if (cfg->dim <= 0 || cfg->n_dim_mults < 1 || cfg->n_dim_mults > CDC_MAX_LEVELS ||
        cfg->n_context_dim_mults < 0 || cfg->n_context_dim_mults > CDC_MAX_LEVELS || cfg->channels < 1)
        return fail(nullptr, CDC_ERR_INVALID, "bad cdc_unet_config");
    if (device < 0) return fail(nullptr, CDC_ERR_INVALID, "device %d out of range", device);
    std::unique_ptr<cdc_handle> h(new cdc_handle);
    h->cfg = *cfg;
    h->device = device;
    h->out_dim = cfg->out_dim > 0 ? cfg->out_dim : cfg->channels;
    h->dims.push_back(cfg->channels);
    for (int i = 0; i < cfg->n_dim_mults; ++i) h->dims.push_back(cfg->dim * cfg->dim_mults[i]);
    h->context_dims.push_back(cfg->context_channels);
    for (int i = 0; i < cfg->n_context_dim_mults; ++i)
        h->context_dims.push_back(cfg->dim * cfg->context_dim_mults[i]);
    h->n_res = cfg->n_dim_mults;
    build_manifest(h.get());
    // The HIP device is first touched by cdc_finalize_weights / cdc_op_* (ensure_device), so the
    // manifest and load_tensor calls work on a host without a GPU; compute never does.
    *out = h.release();
    return CDC_OK;
}

void cdc_destroy(cdc_handle *h) {
    if (!h) return;
    if (!h->own_stream) { delete h; return; }
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    free_program(h);
    free_pool(&h->weight_allocs);
    if (h->d_tab) (void)hipFree(h->d_tab);
    if (h->d_tab_v) (void)hipFree(h->d_tab_v);
    if (h->d_time_steps) (void)hipFree(h->d_time_steps);
    if (h->d_shift_tab) (void)hipFree(h->d_shift_tab);
    (void)resolve_pending(h);
    for (hipEvent_t e : h->ev_free) (void)hipEventDestroy(e);
    if (h->gev_in) (void)hipEventDestroy(h->gev_in);
    if (h->gev_out) (void)hipEventDestroy(h->gev_out);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int cdc_set_arith(cdc_handle *h, int mode) {
    if (!h) return CDC_ERR_INVALID;
    if (mode != CDC_ARITH_BF16X3 && mode != CDC_ARITH_F16X2) return fail(h, CDC_ERR_INVALID, "arith mode %d", mode);
    if (mode != h->arith) {
        if (h->own_stream) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
        free_program(h);          // launch plans and LDS carve-up depend on the operand format
        h->arith = mode;
    }
    return CDC_OK;
}

int cdc_get_arith(const cdc_handle *h) { return h ? h->arith : CDC_ERR_INVALID; }
int cdc_get_range_faults(const cdc_handle *h) { return h ? h->range_faults : CDC_ERR_INVALID; }
int cdc_get_nonfinite_results(const cdc_handle *h) { return h ? h->nonfinite_results : CDC_ERR_INVALID; }

int cdc_num_tensors(const cdc_handle *h) {
    if (!h) return CDC_ERR_INVALID;
    int n = 0;
    for (const Param &p : h->params) n += p.optional ? 0 : 1;     // optional entries sit at the end
    return n;
}

int cdc_tensor_info(const cdc_handle *h, int index, const char **name, int64_t shape[4], int *ndim) {
    if (!h || index < 0 || index >= (int)h->params.size()) return CDC_ERR_INVALID;
    const Param &p = h->params[index];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int)p.shape.size();
    if (shape)
        for (size_t i = 0; i < p.shape.size() && i < 4; ++i) shape[i] = p.shape[i];
    return CDC_OK;
}

int cdc_load_tensor(cdc_handle *h, const char *name, const float *data, const int64_t *shape,
                    int ndim) {
    if (!h || !name || !data || !shape) return CDC_ERR_INVALID;
    auto it = h->pindex.find(name);
    if (it == h->pindex.end()) return fail(h, CDC_ERR_INVALID, "unexpected key \"%s\"", name);
    Param &p = h->params[it->second];
    bool same = (int)p.shape.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = p.shape[i] == shape[i];
    if (!same) {
        std::string got, want;
        for (int i = 0; i < ndim; ++i) got += (i ? "," : "") + std::to_string(shape[i]);
        for (auto d : p.shape) want += (want.empty() ? "" : ",") + std::to_string(d);
        return fail(h, CDC_ERR_INVALID, "size mismatch for %s: got [%s], expected [%s]", name,
                    got.c_str(), want.c_str());
    }
    p.host.assign(data, data + p.numel());
    p.loaded = true;
    h->finalized = false;
    return CDC_OK;
}

int cdc_finalize_weights(cdc_handle *h) {
    if (!h) return CDC_ERR_INVALID;
    for (const Param &p : h->params)
        if (!p.loaded && !p.optional) return fail(h, CDC_ERR_STATE, "missing key \"%s\"", p.name.c_str());
    int rc = ensure_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    free_program(h);
    free_pool(&h->weight_allocs);
    h->d_fault = nullptr; h->d_step = nullptr;          // (they lived in that pool)
    h->rbs.clear(); h->attns.clear(); h->downs.clear(); h->ups.clear();
    if (h->kind == 3) {
        const int n = (int)h->enc_dims.size() - 1;
        int shift_off = 0;
        for (int i = 0; i < n; ++i) {
            const std::string p = "enc." + std::to_string(i);
            if ((rc = pack_resblock(h, p + ".0", h->enc_dims[i], h->enc_dims[i + 1], i == 0 ? 7 : 3, &shift_off, 0, false)))
                return rc;
            ConvW dw;
            const std::string d = p + "." + std::to_string(h->down_index) + ".conv";
            if ((rc = pack_named_conv(h, d + ".weight", d + ".bias", 2, 1, false, &dw))) return rc;
            h->downs.push_back(dw);
        }
        h->hconvs.clear();
        const int nh = (int)h->henc_dims.size() - 1;
        for (int i = 0; i < nh; ++i) {
            const std::string p = "hyper_enc." + std::to_string(i) + ".0";
            ConvW cw;
            if ((rc = pack_named_conv(h, p + ".weight", p + ".bias", i == 0 ? 1 : 2, i == 0 ? 1 : 2, false, &cw))) return rc;
            h->hconvs.push_back(cw);
        }
        h->shift_bs = 0;
        h->finalized = true;
        return CDC_OK;
    }
    if (h->kind == 2) {
        const int n = (int)h->hyper_dims.size() - 1;
        h->hconvs.clear();
        for (int i = 0; i < n; ++i) {
            const std::string p = "hyper_dec." + std::to_string(i) + ".0";
            ConvW cw;
            const bool last = i == n - 1;
            if ((rc = pack_named_conv(h, p + ".weight", p + ".bias", last ? 1 : 2, last ? 1 : 2, !last, &cw))) return rc;
            h->hconvs.push_back(cw);
        }
        h->d_prior = nullptr;
        if (h->params[h->pindex.at("prior.affine.0.weight")].loaded) {
            // per channel: softplus(W0)[3] b0[3] tanh(a0)[3] | softplus(W1)[9] b1[3] tanh(a1)[3] | softplus(W2)[9] b2[3]
            // tanh(a2)[3] | softplus(W3)[3] b3[1] | pad  = 44 floats  (PriorFunction.forward, FlexiblePrior.cdf)
            const int pc = h->hyper_dims[0];
            std::vector<float> pk((size_t)pc * 44, 0.f);
            h->h_prior.assign((size_t)pc * 44, 0.0);
            h->ent.reset();
            auto sp = [](float v) { return v > 20.f ? v : (float)log1p(exp((double)v)); };      // F.softplus (threshold 20)
            auto spd = [](float v) { return v > 20.f ? (double)v : log1p(exp((double)v)); };
            const int pd[5] = {1, 3, 3, 3, 1};
            for (int c = 0; c < pc; ++c) {
                float *o = &pk[(size_t)c * 44];
                double *od = &h->h_prior[(size_t)c * 44];
                for (int i = 0; i < 4; ++i) {
                    const std::string pi = "prior.affine." + std::to_string(i);
                    for (const char *suf : {".weight", ".bias"})
                        if (!h->params[h->pindex.at(pi + suf)].loaded) return fail(h, CDC_ERR_STATE, "missing key \"%s%s\"", pi.c_str(), suf);
                    const auto &w = hostp(h, pi + ".weight");
                    const auto &bb = hostp(h, pi + ".bias");
                    const int nin = pd[i], nout = pd[i + 1];
                    for (int k = 0; k < nin * nout; ++k) { *o++ = sp(w[(size_t)c * nin * nout + k]); *od++ = spd(w[(size_t)c * nin * nout + k]); }
                    for (int k = 0; k < nout; ++k) { *o++ = bb[(size_t)c * nout + k]; *od++ = (double)bb[(size_t)c * nout + k]; }
                    if (i < 3) {
                        const std::string ai = "prior.a." + std::to_string(i);
                        if (!h->params[h->pindex.at(ai)].loaded) return fail(h, CDC_ERR_STATE, "missing key \"%s\"", ai.c_str());
                        const auto &a = hostp(h, ai);
                        for (int k = 0; k < nout; ++k) { *o++ = (float)tanh((double)a[(size_t)c * nout + k]); *od++ = tanh((double)a[(size_t)c * nout + k]); }
                    }
                }
            }
            if ((rc = upload(h, pk.data(), pk.size(), &h->d_prior, &h->weight_allocs))) return rc;
        }
        h->shift_bs = 0;
        h->finalized = true;
        return CDC_OK;
    }
    if (h->kind == 1) {
        // Compressor.dec: ResnetBlock(rev[i] -> rev[i+1] | rev[i] on the last level) + Upsample(-> rev[i+1])
        const int n = (int)h->rev_dims.size() - 1;
        int shift_off = 0;
        for (int i = 0; i < n; ++i) {
            const std::string p = "dec." + std::to_string(i);
            const int din = h->rev_dims[i], dout = h->rev_dims[i + 1], dmid = i == n - 1 ? din : dout;
            if ((rc = pack_resblock(h, p + ".0", din, dmid, 3, &shift_off, 0, false))) return rc;
            ConvW uw;
            const std::string u = p + "." + std::to_string(h->up_index);
            if ((rc = pack_named_conv(h, u + ".conv.weight", u + ".conv.bias", 2, 1, true, &uw))) return rc;
            h->ups.push_back(uw);
        }
        h->shift_bs = 0;
        h->finalized = true;
        return CDC_OK;
    }
    if ((rc = upload_param(h, "time_mlp.0.weight", &h->tm_w0))) return rc;
    if ((rc = upload_param(h, "time_mlp.0.bias", &h->tm_b0))) return rc;
    if ((rc = upload_param(h, "time_mlp.2.weight", &h->tm_w2))) return rc;
    if ((rc = upload_param(h, "time_mlp.2.bias", &h->tm_b2))) return rc;
    const int n = h->n_res;
    int shift_off = 0;
    // FORWARD order: downs (rb, rb, attn, down) x n ; mid_block1, mid_attn, mid_block2 ; ups
    for (int i = 0; i < n; ++i) {
        const std::string p = "downs." + std::to_string(i);
        const int dout = h->dims[i + 1];
        const int cin0 = down_in_channels(h, i);
        const int hoist_cx = (cin0 != h->dims[i] && !dev_env("CDC_NO_HOIST")) ? h->dims[i] : 0;
        if ((rc = pack_resblock(h, p + ".0", cin0, dout, i == 0 ? 7 : 3, &shift_off, hoist_cx)))
            return rc;
        if ((rc = pack_resblock(h, p + ".1", dout, dout, 3, &shift_off))) return rc;
        if ((rc = pack_attn(h, p + ".2", dout))) return rc;
        if (i < n - 1) {
            ConvW dw;
            if ((rc = pack_named_conv(h, p + ".3.conv.weight", p + ".3.conv.bias", 2, 1, false, &dw)))
                return rc;
            h->downs.push_back(dw);
        }
    }
    const int mid = h->dims[n];
    if ((rc = pack_resblock(h, "mid_block1", mid, mid, 3, &shift_off))) return rc;
    if ((rc = pack_attn(h, "mid_attn", mid))) return rc;
    if ((rc = pack_resblock(h, "mid_block2", mid, mid, 3, &shift_off))) return rc;
    for (int i = 0; i < n - 1; ++i) {
        const int lvl = n - 1 - i;
        const int din = h->dims[lvl], dout = h->dims[lvl + 1];
        const std::string p = "ups." + std::to_string(i);
        if ((rc = pack_resblock(h, p + ".0", dout * 2, din, 3, &shift_off))) return rc;
        if ((rc = pack_resblock(h, p + ".1", din, din, 3, &shift_off))) return rc;
        if ((rc = pack_attn(h, p + ".2", din))) return rc;
        ConvW uw;
        if ((rc = pack_named_conv(h, p + ".3.conv.weight", p + ".3.conv.bias", 2, 1, true, &uw)))
            return rc;
        h->ups.push_back(uw);
    }
    if ((rc = upload_param(h, "final_conv.0.g", &h->fin_g))) return rc;
    if ((rc = upload_param(h, "final_conv.0.b", &h->fin_b))) return rc;
    {
        // row-folded final convolution: w'[(co*7+ky)][ci][0][kx] = w[co][ci][ky][kx]
        const Param &pw = h->params[h->pindex.at("final_conv.1.weight")];
        const int co_n = (int)pw.shape[0], ci_n = (int)pw.shape[1], kh = (int)pw.shape[2], kw = (int)pw.shape[3];
        std::vector<float> wf((size_t)co_n * kh * ci_n * kw);
        for (int co = 0; co < co_n; ++co)
            for (int ci = 0; ci < ci_n; ++ci)
                for (int ky = 0; ky < kh; ++ky)
                    for (int kx = 0; kx < kw; ++kx)
                        wf[(((size_t)(co * kh + ky)) * ci_n + ci) * kw + kx] =
                            pw.host[(((size_t)co * ci_n + ci) * kh + ky) * kw + kx];
        if ((rc = pack_conv(h, wf.data(), nullptr, co_n * kh, ci_n, 1, kw, 1, 0, false, &h->fin_conv,
                            &h->weight_allocs))) return rc;
        h->fin_conv.pad_y = 0; h->fin_conv.pad_x = kw / 2;
        if ((rc = upload_param(h, "final_conv.1.bias", &h->fin_bias))) return rc;
    }
    h->shift_bs = shift_off;
    std::vector<TembLayer> tl;
    for (const ResBlockW &rb : h->rbs) tl.push_back({rb.mlp_w, rb.mlp_b, rb.cout, rb.shift_off});
    float *dl = nullptr;
    if ((rc = upload(h, nullptr, tl.size() * sizeof(TembLayer) / sizeof(float) + 1, &dl,
                     &h->weight_allocs))) return rc;
    HIP_TRY(h, hipMemcpy(dl, tl.data(), tl.size() * sizeof(TembLayer), hipMemcpyHostToDevice));
    h->d_temb_layers = (TembLayer *)dl;
    h->finalized = true;
    return CDC_OK;
}

int cdc_encoder_create(const cdc_encoder_config *cfg, int device, cdc_handle **out) {
    if (!cfg || !out) return fail(nullptr, CDC_ERR_INVALID, "null argument");
    if (cfg->dim <= 0 || cfg->channels < 1 || cfg->n_dim_mults < 1 || cfg->n_dim_mults > CDC_MAX_LEVELS ||
        cfg->n_hyper_mults < 1 || cfg->n_hyper_mults > CDC_MAX_LEVELS || cfg->down_index < 1 || cfg->down_index > 2)
        return fail(nullptr, CDC_ERR_INVALID, "bad cdc_encoder_config");
    if (device < 0) return fail(nullptr, CDC_ERR_INVALID, "device %d out of range", device);
    std::unique_ptr<cdc_handle> h(new cdc_handle);
    memset(&h->cfg, 0, sizeof h->cfg);
    h->cfg.dim = cfg->dim;
    h->kind = 3;
    h->device = device;
    h->down_index = cfg->down_index;
    h->enc_dims.push_back(cfg->channels);
    for (int i = 0; i < cfg->n_dim_mults; ++i) h->enc_dims.push_back(cfg->dim * cfg->dim_mults[i]);
    h->henc_dims.push_back(h->enc_dims.back());
    for (int i = 0; i < cfg->n_hyper_mults; ++i) h->henc_dims.push_back(cfg->dim * cfg->hyper_mults[i]);
    for (int i = 0; i < cfg->n_dim_mults; ++i) {          // registration order of Compressor.enc (:131-141)
        const std::string p = "enc." + std::to_string(i);
        add_resblock_params(h.get(), p + ".0", h->enc_dims[i], h->enc_dims[i + 1], i == 0 ? 7 : 3, false);
        const std::string d = p + "." + std::to_string(cfg->down_index) + ".conv";
        add_param(h.get(), d + ".weight", {h->enc_dims[i + 1], h->enc_dims[i + 1], 3, 3});
        add_param(h.get(), d + ".bias", {h->enc_dims[i + 1]});
    }
    for (int i = 0; i < cfg->n_hyper_mults; ++i) {        // Compressor.hyper_enc (:155-165)
        const std::string p = "hyper_enc." + std::to_string(i) + ".0";
        const int k = i == 0 ? 3 : 5;
        add_param(h.get(), p + ".weight", {h->henc_dims[i + 1], h->henc_dims[i], k, k});
        add_param(h.get(), p + ".bias", {h->henc_dims[i + 1]});
    }
    *out = h.release();
    return CDC_OK;
}

int cdc_encoder_encode(cdc_handle *h, const float *images, float *latent, float *hyper_latent, int B, int H, int W,
                       int mem, void *stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->kind != 3) return fail(h, CDC_ERR_STATE, "handle is not an encoder");
    if (!images || !latent || !hyper_latent || B < 1) return fail(h, CDC_ERR_INVALID, "null/invalid argument");
    if ((rc = build_encoder_program(h, B, H, W))) return rc;
    hipStream_t st = pick_stream(h, stream, mem);
    if ((rc = copy_in(h, h->in_x, images, (size_t)B * h->enc_dims[0] * H * W, mem, st))) return rc;
    const bool guard = guard_enabled(h);
    if (guard) { if ((rc = ensure_fault_flag(h))) return rc; HIP_TRY(h, hipMemsetAsync(h->d_fault, 0, sizeof(int), st)); }
    h->prof_now = true;
    for (const Op &op : h->ops)
        if ((rc = run_op(h, op, h->pB, st))) return rc;
    const Act &l = h->dec_outs[0], &hl = h->dec_outs[1];
    if (guard) {
        int fault = 0;
        if ((rc = guard_check(h, {{l.p, l.bs(), (long long)l.C * l.H * l.W}, {hl.p, hl.bs(), (long long)hl.C * hl.H * hl.W}}, B, st, &fault))) return rc;
        if (fault) {
            if (guard_escalate(h, &rc)) { RetryScope r(h); return cdc_encoder_encode(h, images, latent, hyper_latent, B, H, W, mem, stream); }
            if (rc) return rc;
        }
    }
    if ((rc = copy_out(h, latent, l.p, (size_t)B * l.C * l.H * l.W, mem, st))) return rc;
    return copy_out(h, hyper_latent, hl.p, (size_t)B * hl.C * hl.H * hl.W, mem, st);
}

int cdc_hyperdec_create(const cdc_hyperdec_config *cfg, int device, cdc_handle **out) {
    if (!cfg || !out) return fail(nullptr, CDC_ERR_INVALID, "null argument");
    if (cfg->n_layers < 1 || cfg->n_layers > CDC_MAX_LEVELS) return fail(nullptr, CDC_ERR_INVALID, "bad cdc_hyperdec_config");
    if (device < 0) return fail(nullptr, CDC_ERR_INVALID, "device %d out of range", device);
    std::unique_ptr<cdc_handle> h(new cdc_handle);
    memset(&h->cfg, 0, sizeof h->cfg);
    h->kind = 2;
    h->device = device;
    for (int i = 0; i <= cfg->n_layers; ++i) {
        if (cfg->dims[i] < 1) return fail(nullptr, CDC_ERR_INVALID, "bad cdc_hyperdec_config");
        h->hyper_dims.push_back(cfg->dims[i]);
    }
    if (h->hyper_dims.back() % 2) return fail(nullptr, CDC_ERR_INVALID, "the last layer must produce mean and scale");
    for (int i = 0; i < cfg->n_layers; ++i) {
        const std::string p = "hyper_dec." + std::to_string(i) + ".0";
        const int din = h->hyper_dims[i], dout = h->hyper_dims[i + 1];
        if (i == cfg->n_layers - 1) add_param(h.get(), p + ".weight", {dout, din, 3, 3});      // Conv2d
        else add_param(h.get(), p + ".weight", {din, dout, 5, 5});                          // ConvTranspose2d
        add_param(h.get(), p + ".bias", {dout});
    }
    // FlexiblePrior(channels = dims[0], dims = [3, 3, 3]) (network_components.py:316-336), squeezed shapes;
    // optional: only cdc_bpp needs it
    const int pc = h->hyper_dims[0], pd[5] = {1, 3, 3, 3, 1};
    for (int i = 0; i < 4; ++i) {
        add_param(h.get(), "prior.affine." + std::to_string(i) + ".weight", {pc, pd[i], pd[i + 1]}, true);
        add_param(h.get(), "prior.affine." + std::to_string(i) + ".bias", {pc, pd[i + 1]}, true);
        if (i < 3) add_param(h.get(), "prior.a." + std::to_string(i), {pc, pd[i + 1]}, true);
    }
    *out = h.release();
    return CDC_OK;
}

int cdc_bpp(cdc_handle *h, const float *q_hyper_latent, const float *q_latent, const float *mean, const float *scale,
            float *bpp, int B, int hh, int wh, int H_img, int W_img, int mem, void *stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->kind != 2) return fail(h, CDC_ERR_STATE, "handle is not a hyper decoder");
    if (!h->d_prior) return fail(h, CDC_ERR_STATE, "the prior.* tensors were not loaded");
    if (!q_hyper_latent || !q_latent || !mean || !scale || !bpp || B < 1 || hh < 1 || wh < 1 || H_img < 1 || W_img < 1)
        return fail(h, CDC_ERR_INVALID, "null/invalid argument");
    const int Ch = h->hyper_dims[0], Cl = h->hyper_dims.back() / 2;
    const long long nh = (long long)Ch * hh * wh, up = 1LL << ((int)h->hyper_dims.size() - 2),
                    nl = (long long)Cl * up * up * hh * wh;   // hyper_dec upsamples by 2 per layer but the last
    hipStream_t st = pick_stream(h, stream, mem);
    std::vector<void *> tmp;
    auto dev = [&](const float *p, long long n) -> const float * {
        if (mem == CDC_MEM_DEVICE) return p;
        void *d = nullptr;
        if (hipMalloc(&d, sizeof(float) * n) != hipSuccess) return nullptr;
        tmp.push_back(d);
        (void)hipMemcpyAsync(d, p, sizeof(float) * n, hipMemcpyHostToDevice, st);
        return (const float *)d;
    };
    const float *dqh = dev(q_hyper_latent, B * nh), *dql = dev(q_latent, B * nl), *dm = dev(mean, B * nl),
                *ds = dev(scale, B * nl);
    float *dout = nullptr;
    hipError_t e = hipSuccess;
    if (!dqh || !dql || !dm || !ds) e = hipErrorOutOfMemory;
    if (e == hipSuccess) {
        if (mem == CDC_MEM_DEVICE) dout = bpp;
        else { e = hipMalloc(&dout, sizeof(float) * B); if (e == hipSuccess) tmp.push_back(dout); }
    }
    if (e == hipSuccess)
        e = bpp_launch(dqh, nh, hh * wh, h->d_prior, dql, dm, ds, nl, 1.0f / ((float)H_img * (float)W_img), dout, B, st);
    if (e == hipSuccess && mem != CDC_MEM_DEVICE) {
        e = hipMemcpyAsync(bpp, dout, sizeof(float) * B, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (mem != CDC_MEM_DEVICE) { (void)hipStreamSynchronize(st); for (void *d : tmp) (void)hipFree(d); }
    if (e != hipSuccess) return fail(h, CDC_ERR_HIP, "cdc_bpp: %s", hipGetErrorString(e));
    return CDC_OK;
}

int cdc_hyperdec_decode(cdc_handle *h, const float *q_hyper_latent, float *mean, float *scale, int B, int hh,
                        int wh, float scale_min, int mem, void *stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->kind != 2) return fail(h, CDC_ERR_STATE, "handle is not a hyper decoder");
    if (!q_hyper_latent || !mean || !scale || B < 1 || hh < 1 || wh < 1)
        return fail(h, CDC_ERR_INVALID, "null/invalid argument");
    if ((rc = build_hyperdec_program(h, B, hh, wh))) return rc;
    hipStream_t st = pick_stream(h, stream, mem);
    if ((rc = copy_in(h, h->in_x, q_hyper_latent, (size_t)B * h->hyper_dims[0] * hh * wh, mem, st))) return rc;
    const bool guard = guard_enabled(h);
    if (guard) { if ((rc = ensure_fault_flag(h))) return rc; HIP_TRY(h, hipMemsetAsync(h->d_fault, 0, sizeof(int), st)); }
    h->prof_now = true;
    for (const Op &op : h->ops)
        if ((rc = run_op(h, op, h->pB, st))) return rc;
    const Act &o = h->dec_outs[0];                  // [B][2C][4hh][4wh]: mean = channels [0, C), scale = [C, 2C)
    const int C = o.C / 2;
    const long long half = (long long)C * o.H * o.W;
    if (guard) {
        int fault = 0;
        if ((rc = guard_check(h, {{o.p, o.bs(), 2 * half}}, B, st, &fault))) return rc;
        if (fault) {
            if (guard_escalate(h, &rc)) { RetryScope r(h); return cdc_hyperdec_decode(h, q_hyper_latent, mean, scale, B, hh, wh, scale_min, mem, stream); }
            if (rc) return rc;
        }
    }
    HIP_TRY(h, clamp_min_launch(o.p + half, o.bs(), half, scale_min, B, st));
    for (int b = 0; b < B; ++b) {
        if ((rc = copy_out(h, mean + (size_t)b * half, o.p + (size_t)b * o.bs(), (size_t)half, mem, st))) return rc;
        if ((rc = copy_out(h, scale + (size_t)b * half, o.p + (size_t)b * o.bs() + half, (size_t)half, mem, st))) return rc;
    }
    return CDC_OK;
}

// ---- entropy coder (entropy.hip) ---------------------------------------------------------
namespace {
constexpr int kStreamVersion = 3;
// 'C' 'D' 'C' 3 | arith | 0 | hh u16 | wh u16 | n_hyper u32 | n_latent u32 | model hash u32 | symbol checksum u32 | hyper escapes u32 | latent escapes u32
constexpr int kStreamHeader = 34;

int ensure_entropy(cdc_handle *h, const float *medians) {
    if (h->kind != 2) return fail(h, CDC_ERR_STATE, "handle is not a hyper decoder");
    if (h->h_prior.empty()) return fail(h, CDC_ERR_STATE, "the prior.* tensors were not loaded");
    const int C = h->hyper_dims[0];
    if (!h->ent) h->ent.reset(new cdc::EntropyModel);
    cdc::entropy_init(h->ent.get());
    if ((int)h->ent->medians.size() != C || memcmp(h->ent->medians.data(), medians, sizeof(float) * C) != 0) {
        cdc::entropy_build_hyper(h->ent.get(), h->h_prior.data(), medians, C);
        h->ent->dev_stale = true;
    }
    if (!h->ent->d_edges) {
        int rc = upload(h, h->ent->edges, cdc::kEntropyBins, &h->ent->d_edges, &h->weight_allocs);
        if (rc) return rc;
    }
    if (h->ent->dev_stale) {
        HIP_TRY(h, hipDeviceSynchronize());                   // nothing in flight may still read the tables being replaced
        HIP_TRY(h, cdc::entropy_upload(h->ent.get(), &h->weight_allocs));
        h->ent_model_hash = cdc::entropy_model_hash(h->ent.get());
    }
    return CDC_OK;
}

inline uint32_t get_u32(const unsigned char *s) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)s[i] << (8 * i); return v; }
constexpr int kMaxHyperPositions = 1 << 22;       // hh * wh of a 131072 x 131072 image; bounds every allocation of the decoder
inline long long section_cap(long long n) { return (2 * n + 256 + 15) & ~15ll; }   // <= 2 renormalisation bytes per symbol + 64 states

// hyper_dec over the batch through the batch-1 launch plan: h->in_x (filled by the caller) -> dec_outs[0] = (mean | scale).
// *fault: results left the F16X2 range (checked only when `guard`).
int hyperdec_batch(cdc_handle *h, int B, hipStream_t st, bool guard, int *fault) {
    int rc;
    if (guard) { if ((rc = ensure_fault_flag(h))) return rc; HIP_TRY(h, hipMemsetAsync(h->d_fault, 0, sizeof(int), st)); }
    h->prof_now = false;
    for (const Op &op : h->ops)
        if ((rc = run_op(h, op, B, st))) return rc;
    const Act &o = h->dec_outs[0];
    const long long half = (long long)(o.C / 2) * o.H * o.W;
    *fault = 0;
    if (guard && (rc = guard_check(h, {{o.p, o.bs(), 2 * half}}, B, st, fault))) return rc;
    if (!*fault) HIP_TRY(h, clamp_min_launch(o.p + half, o.bs(), half, 0.1f, B, st));   // scale.clamp(min=0.1), compress_modules.py:59
    return CDC_OK;
}

int entropy_encode_impl(cdc_handle *h, const float *latent, const float *hyper_latent, const float *medians, int B,
                        int hh, int wh, unsigned char *out, size_t cap, size_t *offsets, int mem, void *stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!latent || !hyper_latent || !medians || !out || !offsets || B < 1 || hh < 1 || wh < 1 || hh > 65535 || wh > 65535 ||
        (long long)hh * wh > kMaxHyperPositions)
        return fail(h, CDC_ERR_INVALID, "null/invalid argument");
    if ((rc = ensure_entropy(h, medians))) return rc;
    hipStream_t st = h->own_stream;                       // synchronous entry point
    if (mem == CDC_MEM_DEVICE) HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    const int Ch = h->hyper_dims[0], per = hh * wh;
    const long long nh = (long long)Ch * per;
    if ((rc = build_hyperdec_program(h, B, hh, wh, true))) return rc;
    const Act &o = h->dec_outs[0];
    const long long nl = (long long)(o.C / 2) * o.H * o.W;
    if (nh > (1ll << 29) || nl > (1ll << 29)) return fail(h, CDC_ERR_INVALID, "image too large for one coder section");   // (section offsets are 32-bit)
    DevPool d;
    const float *d_hl = hyper_latent, *d_lat = latent;
    if (mem != CDC_MEM_DEVICE) {
        float *a, *b;
        HIP_TRY(h, d.get(&a, (size_t)B * nh)); HIP_TRY(h, d.get(&b, (size_t)B * nl));
        HIP_TRY(h, hipMemcpyAsync(a, hyper_latent, (size_t)B * nh * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(b, latent, (size_t)B * nl * sizeof(float), hipMemcpyHostToDevice, st));
        d_hl = a; d_lat = b;
    }
    int32_t *symh, *syml;
    uint8_t *bin, *sec_h, *sec_l, *packed;
    uint32_t *sf, *ew, *esc_h, *esc_l;
    int *bad;
    cdc::RansMeta *meta;
    long long *d_off;
    const long long cap_h = section_cap(nh), cap_l = section_cap(nl);
    const long long pack_cap = (long long)B * (kStreamHeader + cap_h + 4 * nh + cap_l + 4 * nl);
    HIP_TRY(h, d.get(&symh, (size_t)B * nh)); HIP_TRY(h, d.get(&syml, (size_t)B * nl)); HIP_TRY(h, d.get(&bin, (size_t)B * nl));
    HIP_TRY(h, d.get(&sf, (size_t)B * std::max(nh, nl))); HIP_TRY(h, d.get(&ew, (size_t)B * std::max(nh, nl)));
    HIP_TRY(h, d.get(&sec_h, (size_t)B * cap_h)); HIP_TRY(h, d.get(&sec_l, (size_t)B * cap_l));
    HIP_TRY(h, d.get(&esc_h, (size_t)B * nh)); HIP_TRY(h, d.get(&esc_l, (size_t)B * nl));
    HIP_TRY(h, d.get(&bad, 1)); HIP_TRY(h, d.get(&meta, 2 * (size_t)B)); HIP_TRY(h, d.get(&d_off, (size_t)B + 1));
    HIP_TRY(h, hipMemsetAsync(bad, 0, sizeof(int), st));
    // hyper symbols; their dequantised values are hyper_dec's input (quantize(.., "dequantize", medians), utils.py:72-85)
    HIP_TRY(h, cdc::hyper_symbols_launch(d_hl, h->ent->d_medians, Ch, per, B, symh, h->in_x, bad, st));
    int hbad = 0;                                         // (before hyper_dec: garbage input must not trip the range guard)
    HIP_TRY(h, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    if (hbad) return fail(h, CDC_ERR_INVALID, "non-finite or out-of-range hyper-latent (nothing to code)");
    int fault = 0;
    if ((rc = hyperdec_batch(h, B, st, guard_enabled(h) && !h->in_retry, &fault))) return rc;
    if (fault) {
        // the encoder may leave CDC_ARITH_F16X2 when hyper_dec overflows its range: the stream header records the arithmetic
        // that was finally used and the decoder runs what the header says
        if (guard_escalate(h, &rc)) { RetryScope r(h); return entropy_encode_impl(h, latent, hyper_latent, medians, B, hh, wh, out, cap, offsets, mem, stream); }
        if (rc) return rc;
    }
    HIP_TRY(h, cdc::latent_symbols_launch(d_lat, nl, o.p, o.p + nl, o.bs(), h->ent->d_edges, nl, B, syml, bin, bad, st));
    const cdc::EntropyDev T = h->ent->dev();
    HIP_TRY(h, cdc::rans_encode_launch(T, symh, nh, nullptr, 0, per, 0, (int)nh, 0u, B, sf, ew, sec_h, cap_h, esc_h, nh, meta, st));
    HIP_TRY(h, cdc::rans_encode_launch(T, syml, nl, bin, nl, 0, Ch, (int)nl, 1u, B, sf, ew, sec_l, cap_l, esc_l, nl, meta + B, st));
    const long long dev_cap = (long long)std::min<unsigned long long>((unsigned long long)cap, (unsigned long long)pack_cap);
    HIP_TRY(h, d.get(&packed, (size_t)dev_cap));
    cdc::RansPack P{sec_h, sec_l, esc_h, esc_l, meta, meta + B, cap_h, cap_l, nh, nl, dev_cap, packed, d_off, h->ent_model_hash, h->arith, hh, wh};
    HIP_TRY(h, cdc::rans_pack_launch(P, B, st));
    std::vector<long long> hoff((size_t)B + 1);
    HIP_TRY(h, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(hoff.data(), d_off, sizeof(long long) * ((size_t)B + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    if (hbad) {
        // (in the BF16X3 repetition of a range fault: the range was not the cause -- RetryScope puts the handle back into F16X2
        // and takes the fault off the count, as include/cdc_hip.h promises for every entry point)
        if (h->in_retry) h->retry_futile = true;
        return fail(h, CDC_ERR_INVALID, "non-finite or out-of-range latent, mean or scale (nothing to code)");
    }
    if ((unsigned long long)hoff[B] > (unsigned long long)cap)
        return fail(h, CDC_ERR_NOMEM, "bitstream buffer too small: %d image(s) need %lld bytes of %zu", B, hoff[B], cap);
    HIP_TRY(h, hipMemcpy(out, packed, (size_t)hoff[B], hipMemcpyDeviceToHost));
    for (int b = 0; b <= B; ++b) offsets[b] = (size_t)hoff[b];
    return CDC_OK;
}

int entropy_decode_impl(cdc_handle *h, const unsigned char *in, const size_t *offsets, const float *medians, int B,
                        float *q_latent, float *q_hyper_latent, int mem, void *stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!in || !offsets || !medians || !q_latent || B < 1) return fail(h, CDC_ERR_INVALID, "null/invalid argument");
    if ((rc = ensure_entropy(h, medians))) return rc;
    hipStream_t st = h->own_stream;
    // the outputs may be device buffers that queued work of the caller's stream still uses
    if (mem == CDC_MEM_DEVICE) HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    const int Ch = h->hyper_dims[0];
    // the decoder runs hyper_dec in the encoder's arithmetic (see the contract in entropy.hip); the handle's own mode comes back
    struct Restore { cdc_handle *h; int a; ~Restore() { if (h->arith != a) (void)cdc_set_arith(h, a); } } restore{h, h->arith};
    // ---- headers: everything that sizes an allocation is validated here ----
    struct Hdr { int hh, wh, ar; uint32_t nbh, nbl, sum, eh, el; };
    std::vector<Hdr> hd((size_t)B);
    for (int b = 0; b < B; ++b) {
        if (offsets[b + 1] < offsets[b]) return fail(h, CDC_ERR_INVALID, "image %d: offsets decrease", b);
        const unsigned char *s = in + offsets[b];
        const size_t n = offsets[b + 1] - offsets[b];
        Hdr &q = hd[b];
        if (cdc_entropy_peek(s, n, &q.hh, &q.wh, &q.ar)) return fail(h, CDC_ERR_INVALID, "image %d: not a CDC bitstream (version %d container)", b, kStreamVersion);
        if (q.hh < 1 || q.wh < 1 || (long long)q.hh * q.wh > std::min(kMaxHyperPositions, h->ent_max_positions))
            return fail(h, CDC_ERR_INVALID, "image %d: hyper-latent size %d x %d in the stream header exceeds the decoder's limit of %d positions "
                        "(cdc_entropy_set_limit)", b, q.hh, q.wh, std::min(kMaxHyperPositions, h->ent_max_positions));
        if (q.ar != CDC_ARITH_BF16X3 && q.ar != CDC_ARITH_F16X2) return fail(h, CDC_ERR_INVALID, "image %d: unknown arithmetic %d", b, q.ar);
        q.nbh = get_u32(s + 10); q.nbl = get_u32(s + 14); q.sum = get_u32(s + 22); q.eh = get_u32(s + 26); q.el = get_u32(s + 30);
        if ((unsigned long long)kStreamHeader + q.nbh + q.nbl != n) return fail(h, CDC_ERR_INVALID, "image %d: truncated bitstream", b);
        if (get_u32(s + 18) != h->ent_model_hash)
            return fail(h, CDC_ERR_INVALID, "image %d: the stream was coded with other probability tables (prior parameters, medians, library build or libm differ)", b);
        if (4ull * q.eh + 256 > q.nbh || 4ull * q.el + 256 > q.nbl) return fail(h, CDC_ERR_INVALID, "image %d: corrupt section sizes", b);
        if (q.hh != hd[0].hh || q.wh != hd[0].wh)
            return fail(h, CDC_ERR_INVALID, "image %d: %d x %d hyper-latent in a batch of %d x %d (one call decodes one image size)", b, q.hh, q.wh, hd[0].hh, hd[0].wh);
    }
    const int hh = hd[0].hh, wh = hd[0].wh, per = hh * wh;
    const long long nh = (long long)Ch * per;
    // the whole input goes to the device once (+ slack: nothing reads past the end, but sections are addressed by offset)
    const size_t total = offsets[B] - offsets[0];
    DevPool d;
    uint8_t *d_in;
    HIP_TRY(h, d.get(&d_in, total + 16));
    HIP_TRY(h, hipMemcpyAsync(d_in, in + offsets[0], total, hipMemcpyHostToDevice, st));
    // images that share an arithmetic decode together (normally all of them)
    for (int b0 = 0; b0 < B;) {
        int b1 = b0 + 1;
        while (b1 < B && hd[b1].ar == hd[b0].ar) ++b1;
        const int nb = b1 - b0;
        if (hd[b0].ar != h->arith && (rc = cdc_set_arith(h, hd[b0].ar))) return rc;
        if ((rc = build_hyperdec_program(h, nb, hh, wh, true))) return rc;
        const Act &o = h->dec_outs[0];
        const long long nl = (long long)(o.C / 2) * o.H * o.W;
        if (nh > (1ll << 29) || nl > (1ll << 29)) return fail(h, CDC_ERR_INVALID, "image too large for one coder section");
        std::vector<long long> off(2 * (size_t)nb);
        std::vector<int> len(2 * (size_t)nb), esc(2 * (size_t)nb);
        for (int b = b0; b < b1; ++b) {
            const long long base = (long long)(offsets[b] - offsets[0]) + kStreamHeader;
            off[b - b0] = base; len[b - b0] = (int)hd[b].nbh; esc[b - b0] = (int)hd[b].eh;
            off[nb + b - b0] = base + hd[b].nbh; len[nb + b - b0] = (int)hd[b].nbl; esc[nb + b - b0] = (int)hd[b].el;
        }
        long long *d_off;
        int *d_len, *d_esc;
        int32_t *symh, *syml;
        uint8_t *bin;
        cdc::RansMeta *meta;
        float *ql = nullptr;
        HIP_TRY(h, d.get(&d_off, 2 * (size_t)nb)); HIP_TRY(h, d.get(&d_len, 2 * (size_t)nb)); HIP_TRY(h, d.get(&d_esc, 2 * (size_t)nb));
        HIP_TRY(h, d.get(&symh, (size_t)nb * nh)); HIP_TRY(h, d.get(&syml, (size_t)nb * nl)); HIP_TRY(h, d.get(&bin, (size_t)nb * nl));
        HIP_TRY(h, d.get(&meta, 2 * (size_t)nb));
        HIP_TRY(h, hipMemcpyAsync(d_off, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(d_len, len.data(), sizeof(int) * len.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(d_esc, esc.data(), sizeof(int) * esc.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemsetAsync(meta, 0, sizeof(cdc::RansMeta) * 2 * nb, st));
        const cdc::EntropyDev T = h->ent->dev();
        HIP_TRY(h, cdc::rans_decode_launch(T, d_in, d_off, d_len, d_esc, nullptr, 0, per, 0, (int)nh, 0u, nb, symh, nh, meta, st));
        HIP_TRY(h, cdc::symbols_to_hyper_launch(symh, h->ent->d_medians, Ch, per, nb, h->in_x, st));
        int fault = 0;
        if ((rc = hyperdec_batch(h, nb, st, false, &fault))) return rc;
        HIP_TRY(h, cdc::latent_symbols_launch(nullptr, 0, o.p, o.p + nl, o.bs(), h->ent->d_edges, nl, nb, nullptr, bin, nullptr, st));
        HIP_TRY(h, cdc::rans_decode_launch(T, d_in, d_off + nb, d_len + nb, d_esc + nb, bin, nl, 0, Ch, (int)nl, 1u, nb, syml, nl, meta + nb, st));
        std::vector<cdc::RansMeta> hm(2 * (size_t)nb);
        HIP_TRY(h, hipMemcpyAsync(hm.data(), meta, sizeof(cdc::RansMeta) * hm.size(), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        for (int b = b0; b < b1; ++b) {
            if (hm[b - b0].bad) return fail(h, CDC_ERR_INVALID, "image %d: corrupt hyper stream", b);
            if (hm[nb + b - b0].bad)
                return fail(h, CDC_ERR_INVALID, "image %d: corrupt latent stream (or this decoder's hyper-decoder output differs from the encoder's)", b);
            if (hm[b - b0].checksum + hm[nb + b - b0].checksum != hd[b].sum)
                return fail(h, CDC_ERR_INVALID, "image %d: symbol checksum mismatch -- this decoder's hyper-decoder output differs from the encoder's "
                                               "(other library build, development switches or GPU), or the payload is corrupt", b);
        }
        float *dst_l = q_latent + (size_t)b0 * nl, *dst_h = q_hyper_latent ? q_hyper_latent + (size_t)b0 * nh : nullptr;
        if (mem != CDC_MEM_DEVICE) { HIP_TRY(h, d.get(&ql, (size_t)nb * nl)); }
        HIP_TRY(h, cdc::symbols_to_latent_launch(syml, o.p, o.bs(), nl, nb, mem == CDC_MEM_DEVICE ? dst_l : ql, st));
        if (mem != CDC_MEM_DEVICE) HIP_TRY(h, hipMemcpyAsync(dst_l, ql, (size_t)nb * nl * sizeof(float), hipMemcpyDeviceToHost, st));
        if (dst_h) {
            HIP_TRY(h, hipMemcpyAsync(dst_h, h->in_x, (size_t)nb * nh * sizeof(float),
                                      mem == CDC_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(h, hipStreamSynchronize(st));
        b0 = b1;
    }
    return CDC_OK;
}

}  // namespace

int cdc_entropy_encode(cdc_handle *h, const float *latent, const float *hyper_latent, const float *medians, int B,
                       int hh, int wh, unsigned char *out, size_t cap, size_t *offsets, int mem, void *stream) {
    return no_throw(h, [&] { return entropy_encode_impl(h, latent, hyper_latent, medians, B, hh, wh, out, cap, offsets, mem, stream); });
}

int cdc_entropy_set_limit(cdc_handle *h, int max_hyper_positions) {
    if (!h || h->kind != 2) return h ? fail(h, CDC_ERR_STATE, "handle is not a hyper decoder") : CDC_ERR_INVALID;
    if (max_hyper_positions < 1) return fail(h, CDC_ERR_INVALID, "limit %d", max_hyper_positions);
    h->ent_max_positions = std::min(max_hyper_positions, kMaxHyperPositions);
    return CDC_OK;
}

int cdc_entropy_peek(const unsigned char *in, size_t n, int *hh, int *wh, int *arith) {
    if (!in || n < (size_t)kStreamHeader || in[0] != 'C' || in[1] != 'D' || in[2] != 'C' || in[3] != kStreamVersion) return CDC_ERR_INVALID;
    if (arith) *arith = in[4];
    if (hh) *hh = in[6] | (in[7] << 8);
    if (wh) *wh = in[8] | (in[9] << 8);
    return CDC_OK;
}

int cdc_entropy_decode(cdc_handle *h, const unsigned char *in, const size_t *offsets, const float *medians, int B,
                       float *q_latent, float *q_hyper_latent, int mem, void *stream) {
    return no_throw(h, [&] { return entropy_decode_impl(h, in, offsets, medians, B, q_latent, q_hyper_latent, mem, stream); });
}

int cdc_dequantize(cdc_handle *h, const float *x, const float *offset, float *out, long long n, int mem, void *stream) {
    if (!h) return CDC_ERR_INVALID;
    int rc = ensure_device(h);
    if (rc) return rc;
    if (!x || !offset || !out || n < 1) return fail(h, CDC_ERR_INVALID, "null/invalid argument");
    hipStream_t st = pick_stream(h, stream, mem);
    if (mem == CDC_MEM_DEVICE) {
        HIP_TRY(h, dequantize_launch(x, offset, out, n, st));
        return CDC_OK;
    }
    float *dx = nullptr, *dl = nullptr;
    HIP_TRY(h, hipMalloc(&dx, sizeof(float) * n));
    if (hipMalloc(&dl, sizeof(float) * n) != hipSuccess) { (void)hipFree(dx); return fail(h, CDC_ERR_NOMEM, "hipMalloc failed"); }
    hipError_t e = hipMemcpyAsync(dx, x, sizeof(float) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dl, offset, sizeof(float) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = dequantize_launch(dx, dl, dx, n, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dx, sizeof(float) * n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dx); (void)hipFree(dl);
    if (e != hipSuccess) return fail(h, CDC_ERR_HIP, "dequantize: %s", hipGetErrorString(e));
    return CDC_OK;
}

int cdc_ctxdec_create(const cdc_ctxdec_config *cfg, int device, cdc_handle **out) {
    if (!cfg || !out) return fail(nullptr, CDC_ERR_INVALID, "null argument");
    if (cfg->dim <= 0 || cfg->n_rev_mults < 1 || cfg->n_rev_mults > CDC_MAX_LEVELS || cfg->out_channels < 1 ||
        cfg->up_index < 1 || cfg->up_index > 2)
        return fail(nullptr, CDC_ERR_INVALID, "bad cdc_ctxdec_config");
    if (device < 0) return fail(nullptr, CDC_ERR_INVALID, "device %d out of range", device);
    std::unique_ptr<cdc_handle> h(new cdc_handle);
    memset(&h->cfg, 0, sizeof h->cfg);
    h->cfg.dim = cfg->dim;
    h->kind = 1;
    h->device = device;
    h->up_index = cfg->up_index;
    for (int i = 0; i < cfg->n_rev_mults; ++i) {
        if (cfg->rev_mults[i] < 1) return fail(nullptr, CDC_ERR_INVALID, "bad cdc_ctxdec_config");
        h->rev_dims.push_back(cfg->dim * cfg->rev_mults[i]);
    }
    h->rev_dims.push_back(cfg->out_channels);
    const int n = cfg->n_rev_mults;
    for (int i = 0; i < n; ++i) {       // registration order of Compressor.dec (compress_modules.py:147-156)
        const std::string p = "dec." + std::to_string(i);
        const int din = h->rev_dims[i], dout = h->rev_dims[i + 1], dmid = i == n - 1 ? din : dout;
        add_resblock_params(h.get(), p + ".0", din, dmid, 3, false);
        const std::string u = p + "." + std::to_string(cfg->up_index);
        add_param(h.get(), u + ".conv.weight", {dmid, dout, 4, 4});     // ConvTranspose2d: [Cin][Cout][4][4]
        add_param(h.get(), u + ".conv.bias", {dout});
    }
    *out = h.release();
    return CDC_OK;
}

int cdc_ctxdec_decode(cdc_handle *h, const float *q_latent, float *const *outs, int n_outs, int B,
                      int hl, int wl, int mem, void *stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->kind != 1) return fail(h, CDC_ERR_STATE, "handle is not a context decoder");
    const int n = (int)h->rev_dims.size() - 1;
    if (!q_latent || !outs || n_outs != n || B < 1 || hl < 1 || wl < 1)
        return fail(h, CDC_ERR_INVALID, "null/invalid argument (the decoder has %d outputs)", n);
    for (int i = 0; i < n; ++i)
        if (!outs[i]) return fail(h, CDC_ERR_INVALID, "null output %d", i);
    if ((rc = build_ctxdec_program(h, B, hl, wl))) return rc;
    hipStream_t st = pick_stream(h, stream, mem);
    if ((rc = copy_in(h, h->in_x, q_latent, (size_t)B * h->rev_dims[0] * hl * wl, mem, st))) return rc;
    const bool guard = guard_enabled(h);
    if (guard) { if ((rc = ensure_fault_flag(h))) return rc; HIP_TRY(h, hipMemsetAsync(h->d_fault, 0, sizeof(int), st)); }
    h->prof_now = true;
    for (const Op &op : h->ops)
        if ((rc = run_op(h, op, h->pB, st))) return rc;
    if (guard) {
        for (int i = 0; i < n; ++i) {
            const Act &a = h->dec_outs[i];
            HIP_TRY(h, cdc::nonfinite_launch(a.p, a.bs(), (long long)a.C * a.H * a.W, B, h->d_fault, st));
        }
        int fault = 0;
        if ((rc = guard_check(h, {}, B, st, &fault))) return rc;
        if (fault) {
            if (guard_escalate(h, &rc)) { RetryScope r(h); return cdc_ctxdec_decode(h, q_latent, outs, n_outs, B, hl, wl, mem, stream); }
            if (rc) return rc;
        }
    }
    for (int i = 0; i < n; ++i) {       // outs[0] = finest = the last level's output (output[::-1])
        const Act &a = h->dec_outs[n - 1 - i];
        if ((rc = copy_out(h, outs[i], a.p, (size_t)B * a.C * a.H * a.W, mem, st))) return rc;
    }
    return CDC_OK;
}

int cdc_unet_forward(cdc_handle *h, const float *x, const float *time, const float *const *ctx,
                     int n_ctx, float *out, int B, int H, int W, int mem, void *stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->kind != 0) return fail(h, CDC_ERR_STATE, "handle is not a U-Net");
    if (!x || !time || !out || B < 1) return fail(h, CDC_ERR_INVALID, "null/invalid argument");
    if ((rc = build_program(h, B, H, W))) return rc;
    hipStream_t st = pick_stream(h, stream, mem);
    if ((rc = copy_in(h, h->in_x, x, (size_t)B * h->cfg.channels * H * W, mem, st))) return rc;
    if ((rc = copy_in(h, h->in_time, time, B, mem, st))) return rc;
    if ((rc = stage_ctx(h, ctx, n_ctx, B, mem, st))) return rc;
    const bool guard = guard_enabled(h);
    if (guard) { if ((rc = ensure_fault_flag(h))) return rc; HIP_TRY(h, hipMemsetAsync(h->d_fault, 0, sizeof(int), st)); }
    h->prof_now = true;
    if ((rc = run_pre(h, st))) return rc;
    if ((rc = run_unet(h, st, -1))) return rc;
    if (guard) {
        int fault = 0;
        if ((rc = guard_check(h, {{h->out_fx, 0, (long long)B * h->out_dim * H * W}}, 1, st, &fault))) return rc;
        if (fault) {
            if (guard_escalate(h, &rc)) { RetryScope r(h); return cdc_unet_forward(h, x, time, ctx, n_ctx, out, B, H, W, mem, stream); }
            if (rc) return rc;
        }
    }
    return copy_out(h, out, h->out_fx, (size_t)B * h->out_dim * H * W, mem, st);
}

int cdc_unet_tap(cdc_handle *h, const char *name, float *out, int64_t shape[4]) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!name || !shape) return fail(h, CDC_ERR_INVALID, "null argument");
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return fail(h, CDC_ERR_INVALID, "no tap named '%s' in the current program", name);
    const Act &a = it->second;
    shape[0] = h->pB; shape[1] = a.C; shape[2] = a.H; shape[3] = a.W;
    if (out) {
        HIP_TRY(h, hipDeviceSynchronize());
        if (a.pf) {         // a planes-only tensor: h + l 2^-11 into its (otherwise unwritten) fp32 buffer
            HIP_TRY(h, pf_unpack_launch(a.pf, a.pf_bs, a.p, a.bs(), a.C, a.H, a.W, h->pB, nullptr));
            HIP_TRY(h, hipDeviceSynchronize());
        }
        HIP_TRY(h, hipMemcpy(out, a.p, (size_t)h->pB * a.bs() * sizeof(float), hipMemcpyDeviceToHost));
    }
    return CDC_OK;
}

int cdc_set_schedule(cdc_handle *h, int steps, const float *time_in, const float *sqrt_recip,
                     const float *sqrt_recipm1, const float *sqrt_ac_prev,
                     const float *one_minus_ac_prev, const float *sigma) {
    if (h && h->kind != 0) return fail(h, CDC_ERR_STATE, "handle is not a U-Net");
    if (!h || steps < 1 || !time_in || !sqrt_recip || !sqrt_recipm1 || !sqrt_ac_prev ||
        !one_minus_ac_prev || !sigma)
        return fail(h, CDC_ERR_INVALID, "null/invalid argument");
    int rc0 = ensure_device(h);
    if (rc0) return rc0;
    std::vector<float> tab((size_t)5 * steps);
    const float *srcs[5] = {sqrt_recip, sqrt_recipm1, sqrt_ac_prev, one_minus_ac_prev, sigma};
    for (int k = 0; k < 5; ++k) memcpy(&tab[(size_t)k * steps], srcs[k], sizeof(float) * steps);
    // compress() / decompress() set the schedule on every call: the same tables again change nothing
    if (h->d_tab && h->steps == steps && tab == h->h_tab && (int)h->h_time_in.size() == steps &&
        memcmp(h->h_time_in.data(), time_in, sizeof(float) * steps) == 0)
        return CDC_OK;
    HIP_TRY(h, hipDeviceSynchronize());
    if (tab.size() > h->tab_cap) {       // grow only; the buffer keeps its address otherwise
        if (h->d_tab) { (void)hipFree(h->d_tab); h->d_tab = nullptr; }
        HIP_TRY(h, hipMalloc((void **)&h->d_tab, tab.size() * sizeof(float)));
        h->tab_cap = tab.size();
    }
    HIP_TRY(h, hipMemcpy(h->d_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    h->h_tab.swap(tab);
    h->h_time_in.assign(time_in, time_in + steps);
    h->steps = steps;
    h->time_steps_B = 0;     // forces re-evaluation of the per-step time rows
    ++h->sched_gen;
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    return CDC_OK;
}

// All images of a decode share the step's time value, and the time-embedding MLPs (unet.py:41,
// network_components.py:96-100) depend on nothing else: evaluate them for every sample step in one
// launch (one workgroup per step) and broadcast row i at iteration i.
static int ensure_time_rows(cdc_handle *h, int B) {
    if (h->d_shift_tab && h->time_steps_B == B) return CDC_OK;
    const size_t need = (size_t)h->steps * (h->shift_bs + 1);
    if (need > h->trows_cap) {           // owned by the handle (not by the launch program): reused across schedules
        if (h->d_time_steps) (void)hipFree(h->d_time_steps);
        if (h->d_shift_tab) (void)hipFree(h->d_shift_tab);
        h->d_time_steps = h->d_shift_tab = nullptr;
        h->trows_cap = 0;
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }   // it baked the old addresses
        void *p = nullptr, *q = nullptr;
        HIP_TRY(h, hipMalloc(&p, (size_t)h->steps * sizeof(float)));
        h->d_time_steps = (float *)p;
        HIP_TRY(h, hipMalloc(&q, (size_t)h->steps * h->shift_bs * sizeof(float)));
        h->d_shift_tab = (float *)q;
        h->trows_cap = need;
    }
    HIP_TRY(h, hipMemcpy(h->d_time_steps, h->h_time_in.data(), (size_t)h->steps * sizeof(float), hipMemcpyHostToDevice));
    h->time_steps_B = B;
    TembArgs t;
    t.time = h->d_time_steps; t.w0 = h->tm_w0; t.b0 = h->tm_b0; t.w2 = h->tm_w2; t.b2 = h->tm_b2;
    t.dim = h->cfg.dim; t.layers = h->d_temb_layers; t.n_layers = (int)h->rbs.size();
    t.shift = h->d_shift_tab; t.shift_bs = h->shift_bs;
    HIP_TRY(h, temb_launch(t, h->steps, h->own_stream));
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    return CDC_OK;
}

int cdc_set_schedule_v(cdc_handle *h, int steps, const float *sqrt_ac, const float *sqrt_one_minus_ac) {
    if (h && h->kind != 0) return fail(h, CDC_ERR_STATE, "handle is not a U-Net");
    if (!h || !sqrt_ac || !sqrt_one_minus_ac) return fail(h, CDC_ERR_INVALID, "null argument");
    if (!h->steps || steps != h->steps) return fail(h, CDC_ERR_STATE, "cdc_set_schedule_v follows cdc_set_schedule with the same number of steps");
    int rc0 = ensure_device(h);
    if (rc0) return rc0;
    std::vector<float> tab((size_t)2 * steps);
    memcpy(&tab[0], sqrt_ac, sizeof(float) * steps);
    memcpy(&tab[steps], sqrt_one_minus_ac, sizeof(float) * steps);
    HIP_TRY(h, hipDeviceSynchronize());
    if (steps > h->tab_v_steps) {
        if (h->d_tab_v) { (void)hipFree(h->d_tab_v); h->d_tab_v = nullptr; }
        HIP_TRY(h, hipMalloc((void **)&h->d_tab_v, tab.size() * sizeof(float)));
        h->tab_v_steps = steps;
    }
    HIP_TRY(h, hipMemcpy(h->d_tab_v, tab.data(), sizeof(float) * steps, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_tab_v + h->steps, tab.data() + steps, sizeof(float) * steps, hipMemcpyHostToDevice));
    h->tab_v_gen = h->sched_gen;
    return CDC_OK;
}

static int ddim_on_device(cdc_handle *h, const float *x_in, int i, const float *noise, float eta,
                          float *x_out, int B, int H, int W, int pred_mode, int clip,
                          hipStream_t st) {
    int rc;
    const size_t n = (size_t)B * h->cfg.channels * H * W;
    if (x_in != h->in_x)
        HIP_TRY(h, hipMemcpyAsync(h->in_x, x_in, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    // the 7-row combine of the final convolution rides in the sampler kernel (one launch and a 3-channel tensor less per iteration)
    static const bool fuse_combine = getenv("CDC_NO_COMBINE_FUSE") == nullptr;
    const Op *cb = nullptr;
    if (fuse_combine && !h->ops.empty() && h->ops.back().kind == Op::COMBINE && h->ops.back().cb.out == h->out_fx) cb = &h->ops.back();
    if ((rc = run_unet(h, st, i, cb != nullptr))) return rc;
    Op op;
    op.kind = Op::DDIM; op.prof = PC_SMALL;
    op.ddim = {h->out_fx, h->in_x, (eta != 0.f) ? noise : nullptr, x_out, h->d_tab, h->steps, i < 0 ? 0 : i,
               i == -2 ? h->d_step : nullptr, pred_mode, clip, eta, (long long)n,
               (long long)(B / 2) * h->cfg.channels * H * W, h->d_fault, pred_mode == CDC_PRED_V ? h->d_tab_v : nullptr};
    if (cb) {
        op.ddim.P = cb->cb.P; op.ddim.P_bias = cb->cb.bias; op.ddim.pC = cb->cb.Cout; op.ddim.pKH = cb->cb.KH; op.ddim.pPad = cb->cb.pad;
        op.ddim.pH = cb->cb.H; op.ddim.pW = cb->cb.W;
    }
    op.bytes = 16.0 * n;
    return run_op(h, op, B, st);
}

int cdc_ddim_step(cdc_handle *h, const float *x_in, int i, const float *const *ctx, int n_ctx,
                  const float *noise, float eta, float *x_out, int B, int H, int W, int pred_mode,
                  int clip, int mem, void *stream) {
    if (h && h->kind != 0) return fail(h, CDC_ERR_STATE, "handle is not a U-Net");
    int rc = check_ready(h);
    if (rc) return rc;
    if (!h->steps) return fail(h, CDC_ERR_STATE, "cdc_set_schedule has not been called");
    if (i < 0 || i >= h->steps) return fail(h, CDC_ERR_INVALID, "step index %d out of [0,%d)", i, h->steps);
    if (h->out_dim != h->cfg.channels)
        return fail(h, CDC_ERR_UNSUPPORTED, "sampler needs out_dim == channels");
    if (eta != 0.f && !noise) return fail(h, CDC_ERR_INVALID, "eta != 0 needs the noise draw");
    if (pred_mode < 0 || pred_mode > 3 || clip < 0 || clip > 2) return fail(h, CDC_ERR_INVALID, "pred_mode %d / clip %d out of range", pred_mode, clip);
    if (pred_mode == CDC_PRED_V && (!h->d_tab_v || h->tab_v_gen != h->sched_gen))
        return fail(h, CDC_ERR_STATE, "pred_mode \"v\" needs cdc_set_schedule_v after cdc_set_schedule");
    if ((rc = build_program(h, B, H, W))) return rc;
    if ((rc = ensure_time_rows(h, B))) return rc;
    hipStream_t st = pick_stream(h, stream, mem);
    const size_t n = (size_t)B * h->cfg.channels * H * W;
    if ((rc = copy_in(h, h->in_x, x_in, n, mem, st))) return rc;
    // the flag is cleared BEFORE the hoisted context convolutions run: they report range faults into it too (as in cdc_decode)
    if ((rc = ensure_fault_flag(h))) return rc;              // (the sampler kernel writes the flag)
    const bool guard = guard_enabled(h);
    HIP_TRY(h, hipMemsetAsync(h->d_fault, 0, sizeof(int), st));
    if (ctx) {
        if ((rc = stage_ctx(h, ctx, n_ctx, B, mem, st))) return rc;
        h->prof_now = true;
        if ((rc = run_pre(h, st))) return rc;
    }
    if (eta != 0.f && (rc = copy_in(h, h->noise_buf, noise, n, mem, st))) return rc;
    if ((rc = ddim_on_device(h, h->in_x, i, h->noise_buf, eta, h->xa, B, H, W, pred_mode, clip, st)))
        return rc;
    if (guard) {
        int fault = 0;
        if ((rc = guard_check(h, {{h->xa, 0, (long long)n}}, 1, st, &fault))) return rc;
        if (fault) {
            const bool esc = guard_escalate(h, &rc);
            if (esc && !ctx)            // the staged context went with the old launch program: the caller has to hand it over again
                return fail(h, CDC_ERR_STATE, "fp16 range overflow in step %d; the handle is now in CDC_ARITH_BF16X3 -- repeat the step WITH the context", i);
            if (esc) { RetryScope r(h); return cdc_ddim_step(h, x_in, i, ctx, n_ctx, noise, eta, x_out, B, H, W, pred_mode, clip, mem, stream); }
            if (rc) return rc;
        }
    }
    return copy_out(h, x_out, h->xa, n, mem, st);
}

int cdc_decode(cdc_handle *h, const float *init, const float *const *ctx, int n_ctx, float *out, int B,
               int H, int W, int pred_mode, int clip, int mem, void *stream) {
    if (h && h->kind != 0) return fail(h, CDC_ERR_STATE, "handle is not a U-Net");
    int rc = check_ready(h);
    if (rc) return rc;
    if (!h->steps) return fail(h, CDC_ERR_STATE, "cdc_set_schedule has not been called");
    if (h->out_dim != h->cfg.channels)
        return fail(h, CDC_ERR_UNSUPPORTED, "sampler needs out_dim == channels");
    if (!out || !ctx) return fail(h, CDC_ERR_INVALID, "null argument");
    if (pred_mode < 0 || pred_mode > 3 || clip < 0 || clip > 2)
        return fail(h, CDC_ERR_INVALID, "pred_mode %d / clip %d out of range", pred_mode, clip);
    if (pred_mode == CDC_PRED_V && (!h->d_tab_v || h->tab_v_gen != h->sched_gen))
        return fail(h, CDC_ERR_STATE, "pred_mode \"v\" needs cdc_set_schedule_v after cdc_set_schedule");
    if ((rc = build_program(h, B, H, W))) return rc;
    if ((rc = ensure_time_rows(h, B))) return rc;
    hipStream_t st = pick_stream(h, stream, mem);
    const size_t n = (size_t)B * h->cfg.channels * H * W;
    if ((rc = ensure_fault_flag(h))) return rc;
    HIP_TRY(h, hipMemsetAsync(h->d_fault, 0, sizeof(int), st));
    if (init) { if ((rc = copy_in(h, h->in_x, init, n, mem, st))) return rc; }
    else HIP_TRY(h, hipMemsetAsync(h->in_x, 0, n * sizeof(float), st));
    if ((rc = stage_ctx(h, ctx, n_ctx, B, mem, st))) return rc;
    h->prof_now = false;
    if ((rc = run_pre(h, st))) return rc;       // hoisted context halves: once per decode
    // for i in reversed(range(steps)): img = ddim(img, i)      (x: :188-200 ; eps: :174-190)
    // Optional (CDC_GRAPH=1): replay one captured DDIM iteration as a hipGraph; the step index lives in device memory
    // and is decremented by the graph's last node.  Measured (tools/gpu_graph_latency.py): bit-identical, and NO
    // faster -- 5.8 ms / iteration at batch 1 either way: the ~170 kernels of an iteration are bound by their own
    // serial latency (tiny grids), not by host launches -- so the eager loop stays the default.
    const char *genv = getenv("CDC_GRAPH");           // 0 / 1 overrides the batch-size rule
    const bool use_graph = !h->prof && h->steps > 2 && genv && atoi(genv) != 0;
    int i = h->steps - 1;
    if (use_graph) {
        // the legacy default stream cannot be captured: iterate on the library's own stream, fenced by events
        hipStream_t cs = st;
        if (!h->gev_in) { HIP_TRY(h, hipEventCreateWithFlags(&h->gev_in, hipEventDisableTiming));
                          HIP_TRY(h, hipEventCreateWithFlags(&h->gev_out, hipEventDisableTiming)); }
        if (st != h->own_stream) {
            HIP_TRY(h, hipEventRecord(h->gev_in, cs));
            st = h->own_stream;
            HIP_TRY(h, hipStreamWaitEvent(st, h->gev_in, 0));
        }
        if (!h->d_step) { void *p = nullptr; HIP_TRY(h, hipMalloc(&p, sizeof(int))); h->d_step = (int *)p; h->weight_allocs.push_back(p); }
        // first iteration eagerly (kernel attributes, code pages), then capture the second and replay it
        if ((rc = ddim_on_device(h, h->in_x, i, nullptr, 0.f, h->in_x, B, H, W, pred_mode, clip, st))) return rc;
        --i;
        const int key[4] = {h->steps, pred_mode, clip, h->sched_gen};
        if (!h->graph_exec || memcmp(key, h->graph_key, sizeof key)) {
            if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            HIP_TRY(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            rc = ddim_on_device(h, h->in_x, -2, nullptr, 0.f, h->in_x, B, H, W, pred_mode, clip, st);
            hipError_t e = rc ? hipSuccess : step_dec_launch(h->d_step, st);
            hipError_t e2 = hipStreamEndCapture(st, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (e != hipSuccess || e2 != hipSuccess || !g)
                return fail(h, CDC_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
            e = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) { h->graph_exec = nullptr; return fail(h, CDC_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
            memcpy(h->graph_key, key, sizeof key);
        }
        HIP_TRY(h, hipMemcpyAsync(h->d_step, &i, sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipStreamSynchronize(st));      // `i` is a stack variable
        for (; i >= 0; --i) HIP_TRY(h, hipGraphLaunch(h->graph_exec, st));
        if (st != cs) {
            HIP_TRY(h, hipEventRecord(h->gev_out, st));
            HIP_TRY(h, hipStreamWaitEvent(cs, h->gev_out, 0));
            st = cs;
        }
    }
    for (; i >= 0; --i) {
        h->prof_now = (i % h->prof_every) == 0;
        if ((rc = ddim_on_device(h, h->in_x, i, nullptr, 0.f, h->in_x, B, H, W, pred_mode, clip, st)))
            return rc;
    }
    h->prof_now = true;
    // Range guard (see guard_enabled above): a non-finite U-Net output is flagged by the sampler kernel of the iteration it
    // occurs in; the final image is checked as well.  One 4-byte read-back per decode.
    if (guard_enabled(h)) {
        int fault = 0;
        if ((rc = guard_check(h, {{h->in_x, 0, (long long)n}}, 1, st, &fault))) return rc;
        if (fault) {
            if (guard_escalate(h, &rc)) { RetryScope r(h); return cdc_decode(h, init, ctx, n_ctx, out, B, H, W, pred_mode, clip, mem, stream); }
            if (rc) return rc;
        }
    }
    return copy_out(h, out, h->in_x, n, mem, st);
}

int cdc_prof_enable(cdc_handle *h, int on) {
    if (!h) return CDC_ERR_INVALID;
    h->prof = on != 0;
    h->prof_every = on > 1 ? on : 1;     // on = n > 1: sample the DDIM iterations with i % n == 0
    h->prof_now = true;
    return CDC_OK;
}
int cdc_prof_num_classes(void) { return PC_COUNT; }
const char *cdc_prof_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }
int cdc_prof_get(cdc_handle *h, int cls, double *ms, int64_t *launches, double *flops, double *bytes) {
    if (!h || cls < 0 || cls >= PC_COUNT) return CDC_ERR_INVALID;
    int rc = resolve_pending(h);
    if (rc) return rc;
    if (cls == 0 && getenv("CDC_PROF_OPS")) {        // per-op table of the instrumented iterations
        for (size_t i = 0; i < h->op_ms.size(); ++i)
            if (h->op_n[i])
                fprintf(stderr, "[op %3zu] %8.3f ms  %7.1f TF  x%ld  %s\n", i, h->op_ms[i] / h->op_n[i],
                        h->op_flops[i] / (h->op_ms[i] / h->op_n[i] * 1e-3) / 1e12, h->op_n[i],
                        h->op_label[i].c_str());
    }
    if (ms) *ms = h->prof_ms[cls];
    if (launches) *launches = h->prof_launches[cls];
    if (flops) *flops = h->prof_flops[cls];
    if (bytes) *bytes = h->prof_bytes[cls];
    return CDC_OK;
}
int cdc_prof_num_ops(cdc_handle *h) { return h ? (int)h->op_ms.size() : CDC_ERR_INVALID; }
int cdc_prof_op(cdc_handle *h, int idx, const char **label, double *ms, int64_t *launches, double *flops) {
    if (!h || idx < 0 || idx >= (int)h->op_ms.size()) return CDC_ERR_INVALID;
    int rc = resolve_pending(h);
    if (rc) return rc;
    if (label) *label = h->op_label[idx].c_str();
    if (ms) *ms = h->op_ms[idx];
    if (launches) *launches = h->op_n[idx];
    if (flops) *flops = h->op_flops[idx];
    return CDC_OK;
}
int cdc_prof_reset(cdc_handle *h) {
    if (!h) return CDC_ERR_INVALID;
    (void)resolve_pending(h);
    std::fill(h->op_ms.begin(), h->op_ms.end(), 0.0);
    std::fill(h->op_n.begin(), h->op_n.end(), 0L);
    for (int i = 0; i < PC_COUNT; ++i) {
        h->prof_ms[i] = h->prof_flops[i] = h->prof_bytes[i] = 0;
        h->prof_launches[i] = 0;
    }
    return CDC_OK;
}

}  // extern "C"

// ---- single operators ----------------------------------------------------------------------------
namespace {

constexpr int kOpRetry = -10000;  // internal: repeat the operator in CDC_ARITH_BF16X3 (never returned to the caller)
template <class F> int op_with_guard(cdc_handle *h, F &&f) {
    int rc = f();
    if (rc == kOpRetry) { RetryScope r(h); rc = f(); if (rc == kOpRetry) rc = CDC_ERR_STATE; }
    return rc;
}

struct OpScope {                 // temporary device pool + op list for the cdc_op_* entry points
    cdc_handle *h;
    std::vector<void *> pool;
    std::vector<Op> saved_ops, saved_pre;
    int saved_shift_bs;
    explicit OpScope(cdc_handle *hh) : h(hh), saved_shift_bs(hh->shift_bs) {
        saved_ops.swap(h->ops);
        saved_pre.swap(h->pre_ops);
    }
    ~OpScope() {
        (void)hipDeviceSynchronize();
        free_pool(&pool);
        h->ops.swap(saved_ops);
        h->pre_ops.swap(saved_pre);
        h->shift_bs = saved_shift_bs;
    }
    int up(const float *src, size_t n, float **dst) { return upload(h, src, n, dst, &pool); }
    // Range guard of the single-operator entry points: the launches report non-finite accumulators (ConvArgs::fault) and the
    // result is checked; a faulting F16X2 call returns kOpRetry and its entry point repeats it in BF16X3.
    int run(int B, float *host_out, const float *dev_out, size_t n) {
        hipStream_t st = h->own_stream;
        const bool guard = guard_enabled(h);
        int rc;
        if (guard) { if ((rc = ensure_fault_flag(h))) return rc; HIP_TRY(h, hipMemsetAsync(h->d_fault, 0, sizeof(int), st)); }
        for (const Op &op : h->ops) {
            rc = run_op(h, op, B, st);
            if (rc) return rc;
        }
        if (guard) {
            int fault = 0;
            if ((rc = guard_check(h, {{dev_out, 0, (long long)n}}, 1, st, &fault))) return rc;
            if (fault) {
                if (guard_escalate(h, &rc)) return kOpRetry;
                if (rc) return rc;
            }
        }
        HIP_TRY(h, hipStreamSynchronize(st));
        HIP_TRY(h, hipMemcpy(host_out, dev_out, n * sizeof(float), hipMemcpyDeviceToHost));
        return CDC_OK;
    }
};

int op_ready(cdc_handle *h) { return ensure_device(h); }

}  // namespace

extern "C" {

static int op_conv2d_impl(cdc_handle *h, const float *x, const float *w, const float *bias, float *y, int B,
                  int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                  const float *ln_g, const float *ln_b, int relu, const float *shift,
                  const float *resid) {
    int rc = op_ready(h);
    if (rc) return rc;
    if (!x || !w || !y) return fail(h, CDC_ERR_INVALID, "null argument");
    OpScope sc(h);
    Builder bd{h, B, &sc.pool};
    ConvW cw;
    if ((rc = pack_conv(h, w, bias, Cout, Cin, KH, KW, stride, pad, false, &cw, &sc.pool))) return rc;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    float *dx, *dg = nullptr, *db = nullptr, *ds = nullptr, *dr = nullptr;
    if ((rc = sc.up(x, (size_t)B * Cin * H * W, &dx))) return rc;
    if (ln_g && (rc = sc.up(ln_g, Cout, &dg))) return rc;
    if (ln_b && (rc = sc.up(ln_b, Cout, &db))) return rc;
    if (shift && (rc = sc.up(shift, (size_t)B * Cout, &ds))) return rc;
    if (resid && (rc = sc.up(resid, (size_t)B * Cout * Ho * Wo, &dr))) return rc;
    h->shift_bs = Cout;
    float *dy = bd.dalloc((size_t)B * Cout * Ho * Wo);
    if (bd.pf_mode() == 1) {                     // CDC_PF=1: qualifying shapes run on the pre-split operand kernel
        bd.add_twin(dx, Cin, H, W);
        bd.pack(dx, (long long)Cin * H * W);
    }
    if (bd.rc) return bd.rc;
    Builder::ConvOpts o;
    o.ln_g = dg; o.ln_b = db; o.relu = relu; o.shift = ds;
    o.resid = dr; o.resid_bs = (long long)Cout * Ho * Wo; o.resid_cs = (long long)Ho * Wo;
    const long long obs = (long long)Cout * Ho * Wo;
    const int prof = KH == 7 ? PC_CONV7 : (KH == 1 ? PC_CONV1 : PC_CONV3);
    // few-pixel maps: the weight-stationary kernel (its raw result + the in-place LayerNorm pass; a bias-only call is the raw result)
    if (dg && relu && bd.try_ws(cw, dx, Cin, (long long)Cin * H * W, nullptr, 0, H, W, dy, obs, prof)) {
        bd.ln(dy, dy, Cout, Ho * Wo, dg, db, relu, ds, dr, nullptr, nullptr);
    } else if (!dg && !relu && !ds && !dr && bd.try_ws(cw, dx, Cin, (long long)Cin * H * W, nullptr, 0, H, W, dy, obs, prof)) {
    } else if (dg) {
        if (!bd.conv(cw, dx, Cin, (long long)Cin * H * W, nullptr, 0, H, W, dy, obs, o, true, prof)) {
            bd.conv(cw, dx, Cin, (long long)Cin * H * W, nullptr, 0, H, W, dy, obs, Builder::ConvOpts(),
                    false, prof);
            bd.ln(dy, dy, Cout, Ho * Wo, dg, db, relu, ds, dr, nullptr, nullptr);
        }
    } else {
        bd.conv(cw, dx, Cin, (long long)Cin * H * W, nullptr, 0, H, W, dy, obs, o, false, prof);
    }
    if (bd.rc) return bd.rc;
    // test aid: the layer must have been planned on the plane-operand kernel (a silent fallback would test nothing)
    if (dev_env("CDC_OP_REQUIRE_PF")) {
        bool on_pf = false;
        for (const Op &q : h->ops) on_pf = on_pf || (q.kind == Op::CONVPF && !q.pw);
        if (!on_pf) return fail(h, CDC_ERR_UNSUPPORTED, "CDC_OP_REQUIRE_PF: the convolution was not planned on conv_pf_kernel");
    }
    if (dev_env("CDC_OP_REQUIRE_WS")) {       // ... or on the weight-stationary kernel of the few-pixel levels
        bool on_ws = false;
        for (const Op &q : h->ops) on_ws = on_ws || q.kind == Op::CONVWS;
        if (!on_ws) return fail(h, CDC_ERR_UNSUPPORTED, "CDC_OP_REQUIRE_WS: the convolution was not planned on conv_ws_kernel");
    }
    return sc.run(B, y, dy, (size_t)B * Cout * Ho * Wo);
}

static int op_conv_transpose2d_impl(cdc_handle *h, const float *x, const float *w, const float *bias, float *y,
                            int B, int Cin, int H, int W, int Cout) {
    int rc = op_ready(h);
    if (rc) return rc;
    if (!x || !w || !y) return fail(h, CDC_ERR_INVALID, "null argument");
    OpScope sc(h);
    Builder bd{h, B, &sc.pool};
    ConvW cw;
    if ((rc = pack_conv(h, w, bias, Cout, Cin, 4, 4, 2, 1, true, &cw, &sc.pool))) return rc;
    float *dx;
    if ((rc = sc.up(x, (size_t)B * Cin * H * W, &dx))) return rc;
    const size_t ny = (size_t)B * Cout * 4 * H * W;
    float *dy = bd.dalloc(ny);
    if (bd.pf_mode() == 1) {
        bd.add_twin(dx, Cin, H, W);
        bd.pack(dx, (long long)Cin * H * W);
    }
    if (bd.rc) return bd.rc;
    bd.conv(cw, dx, Cin, (long long)Cin * H * W, nullptr, 0, H, W, dy, (long long)Cout * 4 * H * W,
            Builder::ConvOpts(), false, PC_UP);
    if (bd.rc) return bd.rc;
    if (dev_env("CDC_OP_REQUIRE_PF")) {       // test aid, see op_conv2d_impl
        bool on_pf = false;
        for (const Op &q : h->ops) on_pf = on_pf || (q.kind == Op::CONVPF && !q.pw);
        if (!on_pf) return fail(h, CDC_ERR_UNSUPPORTED, "CDC_OP_REQUIRE_PF: the convolution was not planned on conv_pf_kernel");
    }
    return sc.run(B, y, dy, ny);
}

static int op_chan_layernorm_impl(cdc_handle *h, const float *x, const float *g, const float *b, float *y, int B,
                          int C, int HW) {
    int rc = op_ready(h);
    if (rc) return rc;
    if (!x || !g || !b || !y) return fail(h, CDC_ERR_INVALID, "null argument");
    OpScope sc(h);
    Builder bd{h, B, &sc.pool};
    float *dx, *dg, *db;
    if ((rc = sc.up(x, (size_t)B * C * HW, &dx))) return rc;
    if ((rc = sc.up(g, C, &dg))) return rc;
    if ((rc = sc.up(b, C, &db))) return rc;
    float *dy = bd.dalloc((size_t)B * C * HW);
    if (bd.rc) return bd.rc;
    bd.ln(dx, dy, C, HW, dg, db, 0, nullptr, nullptr, nullptr, nullptr);
    return sc.run(B, y, dy, (size_t)B * C * HW);
}

static int op_linear_attention_impl(cdc_handle *h, const float *x, const float *norm_g, const float *norm_b,
                            const float *w_qkv, const float *w_out, const float *b_out, float *y, int B,
                            int C, int H, int W) {
    int rc = op_ready(h);
    if (rc) return rc;
    if (!x || !norm_g || !norm_b || !w_qkv || !w_out || !b_out || !y)
        return fail(h, CDC_ERR_INVALID, "null argument");
    OpScope sc(h);
    Builder bd{h, B, &sc.pool};
    AttnW at;
    at.C = C;
    if ((rc = pack_qkv_folded(h, w_qkv, norm_g, norm_b, C, 0, 3 * C, &at.qkv, &sc.pool))) return rc;
    if ((rc = pack_qkv_folded(h, w_qkv, norm_g, norm_b, C, C, 2 * C, &at.kv, &sc.pool))) return rc;
    if ((rc = pack_conv(h, w_out, b_out, C, C, 1, 1, 1, 0, false, &at.out, &sc.pool))) return rc;
    if ((rc = sc.up(norm_g, C, &at.ng))) return rc;
    if ((rc = sc.up(norm_b, C, &at.nb))) return rc;
    {
        std::vector<float> woT((size_t)C * C), wqT((size_t)C * C);
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < C; ++j) {
                woT[(size_t)j * C + i] = w_out[(size_t)i * C + j];
                wqT[(size_t)j * C + i] = w_qkv[(size_t)i * C + j];
            }
        if ((rc = sc.up(woT.data(), woT.size(), &at.WoT))) return rc;
        if ((rc = sc.up(wqT.data(), wqT.size(), &at.WqT))) return rc;
        if ((rc = sc.up(w_qkv, (size_t)C * C, &at.Wq))) return rc;
        std::vector<float> uq(C);
        for (int d = 0; d < C; ++d) {
            double acc = 0;
            for (int ci = 0; ci < C; ++ci) acc += (double)w_qkv[(size_t)d * C + ci] * norm_b[ci];
            uq[d] = (float)acc;
        }
        if ((rc = sc.up(uq.data(), uq.size(), &at.uq))) return rc;
    }
    Act ax;
    ax.C = C; ax.H = H; ax.W = W;
    if ((rc = sc.up(x, (size_t)B * C * H * W, &ax.p))) return rc;
    float *sm = bd.dalloc((size_t)B * H * W), *sr = bd.dalloc((size_t)B * H * W);
    if (bd.rc) return bd.rc;
    bd.ln(ax.p, nullptr, C, H * W, nullptr, nullptr, 0, nullptr, nullptr, sm, sr);   // statistics only
    Act ay = bd.attention(at, ax, sm, sr);
    if (bd.rc) return bd.rc;
    return sc.run(B, y, ay.p, (size_t)B * C * H * W);
}


int cdc_op_conv2d(cdc_handle *h, const float *x, const float *w, const float *bias, float *y, int B,
                  int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                  const float *ln_g, const float *ln_b, int relu, const float *shift,
                  const float *resid) {
    return op_with_guard(h, [&] { return op_conv2d_impl(h, x, w, bias, y, B, Cin, H, W, Cout, KH, KW, stride, pad, ln_g, ln_b, relu, shift, resid); });
}

int cdc_op_conv_transpose2d(cdc_handle *h, const float *x, const float *w, const float *bias, float *y,
                            int B, int Cin, int H, int W, int Cout) {
    return op_with_guard(h, [&] { return op_conv_transpose2d_impl(h, x, w, bias, y, B, Cin, H, W, Cout); });
}

int cdc_op_chan_layernorm(cdc_handle *h, const float *x, const float *g, const float *b, float *y, int B,
                          int C, int HW) {
    return op_with_guard(h, [&] { return op_chan_layernorm_impl(h, x, g, b, y, B, C, HW); });
}

int cdc_op_linear_attention(cdc_handle *h, const float *x, const float *norm_g, const float *norm_b,
                            const float *w_qkv, const float *w_out, const float *b_out, float *y, int B,
                            int C, int H, int W) {
    return op_with_guard(h, [&] { return op_linear_attention_impl(h, x, norm_g, norm_b, w_qkv, w_out, b_out, y, B, C, H, W); });
}

}  // extern "C"
