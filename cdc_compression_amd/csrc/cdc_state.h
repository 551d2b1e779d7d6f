// cdc_state.h -- what the translation units of the C-ABI share (round 5: cdc_api.hip was one 3 600-line file): the packed-parameter and
// launch-program types, the handle, error plumbing, the range guard, and the functions that cross the files:
//   cdc_weights.hip      manifest + parameter repacking (cdc_finalize_weights)
//   cdc_planner.hip      the launch-program builder (Builder), the programs of the four handle kinds, the single-operator entry points
//   cdc_api.hip          running a program, handle life cycle, U-Net / sampler / compressor entry points, profiling
//   cdc_entropy_api.hip  entropy-coder entry points and the stream container
#pragma once
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

namespace cdcapi {


enum ProfClass { PC_CONV3 = 0, PC_CONV7, PC_CONV1, PC_DOWN, PC_UP, PC_ATTN_CTX, PC_LN, PC_SMALL,
                 PC_COUNT };
static const char *const kProfNames[PC_COUNT] = {"conv3x3", "conv7x7", "conv1x1", "downsample", "upsample",
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


}  // namespace cdcapi
using namespace cdcapi;

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
    int op_stress_n = 0;                 // cdc_op_stress: extra executions of every cdc_op_* program, results compared on the device
    long long op_stress_launches = 0, op_stress_differing = 0;
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

namespace cdcapi {

extern std::string g_create_err;
int fail(cdc_handle *h, int code, const char *fmt, ...);

#define HIP_TRY(h, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(h, CDC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                    \
    } while (0)

// no C++ exception crosses the C boundary (std::bad_alloc from a vector sized by a hostile header, ...)
template <class F> int no_throw(cdc_handle *h, F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { return h ? fail(h, CDC_ERR_NOMEM, "out of host memory") : CDC_ERR_NOMEM; }
    catch (const std::exception &e) { return h ? fail(h, CDC_ERR_INVALID, "internal error: %s", e.what()) : CDC_ERR_INVALID; }
    catch (...) { return h ? fail(h, CDC_ERR_INVALID, "internal error") : CDC_ERR_INVALID; }
}

// ---- cdc_weights.hip
void add_param(cdc_handle *h, const std::string &name, std::vector<int64_t> shape, bool optional = false);
void add_resblock_params(cdc_handle *h, const std::string &p, int cin, int cout, int k, bool with_mlp = true);
void add_attn_params(cdc_handle *h, const std::string &p, int c);
int down_in_channels(const cdc_handle *h, int ind);
void build_manifest(cdc_handle *h);
int upload(cdc_handle *h, const float *src, size_t n, float **dst, std::vector<void *> *pool);
const std::vector<float> &hostp(cdc_handle *h, const std::string &name);
int upload_param(cdc_handle *h, const std::string &name, float **dst);
int pack_conv(cdc_handle *h, const float *w, const float *bias, int CoutF, int CinF, int KH, int KW, int stride, int pad, bool transposed,
              ConvW *cw, std::vector<void *> *pool, int ci0 = 0, int ncin = 0, int co0 = 0, int ncout = 0);
int pack_named_conv(cdc_handle *h, const std::string &wname, const std::string &bname, int stride, int pad, bool transposed, ConvW *cw,
                    int ci0 = 0, int ncin = 0, int co0 = 0, int ncout = 0);
int pack_qkv_folded(cdc_handle *h, const float *wq, const float *g, const float *bln, int C, int co0, int nco, ConvW *cw, std::vector<void *> *pool);
void free_pool(std::vector<void *> *pool);
// ---- cdc_planner.hip
void free_program(cdc_handle *h);
int build_program(cdc_handle *h, int B, int H, int W);
int build_encoder_program(cdc_handle *h, int B, int H, int W);
int build_hyperdec_program(cdc_handle *h, int B, int hh, int wh, bool batch1_plan = false);
int build_ctxdec_program(cdc_handle *h, int B, int hl, int wl);
// ---- cdc_api.hip: running a launch program
hipEvent_t get_event(cdc_handle *h);
int resolve_pending(cdc_handle *h);
int run_op(cdc_handle *h, const Op &op, int B, hipStream_t st);
int run_pre(cdc_handle *h, hipStream_t st);
int run_unet(cdc_handle *h, hipStream_t st, int step, bool skip_combine = false);
int copy_in(cdc_handle *h, float *dst, const float *src, size_t n, int mem, hipStream_t st);
int copy_out(cdc_handle *h, float *dst, const float *src, size_t n, int mem, hipStream_t st);
int stage_ctx(cdc_handle *h, const float *const *ctx, int n_ctx, int B, int mem, hipStream_t st);
int ensure_device(cdc_handle *h);
hipStream_t pick_stream(cdc_handle *h, void *stream, int mem);
int check_ready(cdc_handle *h);

// ---- range guard of the two-plane fp16 arithmetic ------------------------------------------------------------------
// |activation| >= 65504 becomes inf / NaN in CDC_ARITH_F16X2 and propagates to the results of the call.  Every entry point
// that runs the arithmetic checks its results (one small kernel + one 4-byte read-back, i.e. a stream synchronisation);
// a call whose results are not finite is repeated ONCE in the full-range three-plane bf16 arithmetic, and the handle stays
// in that mode (cdc_get_arith / cdc_get_range_faults tell).  Results that are non-finite there too -- a non-finite input,
// parameters that overflow fp32 -- are returned as they are, as the reference would (cdc_get_nonfinite_results counts them).
bool guard_enabled(const cdc_handle *h);
int ensure_fault_flag(cdc_handle *h);
struct GuardBuf { const float *p; long long bs, n; };
int guard_check(cdc_handle *h, std::initializer_list<GuardBuf> bufs, int B, hipStream_t st, int *fault);
bool guard_escalate(cdc_handle *h, int *rc);
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

}  // namespace cdcapi
