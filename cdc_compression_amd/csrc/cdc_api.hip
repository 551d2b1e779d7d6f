// cdc_api.hip -- C-ABI of libcdc_hip.so (include/cdc_hip.h): running a launch program, handle life cycle, and the entry points of the
// U-Net forward, the DDIM sampler loop (xparam/modules/denoising_diffusion.py:152-205, epsilonparam/...:137-192), the compressor
// programs and the profiling tables.  Parameter repacking: cdc_weights.hip; the launch-program builder: cdc_planner.hip; the entropy
// coder's entry points: cdc_entropy_api.hip; shared types: cdc_state.h.
#include "cdc_state.h"

namespace cdcapi {

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
    // CDC_DEV_REPEAT=n (development, tools/trace_cold_hot.py): every launch n times in a row, so that a kernel trace shows each
    // kernel cold (first launch: code, operands and argument block as the program leaves them) beside itself hot.
    static const int nrep = [] { const char *e = dev_env("CDC_DEV_REPEAT"); return e ? std::max(1, atoi(e)) : 1; }();
    for (int rep = 0; rep < nrep; ++rep)
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
int run_unet(cdc_handle *h, hipStream_t st, int step, bool skip_combine) {
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


bool guard_enabled(const cdc_handle *h) {
    static const bool no_guard = getenv("CDC_NO_RANGE_GUARD") != nullptr;
    return !no_guard && (h->arith == CDC_ARITH_F16X2 || h->in_retry);
}
int ensure_fault_flag(cdc_handle *h) {
    if (!h->d_fault) { void *p = nullptr; HIP_TRY(h, hipMalloc(&p, sizeof(int))); h->d_fault = (int *)p; h->weight_allocs.push_back(p); }
    return CDC_OK;
}
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

}  // namespace cdcapi

// =================================================================================================
// C-ABI
// =================================================================================================
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
