// cdc_entropy_api.hip -- entry points of the entropy coder (kernels and tables: entropy.hip) and the stream container.
#include "cdc_state.h"

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

extern "C" {

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

}  // extern "C"
