// cdc_weights.hip -- the state_dict manifest of the four handle kinds and the repacking of the reference's parameters into the kernels'
// operand layouts (cdc_finalize_weights): fp32 / three-plane bf16 / fp16 planes {WH, WL, WH2} in MFMA A-operand order, transposed
// convolutions as phase convolutions, the row-folded final convolution, PreNorm parameters folded into the following 1x1 weights, the
// folded attention matrices, context hoisting (unet.py:33-104, network_components.py:34-139).
#include "cdc_state.h"

namespace cdcapi {

// ------------------------------------------------------------------------------------------------
// manifest (reference state_dict order: time_mlp, downs, ups, mid_*, final_conv)
// ------------------------------------------------------------------------------------------------
void add_param(cdc_handle *h, const std::string &name, std::vector<int64_t> shape, bool optional) {
    Param p;
    p.name = name;
    p.optional = optional;
    p.shape = std::move(shape);
    h->pindex[name] = (int)h->params.size();
    h->params.push_back(std::move(p));
}

void add_resblock_params(cdc_handle *h, const std::string &p, int cin, int cout, int k, bool with_mlp) {
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
              int stride, int pad, bool transposed, ConvW *cw, std::vector<void *> *pool, int ci0,
              int ncin, int co0, int ncout) {
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
                    int pad, bool transposed, ConvW *cw, int ci0, int ncin, int co0,
                    int ncout) {
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


}  // namespace cdcapi

extern "C" {

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


}  // extern "C"
