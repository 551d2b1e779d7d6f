// cdc_planner.hip -- the launch-program builder: which kernel, tile shape and tensor form (fp32 / planes) every layer of Unet.forward
// (xparam/modules/unet.py:106-135) and of the compressor programs gets for a given batch and frame size, and the single-operator entry
// points (cdc_op_*: one-layer programs through the same builder).
#include "cdc_state.h"

namespace cdcapi {

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
            snprintf(buf, sizeof buf, "conv 3x3 s%d %4d->%-4d out %3dx%-3d NPB%d waves%d tiles%d g%d WS%s", op.wsplan.stride, op.ws.Cin, op.ws.Cout, op.ws.H, op.wsplan.W,
                     op.wsplan.NPB, op.wsplan.waves, op.wsplan.tiles, op.wsplan.groups, op.ws.pre_add ? " pre_add" : "");
        else if (op.kind == Op::CONVWS1)
            snprintf(buf, sizeof buf, "conv 1x1 s1 %4d->%-4d out %3dx%-3d NPB%d waves%d tiles%d g%d WS1%s%s%s", op.ws1.Cin, op.ws1.Cout, ws1_h, op.ws1.HW / std::max(ws1_h, 1),
                     op.ws1plan.NPB, op.ws1plan.waves, op.ws1plan.tiles, op.ws1plan.groups, op.ws1.pre_mean ? " pre" : "", op.ws1.w_bs ? " perimg" : "",
                     op.ws1.resid ? (op.ws1.resid_is_pre ? " pre_add" : " +res") : "");
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
        // every op of the context-only part of the program says so (the convolutions' formats above already do)
        if (cur == &h->pre_ops && !strstr(buf, " HOIST") && strlen(buf) + 7 < sizeof buf) strcat(buf, " HOIST");
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
        const long long join_max = 16384;
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
    // (H, W: the INPUT map; the stride-2 form is the Downsample, 3x3 / pad 1 on even extents)
    bool ws_would_plan(const ConvW &w, int C0, bool two_src, int H, int W) {
        if (h->arith != 1 || !w.wsh || planB > 0 || w.KH != 3 || w.KW != 3 || (w.stride != 1 && w.stride != 2) || w.transposed || w.nz != 1) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != 1 || (w.pad_x >= 0 ? w.pad_x : w.pad) != 1) return false;
        if (w.COP != w.Cout || w.Cin_pad != w.Cin) return false;
        if (w.stride == 2 && ((H & 1) || (W & 1) || two_src)) return false;
        WsPlan plan;
        return ws_make_plan(w.Cin, two_src ? C0 : w.Cin, w.Cout, H / w.stride, W / w.stride, pb(), w.stride, &plan);
    }
    bool try_ws(const ConvW &w, const float *s0, int C0, long long bs0, const float *s1, long long bs1, int H, int W, float *raw,
                long long raw_bs, int prof, const float *pre_add = nullptr) {
        if (rc || !ws_would_plan(w, C0, s1 != nullptr, H, W)) return false;
        if (!ensure_f32(s0, bs0) || (s1 && !ensure_f32(s1, bs1))) return false;
        const int Ho = H / w.stride, Wo = W / w.stride;
        if (raw_bs != (long long)w.Cout * Ho * Wo) return false;
        Op op;
        op.kind = Op::CONVWS; op.prof = prof;
        if (!ws_make_plan(w.Cin, s1 ? C0 : w.Cin, w.Cout, Ho, Wo, pb(), w.stride, &op.wsplan)) return false;
        WsArgs &a = op.ws;
        memset(&a, 0, sizeof a);
        a.x0 = s0; a.x0_bs = bs0; a.x1 = s1; a.x1_bs = bs1;
        a.C0 = s1 ? C0 : w.Cin; a.Cin = w.Cin; a.H = Ho; a.B = B;
        a.w = w.wsh; a.nchunk = w.Cin / 16; a.COP = w.COP; a.Cout = w.Cout; a.acc_scale = w.wscale_inv;
        a.bias = pre_add ? nullptr : w.bias;            // (a hoisted partial sum already carries the bias)
        a.pre_add = pre_add;
        a.out = raw; a.out_bs = raw_bs;
        a.fault = fault_flag();
        const double px = (double)B * Ho * Wo;
        op.flops = 2.0 * px * w.Cout * w.Cin * 9;
        op.bytes = 4.0 * ((double)B * H * W * w.Cin + px * w.Cout);
        if (getenv("CDC_DEBUG_PLAN"))
            fprintf(stderr, "[plan] conv 3x3 s%d %d->%d out %dx%d on conv_ws_kernel: %d tiles of %d pixels x %d groups, %d waves, %zu bytes of LDS\n", w.stride, w.Cin, w.Cout, Ho, Wo,
                    op.wsplan.tiles, op.wsplan.NPB * 32, op.wsplan.groups, op.wsplan.waves, op.wsplan.lds_bytes);
        last_ksplit = 1;
        last_pf_only = false;
        emit(op);
        return true;
    }

    // 1x1 layer of a few-pixel launch (at most 128 pixel blocks, or per-image weights) on conv_ws1_kernel (conv_ws1_kernel.h): all of K inside the
    // workgroup -- no partial-sum tensors, no sum pass.  Epilogue: bias, folded PreNorm (mean on load, rstd after), per-image shift,
    // residual; shared or per-image weight planes.
    bool try_ws1(const ConvW &w, const float *s0, int C0, long long bs0, const float *s1, long long bs1, int H, int W, float *out, long long out_bs,
                 const ConvOpts &o, bool need_all, int prof) {
        if (rc || h->arith != 1 || !w.wsh || planB > 0 || w.KH != 1 || w.KW != 1 || w.stride != 1 || w.transposed || w.nz != 1) return false;
        if ((w.pad_y >= 0 ? w.pad_y : w.pad) != 0 || (w.pad_x >= 0 ? w.pad_x : w.pad) != 0) return false;
        if (need_all || o.ln_g || o.stat_mean || o.res3_w || o.pf_only || (o.pre_add && o.resid) || o.relu || o.resid1 || o.uf_c) return false;
        if (o.pre_mean && o.pre_mode != 2) return false;
        if (o.w_bs && !o.wsp_bs) return false;               // per-image weights without planes
        if (w.COP != w.Cout || w.Cin_pad != w.Cin || out_bs != (long long)w.Cout * H * W) return false;
        if (o.resid && (o.resid_cs != (long long)H * W)) return false;
        // Measured (batch 32, profiles/per_op_r05*.txt): it wins wherever the alternative is a split-K launch + sum pass (the 8x8 level, the
        // per-image attention products everywhere); at 16x16 and batch 32 the wide folded-PreNorm projections (24 - 36 channel groups, each
        // converting the same activations again) and the res_convs are faster on conv_pw_kernel's 64 - 96-channel workgroups.
        const long long blocks = (long long)pb() * (H * W / 32);
        // (CDC_WS1_MAX_BLOCKS / CDC_WS1_MAX_GROUPS: A/B knobs of round 6 -- layers with few channel groups at up to 256 pixel blocks,
        //  profiles/planner_ab_r06.txt)
        const long long max_blocks = dev_env("CDC_WS1_MAX_BLOCKS") ? atoll(dev_env("CDC_WS1_MAX_BLOCKS")) : 128;
        const int max_groups_wide = dev_env("CDC_WS1_MAX_GROUPS") ? atoi(dev_env("CDC_WS1_MAX_GROUPS")) : 0;
        const bool wide_ok = blocks <= max_blocks || (max_groups_wide > 0 && blocks <= 256 && w.Cout / 32 <= max_groups_wide);
        if (!wide_ok && !(o.wsp_bs && W < 32)) return false;     // (per-image products of the few-pixel levels at any batch)
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
        // (a hoisted partial sum -- the step-invariant context half of a concatenated input, layout of `out` -- is one more addend of this linear epilogue)
        if (o.pre_add) { a.resid = o.pre_add; a.resid_bs = out_bs; a.resid_is_pre = 1; }
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
        // Downsample of a few-pixel level at small batch: all of K inside the workgroup (no split-K, no sum pass)
        if (w.stride == 2 && !need_all && !o.ln_g && !o.relu && !o.shift && !o.stat_mean && !o.pre_add && !o.res3_w && !o.resid && !o.pre_mean &&
            !o.w_bs && !o.uf_c && !(o.emit_pf && twin(out)) && try_ws(w, s0, C0, bs0, s1, bs1, H, W, out, out_bs, prof))
            return true;
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
        if (!want_res3 && !uf_c && !resid1 && w.stride == 1 && out.bs() == (long long)w.Cout * H * W &&
            try_ws(w, s0, C0, bs0, s1, bs1, H, W, out.p, out.bs(), prof, pre_add)) {
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
        // (debugging taps of the level's intermediates: the staged projection, the partial context sums, the per-image matrix)
        if (!fused) h->taps[at.prefix + ".kv"] = qkv;
        { Act ts; ts.p = S; ts.C = nsplit; ts.H = C; ts.W = C; h->taps[at.prefix + ".S"] = ts; }
        { Act ts; ts.p = ksum; ts.C = nsplit; ts.H = 1; ts.W = C; h->taps[at.prefix + ".Z"] = ts; }
        { Act ts; ts.p = kmax; ts.C = 1; ts.H = 1; ts.W = C; h->taps[at.prefix + ".kmax"] = ts; }
        { Act ts; ts.p = ctxw; ts.C = 1; ts.H = Cin_pad; ts.W = COP; h->taps[at.prefix + ".M"] = ts; }
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
        { Act ts; ts.p = sm; ts.C = 1; ts.H = x.H; ts.W = x.W; h->taps[dn + ".1.stat_mean"] = ts; ts.p = sr; h->taps[dn + ".1.stat_rstd"] = ts; }   // (the PreNorm statistics the attention reads)
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
int build_hyperdec_program(cdc_handle *h, int B, int hh, int wh, bool batch1_plan) {
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


}  // namespace cdcapi

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
        if (h->op_stress_n > 0) {      // cdc_op_stress: the program again and again, every result against the first, on the device
            void *first = nullptr, *cnt = nullptr;
            HIP_TRY(h, hipMalloc(&first, n * sizeof(float))); pool.push_back(first);
            HIP_TRY(h, hipMalloc(&cnt, 3 * sizeof(long long))); pool.push_back(cnt);
            HIP_TRY(h, hipMemsetAsync(cnt, 0, 3 * sizeof(long long), st));
            HIP_TRY(h, hipMemcpyAsync(first, dev_out, n * sizeof(float), hipMemcpyDeviceToDevice, st));
            for (int k = 0; k < h->op_stress_n; ++k) {
                for (const Op &op : h->ops) { rc = run_op(h, op, B, st); if (rc) return rc; }
                HIP_TRY(h, bits_differ_launch(dev_out, (const float *)first, (long long)n, (long long *)cnt, st));
            }
            long long c[3] = {0, 0, 0};
            HIP_TRY(h, hipStreamSynchronize(st));
            HIP_TRY(h, hipMemcpy(c, cnt, sizeof c, hipMemcpyDeviceToHost));
            h->op_stress_launches = c[0]; h->op_stress_differing = c[1];
            dev_out = (const float *)first;           // the caller gets the FIRST execution's result
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
    if (dg && relu && stride == 1 && bd.try_ws(cw, dx, Cin, (long long)Cin * H * W, nullptr, 0, H, W, dy, obs, prof)) {
        bd.ln(dy, dy, Cout, Ho * Wo, dg, db, relu, ds, dr, nullptr, nullptr);
    } else if (!dg && !relu && !ds && !dr && stride == 1 && bd.try_ws(cw, dx, Cin, (long long)Cin * H * W, nullptr, 0, H, W, dy, obs, prof)) {
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

int cdc_op_stress(cdc_handle *h, int repeats) {
    if (!h) return CDC_ERR_INVALID;
    if (repeats < 0) return fail(h, CDC_ERR_INVALID, "cdc_op_stress: repeats < 0");
    h->op_stress_n = repeats;
    h->op_stress_launches = h->op_stress_differing = 0;
    return CDC_OK;
}

int cdc_op_stress_result(cdc_handle *h, int64_t *launches, int64_t *differing) {
    if (!h) return CDC_ERR_INVALID;
    if (launches) *launches = h->op_stress_launches;
    if (differing) *differing = h->op_stress_differing;
    return CDC_OK;
}

}  // extern "C"

