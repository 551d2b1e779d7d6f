// conv_launch.hip -- launch-plan chooser and launcher of the implicit-GEMM convolution kernel.
#include <map>
#include <vector>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "cdc_internal.h"

namespace cdc {

static unsigned magic_of(unsigned d) {
    if (d <= 1) return 0u;                       // kernel treats 0 as "divide by 1"
    return (unsigned)(((1ull << 32) + d - 1) / d);
}

static conv_kernel_fn lookup(int MB, int NPW, int lnmode) {
    if (MB <= 3) return conv_lookup_a(MB, NPW, lnmode);
    if (MB <= 6) return conv_lookup_b(MB, NPW, lnmode);
    return conv_lookup_c(MB, NPW, lnmode);
}

// LDS budget per workgroup.  The register-staged pipeline keeps ONE buffer set in LDS (the next
// chunk sits in registers), so up to two workgroups fit a CU's 160 KiB.
// LDS budget per workgroup: two buffer sets (double-buffered LDS-DMA).  <= 78 KiB lets two
// workgroups share a CU's 160 KiB when the register budget allows it.
static constexpr size_t kLdsSmall = 78 * 1024, kLdsBig = 150 * 1024;

static size_t plan_lds(int taps, int kc, int COPT, int PH, int PW, int nthr, int xv) {
    const int n_x = kc * PH * PW / xv, n_w4 = taps * kc * (COPT / 4);
    const size_t buf = (size_t)ceil_div(n_w4, nthr) * nthr * 4 + (size_t)ceil_div(n_x, nthr) * nthr * xv;
    return std::max(sizeof(float) * 2 * buf, sizeof(float) * 7 * (size_t)COPT);
}

static bool try_plan(const ConvShape &s, int MB, int NPW, int lognbw, ConvPlan *p, int force_kc = 0) {
    const int nblocks = ceil_div(s.Cout, 32);
    if (nblocks % MB) return false;
    if (!lookup(MB, NPW, s.lnmode)) return false;
    const int NBW = 1 << lognbw, NBH = 32 >> lognbw;
    const int nb_rows = ceil_div(s.Ho, NBH);
    int WN = std::min(4, ceil_div(nb_rows, NPW));
    if (WN == 3) WN = 4;
    const int nthr = 64 * WN;
    const int TH = WN * NPW * NBH;
    const int PH = (TH - 1) * s.stride + s.KH;
    int PW = (NBW - 1) * s.stride + s.KW;
    const int taps = s.KH * s.KW;
    const int COPT = MB * 32;
    // 16-byte input pieces: tile origins are multiples of 4 columns, so the patch of phase z starts
    // (-pad_x[z]) mod 4 columns after a 16-byte boundary; widen it to start ON the boundary.
    const bool xvec = s.Win > 0 && (s.Win & 3) == 0 && s.lnmode != 1 && ((NBW * s.stride) & 3) == 0;
    int xshift[4] = {0, 0, 0, 0};
    if (xvec) {
        int mx = 0;
        for (int z = 0; z < s.nz; ++z) { xshift[z] = ((-s.pad_x[z]) % 4 + 4) % 4; mx = std::max(mx, xshift[z]); }
        PW = round_up(mx + PW, 4);
    }
    const int xv = xvec ? 4 : 1;
    // prefer a footprint that lets two workgroups share a CU (when registers allow it), then the
    // largest K-chunk; fall back to one workgroup per CU
    int KC = 0;
    const bool two_wg = MB * NPW <= 8;
    for (size_t budget : {two_wg ? kLdsSmall : kLdsBig, kLdsBig}) {
        for (int kc : {16, 8, 4}) {
            if (force_kc && kc != force_kc) continue;
            if (s.C0 % kc) continue;                              // a chunk never straddles the concat seam
            if (kc > round_up(s.Cin, 4) && kc > 4) continue;      // do not over-pad tiny Cin
            const bool x_ok = kc * PH * PW / xv <= kXS * nthr;
            if (x_ok && plan_lds(taps, kc, COPT, PH, PW, nthr, xv) <= budget) { KC = kc; break; }
        }
        if (KC) break;
    }
    if (!KC) return false;
    p->MB = MB; p->NPW = NPW; p->WN = WN;
    p->groups = nblocks / MB;
    p->KC = KC;
    p->nchunk = ceil_div(s.Cin, KC);
    p->lognbw = lognbw;
    p->tiles_x = ceil_div(s.Wo, NBW);
    p->tiles_y = ceil_div(s.Ho, TH);
    p->PH = PH; p->PW = PW;
    p->xvec = xvec ? 1 : 0;
    for (int z = 0; z < 4; ++z) p->xshift[z] = xshift[z];
    p->lds_bytes = plan_lds(taps, KC, COPT, PH, PW, nthr, xv);
    p->lnmode = s.lnmode;
    p->split = 0;
    p->ipw = 1;
    p->tg = 0;
    return true;
}

// Architected VGPRs of a kernel (cached): decides how many waves fit a SIMD (512 registers / lane).
static int kernel_vgprs(conv_kernel_fn f) {
    static std::map<const void *, int> cache;
    auto it = cache.find((const void *)f);
    if (it != cache.end()) return it->second;
    hipFuncAttributes at{};
    const int n = hipFuncGetAttributes(&at, (const void *)f) == hipSuccess ? at.numRegs : 512;
    cache[(const void *)f] = n;
    return n;
}

// Split-bf16 kernel: 16-channel chunks, one workgroup per CU; LDS = split patch (96 B/position) +
// fp32 landing area (64 B/position) + two weight-row stages.
static bool try_plan_split(const ConvShape &s, int MB, int NPW, int lognbw, ConvPlan *p, bool allow_ipw = true) {
    const int ar = s.arith;                          // 1: two fp16 planes (register-staged variant only)
    const int xpl = ar ? 16 : 24;                    // floats of split patch per position
    auto lookup2 = [&](int mb, int npw, int ln, int xu) { return ar ? conv_lookup_split2h(mb, npw, ln, xu) : conv_lookup_split2(mb, npw, ln, xu); };
    // (a software-pipelined tap loop was measured in round 4 -- a workgroup's life -12 ... -33 %, the layers no faster: the second operand set
    //  costs the third workgroup per CU, and the one-block tiles are LDS-bandwidth-bound -- and removed: profiles/split2_pipe_ab_r04.txt)
    const int nblocks = ceil_div(s.Cout, 32);
    if (nblocks % MB) return false;
    const int NBW = 1 << lognbw, NBH = 32 >> lognbw;
    const int nb_rows = ceil_div(s.Ho, NBH);
    int WN = std::min(4, ceil_div(nb_rows, NPW));
    if (WN == 3) WN = 4;
    // small feature maps (one or two 32-pixel row blocks per wave cover the image): pack several images
    // into the workgroup so the weight stages are still shared by four waves
    int ipw = 1;
    if (allow_ipw && !s.per_image_w && s.Wo <= NBW && WN <= 2 && s.B >= 2) { ipw = 4 / WN; WN = 4; }
    const int wpi = WN / ipw;
    const int nthr = 64 * WN;
    const int TH = wpi * NPW * NBH;
    const int PH = (TH - 1) * s.stride + s.KH;
    int PW = (NBW - 1) * s.stride + s.KW;
    int xshift[4] = {0, 0, 0, 0}, mx = 0;
    for (int z = 0; z < s.nz; ++z) { xshift[z] = ((-s.pad_x[z]) % 4 + 4) % 4; mx = std::max(mx, xshift[z]); }
    PW = round_up(mx + PW, 4);
    const int plane = PH * PW, COPT = MB * 32;
    // variant 2 (patch staged through registers, two workgroups per CU) when it fits
    // weight stages of one kernel row if that fits next to a second workgroup, else one tap each
    int tg = s.KW;
    const size_t lds_tap = sizeof(float) * ((size_t)ipw * xpl * plane + (size_t)2 * 24 * COPT);
    size_t lds2 = sizeof(float) * ((size_t)ipw * xpl * plane + (size_t)2 * tg * 24 * COPT);
    if (lds2 > 80 * 1024) { tg = 1; lds2 = lds_tap; }
    // a third workgroup per CU (+12 % measured on 64->64 @256^2) when single-tap stages bring the LDS
    // footprint under 160/3 KiB and the kernel's registers allow three waves per SIMD
    if (tg > 1 && lds_tap <= 53 * 1024 && lds2 > 53 * 1024)
        if (conv_kernel_fn f = lookup2(MB, NPW, s.lnmode, s.stride == 2 ? 2 : 1))
            if (kernel_vgprs(f) <= 168) { tg = 1; lds2 = lds_tap; }
    // Few workgroups (low-resolution levels): the chip cannot hide the weight-stage latency by occupancy,
    // so stage ALL taps of a chunk at once -- one barrier and one DMA wait per 16 channels, and the DMA of
    // the next chunk has a whole chunk of matrix work to land.
    {
        const int taps = s.KH * s.KW;
        const long long wgs = (long long)(ipw > 1 ? ceil_div(s.B, ipw) : ceil_div(s.Wo, NBW) * ceil_div(s.Ho, TH) * s.B) *
                              (nblocks / MB) * s.nz;
        const size_t lds_all = sizeof(float) * ((size_t)ipw * xpl * plane + (size_t)2 * taps * 24 * COPT);
        const bool few = wgs <= 3 * 256 && lds_all <= 72 * 1024;
        if (taps > tg && few) { tg = taps; lds2 = lds_all; }
    }
    // patch units per thread: stride 2 always runs the two-unit / parity-plane variant
    const int xu = s.stride == 2 ? 2 : 1;
    const bool v2 = plane / 2 <= xu * wpi * 64 && s.stride <= 2 && lookup2(MB, NPW, s.lnmode, xu) && lds2 <= 150 * 1024 &&
                    (xu == 1 || ((s.Ho * s.Wo >= 256 || (s.max_ksplit > 1)))) && !dev_env("CDC_NO_SPLIT2");
    if (!v2 && ipw > 1) return try_plan_split(s, MB, NPW, lognbw, p, false);
    if (!v2 && s.per_image_w) return false;          // only the register-staged variant takes per-image planes
    if (!v2 && ar) { ConvShape s0 = s; s0.arith = 0; return try_plan_split(s0, MB, NPW, lognbw, p, allow_ipw); }
    if (!v2 && (s.lnmode != 0 || !conv_lookup_split(MB, NPW) || 4 * plane > kXS * nthr)) return false;
    const size_t lds = v2 ? lds2 : sizeof(float) * ((size_t)40 * plane + (size_t)2 * s.KW * 24 * COPT);
    if (lds > 160 * 1024) return false;
    p->MB = MB; p->NPW = NPW; p->WN = WN;
    p->groups = nblocks / MB;
    p->KC = 16;
    p->nchunk = ceil_div(s.Cin, 16);
    p->lognbw = lognbw;
    p->tiles_x = ceil_div(s.Wo, NBW);
    p->tiles_y = ceil_div(s.Ho, TH);
    p->PH = PH; p->PW = PW;
    p->xvec = 1;
    for (int z = 0; z < 4; ++z) p->xshift[z] = xshift[z];
    p->lds_bytes = std::max(lds, sizeof(float) * (6 + ipw) * (size_t)COPT);
    p->lnmode = s.lnmode;
    p->split = v2 ? 2 : 1;
    p->tg = v2 ? tg : s.KW;
    p->ipw = ipw;
    p->xu = v2 ? xu : 1;
    p->arith = v2 ? ar : 0;
    return true;
}

bool conv_make_plan(const ConvShape &s, ConvPlan *plan) {
    const int nblocks = ceil_div(s.Cout, 32);
    int lognbw = 5;
    while (lognbw > 2 && (1 << (lognbw - 1)) >= s.Wo) --lognbw;   // NBW = smallest pow2 >= Wo (<=32, >=4)
    std::vector<int> mbs;
    if (s.need_all_cout) {
        if (nblocks > 12) return false;
        mbs.push_back(nblocks);
    } else {
        for (int mb = std::min(6, nblocks); mb >= 1; --mb)
            if (nblocks % mb == 0) mbs.push_back(mb);
    }
    const int NBH = 32 >> lognbw;
    const int nb_rows = ceil_div(s.Ho, NBH);
    ConvPlan best;
    double best_score = -1;
    // fp32-exact products on the bf16 matrix cores where the layer is matrix-bound (k x k taps, >= 16
    // input channels, chunks aligned to the concat seam, 16-byte alignable rows)
    const bool split_ok = s.allow_split && (s.lnmode == 0 || s.lnmode == 1 || (s.lnmode == 2 && s.KH * s.KW == 1)) &&
                          s.Cin >= (s.KH * s.KW > 1 ? 16 : 32) && (s.C0 % 16) == 0 &&
                          s.Win > 0 && (s.Win & 3) == 0 && (((1 << lognbw) * s.stride) & 3) == 0 &&
                          !dev_env("CDC_NO_SPLIT");
    if (split_ok) {
        for (int MB : mbs) {
            for (int NPW : {4, 2, 1}) {
                if (MB * NPW > 8 || MB > 8) continue;
                if (NPW > 1 && NPW > nb_rows) continue;
                ConvPlan p;
                p.split = 0;
                if (!try_plan_split(s, MB, NPW, lognbw, &p)) continue;
                const double wgs = (p.ipw > 1 ? (double)ceil_div(s.B, p.ipw) : (double)p.tiles_x * p.tiles_y * s.B) *
                                   p.groups * s.nz;
                const int ks_ok = p.split == 2 ? std::max(1, std::min(s.max_ksplit, p.nchunk / 4)) : 1;
                const double fill = std::min(1.0, wgs * ks_ok * p.WN / (p.split == 2 ? 2048.0 : 1024.0));
                const double reuse = (double)(MB * NPW) / (MB + NPW);
                // two co-resident workgroups overlap conversion / staging with the other's MFMAs
                // 1x1 layers are bandwidth-bound: every extra channel group re-reads the input
                const double grp_pen = s.KH * s.KW == 1 ? 1.0 / (1.0 + 0.2 * (p.groups - 1)) : 1.0;
                const double score = fill * (0.5 + 0.15 * std::min(reuse, 3.0)) * (p.split == 2 ? 1.25 : 1.0) * grp_pen;
                if (score > best_score) { best_score = score; best = p; }
            }
            if (s.need_all_cout) break;
        }
        if (best_score >= 0) { *plan = best; return true; }
    }
    for (int MB : mbs) {
        for (int NPW : {4, 2, 1}) {
            if (MB * NPW > 12) continue;
            if (NPW > 1 && NPW > nb_rows) continue;
            ConvPlan p;
            p.split = 0;
            if (!try_plan(s, MB, NPW, lognbw, &p)) continue;
            // measured (round 1, batch 32): register blocking matters more than the chunk depth
            // (MB2/NPW4/KC4 99 TF vs MB2/NPW2/KC8 91 TF; MB4/NPW2/KC4 103 TF vs KC8 with one WG/CU 91 TF)
            const double wgs = (double)p.tiles_x * p.tiles_y * s.B * p.groups;
            const double fill = std::min(1.0, wgs * p.WN / (256.0 * 4.0));
            const double reuse = (double)(MB * NPW) / (MB + NPW);
            const double score = fill * (0.5 + 0.15 * std::min(reuse, 3.0));
            if (score > best_score) { best_score = score; best = p; }
        }
        if (s.need_all_cout) break;
    }
    if (best_score < 0) return false;
    *plan = best;
    return true;
}

hipError_t conv_launch(ConvArgs a, const ConvPlan &p, int B, int nz, hipStream_t st) {
    a.KC = p.KC;
    a.logKC = p.KC == 16 ? 4 : (p.KC == 8 ? 3 : 2);
    a.nchunk = p.nchunk;
    a.lognbw = p.lognbw;
    a.tiles_x = p.tiles_x;
    a.tiles_y = p.tiles_y;
    a.PH = p.PH; a.PW = p.PW;
    a.xvec = p.xvec;
    a.tg = p.tg;
    a.ipw = p.ipw;
    a.ksplit = p.split == 2 ? std::max(1, p.ksplit) : 1;
    a.nzz = nz;
    a.B = B;
    for (int z = 0; z < 4; ++z) a.xshift[z] = p.xshift[z];
    const int xv = p.xvec ? 4 : 1;
    a.magic_hw = magic_of((unsigned)(p.PH * p.PW / xv));
    a.magic_w = magic_of((unsigned)(p.PW / xv));
    conv_kernel_fn fn = p.split == 2 ? (p.arith ? conv_lookup_split2h(p.MB, p.NPW, p.lnmode, p.xu)
                                                : conv_lookup_split2(p.MB, p.NPW, p.lnmode, p.xu))
                        : (p.split ? conv_lookup_split(p.MB, p.NPW) : lookup(p.MB, p.NPW, p.lnmode));
    if (a.uf_c) fn = (p.split == 2 && p.arith && p.xu == 1 && p.lnmode == 0) ? conv_lookup_split2hu(p.MB, p.NPW) : nullptr;
    if (!fn) return hipErrorInvalidValue;
    if (hipError_t e = ensure_dynamic_lds((const void *)fn, p.lds_bytes); e != hipSuccess) return e;
    dim3 grid((unsigned)(p.ipw > 1 ? ceil_div(B, p.ipw) : p.tiles_x * p.tiles_y * B), (unsigned)p.groups,
              (unsigned)(nz * a.ksplit));
    a.zfold = 0;
    a.xcd_remap = (p.split != 1 && p.ipw == 1 && grid.x % 8 == 0 && grid.x >= 64) ? 1 : 0;
    if (p.split == 2 && nz == 4 && a.ksplit == 1 && p.ipw == 1 && grid.x % 8 == 0) {
        a.zfold = 1; grid.x *= 4; grid.z = 1;
    }
    dim3 block(64 * p.WN);
#ifdef CDC_TIMELINE
    // Development build only: per-workgroup cycle categories of the split2 kernel, summarised on stderr.
    static unsigned long long *tl_dev = nullptr;
    const size_t tl_wgs = (size_t)grid.x * grid.y * grid.z;
    a.tl = nullptr;
    if (p.split == 2 && tl_wgs <= (1u << 18)) {
        if (!tl_dev) hipMalloc(&tl_dev, sizeof(unsigned long long) * 16 * (1u << 18));
        hipMemsetAsync(tl_dev, 0, sizeof(unsigned long long) * 16 * tl_wgs, st);
        a.tl = tl_dev;
    }
#endif
    hipLaunchKernelGGL(fn, grid, block, p.lds_bytes, st, a);
#ifdef CDC_TIMELINE
    if (a.tl) {
        hipStreamSynchronize(st);
        std::vector<unsigned long long> h(16 * tl_wgs);
        hipMemcpy(h.data(), tl_dev, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        double cat[7] = {0}, life = 0;
        for (size_t w = 0; w < tl_wgs; ++w) {
            const unsigned long long *r = &h[w * 16];
            for (int c = 0; c < 7; ++c) cat[c] += (double)r[c];
            t0 = std::min(t0, r[7]); t1 = std::max(t1, r[8]);
            life += (double)(r[8] - r[7]);
        }
        static const char *names[7] = {"prologue", "convert+ds_write", "chunk-head wait+barrier", "tap loops", "group-end dma wait", "group-end barrier", "epilogue"};
        fprintf(stderr, "[timeline] conv %dx%d s%d %d->%d out %dx%d: %zu workgroups x %d threads, lds %zu, kernel span %llu cycles, mean workgroup life %.0f cycles (= %.2f of the span);"
                        " per workgroup (wave 0):", a.KH, a.KW, a.stride, a.Cin, a.Cout, a.Ho, a.Wo, tl_wgs, 64 * p.WN, p.lds_bytes, t1 - t0, life / tl_wgs, life / tl_wgs / (double)(t1 - t0));
        for (int c = 0; c < 7; ++c) fprintf(stderr, "  %s %.0f", names[c], cat[c] / tl_wgs);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

}  // namespace cdc
