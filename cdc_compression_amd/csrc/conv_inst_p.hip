// conv_inst_p.hip -- instantiations, launch-plan chooser and launcher of conv_pf_kernel (pre-split fp16 operands
// by LDS-DMA), plus the fp32 NCHW -> PF packing kernel for tensors whose producer does not emit planes.
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include <stdlib.h>

#include "cdc_internal.h"
#include "conv_pf_kernel.h"

namespace cdc {

template <int KH, int KW>
static pf_kernel_fn pf_lookup_k(int MB, int NPW, int WM, int WP) {
    if (MB == 2 && NPW == 2 && WM == 1 && WP == 4) return conv_pf_kernel<2, 2, 1, 4, KH, KW>;
    if (MB == 2 && NPW == 2 && WM == 2 && WP == 2) return conv_pf_kernel<2, 2, 2, 2, KH, KW>;
    if (MB == 2 && NPW == 2 && WM == 2 && WP == 4) return conv_pf_kernel<2, 2, 2, 4, KH, KW>;
    if (MB == 3 && NPW == 2 && WM == 2 && WP == 4) return conv_pf_kernel<3, 2, 2, 4, KH, KW>;
    if (MB == 3 && NPW == 1 && WM == 2 && WP == 4 && KH == 3) return conv_pf_kernel<3, 1, 2, 4, KH, KW>;
    if (MB == 2 && NPW == 2 && WM == 4 && WP == 2) return conv_pf_kernel<2, 2, 4, 2, KH, KW>;
    return nullptr;
}
// stride-2 3x3 (Downsample): 4-row tiles, one workgroup per CU
static pf_kernel_fn pf_lookup_s2(int MB, int NPW, int WM, int WP) {
    if (MB == 2 && NPW == 1 && WM == 1 && WP == 4) return conv_pf_kernel<2, 1, 1, 4, 3, 3, 2>;
    if (MB == 2 && NPW == 2 && WM == 2 && WP == 2) return conv_pf_kernel<2, 2, 2, 2, 3, 3, 2>;
    if (MB == 3 && NPW == 1 && WM == 2 && WP == 4) return conv_pf_kernel<3, 1, 2, 4, 3, 3, 2>;
    return nullptr;
}
// fused-phase transposed form (TZ = 4): one 32-channel block x two pixel rows per wave, four accumulator sets
static pf_kernel_fn pf_lookup_tz(int MB, int NPW, int WM, int WP) {
    if (MB == 1 && NPW == 2 && WM == 2 && WP == 2) return conv_pf_kernel<1, 2, 2, 2, 2, 2, 1, 4>;
    if (MB == 2 && NPW == 1 && WM == 1 && WP == 4) return conv_pf_kernel<2, 1, 1, 4, 2, 2, 1, 4>;      // all 64 channels in a wave: in-lane LayerNorm
    if (MB == 1 && NPW == 2 && WM == 4 && WP == 2) return conv_pf_kernel<1, 2, 4, 2, 2, 2, 1, 4>;
    return nullptr;
}
static pf_kernel_fn pf_lookup(int MB, int NPW, int WM, int WP, int KH, int KW, int stride = 1, int tz = 1) {
    if (tz == 4) return (KH == 2 && KW == 2 && stride != 2) ? pf_lookup_tz(MB, NPW, WM, WP) : nullptr;
    if (stride == 2) return (KH == 3 && KW == 3) ? pf_lookup_s2(MB, NPW, WM, WP) : nullptr;
    if (KH == 3 && KW == 3) return pf_lookup_k<3, 3>(MB, NPW, WM, WP);
    if (KH == 1 && KW == 1) return pf_lookup_k<1, 1>(MB, NPW, WM, WP);
    if (KH == 2 && KW == 2) return pf_lookup_k<2, 2>(MB, NPW, WM, WP);
    if (KH == 1 && KW == 7 && MB == 1 && NPW == 2 && WM == 1 && WP == 4) return conv_pf_kernel<1, 2, 1, 4, 1, 7>;   // row-folded final convolution
    if (KH == 7 && KW == 1 && MB == 2 && NPW == 2 && WM == 1 && WP == 4) return conv_pf_kernel<2, 2, 1, 4, 7, 1, 1, 1, 3>;   // first layer, built from the image (UF)
    return nullptr;
}

// Candidate shapes, best first for a given channel-group width COPT = WM*MB*32.
struct PfCand { int MB, NPW, WM, WP; };
static const PfCand kCandsS2[] = {
    {2, 2, 2, 2},   // 128 channels, 4 waves, 4 rows
    {3, 1, 2, 4},   // 192 channels, 8 waves, 4 rows
    {2, 1, 1, 4},   //  64 channels, 4 waves, 4 rows
};
static const PfCand kCandsTZ[] = {
    {1, 2, 4, 2},   // 128 channels, 8 waves, 4 input rows
    // 64 channels, 4 waves, 4 input rows: every wave holds all 64 channels of one input row (8 ds_read_b128 per 6 MFMAs instead of
    // 5.25, but the fused LayerNorm stays inside the lane: no cross-wave exchange, three barriers fewer): 0.41 -> 0.38 ms against ...
    {2, 1, 1, 4},
    {1, 2, 2, 2},   // ... two channel parts x two row pairs (kept for CDC_PF_PLAN=1,2,2,2)
};
static const PfCand kCands[] = {
    {2, 2, 4, 2},   // 256 channels, 8 waves, 4 rows
    {3, 2, 2, 4},   // 192 channels, 8 waves, 8 rows
    {3, 1, 2, 4},   // 192 channels, 8 waves, 4 rows: twice the workgroups where 8-row tiles leave half the CUs idle (32 x 32 maps at batch 32)
    {2, 2, 2, 2},   // 128 channels, 4 waves, 4 rows
    {2, 2, 2, 4},   // 128 channels, 8 waves, 8 rows
    {2, 2, 1, 4},   //  64 channels, 4 waves, 8 rows
};

bool pf_make_plan(const PfShape &s, PfPlan *p) {
    if (s.uf) {
        // the first layer as a 7x1 convolution over the 21 kx-unfolded channels of the 3-channel image, patches built in the kernel
        // (conv_pf_kernel, UF = 3): 64 output channels, 8-row tiles, two workgroups per CU
        if (s.uf != 3 || s.KH != 7 || s.KW != 1 || s.Cin != 21 || s.Cout != 64 || s.cop != 64 || s.C0 || s.nz != 1 || s.stride != 1 || s.tz != 1 ||
            s.Wo < 32 || dev_env("CDC_NO_PF_UF"))
            return false;
        const int ring = pf_ring(2, 2, 1, 4, 7, 1);
        const double wgs = (double)((s.Wo + 31) / 32) * ((s.Ho + 7) / 8) * s.B;
        const double min_wgs = dev_env("CDC_PF_UF_MIN_WGS") ? atof(dev_env("CDC_PF_UF_MIN_WGS")) : 512.0;
        if (ring < 3 || wgs < min_wgs) return false;
        p->MB = 2; p->NPW = 2; p->WM = 1; p->WP = 4; p->ring = ring;
        p->tiles_x = (s.Wo + 31) / 32; p->tiles_y = (s.Ho + 7) / 8; p->groups = 1;
        p->lds_bytes = (size_t)2 * pf_patch_units(2, 4, 7, 1) * 16 + (size_t)ring * pf_rows(2, 2) * 64 * 16 + (size_t)pf_uf_stage_floats(2, 4) * 4;
        return true;
    }
    if (s.KH == 1 && s.KW == 7) {
        // the row-folded final convolution (unet.py:104: 7x7 to out_dim channels as 1x7 to 7 * out_dim <= 32 virtual channels):
        // one 32-channel block, 8-row tiles, three workgroups per CU; plain fp32 output with masked channel stores
        if (s.Cout > 32 || s.cop != 32 || s.Cin % 16 || s.C0 || s.nz != 1 || s.stride != 1 || s.tz != 1 || s.need_all_cout || s.Wo < 32 ||
            dev_env("CDC_NO_PF_17"))
            return false;
        const int ring = pf_ring(1, 2, 1, 4, 1, 7);
        const double wgs = (double)((s.Wo + 31) / 32) * ((s.Ho + 7) / 8) * s.B;
        const double min_wgs = dev_env("CDC_PF_17_MIN_WGS") ? atof(dev_env("CDC_PF_17_MIN_WGS")) : 768.0;
        if (!ring || (s.Cin / 16) * 7 < ring - 1 || wgs < min_wgs) return false;
        p->MB = 1; p->NPW = 2; p->WM = 1; p->WP = 4; p->ring = ring;
        p->tiles_x = (s.Wo + 31) / 32; p->tiles_y = (s.Ho + 7) / 8; p->groups = 1;
        p->lds_bytes = (size_t)2 * pf_patch_units(2, 4, 1, 7) * 16 + (size_t)ring * pf_rows(1, 2) * 32 * 16;
        return true;
    }
    if (s.Cout % 32 || s.Cin % 16 || (s.C0 % 16)) return false;
    if (s.Wo < 32) return false;                              // 32-pixel blocks are rows of the image (lognbw = 5)
    const char *force = dev_env("CDC_PF_PLAN");                // tuning aid: "MB,NPW,WM,WP" (read per call: the tests switch it)
    int f[4] = {0, 0, 0, 0};
    if (force) sscanf(force, "%d,%d,%d,%d", &f[0], &f[1], &f[2], &f[3]);
    double best = -1;
    if (s.stride == 2) {
        // Downsample: no fused LayerNorm, so any channel-group width divides the work; the widest that fits reads the patch once
        if (s.KH != 3 || s.KW != 3 || s.nz != 1 || s.C0 || dev_env("CDC_NO_PF_S2")) return false;
        for (const PfCand &c : kCandsS2) {
            if (f[0] && (c.MB != f[0] || c.NPW != f[1] || c.WM != f[2] || c.WP != f[3])) continue;
            const int COPT = c.WM * c.MB * 32, TH = c.WP * c.NPW;
            if (s.Cout % COPT || (s.need_all_cout && COPT != s.Cout)) continue;
            const int ring = pf_ring(c.MB, c.NPW, c.WM, c.WP, 3, 3, 2);
            const int tps = pf_tps(c.MB, c.NPW, c.WM, c.WP, 3, 3, 2);
            const int S = (s.Cin / 16) * 9 / tps;                // weight stages of a tile
            if (!ring || S < ring - 1) continue;
            const double wgs = (double)((s.Wo + 31) / 32) * ((s.Ho + TH - 1) / TH) * s.B * (s.Cout / COPT);
            const double min_wgs = dev_env("CDC_PF_S2_MIN_WGS") ? atof(dev_env("CDC_PF_S2_MIN_WGS")) : 128.0;
            if (wgs < min_wgs) continue;                        // one workgroup per CU: fewer than one round leaves CUs idle
            const double score = (double)COPT;
            if (score <= best) continue;
            best = score;
            const size_t patch = (size_t)pf_patch_bufs(3, 3, 2) * pf_patch_units(c.NPW, c.WP, 3, 3, 2) * 16, wst = (size_t)tps * pf_rows(c.MB, c.NPW) * COPT * 16;
            p->MB = c.MB; p->NPW = c.NPW; p->WM = c.WM; p->WP = c.WP;
            p->ring = ring;
            p->tiles_x = (s.Wo + 31) / 32; p->tiles_y = (s.Ho + TH - 1) / TH;
            p->groups = s.Cout / COPT;
            p->lds_bytes = std::max(patch + ring * wst, sizeof(float) * (size_t)(4 * COPT + 2 * c.WM * c.WP * c.NPW * 32));
        }
        return best >= 0;
    }
    if (s.tz == 4) {
        // ConvTranspose2d 4x4 / stride 2 / pad 1, the four phases fused in one workgroup (Ho x Wo = the INPUT extent)
        if (s.KH != 2 || s.KW != 2 || s.C0 || dev_env("CDC_NO_PF_TZ")) return false;
        for (const PfCand &c : kCandsTZ) {
            if (f[0] && (c.MB != f[0] || c.NPW != f[1] || c.WM != f[2] || c.WP != f[3])) continue;
            const int COPT = c.WM * c.MB * 32, TH = c.WP * c.NPW;
            if (COPT != s.Cout) continue;                       // (a channel LayerNorm may be fused: one group)
            const int ring = pf_ring(c.MB, c.NPW, c.WM, c.WP, 2, 2, 1, 4);
            const int tps = pf_tps(c.MB, c.NPW, c.WM, c.WP, 2, 2, 1, 4);
            const int S = (s.Cin / 16) * 16 / tps;               // weight stages of a tile
            if (!ring || S < ring - 1) continue;
            const double wgs = (double)((s.Wo + 31) / 32) * ((s.Ho + TH - 1) / TH) * s.B;
            const double min_wgs = dev_env("CDC_PF_TZ_MIN_WGS") ? atof(dev_env("CDC_PF_TZ_MIN_WGS")) : 128.0;
            if (wgs < min_wgs) continue;
            best = COPT;
            const size_t patch = (size_t)2 * pf_patch_units(c.NPW, c.WP, 2, 2, 1, 4) * 16, wst = (size_t)tps * pf_rows(c.MB, 4 * c.NPW) * COPT * 16;
            p->MB = c.MB; p->NPW = c.NPW; p->WM = c.WM; p->WP = c.WP;
            p->ring = ring;
            p->tiles_x = (s.Wo + 31) / 32; p->tiles_y = (s.Ho + TH - 1) / TH;
            p->groups = 1;
            p->lds_bytes = std::max(patch + ring * wst, sizeof(float) * (size_t)(4 * COPT + 2 * c.WM * c.WP * 4 * c.NPW * 32));
            break;
        }
        return best >= 0;
    }
    for (const PfCand &c : kCands) {
        if (f[0] && (c.MB != f[0] || c.NPW != f[1] || c.WM != f[2] || c.WP != f[3])) continue;
        const int COPT = c.WM * c.MB * 32, NW = c.WM * c.WP;
        if (s.Cout % COPT) continue;
        if (s.need_all_cout && COPT != s.Cout) continue;
        if (!pf_lookup(c.MB, c.NPW, c.WM, c.WP, s.KH, s.KW)) continue;
        const int TH = c.WP * c.NPW, PH = TH + s.KH - 1, PW = 32 + s.KW - 1;
        const int xsw = (4 * PH * PW + 63) / 64;
        if (xsw > 2 * kPfXS) continue;
        const int taps = s.KH * s.KW, npb = taps == 1 ? 3 : 2;
        const size_t patch = (size_t)npb * xsw * 64 * 16, wst = (size_t)pf_rows(c.MB, c.NPW) * COPT * 16;
        const int ring = pf_ring(c.MB, c.NPW, c.WM, c.WP, s.KH, s.KW);    // compile-time in the kernel
        if (!ring) continue;
        const int S = (s.Cin / 16) * taps;
        if (S < ring - 1) continue;                               // the prologue fills ring - 1 slots
        const double wgs = (double)((s.Wo + 31) / 32) * ((s.Ho + TH - 1) / TH) * s.B * (s.Cout / COPT) * s.nz;
        // Few workgroups (small batches at the 64^2 / 32^2 levels): the register-staged kernel with split-K fills the
        // chip better (batch 1: 192->192 @64^2 0.032 ms against 0.09 here; 128->128 @128^2 with 128 workgroups: 0.07
        // against 0.038 -- the threshold sits between the two)
        const double min_waves = dev_env("CDC_PF_MIN_WAVES") ? atof(dev_env("CDC_PF_MIN_WAVES")) : 256.0;     // (read per plan: the tests switch it)
        if (wgs * NW < min_waves) continue;
        // (the 4-row 192-channel shape exists to fill the chip at batch 32; at batch 1 its 32 workgroups lose to the register-staged
        //  kernel with split-K: 192 -> 192 @64^2 0.054 against 0.032 ms)
        if (c.MB == 3 && c.NPW == 1 && wgs < 128.0) continue;
        const double fill = std::min(1.0, wgs * NW / 2048.0);
        const double reads = (3.0 * c.MB + 2.0 * c.NPW) / (3.0 * c.MB * c.NPW);   // ds_read_b128 per MFMA
        const double score = fill * (1.0 - 0.35 * reads) * (s.Cout / COPT > 1 ? 0.9 : 1.0);
        if (score > best) {
            best = score;
            p->MB = c.MB; p->NPW = c.NPW; p->WM = c.WM; p->WP = c.WP;
            p->ring = ring;
            p->tiles_x = (s.Wo + 31) / 32; p->tiles_y = (s.Ho + TH - 1) / TH;
            p->groups = s.Cout / COPT;
            p->lds_bytes = std::max(patch + ring * wst, sizeof(float) * (size_t)(4 * COPT + 2 * c.WM * c.WP * c.NPW * 32));
        }
    }
    return best >= 0;
}

hipError_t pf_launch(PfArgs a, const PfPlan &p, int B, int nz, hipStream_t st) {
    if (p.pf3_epv) return pf3_launch(a, p, B, st);
    pf_kernel_fn fn = pf_lookup(p.MB, p.NPW, p.WM, p.WP, a.KH, a.KW, a.stride == 2 ? 2 : 1, a.tz == 4 ? 4 : 1);
    if (!fn) return hipErrorInvalidValue;
    if (a.tz == 4) nz = 1;                                   // the phases are evaluated inside the workgroup
    a.lognbw = 5;
    a.dbg = 0;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.B = B; a.ring = p.ring;
    if (hipError_t e = ensure_dynamic_lds((const void *)fn, p.lds_bytes); e != hipSuccess) return e;
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * B), (unsigned)p.groups, (unsigned)nz);
    a.xcd_remap = (grid.x % 8 == 0 && grid.x >= 64) ? 1 : 0;
#ifdef CDC_TIMELINE
    // Development build only (tools/build_variant.sh timeline -DCDC_TIMELINE): per-workgroup cycle categories of conv_pf_kernel, start /
    // end stamps and the CU each workgroup ran on, summarised on stderr for the first launches of every layer shape.
    static unsigned long long *tl_dev = nullptr;
    static std::map<std::string, int> seen;
    const size_t tl_wgs = (size_t)grid.x * grid.y * grid.z;
    char key[96];
    snprintf(key, sizeof key, "%dx%d s%d tz%d %d->%d out %dx%d", a.KH, a.KW, a.stride == 2 ? 2 : 1, a.tz == 4 ? 4 : 1, a.Cin, a.Cout, a.Ho, a.Wo);
    a.tl = nullptr;
    if (tl_wgs <= (1u << 18) && seen[key]++ < 3) {
        if (!tl_dev) hipMalloc(&tl_dev, sizeof(unsigned long long) * 16 * (1u << 18));
        hipMemsetAsync(tl_dev, 0, sizeof(unsigned long long) * 16 * tl_wgs, st);
        a.tl = tl_dev;
    }
#endif
    hipLaunchKernelGGL(fn, grid, dim3(64 * p.WM * p.WP), p.lds_bytes, st, a);
#ifdef CDC_TIMELINE
    if (a.tl) {
        hipStreamSynchronize(st);
        std::vector<unsigned long long> h(16 * tl_wgs);
        hipMemcpy(h.data(), tl_dev, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        double cat[7] = {0}, life = 0;
        std::map<unsigned long long, int> per_cu;
        std::vector<unsigned long long> starts;
        for (size_t w = 0; w < tl_wgs; ++w) {
            const unsigned long long *r = &h[w * 16];
            for (int c = 0; c < 7; ++c) cat[c] += (double)r[c];
            t0 = std::min(t0, r[7]); t1 = std::max(t1, r[8]);
            life += (double)(r[8] - r[7]);
            per_cu[(r[10] & 0xf) << 32 | (r[9] & 0xffff00)]++;      // XCC id, SE / SH / CU bits of HW_ID
            starts.push_back(r[7]);
        }
        std::sort(starts.begin(), starts.end());
        int cu_min = 1 << 30, cu_max = 0;
        for (auto &kv : per_cu) { cu_min = std::min(cu_min, kv.second); cu_max = std::max(cu_max, kv.second); }
        static const char *names[7] = {"setup", "prologue issue", "prologue wait", "main loop", "epilogue parameters", "epilogue arithmetic", "epilogue stores"};
        fprintf(stderr, "[pf timeline] conv %s: %zu workgroups x %d threads, lds %zu; kernel span %llu cycles, mean workgroup life %.0f; %zu CUs used, %d .. %d workgroups per CU;"
                        " start of the median / last workgroup at %llu / %llu; per workgroup (wave 0):", key, tl_wgs, 64 * p.WM * p.WP, p.lds_bytes, t1 - t0, life / tl_wgs,
                per_cu.size(), cu_min, cu_max, starts[starts.size() / 2] - t0, starts.back() - t0);
        for (int c = 0; c < 7; ++c) fprintf(stderr, "  %s %.0f", names[c], cat[c] / tl_wgs);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

// fp32 NCHW -> PF (interior only; the halo stays zero).  One thread = one unit pair (8 channels of a pixel).
__global__ void __launch_bounds__(256) pf_pack_kernel(const float *src, long long src_bs, uint4 *dst, long long dst_bs,
                                                      int C, int H, int W) {
    const int b = blockIdx.y;
    const long long n = (long long)(C / 8) * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % W);
    const long long t = i / W;
    const int y = (int)(t % H), g = (int)(t / H);
    const float *sp = src + (size_t)b * src_bs + ((size_t)g * 8 * H + y) * W + x;
    f16x8 hv, lv;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        _Float16 h, l;
        split2h(sp[(size_t)q * H * W], h, l);
        hv[q] = h; lv[q] = l;
    }
    const long long ps = (long long)(H + 2) * (W + 2);
    uint4 *dp = dst + (size_t)b * dst_bs + (long long)g * 2 * ps + (long long)(y + 1) * (W + 2) + x + 1;
    dp[0] = __builtin_bit_cast(uint4, hv);
    dp[ps] = __builtin_bit_cast(uint4, lv);
}

hipError_t pf_pack_launch(const float *src, long long src_bs, void *dst, long long dst_bs, int C, int H, int W, int B,
                          hipStream_t st) {
    const long long n = (long long)(C / 8) * H * W;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(pf_pack_kernel, grid, dim3(256), 0, st, src, src_bs, reinterpret_cast<uint4 *>(dst), dst_bs, C, H, W);
    return hipGetLastError();
}

// fp32 NCHW -> accumulator order [B][C / 4][HW][4] (a hoisted partial-sum tensor read by conv_pf_kernel's epilogue with 16-byte loads)
__global__ void __launch_bounds__(256) c4_pack_kernel(const float *src, long long src_bs, float4 *dst, int C, long long HW) {
    const int b = blockIdx.y;
    const long long n = (long long)(C / 4) * HW;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long pix = i % HW, g = i / HW;
    const float *sp = src + (size_t)b * src_bs + (size_t)g * 4 * HW + pix;
    dst[(size_t)b * n + i] = make_float4(sp[0], sp[HW], sp[2 * HW], sp[3 * HW]);
}

hipError_t c4_pack_launch(const float *src, long long src_bs, float *dst, int C, long long HW, int B, hipStream_t st) {
    const long long n = (long long)(C / 4) * HW;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(c4_pack_kernel, grid, dim3(256), 0, st, src, src_bs, reinterpret_cast<float4 *>(dst), C, HW);
    return hipGetLastError();
}

// PF -> fp32 NCHW: a = h + l' 2^-11 (interior only).  One thread = one unit pair.
__global__ void __launch_bounds__(256) pf_unpack_kernel(const uint4 *src, long long src_bs, float *dst, long long dst_bs, int C, int H, int W) {
    const int b = blockIdx.y;
    const long long n = (long long)(C / 8) * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % W);
    const long long t = i / W;
    const int y = (int)(t % H), g = (int)(t / H);
    const long long ps = (long long)(H + 2) * (W + 2);
    const uint4 *sp = src + (size_t)b * src_bs + (long long)g * 2 * ps + (long long)(y + 1) * (W + 2) + x + 1;
    const f16x8 hv = __builtin_bit_cast(f16x8, sp[0]), lv = __builtin_bit_cast(f16x8, sp[ps]);
    float *dp = dst + (size_t)b * dst_bs + ((size_t)g * 8 * H + y) * W + x;
#pragma unroll
    for (int q = 0; q < 8; ++q) dp[(size_t)q * H * W] = (float)hv[q] + (float)lv[q] * (1.0f / 2048.0f);
}

hipError_t pf_unpack_launch(const void *src, long long src_bs, float *dst, long long dst_bs, int C, int H, int W, int B, hipStream_t st) {
    const long long n = (long long)(C / 8) * H * W;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(pf_unpack_kernel, grid, dim3(256), 0, st, reinterpret_cast<const uint4 *>(src), src_bs, dst, dst_bs, C, H, W);
    return hipGetLastError();
}

}  // namespace cdc
