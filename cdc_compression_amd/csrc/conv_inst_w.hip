// conv_inst_w.hip -- instantiations, plan chooser and launcher of conv_pw_kernel (pointwise convolution, activations
// staged per wave by 4-byte LDS-DMA from the fp32 tensor).
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include <stdlib.h>

#include "cdc_internal.h"
#include "conv_pw_kernel.h"

namespace cdc {

typedef void (*pw_kernel_fn)(const PfArgs);
static pw_kernel_fn pw_lookup(int MB, int NPW, int WM, int WP, bool x16 = false) {
    if (x16) {      // activations by 16-byte LDS-DMA (conv_pw_kernel.h, X16)
        if (MB == 2 && NPW == 2 && WM == 1 && WP == 4) return conv_pw_kernel<2, 2, 1, 4, true>;
        if (MB == 2 && NPW == 2 && WM == 2 && WP == 2) return conv_pw_kernel<2, 2, 2, 2, true>;
        if (MB == 2 && NPW == 2 && WM == 4 && WP == 2) return conv_pw_kernel<2, 2, 4, 2, true>;
        if (MB == 3 && NPW == 1 && WM == 2 && WP == 4) return conv_pw_kernel<3, 1, 2, 4, true>;
        return nullptr;
    }
    if (MB == 2 && NPW == 2 && WM == 1 && WP == 4) return conv_pw_kernel<2, 2, 1, 4>;
    if (MB == 2 && NPW == 2 && WM == 2 && WP == 2) return conv_pw_kernel<2, 2, 2, 2>;
    if (MB == 2 && NPW == 2 && WM == 4 && WP == 2) return conv_pw_kernel<2, 2, 4, 2>;
    if (MB == 3 && NPW == 1 && WM == 2 && WP == 4) return conv_pw_kernel<3, 1, 2, 4>;
    return nullptr;
}
struct PwCand { int MB, NPW, WM, WP; };
static const PwCand kPwCands[] = {
    {2, 2, 4, 2},   // 256 channels per workgroup, 8 waves, 4 rows
    {3, 1, 2, 4},   // 192 channels, 8 waves, 4 rows
    {2, 2, 2, 2},   // 128 channels, 4 waves, 4 rows
    {2, 2, 1, 4},   //  64 channels, 4 waves, 8 rows
};

bool pw_make_plan(const PfShape &s, PfPlan *p) {
    if (s.KH != 1 || s.KW != 1 || s.nz != 1) return false;
    if (s.Cout % 32 || s.Cin % 16 || (s.C0 % 16) || s.Cin < 32) return false;
    // maps narrower than 32 pixels: 32-pixel blocks of the flattened image, a workgroup may span images ("linear" tiles)
    const bool lin = s.Wo < 32;
    if (lin && ((s.Ho * s.Wo) % 32)) return false;
    const double min_waves = dev_env("CDC_PW_MIN_WAVES") ? atof(dev_env("CDC_PW_MIN_WAVES")) : 512.0;   // (per call: tests switch it)
    const bool x16 = (lin || ((s.Wo & 3) == 0 && s.Wo >= 4)) && !dev_env("CDC_NO_PW_X16");          // rows of 4-pixel granularity
    double best = -1;
    for (const PwCand &c : kPwCands) {
        const int COPT = c.WM * c.MB * 32, NW = c.WM * c.WP;
        if (s.Cout % COPT) continue;
        if (s.need_all_cout && COPT != s.Cout) continue;
        const int ring = pw_ring(c.MB, c.NPW, c.WM, c.WP, x16);
        if (ring < 5 || s.Cin / 16 < 2) continue;
        const int TH = c.WP * c.NPW;
        const int groups = s.Cout / COPT;
        const long long nb = (long long)s.B * (s.Ho * s.Wo / 32);     // linear mode: blocks of the whole batch
        const double wgs = lin ? (double)((nb + TH - 1) / TH) * groups
                               : (double)((s.Wo + 31) / 32) * ((s.Ho + TH - 1) / TH) * s.B * groups;
        if (wgs * NW < min_waves) continue;
        const double fill = std::min(1.0, wgs * NW / 2048.0);
        // every channel group reads and splits the activations again: prefer few groups; wider tiles reuse the weights
        const double score = fill / (1.0 + 0.25 * (groups - 1)) * (c.MB * c.NPW >= 4 ? 1.0 : 0.8);
        if (score > best) {
            best = score;
            p->MB = c.MB; p->NPW = c.NPW; p->WM = c.WM; p->WP = c.WP;
            p->ring = ring;
            p->lin = lin ? 1 : 0;
            p->tiles_x = lin ? (int)((nb + TH - 1) / TH) : (s.Wo + 31) / 32;
            p->tiles_y = lin ? 1 : (s.Ho + TH - 1) / TH;
            p->groups = groups;
            p->lds_bytes = (size_t)ring * pf_rows(c.MB, c.NPW) * COPT * 16 + pw_x_bytes(c.NPW, c.WM, c.WP, x16);
            p->x16 = x16 ? 1 : 0;
        }
    }
    return best >= 0;
}

hipError_t pw_launch(PfArgs a, const PfPlan &p, int B, hipStream_t st) {
    pw_kernel_fn fn = pw_lookup(p.MB, p.NPW, p.WM, p.WP, p.x16 != 0);
    if (!fn) return hipErrorInvalidValue;
    a.lognbw = 5;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.B = B; a.ring = p.ring;
    if (hipError_t e = ensure_dynamic_lds((const void *)fn, p.lds_bytes); e != hipSuccess) return e;
    a.lin = p.lin;
    if (const char *e = dev_env("CDC_PW_DBG")) a.dbg = atoi(e); else
    a.dbg = dev_env("CDC_PW_COUNTED_WAIT") ? 1024 : 0;  // (A/B: the counted s_waitcnt of round 4 instead of vmcnt(0) per step, conv_pw_kernel.h)
    dim3 grid((unsigned)(p.lin ? p.tiles_x : p.tiles_x * p.tiles_y * B), (unsigned)p.groups, 1);
    a.xcd_remap = (grid.x % 8 == 0 && grid.x >= 64) ? 1 : 0;
#ifdef CDC_TIMELINE
    // Development build only: cycle categories per workgroup (see pf_launch)
    static unsigned long long *tl_dev = nullptr;
    static std::map<std::string, int> seen;
    const size_t tl_wgs = (size_t)grid.x * grid.y;
    char key[96];
    snprintf(key, sizeof key, "pw 1x1 %d->%d out %dx%d", a.Cin, a.Cout, a.Ho, a.Wo);
    a.tl = nullptr;
    if (tl_wgs <= (1u << 18) && seen[key]++ < 2) {
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
        double cat[7] = {0}, life = 0;
        for (size_t w = 0; w < tl_wgs; ++w) {
            const unsigned long long *r = &h[w * 16];
            for (int c = 0; c < 7; ++c) cat[c] += (double)r[c];
            life += (double)(r[8] - r[7]);
        }
        static const char *names[7] = {"setup", "prologue issue", "prologue wait", "main loop", "epilogue parameters", "-", "epilogue blocks"};
        fprintf(stderr, "[pf timeline] conv %s: %zu workgroups x %d threads, lds %zu; mean workgroup life %.0f; per workgroup (wave 0):", key, tl_wgs, 64 * p.WM * p.WP, p.lds_bytes,
                life / tl_wgs);
        for (int c = 0; c < 7; ++c) fprintf(stderr, "  %s %.0f", names[c], cat[c] / tl_wgs);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

}  // namespace cdc
