// conv_inst_t.hip -- instantiations, launch plan and launcher of conv_ws_kernel (conv_ws_kernel.h): the weight-stationary 3x3
// convolution of the few-pixel levels.
#include "cdc_internal.h"
#include "conv_ws_kernel.h"

namespace cdc {

namespace {
constexpr int kWsNPB = 4;                 // pixel blocks per workgroup: 128 pixels (72 weight registers feed 4 x 27 MFMAs per chunk)

ws_kernel_fn ws_lookup(int W, bool ln) {
    switch (W) {
        case 8: return ln ? conv_ws_kernel<8, kWsNPB, true> : conv_ws_kernel<8, kWsNPB, false>;
        case 16: return ln ? conv_ws_kernel<16, kWsNPB, true> : conv_ws_kernel<16, kWsNPB, false>;
        case 32: return ln ? conv_ws_kernel<32, kWsNPB, true> : conv_ws_kernel<32, kWsNPB, false>;
    }
    return nullptr;
}
}  // namespace

// Cin / C0 / Cout: channels (C0 = those of the first source, Cin when there is one); H x W: the map; B: batch the plan is made for.
bool ws_make_plan(int Cin, int C0, int Cout, int H, int W, int B, WsPlan *p) {
    static const bool off = dev_env("CDC_NO_WS") != nullptr;
    if (off || (W != 8 && W != 16 && W != 32) || H < 2 || (Cin % 16) || (C0 % 16) || (Cout % 32) || Cin < 32) return false;
    const int hw = H * W, tpx = kWsNPB * 32;
    if (hw % 32) return false;
    if (tpx >= hw ? ((tpx % hw) || (hw % 64)) : ((hw % tpx) || ((tpx / W) < 1))) return false;     // whole images (each a multiple of a loader pass), or an image in equal row bands
    const long long tot_px = (long long)B * hw;
    if (tot_px % tpx) return false;
    const int tiles = (int)(tot_px / tpx), groups = Cout / 32, nchunk = Cin / 16;
    // K slices over the waves of the workgroup: up to 8 waves (two per SIMD: the kernel holds 200+ registers), equal shares
    int waves = 0;
    for (int w = 8; w >= 2; --w)
        if (nchunk % w == 0) { waves = w; break; }
    if (!waves) return false;
    // two workgroups per CU when both fit (LDS, 8 waves): fewer waves per workgroup then
    const long long wgs = (long long)tiles * groups;
    if (wgs > 2 * device_cus() && waves > 4)
        for (int w = 4; w >= 2; --w)
            if (nchunk % w == 0) { waves = w; break; }
    const long long min_wgs = dev_env("CDC_WS_MIN_WGS") ? atoll(dev_env("CDC_WS_MIN_WGS")) : 96;
    if (wgs < min_wgs) return false;
    const size_t lds = ws_lds_bytes(W, H, kWsNPB, waves, Cin, true);      // (with the LayerNorm-on-load tables: the larger of the two forms)
    if (lds > 160 * 1024) return false;
    p->W = W; p->NPB = kWsNPB; p->waves = waves; p->tiles = tiles; p->groups = groups; p->lds_bytes = lds;
    return true;
}

hipError_t ws_launch(WsArgs a, const WsPlan &p, hipStream_t st) {
    const bool ln = a.ln_part != nullptr;
    ws_kernel_fn fn = ws_lookup(p.W, ln);
    if (!fn) return hipErrorInvalidValue;
    static bool attr_done[16][3][2] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int wi = p.W == 8 ? 0 : (p.W == 16 ? 1 : 2);
    if (dev < 0 || dev >= 16 || !attr_done[dev][wi][ln]) {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) attr_done[dev][wi][ln] = true;
    }
    a.tiles = p.tiles; a.groups = p.groups;
    a.dbg = 0;
#ifdef CDC_WS_LAB
    a.dbg = dev_env("CDC_WS_DBG") ? atoi(dev_env("CDC_WS_DBG")) : 0;
#endif
    a.cpw = a.nchunk / p.waves;
    const unsigned grid = (unsigned)(p.tiles * p.groups);
    a.xcd_remap = (grid % 8 == 0 && grid >= 64 && !dev_env("CDC_NO_XCD")) ? 1 : 0;
#ifdef CDC_WS_LAB
    static unsigned long long *tl = nullptr;
    static int tl_n = 0;
    if (dev_env("CDC_WS_TL")) {
        if (!tl) (void)hipMalloc((void **)&tl, (size_t)8192 * 16 * 8);
        (void)hipMemsetAsync(tl, 0, (size_t)8192 * 16 * 8, st);
        a.tl = grid <= 8192 ? tl : nullptr;
    } else a.tl = nullptr;
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * p.waves), ws_lds_bytes(p.W, a.H, p.NPB, p.waves, a.Cin, ln), st, a);
#ifdef CDC_WS_LAB
    if (a.tl && tl_n++ >= 40 && tl_n < 63) {                 // the launches of the third DDIM iteration: mean stamps over the grid
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)grid * 16);
        (void)hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost);
        double d[16] = {0}; unsigned long long first = ~0ull, last = 0;
        for (unsigned g = 0; g < grid; ++g) {
            const unsigned long long *t = &h[(size_t)g * 16];
            for (int i = 1; i < 16; ++i) if (t[i]) d[i] += (double)(t[i] - t[0]);
            if (t[0] < first) first = t[0];
            if (t[15] > last) last = t[15];
        }
        fprintf(stderr, "[ws-tl] Cin=%d Cout=%d H=%d W=%d waves=%d ln=%d grid=%u | span %llu | mean since start:", a.Cin, a.Cout, a.H, p.W, p.waves, (int)ln, grid, last - first);
        for (int i = 1; i < 16; ++i) fprintf(stderr, " %d:%.0f", i, d[i] / grid);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

}  // namespace cdc
