// conv_inst_t.hip -- instantiations, launch plan and launcher of conv_ws_kernel (conv_ws_kernel.h): the weight-stationary 3x3
// convolution of the few-pixel levels.
#include "cdc_internal.h"
#include "conv_ws_kernel.h"
#include "conv_ws1_kernel.h"

namespace cdc {

namespace {
// pixel blocks per workgroup: 4 (128 pixels: a kernel row's weight registers feed 4 x 9 MFMAs) wherever that fills the chip,
// 2 (64 pixels) for the small batches
ws_kernel_fn ws_lookup(int W, int NPB, int stride) {
    if (stride == 2) {                         // Downsample onto an 8 / 16 / 32-wide map, 64-pixel tiles
        if (NPB != 2) return nullptr;
        switch (W) {
            case 8: return conv_ws_kernel<8, 2, 2>;
            case 16: return conv_ws_kernel<16, 2, 2>;
            case 32: return conv_ws_kernel<32, 2, 2>;
        }
        return nullptr;
    }
    switch (W) {
        case 8: return NPB == 4 ? conv_ws_kernel<8, 4> : conv_ws_kernel<8, 2>;
        case 16: return NPB == 4 ? conv_ws_kernel<16, 4> : conv_ws_kernel<16, 2>;
        case 32: return NPB == 4 ? conv_ws_kernel<32, 4> : conv_ws_kernel<32, 2>;
        case 64: return NPB == 4 ? conv_ws_kernel<64, 4> : conv_ws_kernel<64, 2>;
    }
    return nullptr;
}
}  // namespace

// Cin / C0 / Cout: channels (C0 = those of the first source, Cin when there is one); H x W: the map; B: batch the plan is made for.
constexpr long long kWsMaxWgsDefault = 2048;   // eight rounds of the chip; a bound of 1280 changes nothing measurable at batch 4 / 8 / 16 (3.366 / 4.785 / 7.107 against
                                                // 3.363 / 4.793 / 7.094 ms per iteration, profiles/planner_ab_r06.txt): no measured crossover below the bound
bool ws_make_plan(int Cin, int C0, int Cout, int H, int W, int B, int stride, WsPlan *p) {
    if ((W != 8 && W != 16 && W != 32 && W != 64) || H < 2 || (Cin % 16) || (C0 % 16) || (Cout % 32) || Cin < 32) return false;
    const int hw = H * W, groups = Cout / 32, nchunk = Cin / 16;
    if (hw % 32) return false;
    if (stride != 1 && (stride != 2 || W > 32)) return false;
    const long long min_wgs = dev_env("CDC_WS_MIN_WGS") ? atoll(dev_env("CDC_WS_MIN_WGS")) : 4;
    const int force_npb = dev_env("CDC_WS_NPB") ? atoi(dev_env("CDC_WS_NPB")) : 0;
    for (int npb : {4, 2}) {
        if (force_npb && npb != force_npb) continue;
        if (npb == 4 && (stride == 2 || (W > 16 && !force_npb))) continue;        // (the wider maps' loader holds more values in flight: 128-pixel tiles spill there)
        const int tpx = npb * 32;
        // a tile = whole images (each a multiple of a loader pass of 64 pixels), or an image in equal bands of rows
        if (tpx >= hw ? ((tpx % hw) || (hw % 64)) : ((hw % tpx) || (tpx % W))) continue;
        const long long tot_px = (long long)B * hw;
        if (tot_px % tpx) continue;
        const int tiles = (int)(tot_px / tpx);
        const long long wgs = (long long)tiles * groups;
        // 128-pixel tiles from 5/8 of the chip's CUs on (round 6: the 320-channel layers of the 8 x 8 level at batch 32, 160 workgroups of 128
        // pixels against 320 of 64: 12.02 / 12.02 against 12.06 / 12.05 ms per iteration, profiles/planner_ab_r06.txt); 64-pixel tiles below
        if (npb == 4 && !force_npb && wgs < (5 * device_cus()) / 8) continue;
        if (wgs < min_wgs) return false;
        // upper bound of a stride-1 launch (block() tries this kernel before the fused register-staged one for every map up to
        // 64 wide that the plane-operand path did not take -- encoder / context-decoder programs, mid-size batches at 64 x 64).
        // CDC_WS_MAX_WGS: A/B knob (0 = no bound).
        if (stride == 1 && !dev_env("CDC_WS_MIN_WGS")) {
            const long long max_wgs = dev_env("CDC_WS_MAX_WGS") ? atoll(dev_env("CDC_WS_MAX_WGS")) : kWsMaxWgsDefault;
            if (max_wgs > 0 && wgs > max_wgs) return false;
        }
        if (stride == 2 && wgs > device_cus() && !dev_env("CDC_WS_MIN_WGS")) return false;     // (its one-workgroup-per-CU launches pay as a single round only)
        // K slices over the waves of the workgroup: up to 8 waves (two per SIMD: the kernel holds ~250 registers), equal shares;
        // two workgroups per CU when the launch has more than two per CU (LDS and registers then want <= 4 waves each)
        int waves = 0;
        for (int w = ((wgs > 2 * device_cus() || stride == 2) ? 4 : 8); w >= 2; --w)
            if (nchunk % w == 0 && ws_lds_bytes(W, H, npb, w, stride) <= 160 * 1024) { waves = w; break; }
        if (!waves) continue;
        const size_t lds = ws_lds_bytes(W, H, npb, waves, stride);
        p->W = W; p->NPB = npb; p->stride = stride; p->waves = waves; p->tiles = tiles; p->groups = groups; p->lds_bytes = lds;
        return true;
    }
    return false;
}

hipError_t ws_launch(WsArgs a, const WsPlan &p, hipStream_t st) {
    ws_kernel_fn fn = ws_lookup(p.W, p.NPB, p.stride);
    if (!fn) return hipErrorInvalidValue;
    static bool attr_done[16][4][2][2] = {};               // [device][width][NPB == 4][stride == 2]
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int wi = p.W == 8 ? 0 : (p.W == 16 ? 1 : (p.W == 32 ? 2 : 3));
    if (dev < 0 || dev >= 16 || !attr_done[dev][wi][p.NPB == 4][p.stride == 2]) {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) attr_done[dev][wi][p.NPB == 4][p.stride == 2] = true;
    }
    a.tiles = p.tiles; a.groups = p.groups;
    a.dbg = 0;
#ifdef CDC_WS_LAB
    a.dbg = getenv("CDC_WS_DBG") ? atoi(getenv("CDC_WS_DBG")) : 0;      // (lab build only)
#endif
    a.cpw = a.nchunk / p.waves;
    const unsigned grid = (unsigned)(p.tiles * p.groups);
    a.xcd_remap = (grid % 8 == 0 && grid >= 64) ? 1 : 0;
#ifdef CDC_WS_LAB
    static unsigned long long *tl = nullptr;
    static int tl_n = 0;
    if (getenv("CDC_WS_TL")) {
        if (!tl) (void)hipMalloc((void **)&tl, (size_t)8192 * 16 * 8);
        (void)hipMemsetAsync(tl, 0, (size_t)8192 * 16 * 8, st);
        a.tl = grid <= 8192 ? tl : nullptr;
    } else a.tl = nullptr;
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * p.waves), p.lds_bytes, st, a);
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
        fprintf(stderr, "[ws-tl] Cin=%d Cout=%d H=%d W=%d waves=%d npb=%d grid=%u | span %llu | mean since start:", a.Cin, a.Cout, a.H, p.W, p.waves, p.NPB, grid, last - first);
        for (int i = 1; i < 16; ++i) fprintf(stderr, " %d:%.0f", i, d[i] / grid);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}


// ---- conv_ws1_kernel (conv_ws1_kernel.h): the 1x1 layers of the few-pixel levels --------------------------------------------------
bool ws1_make_plan(int Cin, int C0, int Cout, int HW, int B, bool per_image_w, Ws1Plan *p) {
    if ((Cin % 16) || (C0 % 16) || (Cout % 32) || Cin < 32 || (HW % 32) || HW < 32) return false;
    const int groups = Cout / 32, nchunk = Cin / 16;
    const long long blocks = (long long)B * (HW / 32);
    const long long min_wgs = dev_env("CDC_WS1_MIN_WGS") ? atoll(dev_env("CDC_WS1_MIN_WGS")) : 4;
    const int force_npb = dev_env("CDC_WS1_NPB") ? atoi(dev_env("CDC_WS1_NPB")) : 0;
    for (int npb : {4, 2}) {
        if (force_npb && npb != force_npb) continue;
        if (npb == 4 && per_image_w) continue;             // (a weight set per pixel block: four of them do not fit the registers)
        if (blocks % npb) continue;
        const long long tiles = blocks / npb, wgs = tiles * groups;
        if (npb == 4 && !force_npb && wgs < (3 * device_cus()) / 4) continue;     // 128-pixel tiles only where they fill the chip
        if (wgs < min_wgs || tiles > (1 << 20)) return false;
        int waves = 0;
        for (int w = 8; w >= 1; --w)
            if (nchunk % w == 0) { waves = w; break; }
        p->NPB = npb; p->waves = waves; p->tiles = (int)tiles; p->groups = groups; p->lds_bytes = (size_t)waves * npb * 4096;
        return true;
    }
    return false;
}

hipError_t ws1_launch(Ws1Args a, const Ws1Plan &p, hipStream_t st) {
    const bool pi = a.w_bs != 0;
    ws1_kernel_fn fn = p.NPB == 4 ? (pi ? conv_ws1_kernel<4, true> : conv_ws1_kernel<4, false>) : (pi ? conv_ws1_kernel<2, true> : conv_ws1_kernel<2, false>);
    static bool attr_done[16][2][2] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev][p.NPB == 4][pi]) {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) attr_done[dev][p.NPB == 4][pi] = true;
    }
    a.tiles = p.tiles;
    a.cpw = a.nchunk / p.waves;
    const unsigned grid = (unsigned)(p.tiles * p.groups);
    a.xcd_remap = (grid % 8 == 0 && grid >= 64) ? 1 : 0;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * p.waves), p.lds_bytes, st, a);
    return hipGetLastError();
}

}  // namespace cdc
