// conv_inst_q.hip -- conv_pf3_kernel (persistent ping-ponged 3x3 convolution on PF operands), 64-channel group shape,
// plus the planner and launcher of the kernel family.
#include <stdlib.h>

#include "conv_pf3_inst.h"

namespace cdc {

pf_kernel_fn pf3_lookup_c128(int epv);               // conv_inst_r.hip

static pf_kernel_fn pf3_lookup(int COPT, int epv) {
    if (COPT == 64) return pf3_lookup_shape<2, 2, 1, 4>(epv);
    if (COPT == 128) return pf3_lookup_c128(epv);
    return nullptr;
}

int device_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Takes a fully described stride-1 3x3 layer (the PfArgs of conv_pf_kernel) and decides whether conv_pf3_kernel runs it.
bool pf3_make_plan(const PfArgs &a, int B, int nz, PfPlan *p) {
    p->pf3_epv = 0;
    if (a.KH != 3 || a.KW != 3 || nz != 1) return false;
    if (a.pad_y[0] != 1 || a.pad_x[0] != 1) return false;
    const int COPT = a.Cout;
    if ((COPT != 64 && COPT != 128) || a.COP < COPT) return false;
    if ((a.Cin % 16) || (a.C0 % 16) || a.nchunk != a.Cin / 16 || a.nchunk < 1) return false;
    const int TH = COPT == 64 ? 8 : 4;
    if ((a.Ho % TH) || (a.Wo % 32) || a.Ho != a.H || a.Wo != a.W) return false;
    if ((a.stat_mean != nullptr) != (a.stat_rstd != nullptr)) return false;
    // hoisted partial sums ride on the residual loads (kPf3Pre): not together with a residual
    if (a.pre_add && (a.resid || a.resid_pf || a.res3_w || a.stat_mean || !a.ep_g)) return false;
    if (a.resid_pf && (a.resid || a.res3_w || a.rpf_ps * 16 >= (1ll << 31))) return false;
    if (a.res3_w && (!a.res3_x || a.Ho * (long long)a.Wo * 3 >= (1ll << 30))) return false;
    if (a.out_xs != 1 || a.out_ys != a.Wo || a.out_zoff[0] != 0) return false;
    if (a.out_pf && (a.pf_xs != 1)) return false;
    const int epv = (a.resid ? kPf3Resid : 0) | (a.resid_pf ? (kPf3Resid | kPf3ResPf) : 0) | (a.out ? kPf3F32 : 0) | (a.out_pf ? kPf3Pf : 0) | (a.stat_mean ? kPf3Stat : 0) |
                    (a.res3_w ? kPf3Res3 : 0) | (a.pre_add ? (kPf3Pre | kPf3Resid) : 0);
    if (!pf3_lookup(COPT, epv)) return false;
    if (COPT == 64 ? pf3_lds_used(2, 2, 1, 4, B) > pf3_lds_bytes() : pf3_lds_used(2, 2, 2, 2, B) > pf3_lds_bytes()) return false;
    // lane offsets of the row-layout accesses are 32-bit: 7 channel strides + a row
    if ((long long)8 * a.out_cs * 4 >= (1ll << 31) || (a.resid && (long long)8 * a.resid_cs * 4 >= (1ll << 31)) || (a.out_pf && a.pf_ps * 16 >= (1ll << 31))) return false;
    const int ntiles = (a.Wo / 32) * (a.Ho / TH) * B;
    int G = 0;                                            // most workgroups with >= 2 tiles per group (one gains nothing from persistence)
    for (int g = device_cus() & ~7; g >= device_cus() / 2; g -= 8)
        if (ntiles % (2 * g) == 0 && ntiles / (2 * g) >= 2) { G = g; break; }
    if (!G) return false;
    p->pf3_epv = epv; p->pf3_G = G; p->pf3_iters = ntiles / (2 * G);
    return true;
}

hipError_t pf3_launch(PfArgs a, const PfPlan &p, int B, hipStream_t st) {
    const int COPT = a.Cout, TH = COPT == 64 ? 8 : 4;
    pf_kernel_fn fn = pf3_lookup(COPT, p.pf3_epv);
    if (!fn) return hipErrorInvalidValue;
    a.lognbw = 5; a.dbg = 0; a.B = B;
    // tile walk order inside a workgroup's range (conv_pf3_kernel reads `lin`): column-major -- the two groups' concurrent tiles are
    // vertical neighbours and their shared halo rows meet in L2 (measured, whole model: 12.96 -> 12.88 ms per iteration, the 64-channel
    // layers at 256^2 -2 %); CDC_PF3_XMAJOR=1 restores the row-major walk
    a.lin = dev_env("CDC_PF3_XMAJOR") ? 0 : 1;
    a.tiles_x = a.Wo / 32; a.tiles_y = a.Ho / TH;
    a.n_iter = p.pf3_iters; a.xcd_remap = 1;
    if (p.pf3_epv & kPf3Pre) {                            // the partial sums take the residual operand's place (same layout as `out`)
        a.resid = a.pre_add; a.resid_bs = a.out_bs; a.resid_cs = a.out_cs;
        a.pre_add = nullptr;
    }
    static bool attr_done[16][2][128];                     // per device: a function attribute belongs to the device's copy of the code
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev][COPT == 128][p.pf3_epv]) {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf3_lds_bytes());
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) attr_done[dev][COPT == 128][p.pf3_epv] = true;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)p.pf3_G, 1, 1), dim3(512), pf3_lds_bytes(), st, a);
    return hipGetLastError();
}

}  // namespace cdc
