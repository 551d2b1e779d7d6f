// conv_pf3_inst.h -- explicit instantiations of conv_pf3_kernel for one group shape (included by conv_inst_q.hip /
// conv_inst_r.hip: one translation unit per shape, so the two compile in parallel).
#pragma once
#include "cdc_internal.h"
#include "conv_pf3_kernel.h"

namespace cdc {

template <int MB, int NPW, int WM, int WP> static pf_kernel_fn pf3_lookup_shape(int epv) {
    switch (epv) {                                    // kPf3Resid | kPf3F32 | kPf3Pf
    case 2: return conv_pf3_kernel<MB, NPW, WM, WP, 2, true>;
    case 3: return conv_pf3_kernel<MB, NPW, WM, WP, 3, true>;
#ifdef CDC_PF3_NOSYNC4
    case 4: return conv_pf3_kernel<MB, NPW, WM, WP, 4, false>;      // (experiment: the epilogue beside the partner group's main loop)
#else
    case 4: return conv_pf3_kernel<MB, NPW, WM, WP, 4, true>;
#endif
    case 5: return conv_pf3_kernel<MB, NPW, WM, WP, 5, true>;
    case 6: return conv_pf3_kernel<MB, NPW, WM, WP, 6, true>;
    case 7: return conv_pf3_kernel<MB, NPW, WM, WP, 7, true>;
    case 11: return conv_pf3_kernel<MB, NPW, WM, WP, 11, true>;      // + LayerNorm statistics of the result
    case 15: return conv_pf3_kernel<MB, NPW, WM, WP, 15, true>;
    case 67: return conv_pf3_kernel<MB, NPW, WM, WP, 67, true>;      // kPf3ResPf: the residual from a PF tensor (3, 7, 11, 15 + 64)
    case 71: return conv_pf3_kernel<MB, NPW, WM, WP, 71, true>;
    case 75: return conv_pf3_kernel<MB, NPW, WM, WP, 75, true>;
    case 79: return conv_pf3_kernel<MB, NPW, WM, WP, 79, true>;
    case 37:                                                          // kPf3Pre: hoisted partial sums before the LayerNorm, planes out (128-channel shape)
        if constexpr (WM == 2) return conv_pf3_kernel<MB, NPW, WM, WP, 37, true>;
        else return nullptr;
    case 23:                                                          // + the 3-channel res_conv of the first ResnetBlock (64-channel shape)
        if constexpr (WM == 1) return conv_pf3_kernel<MB, NPW, WM, WP, 23, true>;
        else return nullptr;
    case 21:                                                          // ... its output as planes only
        if constexpr (WM == 1) return conv_pf3_kernel<MB, NPW, WM, WP, 21, true>;
        else return nullptr;
    }
    return nullptr;
}

}  // namespace cdc
