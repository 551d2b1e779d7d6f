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
    case 4: return conv_pf3_kernel<MB, NPW, WM, WP, 4, true>;
    case 5: return conv_pf3_kernel<MB, NPW, WM, WP, 5, true>;
    case 6: return conv_pf3_kernel<MB, NPW, WM, WP, 6, true>;
    case 7: return conv_pf3_kernel<MB, NPW, WM, WP, 7, true>;
    case 11: return conv_pf3_kernel<MB, NPW, WM, WP, 11, true>;      // + LayerNorm statistics of the result
    case 15: return conv_pf3_kernel<MB, NPW, WM, WP, 15, true>;
    case 37:                                                          // kPf3Pre: hoisted partial sums before the LayerNorm, planes out (128-channel shape)
        if constexpr (WM == 2) return conv_pf3_kernel<MB, NPW, WM, WP, 37, true>;
        else return nullptr;
    case 23:                                                          // + the 3-channel res_conv of the first ResnetBlock (64-channel shape)
        if constexpr (WM == 1) return conv_pf3_kernel<MB, NPW, WM, WP, 23, true>;
        else return nullptr;
    }
    return nullptr;
}

}  // namespace cdc
