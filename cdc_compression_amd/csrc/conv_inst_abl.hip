// tuning-aid instantiations of the conv kernel with compile-time ablations (CDC_ABLATE=n)
#include "cdc_internal.h"
#include "conv_split_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_split_abl(int MB, int NPW, int abl) {
    if (MB == 2 && NPW == 4) {
        switch (abl) {
            case 1: return conv_split_kernel<2, 4, 1>;
            case 2: return conv_split_kernel<2, 4, 2>;
            case 4: return conv_split_kernel<2, 4, 4>;
            case 8: return conv_split_kernel<2, 4, 8>;
            case 15: return conv_split_kernel<2, 4, 15>;
            case 16: return conv_split_kernel<2, 4, 16>;
            case 31: return conv_split_kernel<2, 4, 31>;
        }
    }
    return nullptr;
}
conv_kernel_fn conv_lookup_abl(int MB, int NPW, int abl) {
    if (MB == 2 && NPW == 4) {
        switch (abl) {
            case 1: return conv_mfma_kernel<2, 4, 0, 1>;
            case 2: return conv_mfma_kernel<2, 4, 0, 2>;
            case 4: return conv_mfma_kernel<2, 4, 0, 4>;
            case 7: return conv_mfma_kernel<2, 4, 0, 7>;
            case 8: return conv_mfma_kernel<2, 4, 0, 8>;
            case 15: return conv_mfma_kernel<2, 4, 0, 15>;
        }
    }
    return nullptr;
}
}  // namespace cdc
