// generated instantiation unit of the conv kernel (see conv_kernel.h)
#include "cdc_internal.h"
#include "conv_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_a(int MB, int NPW, bool lnload) {
    if (MB == 1 && NPW == 1 && !lnload) return conv_mfma_kernel<1, 1, false>;
    if (MB == 1 && NPW == 1 && lnload) return conv_mfma_kernel<1, 1, true>;
    if (MB == 1 && NPW == 2 && !lnload) return conv_mfma_kernel<1, 2, false>;
    if (MB == 1 && NPW == 2 && lnload) return conv_mfma_kernel<1, 2, true>;
    if (MB == 1 && NPW == 4 && !lnload) return conv_mfma_kernel<1, 4, false>;
    if (MB == 1 && NPW == 4 && lnload) return conv_mfma_kernel<1, 4, true>;
    if (MB == 2 && NPW == 1 && !lnload) return conv_mfma_kernel<2, 1, false>;
    if (MB == 2 && NPW == 1 && lnload) return conv_mfma_kernel<2, 1, true>;
    if (MB == 2 && NPW == 2 && !lnload) return conv_mfma_kernel<2, 2, false>;
    if (MB == 2 && NPW == 2 && lnload) return conv_mfma_kernel<2, 2, true>;
    if (MB == 2 && NPW == 4 && !lnload) return conv_mfma_kernel<2, 4, false>;
    if (MB == 2 && NPW == 4 && lnload) return conv_mfma_kernel<2, 4, true>;
    if (MB == 3 && NPW == 1 && !lnload) return conv_mfma_kernel<3, 1, false>;
    if (MB == 3 && NPW == 1 && lnload) return conv_mfma_kernel<3, 1, true>;
    if (MB == 3 && NPW == 2 && !lnload) return conv_mfma_kernel<3, 2, false>;
    if (MB == 3 && NPW == 2 && lnload) return conv_mfma_kernel<3, 2, true>;
    if (MB == 3 && NPW == 4 && !lnload) return conv_mfma_kernel<3, 4, false>;
    if (MB == 3 && NPW == 4 && lnload) return conv_mfma_kernel<3, 4, true>;
    return nullptr;
}
}  // namespace cdc
