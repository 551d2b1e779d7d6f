// generated instantiation unit of the conv kernel (see conv_kernel.h)
#include "cdc_internal.h"
#include "conv_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_b(int MB, int NPW, bool lnload) {
    if (MB == 4 && NPW == 1 && !lnload) return conv_mfma_kernel<4, 1, false>;
    if (MB == 4 && NPW == 1 && lnload) return conv_mfma_kernel<4, 1, true>;
    if (MB == 4 && NPW == 2 && !lnload) return conv_mfma_kernel<4, 2, false>;
    if (MB == 4 && NPW == 2 && lnload) return conv_mfma_kernel<4, 2, true>;
    if (MB == 5 && NPW == 1 && !lnload) return conv_mfma_kernel<5, 1, false>;
    if (MB == 5 && NPW == 1 && lnload) return conv_mfma_kernel<5, 1, true>;
    if (MB == 5 && NPW == 2 && !lnload) return conv_mfma_kernel<5, 2, false>;
    if (MB == 5 && NPW == 2 && lnload) return conv_mfma_kernel<5, 2, true>;
    if (MB == 6 && NPW == 1 && !lnload) return conv_mfma_kernel<6, 1, false>;
    if (MB == 6 && NPW == 1 && lnload) return conv_mfma_kernel<6, 1, true>;
    if (MB == 6 && NPW == 2 && !lnload) return conv_mfma_kernel<6, 2, false>;
    if (MB == 6 && NPW == 2 && lnload) return conv_mfma_kernel<6, 2, true>;
    return nullptr;
}
}  // namespace cdc
