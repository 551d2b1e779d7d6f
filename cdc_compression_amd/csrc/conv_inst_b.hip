// generated instantiation unit of the conv kernel (see conv_kernel.h)
#include "cdc_internal.h"
#include "conv_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_b(int MB, int NPW, int lnmode) {
    if (MB == 4 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<4, 1, 0>;
    if (MB == 4 && NPW == 1 && lnmode == 2) return conv_mfma_kernel<4, 1, 2>;
    if (MB == 4 && NPW == 2 && lnmode == 0) return conv_mfma_kernel<4, 2, 0>;
    if (MB == 4 && NPW == 2 && lnmode == 2) return conv_mfma_kernel<4, 2, 2>;
    if (MB == 5 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<5, 1, 0>;
    if (MB == 5 && NPW == 1 && lnmode == 2) return conv_mfma_kernel<5, 1, 2>;
    if (MB == 5 && NPW == 2 && lnmode == 0) return conv_mfma_kernel<5, 2, 0>;
    if (MB == 5 && NPW == 2 && lnmode == 2) return conv_mfma_kernel<5, 2, 2>;
    if (MB == 6 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<6, 1, 0>;
    if (MB == 6 && NPW == 1 && lnmode == 2) return conv_mfma_kernel<6, 1, 2>;
    if (MB == 6 && NPW == 2 && lnmode == 0) return conv_mfma_kernel<6, 2, 0>;
    if (MB == 6 && NPW == 2 && lnmode == 2) return conv_mfma_kernel<6, 2, 2>;
    return nullptr;
}
}  // namespace cdc
