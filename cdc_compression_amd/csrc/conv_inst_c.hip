// generated instantiation unit of the conv kernel (see conv_kernel.h)
#include "cdc_internal.h"
#include "conv_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_c(int MB, int NPW, int lnmode) {
    if (MB == 7 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<7, 1, 0>;
    if (MB == 8 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<8, 1, 0>;
    if (MB == 9 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<9, 1, 0>;
    if (MB == 10 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<10, 1, 0>;
    if (MB == 11 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<11, 1, 0>;
    if (MB == 12 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<12, 1, 0>;
    return nullptr;
}
}  // namespace cdc
