// generated instantiation unit of the conv kernel (see conv_kernel.h)
#include "cdc_internal.h"
#include "conv_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_a(int MB, int NPW, int lnmode) {
    if (MB == 1 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<1, 1, 0>;
    if (MB == 1 && NPW == 1 && lnmode == 1) return conv_mfma_kernel<1, 1, 1>;
    if (MB == 1 && NPW == 1 && lnmode == 2) return conv_mfma_kernel<1, 1, 2>;
    if (MB == 1 && NPW == 2 && lnmode == 0) return conv_mfma_kernel<1, 2, 0>;
    if (MB == 1 && NPW == 2 && lnmode == 1) return conv_mfma_kernel<1, 2, 1>;
    if (MB == 1 && NPW == 2 && lnmode == 2) return conv_mfma_kernel<1, 2, 2>;
    if (MB == 1 && NPW == 4 && lnmode == 0) return conv_mfma_kernel<1, 4, 0>;
    if (MB == 1 && NPW == 4 && lnmode == 1) return conv_mfma_kernel<1, 4, 1>;
    if (MB == 1 && NPW == 4 && lnmode == 2) return conv_mfma_kernel<1, 4, 2>;
    if (MB == 2 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<2, 1, 0>;
    if (MB == 2 && NPW == 1 && lnmode == 1) return conv_mfma_kernel<2, 1, 1>;
    if (MB == 2 && NPW == 1 && lnmode == 2) return conv_mfma_kernel<2, 1, 2>;
    if (MB == 2 && NPW == 2 && lnmode == 0) return conv_mfma_kernel<2, 2, 0>;
    if (MB == 2 && NPW == 2 && lnmode == 1) return conv_mfma_kernel<2, 2, 1>;
    if (MB == 2 && NPW == 2 && lnmode == 2) return conv_mfma_kernel<2, 2, 2>;
    if (MB == 2 && NPW == 4 && lnmode == 0) return conv_mfma_kernel<2, 4, 0>;
    if (MB == 2 && NPW == 4 && lnmode == 1) return conv_mfma_kernel<2, 4, 1>;
    if (MB == 2 && NPW == 4 && lnmode == 2) return conv_mfma_kernel<2, 4, 2>;
    if (MB == 3 && NPW == 1 && lnmode == 0) return conv_mfma_kernel<3, 1, 0>;
    if (MB == 3 && NPW == 1 && lnmode == 2) return conv_mfma_kernel<3, 1, 2>;
    if (MB == 3 && NPW == 2 && lnmode == 0) return conv_mfma_kernel<3, 2, 0>;
    if (MB == 3 && NPW == 2 && lnmode == 2) return conv_mfma_kernel<3, 2, 2>;
    if (MB == 3 && NPW == 4 && lnmode == 0) return conv_mfma_kernel<3, 4, 0>;
    if (MB == 3 && NPW == 4 && lnmode == 2) return conv_mfma_kernel<3, 4, 2>;
    return nullptr;
}
}  // namespace cdc
