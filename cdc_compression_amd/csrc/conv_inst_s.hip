// generated instantiation unit of the split-bf16 conv kernel (see conv_split_kernel.h)
#include "cdc_internal.h"
#include "conv_split_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_split(int MB, int NPW) {
    if (MB == 1 && NPW == 1) return conv_split_kernel<1, 1>;
    if (MB == 1 && NPW == 2) return conv_split_kernel<1, 2>;
    if (MB == 1 && NPW == 4) return conv_split_kernel<1, 4>;
    if (MB == 2 && NPW == 1) return conv_split_kernel<2, 1>;
    if (MB == 2 && NPW == 2) return conv_split_kernel<2, 2>;
    if (MB == 2 && NPW == 4) return conv_split_kernel<2, 4>;
    if (MB == 3 && NPW == 1) return conv_split_kernel<3, 1>;
    if (MB == 3 && NPW == 2) return conv_split_kernel<3, 2>;
    if (MB == 4 && NPW == 1) return conv_split_kernel<4, 1>;
    if (MB == 4 && NPW == 2) return conv_split_kernel<4, 2>;
    if (MB == 5 && NPW == 1) return conv_split_kernel<5, 1>;
    if (MB == 6 && NPW == 1) return conv_split_kernel<6, 1>;
    return nullptr;
}
conv_kernel_fn conv_lookup_split2(int MB, int NPW, int lnmode, int xu) {
    if (xu == 2) {
        if (lnmode != 0) return nullptr;
        if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1, 0, 2>;
        if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1, 0, 2>;
        if (MB == 3 && NPW == 1) return conv_split2_kernel<3, 1, 0, 2>;
        return nullptr;
    }
    if (lnmode == 0) {
        if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1>;
        if (MB == 1 && NPW == 2) return conv_split2_kernel<1, 2>;
        if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1>;
        if (MB == 2 && NPW == 2) return conv_split2_kernel<2, 2>;
        if (MB == 3 && NPW == 1) return conv_split2_kernel<3, 1>;
        if (MB == 4 && NPW == 1) return conv_split2_kernel<4, 1>;
        if (MB == 3 && NPW == 2) return conv_split2_kernel<3, 2>;
        if (MB == 5 && NPW == 1) return conv_split2_kernel<5, 1>;
        if (MB == 6 && NPW == 1) return conv_split2_kernel<6, 1>;
        if (MB == 8 && NPW == 1) return conv_split2_kernel<8, 1>;
    } else if (lnmode == 1) {
        if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1, 1>;
        if (MB == 1 && NPW == 2) return conv_split2_kernel<1, 2, 1>;
        if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1, 1>;
        if (MB == 2 && NPW == 2) return conv_split2_kernel<2, 2, 1>;
    } else if (lnmode == 2) {
        if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1, 2>;
        if (MB == 1 && NPW == 2) return conv_split2_kernel<1, 2, 2>;
        if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1, 2>;
        if (MB == 2 && NPW == 2) return conv_split2_kernel<2, 2, 2>;
        if (MB == 3 && NPW == 1) return conv_split2_kernel<3, 1, 2>;
        if (MB == 4 && NPW == 1) return conv_split2_kernel<4, 1, 2>;
        if (MB == 3 && NPW == 2) return conv_split2_kernel<3, 2, 2>;
        if (MB == 6 && NPW == 1) return conv_split2_kernel<6, 1, 2>;
    }
    return nullptr;
}
}  // namespace cdc
