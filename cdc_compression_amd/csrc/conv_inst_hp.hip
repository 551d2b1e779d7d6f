// instantiation unit of the split conv kernel, two-plane fp16 arithmetic with the software-pipelined tap loop (AR = 1, PIPE = 1;
// see conv_split_kernel.h)
#include "cdc_internal.h"
#include "conv_split_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_split2hp(int MB, int NPW, int lnmode, int xu) {
    if (xu == 2) {
        if (lnmode != 0) return nullptr;
        if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1, 0, 2, 1, 1>;
        if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1, 0, 2, 1, 1>;
        if (MB == 3 && NPW == 1) return conv_split2_kernel<3, 1, 0, 2, 1, 1>;
        return nullptr;
    }
    if (lnmode == 0) {
        if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1, 0, 1, 1, 1>;
        if (MB == 1 && NPW == 2) return conv_split2_kernel<1, 2, 0, 1, 1, 1>;
        if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1, 0, 1, 1, 1>;
        if (MB == 2 && NPW == 2) return conv_split2_kernel<2, 2, 0, 1, 1, 1>;
        if (MB == 3 && NPW == 1) return conv_split2_kernel<3, 1, 0, 1, 1, 1>;
        if (MB == 4 && NPW == 1) return conv_split2_kernel<4, 1, 0, 1, 1, 1>;
        if (MB == 3 && NPW == 2) return conv_split2_kernel<3, 2, 0, 1, 1, 1>;
        if (MB == 5 && NPW == 1) return conv_split2_kernel<5, 1, 0, 1, 1, 1>;
        if (MB == 6 && NPW == 1) return conv_split2_kernel<6, 1, 0, 1, 1, 1>;
    } else if (lnmode == 1) {
        if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1, 1, 1, 1, 1>;
        if (MB == 1 && NPW == 2) return conv_split2_kernel<1, 2, 1, 1, 1, 1>;
        if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1, 1, 1, 1, 1>;
        if (MB == 2 && NPW == 2) return conv_split2_kernel<2, 2, 1, 1, 1, 1>;
    }
    return nullptr;          // (lnmode 2 is 1x1 only: one tap per chunk, nothing to pipeline)
}
// unfold on load (UF = 1): the first 7x7 layer as a 7x1 convolution over the kx-unfolded image
conv_kernel_fn conv_lookup_split2hu(int MB, int NPW) {
    if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1, 0, 1, 1, 0, 1>;
    if (MB == 1 && NPW == 2) return conv_split2_kernel<1, 2, 0, 1, 1, 0, 1>;
    if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1, 0, 1, 1, 0, 1>;
    if (MB == 2 && NPW == 2) return conv_split2_kernel<2, 2, 0, 1, 1, 0, 1>;
    return nullptr;
}
}  // namespace cdc
