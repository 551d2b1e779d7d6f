// instantiation unit of the split conv kernel, two-plane fp16 arithmetic with unfold on load (AR = 1, UF = 1; see conv_split_kernel.h):
// the first 7x7 layer as a 7x1 convolution over the kx-unfolded 3-channel image, where conv_pf_kernel's UF form does not take it
#include "cdc_internal.h"
#include "conv_split_kernel.h"
namespace cdc {
conv_kernel_fn conv_lookup_split2hu(int MB, int NPW) {
    if (MB == 1 && NPW == 1) return conv_split2_kernel<1, 1, 0, 1, 1, 0, 1>;
    if (MB == 1 && NPW == 2) return conv_split2_kernel<1, 2, 0, 1, 1, 0, 1>;
    if (MB == 2 && NPW == 1) return conv_split2_kernel<2, 1, 0, 1, 1, 0, 1>;
    if (MB == 2 && NPW == 2) return conv_split2_kernel<2, 2, 0, 1, 1, 0, 1>;
    return nullptr;
}
}  // namespace cdc
