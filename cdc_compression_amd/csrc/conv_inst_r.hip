// conv_inst_r.hip -- conv_pf3_kernel, 128-channel group shape (two channel halves x two row pairs per group).
#include "conv_pf3_inst.h"

namespace cdc {

pf_kernel_fn pf3_lookup_c128(int epv) { return pf3_lookup_shape<2, 2, 2, 2>(epv); }

}  // namespace cdc
