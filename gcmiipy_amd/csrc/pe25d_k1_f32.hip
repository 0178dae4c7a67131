// GCM_PE25D, K1 kernels in float (pe25d_k1.h): one translation unit per real type, so that the
// many instantiations compile in parallel.
#include "pe25d_k1.h"

namespace gcm {
template FilterKernel<float> spu_filter_kernel_for<float>(const SuperPlan &);
template FilterLoopKernel<float> spu_filter_loop_kernel_for<float>(const SuperPlan &);
}  // namespace gcm
