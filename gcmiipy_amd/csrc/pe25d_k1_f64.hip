// GCM_PE25D, K1 kernels in double (pe25d_k1.h): one translation unit per real type, so that the
// many instantiations compile in parallel.
#include "pe25d_k1.h"

namespace gcm {
template FilterKernel<double> spu_filter_kernel_for<double>(const SuperPlan &);
template FilterLoopKernel<double> spu_filter_loop_kernel_for<double>(const SuperPlan &);
}  // namespace gcm
