// GCM_PE25D, K3 kernels in float (pe25d_k3.h): one translation unit per real type, so that the
// many instantiations compile in parallel.
#include "pe25d_k3.h"

namespace gcm {
template FilterKernel<float> pgf_filter_kernel_for<float>(const SuperPlan &);
template FilterKernel<float> pit2d_kernel_for<float>(const SuperPlan &);
}  // namespace gcm
