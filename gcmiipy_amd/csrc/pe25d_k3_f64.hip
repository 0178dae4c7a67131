// GCM_PE25D, K3 kernels in double (pe25d_k3.h): one translation unit per real type, so that the
// many instantiations compile in parallel.
#include "pe25d_k3.h"

namespace gcm {
template FilterKernel<double> pgf_filter_kernel_for<double>(const SuperPlan &);
template FilterKernel<double> pit2d_kernel_for<double>(const SuperPlan &);
}  // namespace gcm
