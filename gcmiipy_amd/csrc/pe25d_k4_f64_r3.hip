// GCM_PE25D, K4 kernels in double, 3-row workgroups (pe25d_k4.h): one translation unit per real type and group
// height, so that the instantiations compile in parallel.
#include "pe25d_k4.h"

namespace gcm {
template FilterKernel<double> update_rows_kernel_rt<double, 3>(bool, bool);
}  // namespace gcm
