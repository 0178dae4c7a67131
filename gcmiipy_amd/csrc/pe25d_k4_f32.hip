// GCM_PE25D, K4 kernels in float (pe25d_k4.h): one translation unit per real type, so that the
// many instantiations compile in parallel.
#include "pe25d_k4.h"

namespace gcm {
template FilterKernel<float> update_rows_kernel_for<float>(int, bool);
}  // namespace gcm
