// GCM_PE25D, K4 kernels in double (pe25d_k4.h): one translation unit per real type, so that the
// many instantiations compile in parallel.
#include "pe25d_k4.h"

namespace gcm {
template FilterKernel<double> update_rows_kernel_for<double>(int, bool);
}  // namespace gcm
