#include "pe25d_kernels.h"
namespace gcm {
struct Pe25d {};
Pe25d *pe25d_create(const gcm_config &, hipStream_t, std::string *err) { *err = "GCM_PE25D not built yet"; return nullptr; }
void pe25d_destroy(Pe25d *) {}
int pe25d_set(Pe25d *, bool, const double *, const double *, const double *, const double *, const double *, std::string *) { return GCM_ERR_UNSUPPORTED; }
int pe25d_get(Pe25d *, bool, double *, double *, double *, double *, double *, std::string *) { return GCM_ERR_UNSUPPORTED; }
int pe25d_step(Pe25d *, double, hipStream_t, std::string *) { return GCM_ERR_UNSUPPORTED; }
int pe25d_step_part(Pe25d *, int, double, hipStream_t, std::string *) { return GCM_ERR_UNSUPPORTED; }
int pe25d_half(Pe25d *, int, double, hipStream_t, std::string *) { return GCM_ERR_UNSUPPORTED; }
size_t pe25d_halo_bytes(const Pe25d *) { return 0; }
int pe25d_halo(Pe25d *, bool, int, void *, hipStream_t, std::string *) { return GCM_ERR_UNSUPPORTED; }
const double *pe25d_field(Pe25d *, int, long *n) { *n = 0; return nullptr; }
void pe25d_timing(Pe25d *, std::vector<hipEvent_t> *, size_t *) {}
}
