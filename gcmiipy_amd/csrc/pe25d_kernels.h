// 2.5-D sigma-level primitive equations (GCM_PE25D): host-visible interface of
// pe25d_kernels.hip, used by gcmcore.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/gcmcore.h"

namespace gcm {

struct Pe25d;
struct SegCopy;   // sw2d_kernels.h

Pe25d *pe25d_create(const gcm_config &cfg, hipStream_t s, std::string *err);
void pe25d_destroy(Pe25d *m);
int pe25d_set(Pe25d *m, bool star, const double *p, const double *u, const double *v,
              const double *t, const double *q, hipStream_t s, std::string *err);
int pe25d_get(Pe25d *m, bool star, double *p, double *u, double *v, double *t, double *q,
              hipStream_t s, std::string *err);
int pe25d_step(Pe25d *m, double dt, hipStream_t s, std::string *err);
int pe25d_step_part(Pe25d *m, int part, double dt, hipStream_t s, std::string *err);
// chained: the call belongs to gcm_band_run's own sequence (the library knows everything queued between two stages)
int pe25d_step_phase(Pe25d *m, int phase, double dt, hipStream_t s, std::string *err, bool chained = false);
int pe25d_set_halo_buffers(Pe25d *m, void *north, void *south, hipStream_t s, std::string *err);
int pe25d_wait_edges(Pe25d *m, hipStream_t s, std::string *err);
int pe25d_prep_ghost_rows(Pe25d *m, std::string *err);   // gcm_band_run: behind the unpack on the second stream
hipStream_t pe25d_aux_stream(const Pe25d *m);
void pe25d_join_third_stream(Pe25d *m, hipStream_t s);
void pe25d_fork_invalidate(Pe25d *m);
void pe25d_set_edges_first(Pe25d *m, bool on);   // gcm_set_band_overlap on a GCM_PE25D band
// a new non-blocking stream that demonstrably runs beside `main` (and `other`, may be null)
hipStream_t concurrent_stream(hipStream_t main, hipStream_t other);
void launch_spin(hipStream_t s, double us);   // GCM_BAND_EXCHANGE_DELAY_US: the loopback exchange takes that long
int pe25d_half(Pe25d *m, int stage, double dt, hipStream_t s, std::string *err);
size_t pe25d_halo_bytes(const Pe25d *m);
int pe25d_halo_segments(Pe25d *m, bool pack, int side, void *dev_buf, SegCopy *c, std::string *err);
int pe25d_ground(Pe25d *m, bool set, const double *in, double *out, hipStream_t s, std::string *err);
int pe25d_intermediate(Pe25d *m, int kind, double *out, hipStream_t s, std::string *err);
int pe25d_filter_field(Pe25d *m, int nlev, const double *in, double *out, hipStream_t s, std::string *err);
int pe25d_radiation(Pe25d *m, bool apply, double dt, double utc, double t_lw, double t_sw, double albedo,
                    const double *lat, const double *lon, double *dTdt_host, double *dtg_host,
                    hipStream_t s, std::string *err);
int pe25d_physics_tables(Pe25d *m, double t_lw, double t_sw, const double *lat, const double *lon, hipStream_t s,
                         std::string *err);
int pe25d_solar_rows(Pe25d *m, int set, int j0, int j1, int jb0, int jb1, bool keep_ghosts, double dt, double utc, double albedo,
                     hipStream_t s, std::string *err);
int pe25d_new_state_set(const Pe25d *m);    // the set a corrector stage in flight writes (before the swap), else the current one
int pe25d_stats(Pe25d *m, const double *area_host, int area_len, double out[9], hipStream_t s, std::string *err);
int pe25d_filter_plan(int n, unsigned *out, int cap);   // gcm_filter_plan
void pe25d_tv_shape(const Pe25d *m, int field, long *n_outer, long *n_axis, long *n_inner, int *wrap);
const void *pe25d_field(Pe25d *m, int field, long *n, int *f32);
void pe25d_timing(Pe25d *m, std::vector<hipEvent_t> *ev, size_t *used);

}  // namespace gcm
