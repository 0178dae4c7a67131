/* gcmcore.h -- C ABI of libgcmcore.so, the MI355X-native Matsuno C-grid
 * dynamical core for gcmiipy.
 *
 * The reference (marthinwurer/gcmiipy) is pure Python/NumPy and has no FFI of
 * its own; the entry points below are what a ctypes binding for its time-step
 * functions needs (INTEGRATION.md shows that binding).  Each one names the
 * reference interface it stands behind.  Plain pointers and sizes only; no
 * torch / numpy types.  Every function returns 0 on success or a negative
 * gcm_status; nothing throws across the ABI; the message of the last failure
 * is available from gcm_last_error().
 *
 * Threading: a handle is driven by one host thread at a time.  Calls that
 * launch work (gcm_step, gcm_half_step, gcm_halo_*) are asynchronous on the
 * handle's stream; gcm_get_state / gcm_diag / gcm_sync synchronise.
 *
 * Memory: all arrays are float64, C-contiguous, reference layout:
 *   2-D fields [j][i]   (H rows x W columns, i fastest)
 *   3-D fields [k][j][i] (L levels, k = 0 is the bottom layer)
 * Host buffers stay caller-owned and are only touched inside set/get calls.
 */
#ifndef GCMCORE_H
#define GCMCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCM_ABI_VERSION 1

typedef struct gcm_handle gcm_handle;

typedef enum {
    GCM_OK = 0,
    GCM_ERR_ARG = -1,      /* bad argument / shape (mirrors the reference's shape asserts) */
    GCM_ERR_HIP = -2,      /* a HIP runtime call failed                                     */
    GCM_ERR_NODEVICE = -3, /* no gfx950 device visible: there is NO CPU fallback            */
    GCM_ERR_STATE = -4,    /* call sequence error (e.g. half_step corrector before predictor) */
    GCM_ERR_UNSUPPORTED = -5
} gcm_status;

/* Which reference time-step the handle integrates. */
typedef enum {
    GCM_SW2D = 1,      /* matsuno_c_grid.matsumo_scheme(u,v,p,dx,dt)        matsuno_c_grid.py:125-142 */
    GCM_SW2D_TEMP = 2, /* matsumo_temp.matsumo_scheme(u,v,p,t,dx,dt)        matsumo_temp.py:66-99
                          (+ optional tracer q, two_d.py:198-207 / flux_limiter.py:10-32)            */
    GCM_PE2D = 3,      /* no_limits_2d.matsuno_timestep(p,u,v,t,q,dt,dx)    no_limits_2d.py:129-131   */
    GCM_PE25D = 4      /* dynamics.matsuno_timestep(p,u,v,t,q,dt,geom)      dynamics.py:230-237       */
} gcm_model;

typedef enum { GCM_P = 0, GCM_U = 1, GCM_V = 2, GCM_T = 3, GCM_Q = 4, GCM_NFIELDS = 5 } gcm_field;

/* Tracer scheme carried by GCM_SW2D_TEMP alongside the dynamics (advected by
 * the time-n winds, V = (v, u) in two_d.py's axis convention). */
typedef enum {
    GCM_TRACER_NONE = 0,
    GCM_TRACER_UPWIND = 1,  /* two_d.finite_volume_advection            two_d.py:198-207          */
    GCM_TRACER_VANLEER = 2  /* the same split step with van_leer(calc_r) limiting of the centred
                               flux (flux_limiter.py:10-20, two_d.py:135-149); composition by this
                               build, see DESIGN.md                                               */
} gcm_tracer;

typedef enum { GCM_F64 = 0, GCM_F32 = 1 } gcm_dtype;

/* Kernel variant (same arithmetic, different data movement). */
typedef enum {
    GCM_VARIANT_AUTO = 0,
    GCM_VARIANT_STAGED = 1, /* one launch per Matsuno stage, predicted state materialised in HBM */
    GCM_VARIANT_FUSED = 2   /* predictor + corrector in one launch, predicted state in registers */
} gcm_variant;

typedef struct {
    int32_t abi_version;   /* GCM_ABI_VERSION */
    int32_t model;         /* gcm_model */
    int32_t width;         /* W: cells along i (longitude), contiguous                          */
    int32_t height;        /* H: rows along j owned by THIS handle (a latitude band if nranks>1) */
    int32_t layers;        /* L: sigma levels (GCM_PE25D), else 1                                */
    int32_t tracer;        /* gcm_tracer (GCM_SW2D_TEMP only)                                    */
    int32_t variant;       /* gcm_variant                                                        */
    int32_t filter;        /* GCM_PE25D: 1 = apply low_pass.arakawa_1977 (low_pass.py:41-78)     */
    /* Latitude-band decomposition.  nranks == 1: the handle owns the whole grid
     * and np.roll's pole-to-pole periodicity along j is done by index arithmetic.  nranks > 1:
     * rows [row0, row0+height) of a global_height-row grid; the ghost rows on either side
     * are filled by the caller between steps through gcm_halo_* (RCCL ring incl. the wrap).    */
    int32_t nranks;
    int32_t rank;
    int32_t global_height;
    int32_t row0;
    int32_t device;        /* HIP device ordinal; -1 = current                                   */
    int32_t dtype;         /* gcm_dtype: GCM_F64 (default) or GCM_F32 (GCM_PE25D only: arithmetic and
                              storage in fp32 for the tolerance sweep; the host API stays float64) */
    int32_t halo_steps;    /* 2-D bands: Matsuno steps per ghost-row exchange (ghost depth = 2 *
                              halo_steps rows per side, deep-halo communication avoiding); 0/1 = 1 */
    double dx;             /* scalar grid spacing in metres (2-D models: both axes)              */
    double dy;             /* GCM_PE25D: geom.dy                            geometry.py:138      */
    double ptop;           /* GCM_PE25D: geom.ptop in Pa                    geometry.py:147      */
    /* GCM_PE25D host tables, copied at create (geometry.py:79-85,136-137,149): */
    const double *dx_j;    /* [global_height]  zonal spacing at cell-centre latitudes            */
    const double *dx_h;    /* [global_height]  zonal spacing at v latitudes                      */
    const double *sig;     /* [L] */
    const double *dsig;    /* [L] */
    const double *sigb;    /* [L] */
    const double *sigt;    /* [L] */
    const double *heightmap; /* [global_height][W] (all rows), or NULL for flat topography       */
    /* Coriolis terms of advec_m_pu (dynamics.py:82-92; switched off by `if False` in the
     * reference): cor_u[j] = 2 sin(lat_j) w, cor_v[j] = 2 sin(jph(lat)_j) w, w = 2 pi / day.
     * Both NULL = the reference's literal 0.                                                   */
    const double *cor_u;   /* [global_height] */
    const double *cor_v;   /* [global_height] */
    void *stream;          /* hipStream_t to launch on; NULL = the null stream                   */
} gcm_config;

/* Library / device probes (no handle needed). */
int gcm_abi_version(void);
int gcm_device_count(void);                 /* gfx950 devices visible; 0 on a CPU-only box */
const char *gcm_build_info(void);           /* compiler, offload arch, build flags          */
/* The 256-double table the kernels use for (p/P0)**kappa (temperature.py:7-19): lets a host
 * test check the device algorithm's accuracy without a GPU.  Returns 0. */
int gcm_exner_table(double *out256);
/* The composite-radix plan of the polar filter's in-LDS transform (low_pass.py:41-78 does numpy.fft.rfft /
 * irfft; csrc/fft_lds.h) for rows of n columns, so that a host test can replay the passes' index arithmetic
 * without a GPU.  out[0] = 1 if the composite path serves n (0: generic mixed-radix path), out[1] = passes,
 * out[2] = threads per workgroup, then four words per pass: r1, r2 (the pass has radix r1 * r2),
 * magic = ceil(2^32 / Ns) of the forward pass, imagic = the same for the inverse transform, whose passes take
 * the radices in reversed order.  cap = words available in out (>= 3 + 4 * 8).  Returns 0. */
int gcm_filter_plan(int n, unsigned *out, int cap);

/* Lifetime.  Device buffers are library-owned inside the handle. */
int gcm_create(const gcm_config *cfg, gcm_handle **out);
int gcm_destroy(gcm_handle *h);
const char *gcm_last_error(const gcm_handle *h); /* h may be NULL: last create() failure */

/* State transfer (host <-> device).  NULL pointers are skipped.  Fields a model
 * does not have must be NULL.  2-D models: all five are [H][W]; GCM_PE25D: p is
 * [H][W], the rest [L][H][W].  Reference tuple orders: (u,v,p[,t]) for the
 * shallow-water schemes, (p,u,v,t,q) for the primitive-equation ones.          */
int gcm_set_state(gcm_handle *h, const double *p, const double *u, const double *v,
                  const double *t, const double *q);
int gcm_get_state(gcm_handle *h, double *p, double *u, double *v, double *t, double *q);

/* One or more full Matsuno steps (predictor + corrector), state stays resident.
 * Stands behind matsumo_scheme / matsuno_timestep (files cited at gcm_model). */
int gcm_step(gcm_handle *h, int nsteps, double dt);

/* One Euler stage, for per-stage parity tests and for the reference's
 * boundary_conditions hook (dynamics.py:232-236): stage == 0 computes the
 * predictor from the current state into the handle's "star" buffers; stage == 1
 * computes the corrector from (current, star) and makes it the current state.
 * gcm_get_star / gcm_set_star expose the predicted state in between.
 * Stands behind half_timestep (dynamics.py:183-227, no_limits_2d.py:104-126). */
int gcm_half_step(gcm_handle *h, int stage, double dt);
int gcm_get_star(gcm_handle *h, double *p, double *u, double *v, double *t, double *q);
int gcm_set_star(gcm_handle *h, const double *p, const double *u, const double *v,
                 const double *t, const double *q);

/* Diagnostics the reference's drivers evaluate on the host every step
 * (SURVEY.md 8f-1); computed by device reductions, result copied to *out. */
typedef enum {
    GCM_DIAG_ANY_NAN = 0,   /* np.isnan(u).any() watch            matsuno_c_grid.py:184-187     */
    GCM_DIAG_MAX_U = 1,     /* np.max(u)                          constants.py:111-112          */
    GCM_DIAG_MEAN_P = 2,    /* np.mean(p)                         constants.py:111-112          */
    GCM_DIAG_SUM_P = 3,     /* conservation check                                                */
    GCM_DIAG_MIN_U = 4, GCM_DIAG_MAX_V = 5, GCM_DIAG_MIN_V = 6, /* STATS, no_limits_2_5d.py:85-88 */
    /* get_total_variation(field) = sum |q - roll(q, -1, 0)|  (constants.py:105-108), the monitor
     * run_2d_with_ft evaluates every step (two_d.py:334-338).  Axis 0 of the REFERENCE layout: rows
     * j for 2-D fields, levels k for the 3-D fields of GCM_PE25D.  On a latitude band the last row
     * is differenced against the south ghost row, which must belong to the CURRENT state: a 2-D band
     * exchanges before a step, so after one the call fails with GCM_ERR_STATE until the ghost rows
     * have been exchanged again (gcm_halo_pack2 / exchange / gcm_halo_unpack2); the band sums then
     * add up to the global figure.                                                                */
    GCM_DIAG_TV_P = 7, GCM_DIAG_TV_U = 8, GCM_DIAG_TV_V = 9, GCM_DIAG_TV_T = 10, GCM_DIAG_TV_Q = 11
} gcm_diag_kind;
int gcm_diag(gcm_handle *h, int kind, double *out);
/* The reductions of constants.py for callers that hold no handle: a host float64 array viewed as
 * [n_axis][n_inner] -> out3 = { get_total_variation (sum |x - roll(x, -1, 0)|, constants.py:105-108),
 * max x, mean x (the two reductions of courant_number, :111-112) }; a NaN anywhere in x makes all
 * three NaN, as np.max / np.mean / np.sum do.  Errors: gcm_last_error(NULL).                     */
int gcm_array_stats(const double *x, long n_axis, long n_inner, double *out3);
/* The whole STATS record of full_timestep (no_limits_2_5d.py:85-91) by ONE launch and ONE
 * synchronisation (GCM_PE25D, fp64, single band): out9 = u_max, u_min, v_max, v_min, ke, ate, geo,
 * total (calc_energy, :35-60; `area` as gcm_energy), count of NaNs in u and v.                   */
int gcm_stats(gcm_handle *h, const double *area, int area_len, double *out9);
/* calc_energy(p,u,v,t,q,g,geom) -> out4 = (ke, ate, geo, total) in J, no_limits_2_5d.py:35-60
 * (GCM_PE25D).  `area` is geom.area; the reference broadcasts its (H,) array against the LAST
 * axis (:49), so area_len must be W (== H) or 1 -- reproduced, not fixed.                      */
int gcm_energy(gcm_handle *h, const double *area, int area_len, double *out4);

/* Column physics next to the dynamics (GCM_PE25D).  The ground temperature
 * gt[H][W] (GroundVars.gt, no_limits_2_5d.py:143) lives in the handle.  `lat` [global_height]
 * and `lon` [W] are geom.lat / geom.long in radians; `utc` in seconds.
 *   gcm_grey_radiation  grey_solar.basic_grey_radiation(p,tp,tt,g,t_lw,t_sw,albedo,utc,geom)
 *                       -> dTdt [L][H][W], dt_ground [H][W] (either may be NULL)  grey_solar.py:358-563
 *   gcm_solar_step      no_limits_2_5d.solar_timestep: theta and gt advanced in place by dt
 *                       (the reference passes t_lw = 0.1, t_sw = 0.9, albedo = 0.3)  no_limits_2_5d.py:66-75 */
/* low_pass.arakawa_1977(q, geom) (low_pass.py:41-78) on its own, GCM_PE25D handles: the zonal Fourier
 * damping of nlev <= layers levels of a field on the handle's grid, host [nlev][H][W] float64 in and
 * out (may alias).  Rows are filtered with the multiplier of their global latitude.              */
int gcm_polar_filter(gcm_handle *h, int nlev, const double *in, double *out);
/* Parity tap (GCM_PE25D, single band): the intermediates of the LAST gcm_half_step as the stage kernels
 * themselves left them in the handle -- not a recomputation -- so that a mistake inside K1 / K2 / K3 shows
 * as a wrong intermediate, not only as a wrong stage output.  Host float64, reference layouts.
 *   GCM_INT_SPU   [L][H][W]  spu = arakawa_1977(calc_pu(sp, su))                 dynamics.py:186-190
 *   GCM_INT_PIT   [H][W]     pit = sum_k conv (from the 2-D column sums)         dynamics.py:35-40
 *   GCM_INT_PN    [H][W]     p_n = p - pit dt                                    dynamics.py:194
 *   GCM_INT_PHI   [L][H][W]  compute_geopotential: the stored anchors on the even levels, the odd levels
 *                            rebuilt from them exactly as K3 and K4 do (phi_up)  dynamics.py:111-143
 *   GCM_INT_PGFU  [L][H][W]  arakawa_1977(pgu + phiu)                            dynamics.py:147-171,203      */
typedef enum { GCM_INT_SPU = 0, GCM_INT_PIT = 1, GCM_INT_PN = 2, GCM_INT_PHI = 3, GCM_INT_PGFU = 4 } gcm_intermediate;
int gcm_get_intermediate(gcm_handle *h, int kind, double *out);
int gcm_set_ground(gcm_handle *h, const double *gt);
int gcm_get_ground(gcm_handle *h, double *gt);
int gcm_grey_radiation(gcm_handle *h, double utc, double t_lw, double t_sw, double albedo,
                       const double *lat, const double *lon, double *dTdt, double *dt_ground);
int gcm_solar_step(gcm_handle *h, double dt, double utc, double t_lw, double t_sw, double albedo,
                   const double *lat, const double *lon);
/* The column physics as the second phase of every step (BASELINE configs[4]: dynamics + solar_timestep;
 * the loop of no_limits_2_5d.run_model, :229-234, with the physics the reference keeps below
 * full_timestep's early return, :94-96).  After gcm_set_physics every step taken by gcm_step and
 * gcm_band_run is  matsuno_timestep(dt)  followed by  solar_timestep(t, p, g, dt, utc, geom)  and
 * utc += dt  (:231).  On a latitude band (nranks > 1) the ghost rows are radiated LOCALLY -- the kernel
 * is column-local, so no third exchange per step is needed: the ghost rows of the ground temperature
 * travel with every ghost-row message (gcm_halo_bytes counts them; pack / unpack move them), the ghost
 * rows of theta that the post-corrector exchange delivers are advanced by the same kernel with the
 * neighbour's own inputs, and so hold the neighbour's own bits.  An explicit gcm_solar_step on a band
 * does the same (own rows and ghost rows), for callers that drive the exchange themselves; the ghost
 * rows of the current state must then be current (the post-corrector exchange unpacked).
 * `lat` [global_height] and `lon` [width] are copied.  NULL switches the physics off.          */
typedef struct {
    double utc;                  /* seconds; advanced by dt after every step                  */
    double t_lw, t_sw, albedo;   /* the reference passes 0.1, 0.9, 0.3  no_limits_2_5d.py:69  */
    const double *lat, *lon;     /* geom.lat [global_height], geom.long [width], radians      */
} gcm_physics;
int gcm_set_physics(gcm_handle *h, const gcm_physics *ph);
int gcm_get_utc(gcm_handle *h, double *utc);   /* the physics clock (GCM_ERR_STATE without gcm_set_physics) */

/* Device-side snapshot / restore of the current state, ghost rows included (2-D models): a long
 * run can restart from a known state without a host round trip.  gcm_restore is asynchronous on
 * the handle's stream.                                                                          */
int gcm_snapshot(gcm_handle *h);
int gcm_restore(gcm_handle *h);

/* Latitude-band ghost rows (nranks > 1).  The library packs the rows a neighbour
 * needs into / unpacks them from caller-owned DEVICE buffers (e.g. torch tensors
 * handed to torch.distributed / RCCL send-recv); it never calls a collective
 * itself.  `side` 0 = towards row 0 (north), 1 = towards the last row (south).
 * gcm_halo_bytes gives the buffer size for one side (GCM_PE25D: the two rows of p and of
 * u, v, t, q on every level in the handle's storage type, then the two rows of the ground
 * temperature in float64 -- see gcm_set_physics).                                            */
size_t gcm_halo_bytes(const gcm_handle *h);
int gcm_halo_pack(gcm_handle *h, int side, void *dev_buf, void *stream);
int gcm_halo_unpack(gcm_handle *h, int side, const void *dev_buf, void *stream);
/* both sides with one launch (buffers as above: north = side 0, south = side 1) */
int gcm_halo_pack2(gcm_handle *h, void *north_buf, void *south_buf, void *stream);
int gcm_halo_unpack2(gcm_handle *h, const void *north_buf, const void *south_buf, void *stream);
/* Step split for comm/compute overlap: rows that need no ghost data, then the rest. */
int gcm_step_interior(gcm_handle *h, double dt, void *stream);
int gcm_step_boundary(gcm_handle *h, double dt, void *stream);
/* GCM_PE25D bands, exchange hidden behind the update kernel's interior rows: each Euler stage is
 * split into "everything the neighbours wait for" and "the rest".  phase 0: predictor K1-K3 + the
 * update of the two edge rows on either side; gcm_halo_pack then packs the PREDICTED edge rows;
 * phase 1: predictor update of the interior rows (overlaps the exchange); gcm_halo_unpack fills the
 * predicted state's ghosts; phase 2 / 3: the same for the corrector (pack = the NEW state's edge
 * rows; phase 3 ends with the swap; unpack then fills the new current state's ghosts, which the
 * next step's phase 0 needs).  Before the first step the current state's ghosts must be exchanged
 * once (pack / unpack with no phase pending).                                                    */
int gcm_step_phase(gcm_handle *h, int phase, double dt, void *stream);
/* Optional: register the two DEVICE send buffers (gcm_halo_bytes each).  Phase 0 / 2 then update the
 * edge rows AND pack them into these buffers on the handle's second stream, concurrently with
 * whatever the caller queues next on `stream` (phase 1 / 3, the interior rows); no gcm_halo_pack
 * call is needed for those phases.  gcm_wait_edges makes a stream (the one the send is posted on)
 * wait for that pack.  NULL, NULL unregisters.                                                  */
int gcm_set_halo_buffers(gcm_handle *h, void *north_send, void *south_send);
/* A stream for the exchange, owned by the handle (created on first request): HIP maps streams onto a
 * few hardware queues and two streams on one queue run in order, so the library hands out one that
 * it has measured to run beside the handle's stream (and its internal second stream).            */
int gcm_comm_stream(gcm_handle *h, void **stream);
int gcm_wait_edges(gcm_handle *h, void *stream);

/* The whole band step inside the library: one call per run instead of ~16 per step from the host.
 * gcm_set_exchange hands the library what it needs to post the ghost-row exchange itself -- the
 * RCCL communicator, the two ring neighbours, the addresses of ncclSend / ncclRecv / ncclGroupStart /
 * ncclGroupEnd in the librccl.so the process already has loaded (the library does not link RCCL), and
 * four DEVICE buffers of gcm_halo_bytes() each.  With all four function pointers NULL the exchange is
 * a device-local copy (the band is its own neighbour on both sides: the periodic single domain;
 * tests and the one-GPU scaling tools).  gcm_band_run then runs `nsteps` full steps, exchanges
 * included (GCM_PE25D: two per step, posted on the handle's comm stream behind the edge rows and
 * overlapped with the interior rows; 2-D models: one per halo_steps steps), asynchronously on the
 * handle's streams.  The sequence is the one gcm_step_phase / gcm_halo_* document, so the results
 * are bit-identical to a host-driven band.                                                       */
typedef int (*gcm_p2p_fn)(const void *buf, size_t count, int datatype, int peer, void *comm, void *stream);
typedef int (*gcm_group_fn)(void);
typedef struct {
    void *comm;               /* ncclComm_t */
    int32_t north, south;     /* ranks of the ring neighbours in that communicator */
    gcm_p2p_fn send, recv;    /* ncclSend, ncclRecv   (bytes are sent as ncclChar) */
    gcm_group_fn group_start, group_end;
    void *send_north, *send_south, *recv_north, *recv_south;
} gcm_exchange;
int gcm_set_exchange(gcm_handle *h, const gcm_exchange *x);   /* NULL: unregister */
int gcm_band_run(gcm_handle *h, int nsteps, double dt);
/* 2-D models with halo_steps > 1 (an exchange every k steps): on = 1 hides the exchange behind the
 * interior rows of the step before and the step after it (the last step of a window produces, packs
 * and sends its edge rows first; the first step of the next starts with the rows that need no ghost
 * data).  Same kernels on the same rows: bit-identical results.  Costs four more launches per window,
 * so it pays where the exchange takes longer than that (measured per run by bench.py --gpus N, which
 * times both and keeps the faster).  Default off, or GCM_BAND_OVERLAP=1 at gcm_set_exchange.       */
/* GCM_PE25D bands: on = 1 holds the interior rows' update kernel back until the library's second stream has reached
 * the edge rows' update kernel, so that the edge rows' workgroups are dispatched first (15-20 us instead of the 60-70
 * they take when the two launches race for the chip): the pack and the exchange of a stage start ~45 us earlier, the
 * interior rows ~10 us later.  Same kernels on the same rows: bit-identical results.  Pays where an exchange takes
 * longer than ~30 us (measured per run by bench.py --gpus N, as above).  Default off.                              */
int gcm_set_band_overlap(gcm_handle *h, int on);

int gcm_sync(gcm_handle *h);

/* Stand-alone 2-D operators of two_d.py on velocity stacks V[axis] (host arrays in/out; V is
 * [2][H][W] with V[0] acting along array axis 0 = rows and dx0 = spatial_change[0], as
 * two_d.py:16-22).  `axes` is a bit mask (1 = axis 0, 2 = axis 1, 3 = both, axis 0 first --
 * the dimension split of corner_transport_2d / finite_volume_advection); `finite` = 1 returns
 * the increment of one axis pass instead of the new field (the *_finite functions).          */
typedef enum {
    GCM_ADV_UPWIND = 0,    /* upwind_axis / corner_transport_2d           two_d.py:11-71            */
    GCM_ADV_FV_UPWIND = 1, /* fv_advect_axis_upwind / finite_volume_advection  two_d.py:103-132,198-207 */
    GCM_ADV_FV_PLAIN = 2,  /* fv_advect_axis_plain                        two_d.py:135-166          */
    GCM_ADV_VANLEER = 3,   /* fv upwind + van_leer(calc_r)-limited centred flux (composition)        */
    GCM_ADV_MOMENTUM = 4   /* advect_with_momentum: V * pressure_at_edge(p), then fv upwind  :277-292 */
} gcm_adv_scheme;
int gcm_advect2d(int scheme, int axes, int finite, int width, int height, int nsteps, double dt,
                 double dx0, double dx1, const double *V, const double *q_in, double *q_out);
/* kind 0: pgf_c_grid_axis gradients (two_d.py:210-220); 1: pgf_c_grid (needs t, :223-245);
 * 2: pgf_templess (:248-261); 3: pressure_at_edge (:264-268; out2[0] alone is
 * pressure_at_edge_one_d, :271-274); 4: gradient, centred (:74-77); 5: pressure_gradient (needs t,
 * :80-100); 6: pgf_one_d along axis 0 in out2[0] and along axis 1 in out2[1] (:295-303; the edge
 * density is taken along axis 0 for either, as the reference does).  out2 is [2][H][W]; a 1-D array
 * of n cells is height n, width 1.                                                          */
int gcm_pgf2d(int kind, int width, int height, double dt, double dx0, double dx1, const double *p,
              const double *t, double *out2);
/* The 2-D operators the step kernels are fused from, one by one (host float64 arrays [H][W] in and
 * out, periodic in both axes as the reference's np.roll shifts): what matsuno_c_grid.py,
 * viscosity.py, matsumo_temp.py and temperature.py export.  Inputs x0, x1, x2 in the reference's
 * argument order; dx, mu where the operator takes them (else ignored).  The elementwise ones take
 * any array as height 1, width n.                                                            */
typedef enum {
    GCM_OP_ADV_U = 0,             /* advection_of_velocity_u(u, v, dx)        matsuno_c_grid.py:15-51  */
    GCM_OP_ADV_V = 1,             /* advection_of_velocity_v(u, v, dx)        :54-80                   */
    GCM_OP_GEO_GRAD_U = 2,        /* geopotential_gradient_u(p, dx)           :97-100                  */
    GCM_OP_GEO_GRAD_V = 3,        /* geopotential_gradient_v(p, dx)           :103-106                 */
    GCM_OP_ADV_GEO = 4,           /* advection_of_geopotential(u, v, p, dx)   :109-118                 */
    GCM_OP_LAPLACIAN = 5,         /* finite_laplacian_2d(q, dx)               viscosity.py:12-19       */
    GCM_OP_VISCOSITY = 6,         /* incompressible_viscosity_2d(u, mu, dx)   :22-25                   */
    GCM_OP_DENSITY_FROM = 7,      /* density_from(p, t)                       matsumo_temp.py:13-19    */
    GCM_OP_GEOPOTENTIAL_FROM = 8, /* geopotential_from(rho, p)                :45-47                   */
    GCM_OP_TO_TRUE_TEMP = 9,      /* to_true_temp(t, p)                       temperature.py:7-12      */
    GCM_OP_TO_POTENTIAL_TEMP = 10,/* to_potential_temp(tt, p)                 :15-19                   */
    GCM_OP_TO_DENSITY = 11,       /* to_density(tt, p)                        :22-24                   */
    GCM_OP_SCALING = 12,          /* scaling(pa, t, dx)                       matsumo_temp.py:28-30    */
    GCM_OP_UNSCALING = 13,        /* unscaling(pb, tt, dx)                    :33-35                   */
    GCM_OP_PE2D_ADVEC_P = 14,     /* advec_p(pu, pv, dx)                      no_limits_2d.py:41-44    */
    GCM_OP_PE2D_DUT = 15,         /* advec_m(p, u, v, dx)[0]                  :47-76                   */
    GCM_OP_PE2D_DVT = 16,         /* advec_m(p, u, v, dx)[1]                                           */
    GCM_OP_PE2D_PGF_U = 17,       /* pgf(p, t, dx)[0]                         :79-92                   */
    GCM_OP_PE2D_PGF_V = 18        /* pgf(p, t, dx)[1]                                                  */
} gcm_sw2d_op_kind;
int gcm_sw2d_op(int kind, int width, int height, double dx, double mu, const double *x0, const double *x1,
                const double *x2, double *out);
/* The operators of dynamics.py one by one (dynamics.py:15-181): host float64 arrays in the
 * reference's layout, 3-D [layers][height][width], 2-D [height][width], periodic in i and j (and in
 * k where the reference rolls k), geometry tables as geometry.gen_geometry builds them.  Inputs and
 * outputs in the reference's argument / return order:
 *   CALC_PU (p, u) -> pu            CALC_PV (p, v) -> pv         UN_PU (pu, p) -> u     UN_PV (pv, p) -> v
 *   AFLUX (pu, pv) -> pit[2-D], sd  ADVEC_SIG (sd, q) -> dq      ADVEC_M_PU (p, u, v, pu, pv) -> dut, dvt
 *   GEOPOTENTIAL (p, t) -> phi      PGF (p, t) -> pgfu, pgfv, phiu, phiv       ADVEC_T (pu, pv, t) -> dt   */
typedef enum {
    GCM_PEOP_CALC_PU = 0, GCM_PEOP_CALC_PV = 1, GCM_PEOP_UN_PU = 2, GCM_PEOP_UN_PV = 3, GCM_PEOP_AFLUX = 4,
    GCM_PEOP_ADVEC_SIG = 5, GCM_PEOP_ADVEC_M_PU = 6, GCM_PEOP_GEOPOTENTIAL = 7, GCM_PEOP_PGF = 8, GCM_PEOP_ADVEC_T = 9
} gcm_pe25d_op_kind;
typedef struct {
    const double *dx_j, *dx_h;                  /* [height]  geometry.py:136-137 */
    const double *dsig, *sig, *sigb, *sigt;     /* [layers]                      */
    const double *heightmap;                    /* [height][width] or NULL       */
    double dy, ptop;
} gcm_pe_geom;
int gcm_pe25d_op(int kind, int width, int height, int layers, const gcm_pe_geom *g, const double *const in[5],
                 double *const out[4]);
const char *gcm_pe25d_op_last_error(void);
/* flux_limiter.py on 1-D arrays of n cells (host arrays in/out; ip/im = np.roll by -1/+1,
 * coordinates_1d.py:25-30).  Results are BIT-identical to NumPy's, masks included: IEEE division,
 * no contraction.
 *   kind 0  van_leer(q)                    (r + |r|) / (1 + |r|)                       :10-11
 *   kind 1  calc_r(q)                      (q - im(q)) / (ip(q) - q), 0 where the denominator == 0  :14-20
 *   kind 2  donor_cell_flux(q, u)          where(u > 0, q, ip(q)) * u                  :23-27
 *   kind 3  donor_cell_advection(q,u,dx,dt) q + (im(flux) - flux) * dt / dx            :30-32
 * `u` is ignored by kinds 0 and 1 (may be NULL).                                                 */
typedef enum { GCM_FL_VAN_LEER = 0, GCM_FL_CALC_R = 1, GCM_FL_DONOR_FLUX = 2, GCM_FL_DONOR_ADVECTION = 3 } gcm_fl_kind;
int gcm_flux_limiter(int kind, int n, const double *q, const double *u, double dx, double dt, double *out);
/* The 1-D model of BASELINE configs[0] (no_limits.py:50-152): p, u, theta, q on a periodic line of n
 * cells, momentum form.  half_only = 1: ONE Euler stage, half_timestep(p,u,t,q, sp,su,st,sq, dt, dx)
 * (:115-147), base and stage given; half_only = 0: nsteps full Matsuno steps, matsuno_timestep
 * (:150-152), `stage` ignored.  Arrays are {p, u, t, q}, host float64.                            */
int gcm_pe1d(int n, int nsteps, int half_only, double dt, double dx, const double *const base[4],
             const double *const stage[4], double *const out[4]);
/* ... and its operators one by one (no_limits.py:50-112), arguments in the reference's order:
 * ADVEC_Q (u, q), CALC_PU (u, p), UN_PU (pu, p), ADVEC_P (pu), ADVEC_PU (p, pu, u), ADVEC_T (pu, t), PGF (p, t) */
typedef enum {
    GCM_OP1D_ADVEC_Q = 0, GCM_OP1D_CALC_PU = 1, GCM_OP1D_UN_PU = 2, GCM_OP1D_ADVEC_P = 3, GCM_OP1D_ADVEC_PU = 4,
    GCM_OP1D_ADVEC_T = 5, GCM_OP1D_PGF = 6
} gcm_pe1d_op_kind;
int gcm_pe1d_op(int kind, int n, double dx, const double *x0, const double *x1, const double *x2, double *out);
const char *gcm_ops_last_error(void);
/* The stand-alone operator entry points above (host arrays in, host arrays out: gcm_sw2d_op, gcm_pe25d_op,
 * gcm_pe1d_op, gcm_flux_limiter, gcm_pgf2d, gcm_advect2d) carve their device operands from a scratch arena the
 * calling thread keeps between calls (at most 256 MB: a call that needed more hands everything back when it
 * ends).  This frees the calling thread's arena now.  Returns 0.                                          */
int gcm_ops_release_scratch(void);

/* Timing helper for bench.py: runs nsteps steps bracketed by HIP events on the
 * handle's stream; returns elapsed milliseconds in *ms and, in *kernel_ms_avg,
 * the mean duration of the dominant kernel's launches measured by per-launch
 * event pairs in a second pass (so the first figure carries no event overhead).
 * NOTE: with kernel_ms_avg != NULL the state advances 2 * nsteps steps (the second
 * pass re-runs the same number of steps).  GCM_DIAG_ANY_NAN looks at u only, as the
 * reference's watch does (np.isnan(u).any(), matsuno_c_grid.py:184-187).            */
int gcm_time_steps(gcm_handle *h, int nsteps, double dt, double *ms, double *kernel_ms_avg);

#ifdef __cplusplus
}
#endif
#endif /* GCMCORE_H */
