// Host side of libgcmcore.so: the C ABI of include/gcmcore.h.
// Owns device buffers, picks kernels, launches on the handle's stream.  There is
// no CPU fallback anywhere in this file: without a HIP device gcm_create fails.
#include "../../include/gcmcore.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pe25d_kernels.h"
#include "sw2d_kernels.h"

using namespace gcm;

namespace {
thread_local std::string g_create_error;
}

struct gcm_handle {
    gcm_config cfg{};
    int W = 0, H = 0, L = 1;
    bool wrap = true;  // nranks == 1: periodic rows by index arithmetic
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void *> allocs;

    // 2-D models: per-field arrays of (H + 2*kGhost) rows; pointers address interior row 0
    double *cur[GCM_NFIELDS] = {}, *nxt[GCM_NFIELDS] = {}, *star[GCM_NFIELDS] = {};
    double *geo = nullptr, *irho = nullptr, *sst = nullptr, *qtmp = nullptr;
    bool has[GCM_NFIELDS] = {};
    double *exner_tab = nullptr;
    int G = kGhost;          // ghost rows per side = 2 * steps between exchanges (2-D bands)
    int since_exchange = 0;  // steps taken on the current ghost rows
    bool ghosts_current = false;  // 2-D bands: the current state's ghost rows were filled after its last step
    bool star_valid = false;
    bool launch_refused = false;      // hipLaunchKernel of the fused step returned an error (reported by launch_status)
    hipStream_t comm = nullptr;       // gcm_comm_stream: owned, created on first request
    double *snap[GCM_NFIELDS] = {};   // gcm_snapshot: device copy of the state, ghost rows included
    int snap_since_exchange = 0;
    int variant = GCM_VARIANT_FUSED;
    int rows_per_band = 32;

    // diagnostics scratch
    double *diag_dev = nullptr;
    static constexpr int kDiagBlocks = 512;

    // per-launch timing of the dominant kernel (gcm_time_steps second pass)
    bool timing = false;
    std::vector<hipEvent_t> ev;
    std::vector<hipEvent_t> region_ev;   // gcm_time_steps: start / end of the timed region
    size_t ev_used = 0;

    Pe25d *pe = nullptr;  // GCM_PE25D state (pe25d_kernels.h)

    // gcm_band_run: the exchange the library posts itself
    gcm_exchange xch{};
    bool xch_set = false, primed = false;
    bool xch_inflight = false;                 // gcm_band_run: an exchange posted, its unpack still to come
    bool band_overlap = false;                 // deep-halo bands: hide the exchange behind interior rows (gcm_set_band_overlap)
    hipEvent_t ev_pack = nullptr, ev_comm = nullptr;
    bool join_pending = false;                 // gcm_band_run (GCM_PE25D): work on the second stream not yet joined
    bool on_comm = false;                      // GCM_BAND_COMM_STREAM=1 at gcm_set_exchange: exchange on the comm stream, a join per stage

    // gcm_set_physics: solar_timestep as the second phase of every step
    bool phys_on = false;
    gcm_physics phys{};
    std::vector<double> phys_lat, phys_lon;
};

#define HIPCHK(h, call)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            char b_[512];                                                                  \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                  \
            (h)->err = b_;                                                                 \
            return GCM_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

static int fail(gcm_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

static int alloc_field(gcm_handle *h, double **p) {
    // Every array starts a different multiple of 256 B past its (2 MiB-aligned) allocation: rows of a
    // power-of-two width would otherwise put the same (row, column) of all fields on the same HBM channel and
    // bank, and a wave of the fused kernel touches that element of ten arrays per row (tools/micro/copy_width.hip:
    // the bare access pattern moves 4-6 % faster with the arrays skewed).  GCM_ALLOC_SKEW=0 switches it off.
    static const long skew_unit = getenv("GCM_ALLOC_SKEW") ? atol(getenv("GCM_ALLOC_SKEW")) : 256;
    const size_t n = (size_t)(h->H + 2 * h->G) * h->W;
    const size_t skew = (size_t)(skew_unit > 0 ? skew_unit : 0) * (h->allocs.size() % 16) / sizeof(double);
    void *d = nullptr;
    HIPCHK(h, hipMalloc(&d, (n + skew) * sizeof(double)));
    HIPCHK(h, hipMemsetAsync(d, 0, (n + skew) * sizeof(double), h->stream));
    h->allocs.push_back(d);
    *p = (double *)d + skew + (size_t)h->G * h->W;
    return GCM_OK;
}

extern "C" {

int gcm_abi_version(void) { return GCM_ABI_VERSION; }

int gcm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *gcm_build_info(void) {
    return "libgcmcore gfx950 (hipcc " __VERSION__ "), fp64, kernels: sw2d staged+fused, pe25d";
}

int gcm_exner_table(double *out256) {
    if (!out256) return GCM_ERR_ARG;
    build_exner_table(out256);
    return GCM_OK;
}

int gcm_filter_plan(int n, unsigned *out, int cap) { return pe25d_filter_plan(n, out, cap); }

const char *gcm_last_error(const gcm_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int gcm_destroy(gcm_handle *h) {
    if (!h) return GCM_OK;
    if (h->cfg.device >= 0) (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    if (h->comm) {
        (void)hipStreamSynchronize(h->comm);
        (void)hipStreamDestroy(h->comm);
    }
    if (h->pe) pe25d_destroy(h->pe);
    for (void *p : h->allocs) (void)hipFree(p);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->region_ev) (void)hipEventDestroy(e);
    if (h->ev_pack) (void)hipEventDestroy(h->ev_pack);
    if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
    delete h;
    return GCM_OK;
}

int gcm_create(const gcm_config *cfg, gcm_handle **out) {
    if (!cfg || !out) return fail(nullptr, GCM_ERR_ARG, "gcm_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != GCM_ABI_VERSION)
        return fail(nullptr, GCM_ERR_ARG, "gcm_create: abi_version mismatch");
    if (cfg->width < 1 || cfg->height < 1 || cfg->layers < 1)
        return fail(nullptr, GCM_ERR_ARG, "gcm_create: width/height/layers must be >= 1");
    if (cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks)
        return fail(nullptr, GCM_ERR_ARG, "gcm_create: bad rank/nranks");
    const int hsteps = cfg->halo_steps < 1 ? 1 : cfg->halo_steps;
    if (cfg->nranks > 1 && cfg->height < kGhost * hsteps)
        return fail(nullptr, GCM_ERR_ARG, "gcm_create: a latitude band needs >= 2 * halo_steps rows");
    if (hsteps > 1 && (cfg->nranks == 1 || (cfg->model != GCM_SW2D && cfg->model != GCM_SW2D_TEMP)))
        return fail(nullptr, GCM_ERR_ARG, "gcm_create: halo_steps > 1 needs a 2-D latitude band");
    if (gcm_device_count() < 1)
        return fail(nullptr, GCM_ERR_NODEVICE,
                    "gcm_create: no HIP device visible; libgcmcore has no CPU fallback");
    gcm_handle *h = new gcm_handle;
    h->cfg = *cfg;
    h->W = cfg->width;
    h->H = cfg->height;
    h->L = cfg->layers;
    h->wrap = cfg->nranks == 1;
    h->G = kGhost * hsteps;
    h->stream = (hipStream_t)cfg->stream;
    int rc = GCM_OK;
    auto bail = [&](int code, const std::string &m) {
        g_create_error = m.empty() ? h->err : m;
        gcm_destroy(h);
        return code;
    };
    if (cfg->device >= 0 && hipSetDevice(cfg->device) != hipSuccess)
        return bail(GCM_ERR_HIP, "gcm_create: hipSetDevice failed");

    switch (cfg->model) {
        case GCM_SW2D:
        case GCM_SW2D_TEMP: {
            if (!(cfg->dx > 0)) return bail(GCM_ERR_ARG, "gcm_create: dx must be > 0");
            const bool temp = cfg->model == GCM_SW2D_TEMP;
            if (!temp && cfg->tracer != GCM_TRACER_NONE)
                return bail(GCM_ERR_ARG, "gcm_create: tracer needs GCM_SW2D_TEMP");
            if (cfg->tracer < 0 || cfg->tracer > GCM_TRACER_VANLEER)
                return bail(GCM_ERR_ARG, "gcm_create: bad tracer");
            h->has[GCM_U] = h->has[GCM_V] = h->has[GCM_P] = true;
            h->has[GCM_T] = temp;
            h->has[GCM_Q] = temp && cfg->tracer != GCM_TRACER_NONE;
            h->variant = cfg->variant == GCM_VARIANT_AUTO ? GCM_VARIANT_FUSED : cfg->variant;
            if (h->variant != GCM_VARIANT_FUSED && h->variant != GCM_VARIANT_STAGED)
                return bail(GCM_ERR_ARG, "gcm_create: bad variant");
            for (int f = 0; f < GCM_NFIELDS; ++f) {
                if (!h->has[f]) continue;
                if ((rc = alloc_field(h, &h->cur[f]))) return bail(rc, "");
                if ((rc = alloc_field(h, &h->nxt[f]))) return bail(rc, "");
                if (f != GCM_Q && (rc = alloc_field(h, &h->star[f]))) return bail(rc, "");
            }
            if (temp) {
                if ((rc = alloc_field(h, &h->geo))) return bail(rc, "");
                if ((rc = alloc_field(h, &h->irho))) return bail(rc, "");
                if ((rc = alloc_field(h, &h->sst))) return bail(rc, "");
            }
            if (h->has[GCM_Q] && (rc = alloc_field(h, &h->qtmp))) return bail(rc, "");
            if (temp) {
                double tab[kExnerTabDoubles];
                build_exner_table(tab);
                void *d = nullptr;
                if (hipMalloc(&d, sizeof tab) != hipSuccess ||
                    hipMemcpy(d, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess)
                    return bail(GCM_ERR_HIP, "gcm_create: exner table upload failed");
                h->allocs.push_back(d);
                h->exner_tab = (double *)d;
            }
            h->rows_per_band = sw2d_fused_rows_per_band(h->W, h->H, temp, h->has[GCM_Q] ? cfg->tracer : 0, h->wrap);
            break;
        }
        case GCM_PE2D: {
            if (!(cfg->dx > 0)) return bail(GCM_ERR_ARG, "gcm_create: dx must be > 0");
            if (cfg->nranks != 1)
                return bail(GCM_ERR_UNSUPPORTED, "gcm_create: GCM_PE2D has no latitude-band mode");
            h->variant = GCM_VARIANT_STAGED;
            for (int f = 0; f < GCM_NFIELDS; ++f) {
                h->has[f] = true;
                if ((rc = alloc_field(h, &h->cur[f]))) return bail(rc, "");
                if ((rc = alloc_field(h, &h->nxt[f]))) return bail(rc, "");
                if ((rc = alloc_field(h, &h->star[f]))) return bail(rc, "");
            }
            double tab[kExnerTabDoubles];
            build_exner_table(tab);
            void *d = nullptr;
            if (hipMalloc(&d, sizeof tab) != hipSuccess ||
                hipMemcpy(d, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess)
                return bail(GCM_ERR_HIP, "gcm_create: exner table upload failed");
            h->allocs.push_back(d);
            h->exner_tab = (double *)d;
            break;
        }
        case GCM_PE25D: {
            std::string msg;
            h->pe = pe25d_create(*cfg, h->stream, &msg);
            if (!h->pe) return bail(msg.find("hip") == 0 ? GCM_ERR_HIP : GCM_ERR_ARG, msg);
            h->has[GCM_P] = h->has[GCM_U] = h->has[GCM_V] = h->has[GCM_T] = h->has[GCM_Q] = true;
            break;
        }
        default:
            return bail(GCM_ERR_UNSUPPORTED, "gcm_create: model not built in this round");
    }
    void *d = nullptr;
    if (hipMalloc(&d, sizeof(double) * 4 * gcm_handle::kDiagBlocks) != hipSuccess)
        return bail(GCM_ERR_HIP, "gcm_create: hipMalloc(diag) failed");
    h->allocs.push_back(d);
    h->diag_dev = (double *)d;
    if (hipStreamSynchronize(h->stream) != hipSuccess)
        return bail(GCM_ERR_HIP, "gcm_create: stream sync failed");
    *out = h;
    return GCM_OK;
}

// ------------------------------------------------------------------ state transfer
static int xfer(gcm_handle *h, double *const dev[GCM_NFIELDS], const double *const hostc[GCM_NFIELDS],
                double *const hostm[GCM_NFIELDS], bool to_device) {
    const size_t bytes = (size_t)h->H * h->W * sizeof(double);
    for (int f = 0; f < GCM_NFIELDS; ++f) {
        const void *src = hostc ? (const void *)hostc[f] : (const void *)hostm[f];
        if (!src) continue;
        if (!h->has[f] || !dev[f])
            return fail(h, GCM_ERR_ARG, "state transfer: field not part of this model");
        if (to_device)
            HIPCHK(h, hipMemcpyAsync(dev[f], hostc[f], bytes, hipMemcpyHostToDevice, h->stream));
        else
            HIPCHK(h, hipMemcpyAsync(hostm[f], dev[f], bytes, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return GCM_OK;
}

int gcm_set_state(gcm_handle *h, const double *p, const double *u, const double *v,
                  const double *t, const double *q) {
    if (!h) return GCM_ERR_ARG;
    h->primed = false;                                    // gcm_band_run: the new state's ghost rows are not exchanged yet
    h->ghosts_current = false;
    if (h->pe) return pe25d_set(h->pe, false, p, u, v, t, q, h->stream, &h->err);
    const double *src[GCM_NFIELDS] = {p, u, v, t, q};
    h->star_valid = false;
    return xfer(h, h->cur, src, nullptr, true);
}

int gcm_get_state(gcm_handle *h, double *p, double *u, double *v, double *t, double *q) {
    if (!h) return GCM_ERR_ARG;
    if (h->pe) return pe25d_get(h->pe, false, p, u, v, t, q, h->stream, &h->err);
    double *dst[GCM_NFIELDS] = {p, u, v, t, q};
    return xfer(h, h->cur, nullptr, dst, false);
}

int gcm_set_star(gcm_handle *h, const double *p, const double *u, const double *v,
                 const double *t, const double *q) {
    if (!h) return GCM_ERR_ARG;
    if (h->pe) return pe25d_set(h->pe, true, p, u, v, t, q, h->stream, &h->err);
    if (q && h->cfg.model != GCM_PE2D)
        return fail(h, GCM_ERR_ARG, "set_star: the tracer has no predicted state");
    if (h->cfg.model == GCM_PE2D) {
        const double *src5[GCM_NFIELDS] = {p, u, v, t, q};
        int rc5 = xfer(h, h->star, src5, nullptr, true);
        if (rc5 == GCM_OK) h->star_valid = true;
        return rc5;
    }
    const double *src[GCM_NFIELDS] = {p, u, v, t, nullptr};
    int rc = xfer(h, h->star, src, nullptr, true);
    if (rc == GCM_OK) h->star_valid = true;
    return rc;
}

int gcm_get_star(gcm_handle *h, double *p, double *u, double *v, double *t, double *q) {
    if (!h) return GCM_ERR_ARG;
    if (h->pe) return pe25d_get(h->pe, true, p, u, v, t, q, h->stream, &h->err);
    if (!h->star_valid) return fail(h, GCM_ERR_STATE, "get_star: no predicted state yet");
    if (h->cfg.model == GCM_PE2D) {
        double *dst5[GCM_NFIELDS] = {p, u, v, t, q};
        return xfer(h, h->star, nullptr, dst5, false);
    }
    if (q) return fail(h, GCM_ERR_ARG, "get_star: the tracer has no predicted state");
    double *dst[GCM_NFIELDS] = {p, u, v, t, nullptr};
    return xfer(h, h->star, nullptr, dst, false);
}

// ------------------------------------------------------------------ stepping (2-D)
static Sw2dArgs base_args(gcm_handle *h, double dt) {
    Sw2dArgs a{};
    a.bu = h->cur[GCM_U];
    a.bv = h->cur[GCM_V];
    a.bp = h->cur[GCM_P];
    a.bt = h->cur[GCM_T];
    a.bq = h->cur[GCM_Q];
    a.exner_tab = h->exner_tab;
    a.W = h->W;
    a.H = h->H;
    a.wrap_j = h->wrap ? 1 : 0;
    a.j0 = 0;
    a.j1 = h->H;
    a.rows_per_band = h->rows_per_band;
    a.dt = dt;
    a.dx = h->cfg.dx;
    a.inv_dx = 1.0 / h->cfg.dx;
    a.dx2 = h->cfg.dx * h->cfg.dx;
    a.inv_dx2 = 1.0 / (h->cfg.dx * h->cfg.dx);
    a.h_dx = 0.5 / h->cfg.dx;
    a.dtdx = dt / h->cfg.dx;
    a.g_dx = 9.8 / h->cfg.dx;
    a.mu_dx2 = (18.5 * 1e-6) * 287.0 / (h->cfg.dx * h->cfg.dx);   // mu_air Rd / dx^2 (see thermo())
    a.inv_dx2_ = a.inv_dx2;
    return a;
}

// gcm_set_physics: the radiation kernel's tables in place before a run queues anything (no-op without physics)
static int physics_tables(gcm_handle *h) {
    if (!h->phys_on) return GCM_OK;
    return pe25d_physics_tables(h->pe, h->phys.t_lw, h->phys.t_sw, h->phys_lat.data(), h->phys_lon.data(), h->stream, &h->err);
}

// what the launches queued since the last check returned: the status hipLaunchKernel handed back for the
// fused step (kept in the handle: step_rows has many callers) and the runtime's sticky last error
static int launch_status(gcm_handle *h) {
    const hipError_t e = hipGetLastError();
    if (h->launch_refused) {
        h->launch_refused = false;
        return fail(h, GCM_ERR_HIP, std::string("sw2d fused kernel: launch refused") +
                                        (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string()));
    }
    HIPCHK(h, e);
    return GCM_OK;
}

static void swap_state(gcm_handle *h) {
    for (int f = 0; f < GCM_NFIELDS; ++f) std::swap(h->cur[f], h->nxt[f]);
    h->ghosts_current = false;
}

static void tick(gcm_handle *h, hipStream_t s) {
    if (h->timing && h->ev_used < h->ev.size()) (void)hipEventRecord(h->ev[h->ev_used++], s);
}

// predictor (stage 0) or corrector (stage 1) of the staged variant over rows [j0, j1)
static void staged_stage(gcm_handle *h, int stage, double dt, int j0, int j1, hipStream_t s) {
    if (h->cfg.model == GCM_PE2D) {
        launch_pe2d_stage(h->cur, stage == 0 ? h->cur : h->star, stage == 0 ? h->star : h->nxt,
                          h->exner_tab, h->W, h->H, dt, h->cfg.dx, s);
        return;
    }
    const bool temp = h->cfg.model == GCM_SW2D_TEMP;
    double *const *S = stage == 0 ? h->cur : h->star;
    double *const *O = stage == 0 ? h->star : h->nxt;
    Sw2dArgs a = base_args(h, dt);
    a.su = S[GCM_U];
    a.sv = S[GCM_V];
    a.sp = S[GCM_P];
    a.st = S[GCM_T];
    a.ou = O[GCM_U];
    a.ov = O[GCM_V];
    a.op = O[GCM_P];
    a.ot = O[GCM_T];
    a.j0 = j0;
    a.j1 = j1;
    if (temp) {
        Sw2dArgs d = a;
        d.dgeo = h->geo;
        d.dirho = h->irho;
        d.dst = h->sst;
        if (!h->wrap) {  // stencil reaches one row beyond the rows produced
            d.j0 = j0 - 1;
            d.j1 = j1 + 1;
        }
        launch_sw2d_derive(d, s);
        a.sgeo = h->geo;
        a.sirho = h->irho;
        a.sst = h->sst;
    }
    launch_sw2d_stage(a, temp, s);
}

static void staged_tracer(gcm_handle *h, double dt, int j0, int j1, hipStream_t s) {
    if (!h->has[GCM_Q] || h->cfg.model == GCM_PE2D) return;
    const bool lim = h->cfg.tracer == GCM_TRACER_VANLEER;
    Sw2dArgs a = base_args(h, dt);
    a.j0 = j0;
    a.j1 = j1;
    launch_tracer_axis(a, 0, lim, h->cur[GCM_Q], h->qtmp, s);
    launch_tracer_axis(a, 1, lim, h->qtmp, h->nxt[GCM_Q], s);
}

// one full Matsuno step producing rows [j0, j1) of nxt from cur (ghost rows already valid)
static void step_rows(gcm_handle *h, double dt, int j0, int j1, hipStream_t s) {
    if (j1 <= j0) return;
    const bool temp = h->cfg.model == GCM_SW2D_TEMP;
    if (h->variant == GCM_VARIANT_FUSED) {
        Sw2dArgs a = base_args(h, dt);
        a.ou = h->nxt[GCM_U];
        a.ov = h->nxt[GCM_V];
        a.op = h->nxt[GCM_P];
        a.ot = h->nxt[GCM_T];
        a.oq = h->nxt[GCM_Q];
        a.j0 = j0;
        a.j1 = j1;
        tick(h, s);
        if (!launch_sw2d_fused(a, temp, h->has[GCM_Q] ? h->cfg.tracer : 0, s)) h->launch_refused = true;
        tick(h, s);
    } else {
        // the predicted state is needed one row beyond the rows produced
        const int e = h->wrap ? 0 : 1;
        staged_stage(h, 0, dt, j0 - e, j1 + e, s);
        tick(h, s);
        staged_stage(h, 1, dt, j0, j1, s);
        tick(h, s);
        staged_tracer(h, dt, j0, j1, s);
    }
}

int gcm_step(gcm_handle *h, int nsteps, double dt) {
    if (!h || nsteps < 0) return GCM_ERR_ARG;
    if (h->cfg.device >= 0) HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->pe) {
        int rc = physics_tables(h);
        if (rc) return rc;
        for (int n = 0; n < nsteps; ++n) {
            if ((rc = pe25d_step(h->pe, dt, h->stream, &h->err))) return rc;
            if (h->phys_on) {
                // no_limits_2_5d.py:229-234 with the physics below full_timestep's early return (:96): the dynamics
                // step, then solar_timestep at the current utc, then utc += dt
                if ((rc = pe25d_solar_rows(h->pe, -1, 0, h->H, 0, 0, false, dt, h->phys.utc, h->phys.albedo, h->stream, &h->err)))
                    return rc;
                h->phys.utc += dt;
            }
        }
        return GCM_OK;
    }
    if (!h->wrap && h->since_exchange + nsteps > h->G / kGhost)
        return fail(h, GCM_ERR_STATE,
                    "gcm_step: a latitude band needs a ghost-row exchange every halo_steps steps");
    int n0 = 0;
    if (h->cfg.model == GCM_SW2D && h->wrap && h->variant == GCM_VARIANT_FUSED && !h->timing &&
        !(getenv("GCM_SW2D_TWO_STEP") && getenv("GCM_SW2D_TWO_STEP")[0] == '0')) {
        // small grids: pairs of steps in one launch (sw2d_fused2_kernel)
        while (nsteps - n0 >= 2) {
            Sw2dArgs a = base_args(h, dt);
            a.ou = h->nxt[GCM_U];
            a.ov = h->nxt[GCM_V];
            a.op = h->nxt[GCM_P];
            a.j0 = 0;
            a.j1 = h->H;
            if (!launch_sw2d_fused2(a, h->stream)) break;
            swap_state(h);
            n0 += 2;
        }
    }
    for (int n = n0; n < nsteps; ++n) {
        // bands: each step consumes two ghost rows per side; the rows still valid shrink towards
        // the interior until the next exchange (communication-avoiding deep halo)
        const int e = h->wrap ? 0 : h->G - kGhost * (h->since_exchange + 1);
        step_rows(h, dt, -e, h->H + e, h->stream);
        swap_state(h);
        if (!h->wrap) ++h->since_exchange;
    }
    h->star_valid = false;
    return launch_status(h);
}

int gcm_step_interior(gcm_handle *h, double dt, void *stream) {
    if (!h) return GCM_ERR_ARG;
    if (h->pe) return pe25d_step_part(h->pe, 0, dt, (hipStream_t)stream, &h->err);
    if (h->wrap) return fail(h, GCM_ERR_STATE, "step_interior: handle is not a latitude band");
    if (h->G != kGhost) return fail(h, GCM_ERR_STATE, "step_interior: halo_steps > 1 steps through gcm_step");
    step_rows(h, dt, kGhost, h->H - kGhost, (hipStream_t)stream);
    return launch_status(h);
}

int gcm_step_boundary(gcm_handle *h, double dt, void *stream) {
    if (!h) return GCM_ERR_ARG;
    if (h->pe) return pe25d_step_part(h->pe, 1, dt, (hipStream_t)stream, &h->err);
    if (h->wrap) return fail(h, GCM_ERR_STATE, "step_boundary: handle is not a latitude band");
    hipStream_t s = (hipStream_t)stream;
    if (h->H <= 2 * kGhost) {
        step_rows(h, dt, 0, h->H, s);
    } else {
        step_rows(h, dt, 0, kGhost, s);
        step_rows(h, dt, h->H - kGhost, h->H, s);
    }
    swap_state(h);
    h->star_valid = false;
    return launch_status(h);
}

int gcm_step_phase(gcm_handle *h, int phase, double dt, void *stream) {
    if (!h) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_step_phase: GCM_PE25D latitude bands only");
    return pe25d_step_phase(h->pe, phase, dt, (hipStream_t)stream, &h->err);
}

int gcm_comm_stream(gcm_handle *h, void **stream) {
    if (!h || !stream) return GCM_ERR_ARG;
    if (!h->comm) {
        if (h->cfg.device >= 0) HIPCHK(h, hipSetDevice(h->cfg.device));
        h->comm = concurrent_stream(h->stream, h->pe ? pe25d_aux_stream(h->pe) : nullptr);
        if (!h->comm) return fail(h, GCM_ERR_HIP, "gcm_comm_stream: stream creation failed");
    }
    *stream = (void *)h->comm;
    return GCM_OK;
}

int gcm_set_halo_buffers(gcm_handle *h, void *north_send, void *south_send) {
    if (!h) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_set_halo_buffers: GCM_PE25D latitude bands only");
    return pe25d_set_halo_buffers(h->pe, north_send, south_send, h->stream, &h->err);
}

int gcm_wait_edges(gcm_handle *h, void *stream) {
    if (!h) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_wait_edges: GCM_PE25D latitude bands only");
    return pe25d_wait_edges(h->pe, (hipStream_t)stream, &h->err);
}

int gcm_half_step(gcm_handle *h, int stage, double dt) {
    if (!h || (stage != 0 && stage != 1)) return GCM_ERR_ARG;
    if (h->pe) return pe25d_half(h->pe, stage, dt, h->stream, &h->err);
    if (!h->wrap) return fail(h, GCM_ERR_UNSUPPORTED, "half_step on a latitude band");
    if (stage == 0) {
        staged_stage(h, 0, dt, 0, h->H, h->stream);
        h->star_valid = true;
    } else {
        if (!h->star_valid) return fail(h, GCM_ERR_STATE, "half_step(1) before half_step(0)");
        staged_stage(h, 1, dt, 0, h->H, h->stream);
        staged_tracer(h, dt, 0, h->H, h->stream);
        swap_state(h);
        h->star_valid = false;
    }
    return launch_status(h);
}

// ------------------------------------------------------------------ ghost rows
size_t gcm_halo_bytes(const gcm_handle *h) {
    if (!h) return 0;
    if (h->pe) return pe25d_halo_bytes(h->pe);
    int nf = 0;
    for (int f = 0; f < GCM_NFIELDS; ++f) nf += h->has[f];
    return (size_t)nf * h->G * h->W * sizeof(double);
}

}  // extern "C"

// side 0: rows [0, G) <-> buffer (pack: they become the north neighbour's south ghost rows;
// unpack: buffer -> ghost rows [-G, 0)); side 1: rows [H-G, H) / ghost rows [H, H+G)
static int halo_segments(gcm_handle *h, bool pack, int side, void *dev_buf, SegCopy *c) {
    if (h->pe) return pe25d_halo_segments(h->pe, pack, side, dev_buf, c, &h->err);
    double *b = (double *)dev_buf;
    const size_t n = (size_t)h->G * h->W;
    for (int f = 0; f < GCM_NFIELDS; ++f) {
        if (!h->has[f]) continue;
        double *edge = side == 0 ? h->cur[f] : h->cur[f] + (size_t)(h->H - h->G) * h->W;
        double *ghost = side == 0 ? h->cur[f] - n : h->cur[f] + (size_t)h->H * h->W;
        c->src[c->nseg] = pack ? edge : b;
        c->dst[c->nseg] = pack ? b : ghost;
        c->n[c->nseg++] = (long)n;
        b += n;
    }
    return GCM_OK;
}

static int halo_run(gcm_handle *h, bool pack, void *north, void *south, void *stream) {
    SegCopy c{};
    int rc = GCM_OK;
    if (north) rc = halo_segments(h, pack, 0, north, &c);
    if (rc == GCM_OK && south) rc = halo_segments(h, pack, 1, south, &c);
    if (rc != GCM_OK) return rc;
    if (!pack && !h->pe) {
        h->since_exchange = 0;
        if (south) h->ghosts_current = true;       // (the total variation reads the south ghost row only)
    }
    // GCM_PE25D: ghost rows filled on any stream but the library's second one (a host-driven exchange, gcm_band_run's
    // first exchange of a run): the next stage's second-stream work must follow THAT, not only the last update kernel
    if (!pack && h->pe && (hipStream_t)stream != pe25d_aux_stream(h->pe)) pe25d_fork_invalidate(h->pe);
    launch_seg_copy(c, (hipStream_t)stream);
    return launch_status(h);
}

extern "C" {

int gcm_halo_pack(gcm_handle *h, int side, void *dev_buf, void *stream) {
    if (!h || !dev_buf || (side != 0 && side != 1)) return GCM_ERR_ARG;
    return halo_run(h, true, side == 0 ? dev_buf : nullptr, side == 1 ? dev_buf : nullptr, stream);
}

int gcm_halo_unpack(gcm_handle *h, int side, const void *dev_buf, void *stream) {
    if (!h || !dev_buf || (side != 0 && side != 1)) return GCM_ERR_ARG;
    return halo_run(h, false, side == 0 ? (void *)dev_buf : nullptr, side == 1 ? (void *)dev_buf : nullptr, stream);
}

// both sides in one launch
int gcm_halo_pack2(gcm_handle *h, void *north_buf, void *south_buf, void *stream) {
    if (!h || !north_buf || !south_buf) return GCM_ERR_ARG;
    return halo_run(h, true, north_buf, south_buf, stream);
}

int gcm_halo_unpack2(gcm_handle *h, const void *north_buf, const void *south_buf, void *stream) {
    if (!h || !north_buf || !south_buf) return GCM_ERR_ARG;
    return halo_run(h, false, (void *)north_buf, (void *)south_buf, stream);
}

// Device-side snapshot of the current state (2-D models): lets a long run restart from a known
// state without a host round trip (bench.py: the SURVEY's noise initial state lives for a few
// hundred steps only).  gcm_restore is asynchronous on the handle's stream.
int gcm_snapshot(gcm_handle *h) {
    if (!h) return GCM_ERR_ARG;
    if (h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_snapshot: 2-D models only");
    const size_t n = (size_t)(h->H + 2 * h->G) * h->W;
    for (int f = 0; f < GCM_NFIELDS; ++f) {
        if (!h->has[f]) continue;
        if (!h->snap[f]) {
            void *d = nullptr;
            HIPCHK(h, hipMalloc(&d, n * sizeof(double)));
            h->allocs.push_back(d);
            h->snap[f] = (double *)d;
        }
        HIPCHK(h, hipMemcpyAsync(h->snap[f], h->cur[f] - (size_t)h->G * h->W, n * sizeof(double),
                                 hipMemcpyDeviceToDevice, h->stream));
    }
    h->snap_since_exchange = h->since_exchange;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return GCM_OK;
}

int gcm_restore(gcm_handle *h) {
    if (!h) return GCM_ERR_ARG;
    if (h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_restore: 2-D models only");
    const size_t n = (size_t)(h->H + 2 * h->G) * h->W;
    for (int f = 0; f < GCM_NFIELDS; ++f) {
        if (!h->has[f]) continue;
        if (!h->snap[f]) return fail(h, GCM_ERR_STATE, "gcm_restore: no snapshot taken");
        HIPCHK(h, hipMemcpyAsync(h->cur[f] - (size_t)h->G * h->W, h->snap[f], n * sizeof(double),
                                 hipMemcpyDeviceToDevice, h->stream));
    }
    h->since_exchange = h->snap_since_exchange;
    h->ghosts_current = false;
    h->star_valid = false;
    h->primed = false;              // gcm_band_run: exchange the restored state's ghost rows first
    return GCM_OK;
}

int gcm_sync(gcm_handle *h) {
    if (!h) return GCM_ERR_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return GCM_OK;
}

int gcm_set_exchange(gcm_handle *h, const gcm_exchange *x) {
    if (!h) return GCM_ERR_ARG;
    if (h->wrap) return fail(h, GCM_ERR_STATE, "gcm_set_exchange: handle is not a latitude band");
    if (!x) {
        h->xch_set = false;
        if (h->pe) return pe25d_set_halo_buffers(h->pe, nullptr, nullptr, h->stream, &h->err);
        return GCM_OK;
    }
    const int nfn = (x->send != nullptr) + (x->recv != nullptr) + (x->group_start != nullptr) + (x->group_end != nullptr);
    if (nfn != 0 && nfn != 4) return fail(h, GCM_ERR_ARG, "gcm_set_exchange: give all four RCCL entry points, or none (loopback)");
    if (nfn == 4 && !x->comm) return fail(h, GCM_ERR_ARG, "gcm_set_exchange: communicator is NULL");
    if (!x->send_north || !x->send_south || !x->recv_north || !x->recv_south)
        return fail(h, GCM_ERR_ARG, "gcm_set_exchange: four device buffers of gcm_halo_bytes() are required");
    if (h->cfg.device >= 0) HIPCHK(h, hipSetDevice(h->cfg.device));
    void *cs = nullptr;
    int rc = gcm_comm_stream(h, &cs);
    if (rc) return rc;
    if (!h->ev_pack) HIPCHK(h, hipEventCreateWithFlags(&h->ev_pack, hipEventDisableTiming));
    if (!h->ev_comm) HIPCHK(h, hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
    h->xch = *x;
    h->xch_set = true;
    h->primed = false;
    if (const char *e = getenv("GCM_BAND_OVERLAP")) h->band_overlap = e[0] == '1';
    if (h->pe) pe25d_set_edges_first(h->pe, h->band_overlap);
    {
        const char *e = getenv("GCM_BAND_COMM_STREAM");     // diagnostic: the exchange on the comm stream, a join per stage (round 1)
        h->on_comm = e && e[0] == '1';
    }
    // GCM_PE25D: the edge rows of a stage are updated and packed into the send buffers on the
    // handle's second stream (gcm_set_halo_buffers)
    if (h->pe) return pe25d_set_halo_buffers(h->pe, x->send_north, x->send_south, h->stream, &h->err);
    return GCM_OK;
}

}  // extern "C"

// the send buffers are packed (or being packed: the caller has made the comm stream wait for that);
// post the ring exchange on the comm stream ...
static int band_post(gcm_handle *h, bool on_compute_stream = false, hipStream_t on = nullptr) {
    const gcm_exchange &x = h->xch;
    const size_t nbytes = gcm_halo_bytes(h);
    hipStream_t cs = on ? on : on_compute_stream ? h->stream : h->comm;
    if (x.send) {
        int rc = x.group_start();
        if (rc == 0) {
            // north edge first, then the south ghost first: with two ranks both neighbours are the
            // same peer and the i-th send must meet the peer's i-th receive
            int r1 = x.send(x.send_north, nbytes, 0 /*ncclChar*/, x.north, x.comm, cs);
            int r2 = r1 ? r1 : x.send(x.send_south, nbytes, 0, x.south, x.comm, cs);
            int r3 = r2 ? r2 : x.recv(x.recv_south, nbytes, 0, x.south, x.comm, cs);
            int r4 = r3 ? r3 : x.recv(x.recv_north, nbytes, 0, x.north, x.comm, cs);
            const int re = x.group_end();                   // always closed, whatever a call returned
            rc = r4 ? r4 : re;
        }
        if (rc != 0) {
            char b[96];
            snprintf(b, sizeof b, "gcm_band_run: RCCL call failed (ncclResult %d)", rc);
            return fail(h, GCM_ERR_HIP, b);
        }
    } else {
        // loopback: what goes north arrives as this band's own south ghost rows and vice versa
        // (GCM_BAND_EXCHANGE_DELAY_US: a stand-in for the transfer time between two devices, which one GPU cannot
        // show -- tools/tools_band_time.py sweeps it to see how much exchange latency an orchestration hides)
        static const double delay_us = getenv("GCM_BAND_EXCHANGE_DELAY_US") ? atof(getenv("GCM_BAND_EXCHANGE_DELAY_US")) : 0.0;
        launch_spin(cs, delay_us);
        HIPCHK(h, hipMemcpyAsync(x.recv_south, x.send_north, nbytes, hipMemcpyDeviceToDevice, cs));
        HIPCHK(h, hipMemcpyAsync(x.recv_north, x.send_south, nbytes, hipMemcpyDeviceToDevice, cs));
    }
    if (!on_compute_stream && !on) HIPCHK(h, hipEventRecord(h->ev_comm, cs));
    return GCM_OK;
}
// ... and the other half: the compute stream waits for the exchange and fills the ghost rows
static int band_finish(gcm_handle *h) {
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_comm, 0));
    return gcm_halo_unpack2(h, h->xch.recv_north, h->xch.recv_south, h->stream);
}
static int band_exchange(gcm_handle *h) {
    const int rc = band_post(h);
    return rc ? rc : band_finish(h);
}

// pack both edges on the compute stream, then exchange (the comm stream waits for the pack only)
static int band_pack_exchange(gcm_handle *h) {
    int rc = gcm_halo_pack2(h, h->xch.send_north, h->xch.send_south, h->stream);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev_pack, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm, h->ev_pack, 0));
    return band_exchange(h);
}

// one GCM_PE25D band step: per Euler stage the edge rows + pack on the library's second stream, the interior
// rows on the compute stream, then the exchange and the unpack behind the pack on that same second stream
// (they overlap the interior rows).  The compute stream carries K2a -> K3 -> K4 of the band's OWN rows and
// never reads a ghost row (pe25d_kernels.hip, half_t), so it does not wait for the exchange: everything that
// reads ghost rows -- the next stage's K1, column sums, edge rows -- is queued on the second stream, behind
// the unpack, in stream order.  gcm_band_run joins the two streams once, when it returns.
// GCM_BAND_COMM_STREAM=1: the exchange on the comm stream and a join per stage, as in round 1.
// With gcm_set_physics the step has a second phase, solar_timestep (no_limits_2_5d.py:66-75), which changes theta and
// the ground temperature in place AFTER the post-corrector exchange has left: the ghost rows are radiated locally
// (column-local kernel, the neighbour's own inputs -- theta and p as the exchange delivered them, the ground
// temperature's ghost rows, the latitude of the global row -- hence the neighbour's own bits), on the second stream
// right behind the unpack and ahead of the ghost rows' column sums and anchors; the band's own rows follow the
// corrector on the compute stream, which by then has waited for the edge rows and their pack.
static int band_step_pe(gcm_handle *h, double dt) {
    int rc = GCM_OK;
    hipStream_t ax = h->on_comm ? nullptr : pe25d_aux_stream(h->pe);
    const int H = h->H;
    for (int stage = 0; stage < 2; ++stage) {
        if ((rc = pe25d_step_phase(h->pe, 2 * stage, dt, h->stream, &h->err, ax != nullptr))) return rc;
        if (ax) {
            if ((rc = band_post(h, false, ax))) return rc;
            if ((rc = gcm_halo_unpack2(h, h->xch.recv_north, h->xch.recv_south, ax))) return rc;
            if (stage == 1 && h->phys_on &&
                (rc = pe25d_solar_rows(h->pe, pe25d_new_state_set(h->pe), -kGhost, 0, H, H + kGhost, true, dt, h->phys.utc,
                                       h->phys.albedo, ax, &h->err)))
                return rc;
            if ((rc = pe25d_prep_ghost_rows(h->pe, &h->err))) return rc;
            h->join_pending = true;
        }
        if ((rc = pe25d_step_phase(h->pe, 2 * stage + 1, dt, h->stream, &h->err, ax != nullptr))) return rc;
        if (!ax) {
            if ((rc = pe25d_wait_edges(h->pe, h->comm, &h->err))) return rc;
            if ((rc = band_exchange(h))) return rc;
        }
    }
    if (h->phys_on) {
        // own rows (and, when the exchange was joined into the compute stream, the ghost rows with them)
        const int g = ax ? 0 : kGhost;
        if ((rc = pe25d_solar_rows(h->pe, -1, -g, H + g, 0, 0, ax != nullptr, dt, h->phys.utc, h->phys.albedo, h->stream, &h->err)))
            return rc;
        h->phys.utc += dt;
    }
    return GCM_OK;
}

extern "C" {

int gcm_set_band_overlap(gcm_handle *h, int on) {
    if (!h) return GCM_ERR_ARG;
    if (h->wrap) return fail(h, GCM_ERR_STATE, "gcm_set_band_overlap: handle is not a latitude band");
    h->band_overlap = on != 0;
    if (h->pe) pe25d_set_edges_first(h->pe, h->band_overlap);
    return GCM_OK;
}

int gcm_band_run(gcm_handle *h, int nsteps, double dt) {
    if (!h || nsteps < 0) return GCM_ERR_ARG;
    if (h->wrap) return fail(h, GCM_ERR_STATE, "gcm_band_run: handle is not a latitude band");
    if (!h->xch_set) return fail(h, GCM_ERR_STATE, "gcm_band_run: no exchange registered (gcm_set_exchange)");
    if (h->cfg.device >= 0) HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = GCM_OK;
    if (h->pe) {
        if ((rc = physics_tables(h))) return rc;
        if (!h->primed) {                                  // ghost rows of the initial state (and of the ground temperature), once
            if ((rc = band_pack_exchange(h))) return rc;
            h->primed = true;
        }
        for (int n = 0; n < nsteps; ++n)
            if ((rc = band_step_pe(h, dt))) return rc;
        if (h->join_pending) {
            // the one join of the run: what follows on the compute stream (the caller's gcm_get_state,
            // diagnostics, the next run) also follows the last unpack on the second stream
            hipStream_t ax = pe25d_aux_stream(h->pe);
            HIPCHK(h, hipEventRecord(h->ev_comm, ax));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_comm, 0));
            h->join_pending = false;
        }
        pe25d_join_third_stream(h->pe, h->stream);
        return GCM_OK;
    }
    const int k = h->G / kGhost;                            // steps per exchange
    const bool overlap = k > 1 && h->band_overlap && h->H > 2 * h->G + 2 * kGhost;
    int done = 0;
    while (done < nsteps) {
        if (k == 1) {
            // one exchange per step, overlapped with the rows that need no ghost data
            if ((rc = gcm_halo_pack2(h, h->xch.send_north, h->xch.send_south, h->stream))) return rc;
            HIPCHK(h, hipEventRecord(h->ev_pack, h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->comm, h->ev_pack, 0));
            if ((rc = gcm_step_interior(h, dt, h->stream))) return rc;
            if ((rc = band_exchange(h))) return rc;
            if ((rc = gcm_step_boundary(h, dt, h->stream))) return rc;
            ++done;
            continue;
        }
        if (!overlap) {
            if (!h->primed || h->since_exchange >= k) {
                // nothing runs beside this exchange, so it goes on the compute stream itself: pack,
                // send/recv group, unpack in stream order (on a second stream the two cross-queue
                // dependencies cost 11 us each -- trace of the N = 8 band -- a quarter of the exchange)
                if ((rc = gcm_halo_pack2(h, h->xch.send_north, h->xch.send_south, h->stream))) return rc;
                if ((rc = band_post(h, true))) return rc;
                if ((rc = gcm_halo_unpack2(h, h->xch.recv_north, h->xch.recv_south, h->stream))) return rc;
                h->primed = true;
            }
            const int n = std::min(k - h->since_exchange, nsteps - done);
            if ((rc = gcm_step(h, n, dt))) return rc;
            done += n;
            continue;
        }
        // Deep halo (an exchange every k steps) with the exchange hidden behind two steps' interior
        // rows.  The LAST step of a window produces the G edge rows of either side first; they are
        // packed and sent while the rest of that step runs.  The FIRST step of the next window starts
        // with the rows that need no ghost data; only then does the compute stream wait for the
        // exchange, fill the ghost rows and produce the rows next to them.  Same kernels on the same
        // rows as the plain sequence: bit-identical.
        hipStream_t st = h->stream;
        const int H = h->H, G = h->G;
        if (!h->primed || (h->since_exchange >= k && !h->xch_inflight)) {
            if ((rc = gcm_halo_pack2(h, h->xch.send_north, h->xch.send_south, st))) return rc;
            HIPCHK(h, hipEventRecord(h->ev_pack, st));
            HIPCHK(h, hipStreamWaitEvent(h->comm, h->ev_pack, 0));
            if ((rc = band_post(h))) return rc;
            h->xch_inflight = true;
            h->primed = true;
        }
        if (h->xch_inflight) {                              // first step of a window, split
            const int e = G - kGhost;
            step_rows(h, dt, kGhost, H - kGhost, st);
            if ((rc = band_finish(h))) return rc;           // (resets since_exchange)
            h->xch_inflight = false;
            step_rows(h, dt, -e, kGhost, st);
            step_rows(h, dt, H - kGhost, H + e, st);
            swap_state(h);
            h->since_exchange = 1;
            ++done;
        }
        while (done < nsteps && h->since_exchange < k - 1) {
            const int e = G - kGhost * (h->since_exchange + 1);
            step_rows(h, dt, -e, H + e, st);
            swap_state(h);
            ++h->since_exchange;
            ++done;
        }
        if (done < nsteps && h->since_exchange == k - 1) {   // last step of the window: no ghost rows left to use
            step_rows(h, dt, 0, G, st);
            step_rows(h, dt, H - G, H, st);
            swap_state(h);                                  // the pack reads the state being produced
            rc = gcm_halo_pack2(h, h->xch.send_north, h->xch.send_south, st);
            swap_state(h);
            if (rc) return rc;
            HIPCHK(h, hipEventRecord(h->ev_pack, st));
            HIPCHK(h, hipStreamWaitEvent(h->comm, h->ev_pack, 0));
            if ((rc = band_post(h))) return rc;
            h->xch_inflight = true;
            step_rows(h, dt, G, H - G, st);
            swap_state(h);
            h->since_exchange = k;
            ++done;
        }
    }
    if (h->xch_inflight) {                                  // nothing is left pending across calls
        if ((rc = band_finish(h))) return rc;
        h->xch_inflight = false;
    }
    h->star_valid = false;
    return launch_status(h);
}

// ------------------------------------------------------------------ diagnostics
}  // extern "C"

template <typename T>
__global__ __launch_bounds__(256) void diag_kernel(const T *x, long n, double *out) {
    // out[4*b + {0,1,2,3}] = max, min, sum, nan-count of this block's grid-stride share
    double mx = -INFINITY, mn = INFINITY, sm = 0.0, nn = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double v = (double)x[i];
        if (v != v) nn += 1.0;
        mx = fmax(mx, v);
        mn = fmin(mn, v);
        sm += v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmax(mx, __shfl_down(mx, o));
        mn = fmin(mn, __shfl_down(mn, o));
        sm += __shfl_down(sm, o);
        nn += __shfl_down(nn, o);
    }
    __shared__ double s[4][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s[w][0] = mx;
        s[w][1] = mn;
        s[w][2] = sm;
        s[w][3] = nn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            s[0][0] = fmax(s[0][0], s[k][0]);
            s[0][1] = fmin(s[0][1], s[k][1]);
            s[0][2] += s[k][2];
            s[0][3] += s[k][3];
        }
        for (int k = 0; k < 4; ++k) out[4 * blockIdx.x + k] = s[0][k];
    }
}

// get_total_variation (constants.py:105-108): sum |x - roll(x, -1, axis)| for an array viewed as
// [n_outer][n_axis][n_inner]; wrap == 0: the slab after the last one (a band's south ghost row) is
// differenced instead of slab 0.  out[4*b + 2] = the block's partial sum, [3] = its NaN count.
template <typename T>
__global__ __launch_bounds__(256) void tv_kernel(const T *x, long n_outer, long n_axis, long n_inner, int wrap, double *out) {
    const long n = n_outer * n_axis * n_inner;
    double sm = 0.0, nn = 0.0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long i = e % n_inner, r = e / n_inner;
        const long a = r % n_axis, o = r / n_axis;
        const long an = (a + 1 == n_axis && wrap) ? 0 : a + 1;
        const double v = (double)x[e], w = (double)x[(o * n_axis + an) * n_inner + i];
        const double d = fabs(v - w);
        if (d != d) nn += 1.0;
        sm += d;
    }
    for (int o = 32; o > 0; o >>= 1) {
        sm += __shfl_down(sm, o);
        nn += __shfl_down(nn, o);
    }
    __shared__ double s[4][2];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[w][0] = sm; s[w][1] = nn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x] = 0.0;
        out[4 * blockIdx.x + 1] = 0.0;
        out[4 * blockIdx.x + 2] = s[0][0] + s[1][0] + s[2][0] + s[3][0];
        out[4 * blockIdx.x + 3] = s[0][1] + s[1][1] + s[2][1] + s[3][1];
    }
}

extern "C" {

int gcm_diag(gcm_handle *h, int kind, double *out) {
    if (!h || !out) return GCM_ERR_ARG;
    const double *x = nullptr;
    long n = (long)h->H * h->W;
    int f;
    const bool tv = kind >= GCM_DIAG_TV_P && kind <= GCM_DIAG_TV_Q;
    switch (kind) {
        case GCM_DIAG_TV_P: case GCM_DIAG_TV_U: case GCM_DIAG_TV_V: case GCM_DIAG_TV_T: case GCM_DIAG_TV_Q:
            f = kind - GCM_DIAG_TV_P;
            if (!h->pe && !h->has[f]) return fail(h, GCM_ERR_ARG, "gcm_diag: the model has no such field");
            // a 2-D band differences its last row against the south ghost row: that row must belong to
            // the CURRENT state (bands exchange before a step, so after a step it is stale)
            if (!h->pe && !h->wrap && !h->ghosts_current)
                return fail(h, GCM_ERR_STATE, "gcm_diag: total variation on a latitude band needs the current state's "
                                              "ghost rows (exchange them first: gcm_halo_pack2 / exchange / gcm_halo_unpack2)");
            break;
        case GCM_DIAG_ANY_NAN: case GCM_DIAG_MAX_U: case GCM_DIAG_MIN_U: f = GCM_U; break;
        case GCM_DIAG_MEAN_P: case GCM_DIAG_SUM_P: f = GCM_P; break;
        case GCM_DIAG_MAX_V: case GCM_DIAG_MIN_V: f = GCM_V; break;
        default: return fail(h, GCM_ERR_ARG, "gcm_diag: unknown kind");
    }
    int f32 = 0;
    const void *xv = nullptr;
    if (h->pe) {
        xv = pe25d_field(h->pe, f, &n, &f32);
    } else {
        x = h->cur[f];
    }
    const int nb = gcm_handle::kDiagBlocks;
    if (tv) {
        long n_outer = 1, n_axis = h->H, n_inner = h->W;
        int wrap = h->wrap ? 1 : 0;
        if (h->pe) pe25d_tv_shape(h->pe, f, &n_outer, &n_axis, &n_inner, &wrap);
        if (f32)
            hipLaunchKernelGGL(tv_kernel<float>, dim3(nb), dim3(256), 0, h->stream, (const float *)xv, n_outer, n_axis,
                               n_inner, wrap, h->diag_dev);
        else
            hipLaunchKernelGGL(tv_kernel<double>, dim3(nb), dim3(256), 0, h->stream, h->pe ? (const double *)xv : x,
                               n_outer, n_axis, n_inner, wrap, h->diag_dev);
    } else if (f32)
        hipLaunchKernelGGL(diag_kernel<float>, dim3(nb), dim3(256), 0, h->stream, (const float *)xv, n, h->diag_dev);
    else
        hipLaunchKernelGGL(diag_kernel<double>, dim3(nb), dim3(256), 0, h->stream,
                           h->pe ? (const double *)xv : x, n, h->diag_dev);
    std::vector<double> part(4 * nb);
    HIPCHK(h, hipMemcpyAsync(part.data(), h->diag_dev, sizeof(double) * 4 * nb,
                             hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double mx = -INFINITY, mn = INFINITY, sm = 0.0, nn = 0.0;
    for (int b = 0; b < nb; ++b) {
        mx = std::fmax(mx, part[4 * b]);
        mn = std::fmin(mn, part[4 * b + 1]);
        sm += part[4 * b + 2];
        nn += part[4 * b + 3];
    }
    switch (kind) {
        case GCM_DIAG_ANY_NAN: *out = nn > 0 ? 1.0 : 0.0; break;
        case GCM_DIAG_MAX_U: case GCM_DIAG_MAX_V: *out = nn > 0 ? NAN : mx; break;
        case GCM_DIAG_MIN_U: case GCM_DIAG_MIN_V: *out = nn > 0 ? NAN : mn; break;
        case GCM_DIAG_MEAN_P: *out = sm / (double)n; break;
        case GCM_DIAG_SUM_P: *out = sm; break;
        default: *out = nn > 0 ? NAN : sm; break;          // total variation
    }
    return GCM_OK;
}

int gcm_energy(gcm_handle *h, const double *area, int area_len, double *out4) {
    if (!h || !area || !out4 || area_len < 1) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_energy: GCM_PE25D only");
    double o9[9];
    int rc = pe25d_stats(h->pe, area, area_len, o9, h->stream, &h->err);
    if (rc == GCM_OK) for (int q = 0; q < 4; ++q) out4[q] = o9[4 + q];
    return rc;
}

// constants.get_total_variation (constants.py:105-108) of ANY host array viewed as [n_axis][n_inner]
// (the roll is along axis 0), and the two reductions of constants.courant_number (:111-112), max and
// mean, for callers that hold no handle.  out3 = {sum |x - roll(x, -1, 0)|, max x, mean x}.
int gcm_array_stats(const double *x, long n_axis, long n_inner, double *out3) {
    if (!x || !out3 || n_axis < 1 || n_inner < 1) return GCM_ERR_ARG;
    if (gcm_device_count() < 1) {
        g_create_error = "gcm_array_stats: no HIP device; no CPU fallback";
        return GCM_ERR_NODEVICE;
    }
    const long n = n_axis * n_inner;
    constexpr int nb = 256;
    double *dx = nullptr, *dp = nullptr;
    std::vector<double> part(8 * nb);
    hipError_t e = hipMalloc((void **)&dx, sizeof(double) * (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void **)&dp, sizeof(double) * 8 * nb);
    if (e == hipSuccess) e = hipMemcpy(dx, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(tv_kernel<double>, dim3(nb), dim3(256), 0, nullptr, dx, 1L, n_axis, n_inner, 1, dp);
        hipLaunchKernelGGL(diag_kernel<double>, dim3(nb), dim3(256), 0, nullptr, dx, n, dp + 4 * nb);
        e = hipMemcpy(part.data(), dp, sizeof(double) * 8 * nb, hipMemcpyDeviceToHost);
    }
    if (dx) (void)hipFree(dx);
    if (dp) (void)hipFree(dp);
    if (e != hipSuccess) {
        g_create_error = std::string("gcm_array_stats: ") + hipGetErrorString(e);
        return GCM_ERR_HIP;
    }
    double tv = 0.0, mx = -INFINITY, sm = 0.0, nn = 0.0;
    for (int b = 0; b < nb; ++b) {
        tv += part[4 * b + 2];
        mx = std::fmax(mx, part[4 * nb + 4 * b]);
        sm += part[4 * nb + 4 * b + 2];
        nn += part[4 * nb + 4 * b + 3];
    }
    // np.max propagates NaN (constants.py:111-112); fmax drops it, so the count decides
    out3[0] = tv; out3[1] = nn > 0 ? NAN : mx; out3[2] = sm / (double)n;
    return GCM_OK;
}

int gcm_stats(gcm_handle *h, const double *area, int area_len, double *out9) {
    if (!h || !area || !out9 || area_len < 1) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_stats: GCM_PE25D only");
    return pe25d_stats(h->pe, area, area_len, out9, h->stream, &h->err);
}

int gcm_set_ground(gcm_handle *h, const double *gt) {
    if (!h || !gt) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_set_ground: GCM_PE25D only");
    h->primed = false;                                     // gcm_band_run: the ghost rows of the ground temperature travel again
    return pe25d_ground(h->pe, true, gt, nullptr, h->stream, &h->err);
}

int gcm_set_physics(gcm_handle *h, const gcm_physics *ph) {
    if (!h) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_set_physics: GCM_PE25D only");
    if (!ph) {
        h->phys_on = false;
        return GCM_OK;
    }
    if (!ph->lat || !ph->lon) return fail(h, GCM_ERR_ARG, "gcm_set_physics: lat and lon tables are required");
    h->phys = *ph;
    h->phys_lat.assign(ph->lat, ph->lat + h->cfg.global_height);
    h->phys_lon.assign(ph->lon, ph->lon + h->W);
    h->phys.lat = h->phys.lon = nullptr;                   // (the copies above are what is used)
    h->phys_on = true;
    return GCM_OK;
}

int gcm_get_utc(gcm_handle *h, double *utc) {
    if (!h || !utc) return GCM_ERR_ARG;
    if (!h->phys_on) return fail(h, GCM_ERR_STATE, "gcm_get_utc: no physics registered (gcm_set_physics)");
    *utc = h->phys.utc;
    return GCM_OK;
}

int gcm_polar_filter(gcm_handle *h, int nlev, const double *in, double *out) {
    if (!h || !in || !out) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_polar_filter: GCM_PE25D only");
    if (h->cfg.device >= 0) HIPCHK(h, hipSetDevice(h->cfg.device));
    return pe25d_filter_field(h->pe, nlev, in, out, h->stream, &h->err);
}

int gcm_get_intermediate(gcm_handle *h, int kind, double *out) {
    if (!h || !out) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_get_intermediate: GCM_PE25D only");
    if (h->cfg.device >= 0) HIPCHK(h, hipSetDevice(h->cfg.device));
    return pe25d_intermediate(h->pe, kind, out, h->stream, &h->err);
}

int gcm_get_ground(gcm_handle *h, double *gt) {
    if (!h || !gt) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_get_ground: GCM_PE25D only");
    return pe25d_ground(h->pe, false, nullptr, gt, h->stream, &h->err);
}

int gcm_grey_radiation(gcm_handle *h, double utc, double t_lw, double t_sw, double albedo,
                       const double *lat, const double *lon, double *dTdt, double *dt_ground) {
    if (!h) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_grey_radiation: GCM_PE25D only");
    return pe25d_radiation(h->pe, false, 0.0, utc, t_lw, t_sw, albedo, lat, lon, dTdt, dt_ground,
                           h->stream, &h->err);
}

int gcm_solar_step(gcm_handle *h, double dt, double utc, double t_lw, double t_sw, double albedo,
                   const double *lat, const double *lon) {
    if (!h) return GCM_ERR_ARG;
    if (!h->pe) return fail(h, GCM_ERR_UNSUPPORTED, "gcm_solar_step: GCM_PE25D only");
    return pe25d_radiation(h->pe, true, dt, utc, t_lw, t_sw, albedo, lat, lon, nullptr, nullptr,
                           h->stream, &h->err);
}

int gcm_time_steps(gcm_handle *h, int nsteps, double dt, double *ms, double *kernel_ms_avg) {
    if (!h || nsteps < 1 || !ms) return GCM_ERR_ARG;
    // the two region events are created once and kept in the handle (freed by gcm_destroy), so no
    // exit of this function leaks them.  NOTE: with kernel_ms_avg != NULL the state advances
    // 2 * nsteps steps (the per-launch pass re-runs the same number of steps).
    while (h->region_ev.size() < 2) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreate(&e));
        h->region_ev.push_back(e);
    }
    const hipEvent_t e0 = h->region_ev[0], e1 = h->region_ev[1];
    HIPCHK(h, hipEventRecord(e0, h->stream));
    int rc = gcm_step(h, nsteps, dt);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    float t = 0;
    HIPCHK(h, hipEventElapsedTime(&t, e0, e1));
    *ms = t;
    if (kernel_ms_avg) {
        // second pass: one event pair around every launch of the dominant kernel
        const size_t need = 2 * (size_t)nsteps;
        while (h->ev.size() < need) {
            hipEvent_t e;
            HIPCHK(h, hipEventCreate(&e));
            h->ev.push_back(e);
        }
        h->ev_used = 0;
        h->timing = true;
        if (h->pe) pe25d_timing(h->pe, &h->ev, &h->ev_used);
        rc = gcm_step(h, nsteps, dt);
        h->timing = false;
        if (h->pe) pe25d_timing(h->pe, nullptr, nullptr);
        if (rc) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        double tot = 0;
        int cnt = 0;
        for (size_t k = 0; k + 1 < h->ev_used; k += 2) {
            float d = 0;
            HIPCHK(h, hipEventElapsedTime(&d, h->ev[k], h->ev[k + 1]));
            tot += d;
            ++cnt;
        }
        *kernel_ms_avg = cnt ? tot / cnt : 0.0;
    }
    return GCM_OK;
}

}  // extern "C"
