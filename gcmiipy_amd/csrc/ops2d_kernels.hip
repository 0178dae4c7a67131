// Stand-alone 2-D C-grid operators of the reference's two_d.py on velocity stacks V[axis]
// (V[0] acts along array axis 0 = rows, with spatial_change[0]; two_d.py:16-22):
//   gcm_advect2d  upwind_axis / corner_transport_2d (:11-71), fv_advect_axis_upwind /
//                 finite_volume_advection (:103-132,198-207), fv_advect_axis_plain (:135-166),
//                 advect_with_momentum (:277-292), + the van-Leer-limited composition
//   gcm_pgf2d     pgf_c_grid_axis / pgf_c_grid / pgf_templess / pressure_at_edge (:210-274)
// Host pointers in, host pointers out (the reference's call shape); the field stays on the
// device for all `nsteps`.
#include "dev_arena.h"
#include "../../include/gcmcore.h"
#include "gcm_math.h"
#include "sw2d_kernels.h"

#include <string>
#include <vector>

namespace gcm {

struct AdvArgs {
    const double *vel;     // V[axis], [H][W]
    const double *qin;
    double *qout;
    int W, H;
    double dt, dx, dtdx, area, volume;
};

__device__ __forceinline__ long at2(int j, int i, int H, int W) {
    j %= H; if (j < 0) j += H;
    i %= W; if (i < 0) i += W;
    return (long)j * W + i;
}

// SCHEME: 0 upwind_axis (non-conservative), 1 fv upwind, 2 fv plain (centred), 3 van Leer
template <int AXIS, int SCHEME, bool FINITE>
__global__ __launch_bounds__(256) void advect_axis_kernel(AdvArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= a.W || j >= a.H) return;
    const int H = a.H, W = a.W;
    auto Q = [&](int d) { return a.qin[AXIS == 0 ? at2(j + d, i, H, W) : at2(j, i + d, H, W)]; };
    auto VEL = [&](int d) { return a.vel[AXIS == 0 ? at2(j + d, i, H, W) : at2(j, i + d, H, W)]; };
    const double q0 = Q(0), qm = Q(-1), qp = Q(1);
    const double v0 = VEL(0);
    double res;
    if (SCHEME == 0) {                                   // two_d.py:11-55
        const double a_plus = fmax(v0, 0.0), a_minus = fmin(v0, 0.0);
        const double mult = a_plus * (q0 - qm) + a_minus * (qp - q0);
        const double fin = mult * (a.dt / a.dx);
        res = FINITE ? fin : q0 - fin;
    } else if (SCHEME == 2) {                            // two_d.py:135-166
        const double vm = VEL(-1);
        const double flux = v0 * ((q0 + qp) * 0.5) * a.dt * a.area;
        const double fm = vm * ((qm + q0) * 0.5) * a.dt * a.area;
        res = FINITE ? fm - flux : q0 - (flux - fm) / a.volume;
    } else {                                             // two_d.py:103-132 (+ limiter)
        const double vm = VEL(-1);
        const double qmm = SCHEME == 3 ? Q(-2) : 0.0, qpp = SCHEME == 3 ? Q(2) : 0.0;
        const double flux = face_flux<SCHEME == 3>(v0, qm, q0, qp, qpp, a.dtdx);
        const double fm = face_flux<SCHEME == 3>(vm, qmm, qm, q0, qp, a.dtdx);
        res = FINITE ? fm - flux : q0 - flux + fm;
    }
    a.qout[(long)j * W + i] = res;
}

// momentum = V * pressure_at_edge(p): edge average along the velocity's own axis (two_d.py:264-279)
__global__ __launch_bounds__(256) void momentum_kernel(double *m0, double *m1, const double *v0, const double *v1,
                                                      const double *p, int W, int H) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= W || j >= H) return;
    const long o = (long)j * W + i;
    m0[o] = v0[o] * ((p[at2(j + 1, i, H, W)] + p[o]) * 0.5);
    m1[o] = v1[o] * ((p[at2(j, i + 1, H, W)] + p[o]) * 0.5);
}

// kind: 0 pgf_c_grid_axis gradients only, 1 pgf_c_grid (needs t), 2 pgf_templess, 3 pressure_at_edge
// (o0 alone = pressure_at_edge_one_d), 4 gradient (centred), 5 pressure_gradient (needs t), 6 pgf_one_d
__global__ __launch_bounds__(256) void pgf_kernel(double *o0, double *o1, const double *p, const double *t,
                                                 const double *etab, int kind, int W, int H, double dt,
                                                 double dx0, double dx1) {
    __shared__ double tab[kExnerTabDoubles];
    tab[threadIdx.y * 64 + threadIdx.x] = etab[threadIdx.y * 64 + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= W || j >= H) return;
    const long o = (long)j * W + i;
    const double pc = p[o], ps = p[at2(j + 1, i, H, W)], pe = p[at2(j, i + 1, H, W)];
    const double g0 = (ps - pc) / dx0, g1 = (pe - pc) / dx1;           // two_d.py:210-220
    if (kind == 0) {
        o0[o] = g0; o1[o] = g1;
    } else if (kind == 1) {                                           // two_d.py:223-245
        const double true_t = t[o] * exner(pc, tab);                  // t / (P0/p)**kappa
        const double rho = pc / (kRd * true_t);
        o0[o] = g0 / rho * dt; o1[o] = g1 / rho * dt;
    } else if (kind == 2) {                                           // two_d.py:248-261
        const double d0 = ((ps + pc) * 0.5) / (kRd * 273.16), d1 = ((pe + pc) * 0.5) / (kRd * 273.16);
        o0[o] = g0 * dt / d0; o1[o] = g1 * dt / d1;
    } else if (kind == 3) {                                           // two_d.py:264-274
        o0[o] = (ps + pc) * 0.5; o1[o] = (pe + pc) * 0.5;
    } else if (kind == 6) {                                           // pgf_one_d, two_d.py:295-303:
        // the edge density is ALWAYS taken along axis 0 (pressure_at_edge_one_d), whatever `axis`
        const double d0 = ((ps + pc) * 0.5) / (kRd * 273.16);
        o0[o] = g0 * dt / d0; o1[o] = g1 * dt / d0;
    } else {                                                          // centred gradient, two_d.py:74-77
        const double pn = p[at2(j - 1, i, H, W)], pw = p[at2(j, i - 1, H, W)];
        const double c0 = (ps - pn) / (2 * dx0), c1 = (pe - pw) / (2 * dx1);
        if (kind == 4) {
            o0[o] = c0; o1[o] = c1;
        } else {                                                      // pressure_gradient, two_d.py:80-100
            const double true_t = t[o] * exner(pc, tab);
            const double rho = pc / (kRd * true_t);
            o0[o] = c0 / rho * dt; o1[o] = c1 / rho * dt;
        }
    }
}

template <int AXIS, bool FIN>
static void launch_axis(int scheme, const AdvArgs &a, dim3 g, dim3 b) {
    switch (scheme) {
        case 0: hipLaunchKernelGGL((advect_axis_kernel<AXIS, 0, FIN>), g, b, 0, nullptr, a); break;
        case 1: hipLaunchKernelGGL((advect_axis_kernel<AXIS, 1, FIN>), g, b, 0, nullptr, a); break;
        case 2: hipLaunchKernelGGL((advect_axis_kernel<AXIS, 2, FIN>), g, b, 0, nullptr, a); break;
        default: hipLaunchKernelGGL((advect_axis_kernel<AXIS, 3, FIN>), g, b, 0, nullptr, a); break;
    }
}

}  // namespace gcm

using namespace gcm;

namespace {
using DevBuf = gcm::DevScratch;      // operands from the calling thread's grow-only arena (dev_arena.h)
// flux_limiter.py:10-32 on a periodic 1-D array; one thread per cell.  IEEE division and no
// contraction: the results (and so the b != 0 / u > 0 masks they carry) are NumPy's bit for bit.
__global__ __launch_bounds__(256) void flux_limiter_kernel(int kind, int n, const double *q, const double *u,
                                                           double dx, double dt, double *out) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int im = i == 0 ? n - 1 : i - 1, ip = i + 1 == n ? 0 : i + 1;
    if (kind == GCM_FL_VAN_LEER) {
        const double r = q[i], ar = fabs(r);
        out[i] = __ddiv_rn(r + ar, 1.0 + ar);                                  // :10-11
    } else if (kind == GCM_FL_CALC_R) {
        const double a = q[i] - q[im], b = q[ip] - q[i];
        out[i] = b != 0.0 ? __ddiv_rn(a, b) : 0.0;                             // :17-19, where=(b != 0)
    } else {
        const double f = (u[i] > 0.0 ? q[i] : q[ip]) * u[i];                    // :24-25, strict u > 0
        if (kind == GCM_FL_DONOR_FLUX) {
            out[i] = f;
        } else {
            const double fm = (u[im] > 0.0 ? q[im] : q[i]) * u[im];
            out[i] = q[i] + __ddiv_rn((fm - f) * dt, dx);                       // :32
        }
    }
}

// 1-D Matsuno of p, u, theta, q in momentum form on a periodic line (no_limits.py:50-152; BASELINE
// configs[0]): one Euler stage, one thread per cell.  iph(x) = (x + ip(x))/2, imh(x) = (im(x) + x)/2,
// div(q_h) = (q_h - im(q_h))/dx, gradh(q) = (ip(q) - q)/dx (coordinates_1d.py:25-53).  u_n needs
// iph(p_n), so a thread forms p_n of its own cell and of the next.
struct Pe1dArgs {
    const double *p, *u, *t, *q;          // base state
    const double *sp, *su, *st, *sq;      // stage state
    double *po, *uo, *to, *qo;
    const double *exner_tab;
    int n;
    double dt, dx;
};
__global__ __launch_bounds__(256) void pe1d_half_kernel(Pe1dArgs a) {
    __shared__ double tab[kExnerTabDoubles];
    tab[threadIdx.x] = a.exner_tab[threadIdx.x];
    __syncthreads();
    const int n = a.n, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int im = i == 0 ? n - 1 : i - 1, ip = i + 1 == n ? 0 : i + 1, ipp = ip + 1 == n ? 0 : ip + 1;
    const double dt = a.dt, dx = a.dx;
    const double sp_m = a.sp[im], sp_c = a.sp[i], sp_p = a.sp[ip], sp_pp = a.sp[ipp];
    const double su_m = a.su[im], su_c = a.su[i], su_p = a.su[ip];
    const double st_m = a.st[im], st_c = a.st[i], st_p = a.st[ip];
    const double sq_m = a.sq[im], sq_c = a.sq[i], sq_p = a.sq[ip];
    const double spu_m = su_m * ((sp_m + sp_c) / 2), spu_c = su_c * ((sp_c + sp_p) / 2), spu_p = su_p * ((sp_p + sp_pp) / 2);
    // advec_q, no_limits.py:50-62
    const double q_n = a.q[i] - (((((sq_c + sq_p) / 2) * su_c) - (((sq_m + sq_c) / 2) * su_m)) / dx) * dt;
    // advec_p, :73-75
    const double p_n = a.p[i] - ((spu_c - spu_m) / dx) * dt;
    const double p_np = a.p[ip] - ((spu_p - spu_c) / dx) * dt;
    // advec_pu, :78-92
    const double um = (su_m + su_c) / 2, up = (su_c + su_p) / 2;
    const double adv_pu = ((up * up) * ((sp_c + sp_p) / 2) - (um * um) * sp_c) / dx;
    // pgf, :102-115: rho at the half point from to_true_temp(iph(t), iph(p))
    const double pph = (sp_c + sp_p) / 2, tph = (st_c + st_p) / 2;
    const double tt = tph * exner(pph, tab);
    const double rho = pph / (kRd * tt);
    const double pgf = pph / rho * ((sp_p - sp_c) / dx);
    const double pu = a.u[i] * ((a.p[i] + a.p[ip]) / 2);
    const double pu_n = pu - (adv_pu + pgf) * dt;
    const double u_n = pu_n / ((p_n + p_np) / 2);
    // advec_t, :95-97
    const double adv_t = (spu_c * ((st_c + st_p) / 2) - spu_m * ((st_m + st_c) / 2)) / dx;
    const double t_n = a.t[i] - (adv_t / p_n) * dt;
    a.po[i] = p_n;
    a.uo[i] = u_n;
    a.to[i] = t_n;
    a.qo[i] = q_n;
}

// The operators of matsuno_c_grid.py / viscosity.py / matsumo_temp.py / temperature.py one by one
// (the fused step kernels evaluate the same device functions): one thread per cell, periodic in
// both axes as the reference's rolls.  c0 = dx, c1 = mu where the operator has them.
__global__ __launch_bounds__(256) void sw2d_op_kernel(int kind, int W, int H, const double *x0, const double *x1,
                                                     const double *x2, double *out, double c0, double c1,
                                                     const double *etab) {
    __shared__ double tab[kExnerTabDoubles];
    tab[threadIdx.y * 64 + threadIdx.x] = etab[threadIdx.y * 64 + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= W || j >= H) return;
    const long o = (long)j * W + i;
    const long ow = at2(j, i - 1, H, W), oe = at2(j, i + 1, H, W), on = at2(j - 1, i, H, W), os = at2(j + 1, i, H, W),
               osw = at2(j + 1, i - 1, H, W);
    const double dx = c0;
    double r = 0.0;
    switch (kind) {
        case GCM_OP_ADV_U:                                            // matsuno_c_grid.py:15-51
            r = adv_vel_u(x0[o], x0[ow], x0[oe], x0[on], x0[os], x1[o], x1[ow], x1[os], x1[osw], 0.5 / dx);
            break;
        case GCM_OP_ADV_V:                                            // :54-80
            r = adv_vel_v(x1[o], x1[ow], x1[oe], x1[on], x1[os], x0[o], x0[on], x0[ow], x0[osw], 0.5 / dx);
            break;
        case GCM_OP_GEO_GRAD_U: r = geo_grad(x0[oe], x0[o], kG / dx); break;      // :97-100
        case GCM_OP_GEO_GRAD_V: r = geo_grad(x0[os], x0[o], kG / dx); break;      // :103-106
        case GCM_OP_ADV_GEO:                                          // :109-118
            r = adv_geo(x0[o], x0[ow], x1[o], x1[on], x2[o], x2[ow], x2[oe], x2[on], x2[os], 0.5 / dx);
            break;
        case GCM_OP_LAPLACIAN: r = visc_u(x0[o], x0[ow], x0[oe], x0[on], x0[os], 1.0 / (dx * dx)); break;   // viscosity.py:12-19
        case GCM_OP_VISCOSITY: r = visc_u(x0[o], x0[ow], x0[oe], x0[on], x0[os], c1 / (dx * dx)); break;    // :22-25
        case GCM_OP_DENSITY_FROM: r = x0[o] / (kRd * (x1[o] * exner(x0[o], tab))); break;                  // matsumo_temp.py:13-19
        case GCM_OP_GEOPOTENTIAL_FROM: r = x1[o] / (kG * x0[o]); break;                                    // :45-47
        case GCM_OP_TO_TRUE_TEMP: r = x0[o] * exner(x1[o], tab); break;           // temperature.py:7-12
        case GCM_OP_TO_POTENTIAL_TEMP: r = x0[o] / exner(x1[o], tab); break;      // :15-19; matsumo_temp.py:22-25
        case GCM_OP_TO_DENSITY: r = x1[o] / (kRd * x0[o]); break;                 // :22-24
        case GCM_OP_SCALING: r = x0[o] * x1[o] * dx * dx; break;                  // matsumo_temp.py:28-30
        case GCM_OP_UNSCALING: r = x1[o] / (x0[o] * dx * dx); break;              // :33-35
        case GCM_OP_PE2D_ADVEC_P:                                                 // no_limits_2d.py:41-44 (pu, pv)
            r = (x0[o] - x0[ow]) / dx + (x1[o] - x1[on]) / dx;
            break;
        case GCM_OP_PE2D_DUT:
        case GCM_OP_PE2D_DVT: {                                                   // advec_m(p, u, v, dx), :47-76
            const double *p = x0, *u = x1, *v = x2;
            const auto P = [&](int jj, int ii) { return p[at2(jj, ii, H, W)]; };
            const auto U = [&](int jj, int ii) { return u[at2(jj, ii, H, W)]; };
            const auto V = [&](int jj, int ii) { return v[at2(jj, ii, H, W)]; };
            const auto jphp = [&](int jj, int ii) { return (P(jj, ii) + P(jj + 1, ii)) / 2; };
            const auto p_mid = [&](int jj, int ii) { return (jphp(jj, ii) + jphp(jj, ii + 1)) / 2; };          // iph(jph(p))
            if (kind == GCM_OP_PE2D_DUT) {
                const auto puum = [&](int jj, int ii) { const double a = (U(jj, ii) + U(jj, ii - 1)) / 2; return a * a * P(jj, ii); };
                const auto puvm = [&](int jj, int ii) {
                    return ((U(jj, ii) + U(jj - 1, ii)) / 2) * ((V(jj - 1, ii) + V(jj - 1, ii + 1)) / 2) * p_mid(jj - 1, ii);
                };
                r = (puum(j, i) - puum(j, i + 1)) / dx + (puvm(j, i) - puvm(j, i + 1)) / dx;
            } else {
                const auto pvvm = [&](int jj, int ii) { const double a = (V(jj, ii) + V(jj - 1, ii)) / 2; return a * a * P(jj, ii); };
                const auto pvum = [&](int jj, int ii) {
                    return p_mid(jj, ii - 1) * ((V(jj, ii) + V(jj, ii - 1)) / 2) * ((U(jj, ii - 1) + U(jj + 1, ii - 1)) / 2);
                };
                r = (pvvm(j, i) - pvvm(j + 1, i)) / dx + (pvum(j, i) - pvum(j, i + 1)) / dx;
            }
            break;
        }
        case GCM_OP_PE2D_PGF_U:
        case GCM_OP_PE2D_PGF_V: {                                                 // pgf(p, t, dx), :79-92
            const long o2 = kind == GCM_OP_PE2D_PGF_U ? oe : os;
            const double ph = (x0[o] + x0[o2]) / 2;
            const double tt = ((x1[o] + x1[o2]) / 2) * exner(ph, tab);            // to_true_temp(iph(t), iph(p))
            const double rho = ph / (kRd * tt);
            r = ph / rho * ((x0[o2] - x0[o]) / dx);
            break;
        }
        default: break;
    }
    out[o] = r;
}

// The operators of the 1-D model one by one (no_limits.py:50-112) on a periodic line of n cells.
__global__ __launch_bounds__(256) void pe1d_op_kernel(int kind, int n, const double *x0, const double *x1, const double *x2,
                                                     double *out, double dx, const double *etab) {
    __shared__ double tab[kExnerTabDoubles];
    tab[threadIdx.x] = etab[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int im = i == 0 ? n - 1 : i - 1, ip = i + 1 == n ? 0 : i + 1;
    double r = 0.0;
    switch (kind) {
        case GCM_OP1D_ADVEC_Q:                                        // advec_q(u, q, dx), :50-62
            r = ((((x1[i] + x1[ip]) / 2) * x0[i]) - (((x1[im] + x1[i]) / 2) * x0[im])) / dx;
            break;
        case GCM_OP1D_CALC_PU: r = x0[i] * ((x1[i] + x1[ip]) / 2); break;       // calc_pu(u, p), :65-67
        case GCM_OP1D_UN_PU: r = x0[i] / ((x1[i] + x1[ip]) / 2); break;         // un_pu(pu, p), :69-70
        case GCM_OP1D_ADVEC_P: r = (x0[i] - x0[im]) / dx; break;                // advec_p(pu, dx), :73-75
        case GCM_OP1D_ADVEC_PU: {                                     // advec_pu(p, pu, u, dx), :78-92 (pu itself unused)
            const double um = (x2[im] + x2[i]) / 2, up = (x2[i] + x2[ip]) / 2;
            r = ((up * up) * ((x0[i] + x0[ip]) / 2) - (um * um) * x0[i]) / dx;
            break;
        }
        case GCM_OP1D_ADVEC_T:                                        // advec_t(pu, t, dx), :95-97
            r = (x0[i] * ((x1[i] + x1[ip]) / 2) - x0[im] * ((x1[im] + x1[i]) / 2)) / dx;
            break;
        case GCM_OP1D_PGF: {                                          // pgf(p, t, dx), :102-112
            const double pph = (x0[i] + x0[ip]) / 2, tph = (x1[i] + x1[ip]) / 2;
            const double rho = pph / (kRd * (tph * exner(pph, tab)));
            r = pph / rho * ((x0[ip] - x0[i]) / dx);
            break;
        }
        default: break;
    }
    out[i] = r;
}

thread_local std::string g_ops_error;
int ops_fail(int code, const char *m) { g_ops_error = m; return code; }
}  // namespace

extern "C" {

const char *gcm_ops_last_error(void) { return g_ops_error.c_str(); }

int gcm_ops_release_scratch(void) {
    gcm::thread_arena().release();
    return GCM_OK;
}

int gcm_advect2d(int scheme, int axes, int finite, int width, int height, int nsteps, double dt,
                 double dx0, double dx1, const double *V, const double *q_in, double *q_out) {
    if (!V || !q_in || !q_out || width < 1 || height < 1 || nsteps < 0 || !(dx0 > 0) || !(dx1 > 0))
        return ops_fail(GCM_ERR_ARG, "gcm_advect2d: bad argument");
    if (scheme < GCM_ADV_UPWIND || scheme > GCM_ADV_MOMENTUM || axes < 1 || axes > 3)
        return ops_fail(GCM_ERR_ARG, "gcm_advect2d: bad scheme/axes");
    if (finite && (nsteps != 1 || axes == 3))
        return ops_fail(GCM_ERR_ARG, "gcm_advect2d: finite (increment) output is per axis, one step");
    if (gcm_device_count() < 1) return ops_fail(GCM_ERR_NODEVICE, "gcm_advect2d: no HIP device; no CPU fallback");
    const size_t n = (size_t)width * height;
    DevBuf mem;
    double *v0 = mem.get(n, V), *v1 = mem.get(n, V + n);
    double *qa = mem.get(n, q_in), *qb = mem.get(n);
    double *m0 = nullptr, *m1 = nullptr;
    if (scheme == GCM_ADV_MOMENTUM) { m0 = mem.get(n); m1 = mem.get(n); }
    if (!v0 || !v1 || !qa || !qb || (scheme == GCM_ADV_MOMENTUM && (!m0 || !m1)))
        return ops_fail(GCM_ERR_HIP, "gcm_advect2d: device allocation/upload failed");
    const dim3 g((width + 63) / 64, (height + 3) / 4), b(64, 4);
    const int kscheme = scheme == GCM_ADV_MOMENTUM ? 1 : scheme;   // momentum: fv upwind of V*p_edge
    for (int s = 0; s < nsteps; ++s) {
        const double *u0 = v0, *u1 = v1;
        if (scheme == GCM_ADV_MOMENTUM) {
            hipLaunchKernelGGL(momentum_kernel, g, b, 0, nullptr, m0, m1, v0, v1, qa, width, height);
            u0 = m0; u1 = m1;
        }
        for (int axis = 0; axis < 2; ++axis) {
            if (!(axes & (1 << axis))) continue;
            AdvArgs a{axis == 0 ? u0 : u1, qa, qb, width, height, dt, axis == 0 ? dx0 : dx1,
                      dt / (axis == 0 ? dx0 : dx1), 0.0, 1.0 * dx0 * dx1};
            a.area = a.volume / a.dx;
            if (axis == 0) { if (finite) launch_axis<0, true>(kscheme, a, g, b); else launch_axis<0, false>(kscheme, a, g, b); }
            else { if (finite) launch_axis<1, true>(kscheme, a, g, b); else launch_axis<1, false>(kscheme, a, g, b); }
            std::swap(qa, qb);
        }
    }
    if (hipMemcpy(q_out, qa, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return ops_fail(GCM_ERR_HIP, "gcm_advect2d: kernel or copy-back failed");
    return GCM_OK;
}

int gcm_pe1d(int n, int nsteps, int half_only, double dt, double dx, const double *const base[4],
             const double *const stage[4], double *const out[4]) {
    if (n < 1 || nsteps < 0 || !(dx != 0.0) || !base || !out) return ops_fail(GCM_ERR_ARG, "gcm_pe1d: bad argument");
    for (int f = 0; f < 4; ++f)
        if (!base[f] || !out[f] || (half_only && (!stage || !stage[f]))) return ops_fail(GCM_ERR_ARG, "gcm_pe1d: null array");
    if (gcm_device_count() < 1) return ops_fail(GCM_ERR_NODEVICE, "gcm_pe1d: no HIP device; no CPU fallback");
    DevBuf mem;
    double tab[kExnerTabDoubles];
    build_exner_table(tab);
    double *dtab = mem.get(kExnerTabDoubles, tab);
    double *b[4], *s[4], *o[4];
    for (int f = 0; f < 4; ++f) {
        b[f] = mem.get(n, base[f]);
        s[f] = half_only ? mem.get(n, stage[f]) : mem.get(n);
        o[f] = mem.get(n);
        if (!b[f] || !s[f] || !o[f] || !dtab) return ops_fail(GCM_ERR_HIP, "gcm_pe1d: device allocation/upload failed");
    }
    const dim3 g((n + 255) / 256), blk(256);
    auto stage_launch = [&](double *const st[4], double *const dst[4]) {
        Pe1dArgs a{b[0], b[1], b[2], b[3], st[0], st[1], st[2], st[3], dst[0], dst[1], dst[2], dst[3], dtab, n, dt, dx};
        hipLaunchKernelGGL(pe1d_half_kernel, g, blk, 0, nullptr, a);
    };
    if (half_only) {
        stage_launch(s, o);
    } else {
        for (int k = 0; k < nsteps; ++k) {          // matsuno_timestep, no_limits.py:150-152
            stage_launch(b, s);                     // predictor: stage = base
            stage_launch(s, o);                     // corrector
            for (int f = 0; f < 4; ++f) std::swap(b[f], o[f]);      // the new state is the next base
        }
        for (int f = 0; f < 4; ++f) o[f] = b[f];
    }
    for (int f = 0; f < 4; ++f)
        if (hipMemcpy(out[f], o[f], (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            return ops_fail(GCM_ERR_HIP, "gcm_pe1d: kernel or copy-back failed");
    return GCM_OK;
}

int gcm_sw2d_op(int kind, int width, int height, double dx, double mu, const double *x0, const double *x1,
                const double *x2, double *out) {
    static const int nin[] = {2, 2, 1, 1, 3, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 2, 2};
    if (kind < 0 || kind > GCM_OP_PE2D_PGF_V || width < 1 || height < 1 || !out || !x0 || (nin[kind] > 1 && !x1) ||
        (nin[kind] > 2 && !x2))
        return ops_fail(GCM_ERR_ARG, "gcm_sw2d_op: bad argument");
    const bool uses_dx = kind <= GCM_OP_VISCOSITY || kind >= GCM_OP_SCALING;      // (all of the no_limits_2d ones do)
    if (uses_dx && !(dx != 0.0)) return ops_fail(GCM_ERR_ARG, "gcm_sw2d_op: dx must be non-zero");
    if (gcm_device_count() < 1) return ops_fail(GCM_ERR_NODEVICE, "gcm_sw2d_op: no HIP device; no CPU fallback");
    const size_t n = (size_t)width * height;
    DevBuf mem;
    double tab[kExnerTabDoubles];
    build_exner_table(tab);
    double *d0 = mem.get(n, x0), *d1 = nin[kind] > 1 ? mem.get(n, x1) : nullptr, *d2 = nin[kind] > 2 ? mem.get(n, x2) : nullptr,
           *o = mem.get(n), *dtab = mem.get(kExnerTabDoubles, tab);
    if (!d0 || (nin[kind] > 1 && !d1) || (nin[kind] > 2 && !d2) || !o || !dtab)
        return ops_fail(GCM_ERR_HIP, "gcm_sw2d_op: device allocation failed");
    hipLaunchKernelGGL(sw2d_op_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(64, 4), 0, nullptr, kind, width, height,
                       d0, d1, d2, o, dx, mu, dtab);
    if (hipMemcpy(out, o, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return ops_fail(GCM_ERR_HIP, "gcm_sw2d_op: kernel or copy-back failed");
    return GCM_OK;
}

int gcm_pe1d_op(int kind, int n, double dx, const double *x0, const double *x1, const double *x2, double *out) {
    static const int nin[] = {2, 2, 2, 1, 3, 2, 2};
    if (kind < 0 || kind > GCM_OP1D_PGF || n < 1 || !out || !x0 || (nin[kind] > 1 && !x1) || (nin[kind] > 2 && !x2))
        return ops_fail(GCM_ERR_ARG, "gcm_pe1d_op: bad argument");
    if (kind != GCM_OP1D_CALC_PU && kind != GCM_OP1D_UN_PU && !(dx != 0.0))
        return ops_fail(GCM_ERR_ARG, "gcm_pe1d_op: dx must be non-zero");
    if (gcm_device_count() < 1) return ops_fail(GCM_ERR_NODEVICE, "gcm_pe1d_op: no HIP device; no CPU fallback");
    DevBuf mem;
    double tab[kExnerTabDoubles];
    build_exner_table(tab);
    double *d0 = mem.get(n, x0), *d1 = nin[kind] > 1 ? mem.get(n, x1) : nullptr, *d2 = nin[kind] > 2 ? mem.get(n, x2) : nullptr,
           *o = mem.get(n), *dtab = mem.get(kExnerTabDoubles, tab);
    if (!d0 || (nin[kind] > 1 && !d1) || (nin[kind] > 2 && !d2) || !o || !dtab)
        return ops_fail(GCM_ERR_HIP, "gcm_pe1d_op: device allocation failed");
    hipLaunchKernelGGL(pe1d_op_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, kind, n, d0, d1, d2, o, dx, dtab);
    if (hipMemcpy(out, o, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return ops_fail(GCM_ERR_HIP, "gcm_pe1d_op: kernel or copy-back failed");
    return GCM_OK;
}

int gcm_flux_limiter(int kind, int n, const double *q, const double *u, double dx, double dt, double *out) {
    if (!q || !out || n < 1 || kind < GCM_FL_VAN_LEER || kind > GCM_FL_DONOR_ADVECTION)
        return ops_fail(GCM_ERR_ARG, "gcm_flux_limiter: bad argument");
    if (kind >= GCM_FL_DONOR_FLUX && !u) return ops_fail(GCM_ERR_ARG, "gcm_flux_limiter: u is required");
    if (kind == GCM_FL_DONOR_ADVECTION && !(dx != 0.0)) return ops_fail(GCM_ERR_ARG, "gcm_flux_limiter: dx == 0");
    if (gcm_device_count() < 1) return ops_fail(GCM_ERR_NODEVICE, "gcm_flux_limiter: no HIP device; no CPU fallback");
    DevBuf mem;
    double *dq = mem.get(n, q), *du = u ? mem.get(n, u) : nullptr, *o = mem.get(n);
    if (!dq || (u && !du) || !o) return ops_fail(GCM_ERR_HIP, "gcm_flux_limiter: device allocation/upload failed");
    hipLaunchKernelGGL(flux_limiter_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, kind, n, dq, du, dx, dt, o);
    if (hipMemcpy(out, o, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return ops_fail(GCM_ERR_HIP, "gcm_flux_limiter: kernel or copy-back failed");
    return GCM_OK;
}

int gcm_pgf2d(int kind, int width, int height, double dt, double dx0, double dx1, const double *p,
              const double *t, double *out2) {
    if (!p || !out2 || width < 1 || height < 1 || kind < 0 || kind > 6 || ((kind == 1 || kind == 5) && !t))
        return ops_fail(GCM_ERR_ARG, "gcm_pgf2d: bad argument");
    if (gcm_device_count() < 1) return ops_fail(GCM_ERR_NODEVICE, "gcm_pgf2d: no HIP device; no CPU fallback");
    const size_t n = (size_t)width * height;
    DevBuf mem;
    double tab[kExnerTabDoubles];
    build_exner_table(tab);
    double *dp = mem.get(n, p), *dt_ = t ? mem.get(n, t) : nullptr, *o = mem.get(2 * n),
           *dtab = mem.get(kExnerTabDoubles, tab);
    if (!dp || (t && !dt_) || !o || !dtab) return ops_fail(GCM_ERR_HIP, "gcm_pgf2d: device allocation failed");
    hipLaunchKernelGGL(pgf_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(64, 4), 0, nullptr, o, o + n, dp,
                       dt_, dtab, kind, width, height, dt, dx0, dx1);
    if (hipMemcpy(out2, o, 2 * n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return ops_fail(GCM_ERR_HIP, "gcm_pgf2d: kernel or copy-back failed");
    return GCM_OK;
}

}  // extern "C"
