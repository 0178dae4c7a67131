// Device arithmetic shared by the staged and the fused kernels, so that the two
// variants give bit-identical results (tests compare them at full size).
//
// Expression order follows the reference (file:line at each function).  Allowed
// departures, all far inside the 1e-10 relative tolerance (they move results by
// O(1e-16) relative): division by a loop-invariant scalar is a multiply by its
// host-computed reciprocal; x / y per cell is x * rcp(y) with two Newton steps;
// (P0/p)**kappa is exp(kappa * (log P0 - log p)); hipcc contracts a*b+c to fma.
#pragma once
#include <hip/hip_runtime.h>

#include "sw2d_kernels.h"

namespace gcm {

constexpr double kG = 9.8;                    // constants.py:45
constexpr double kRd = 287.0;                 // constants.py:16
constexpr double kCp = 1004.0;                // constants.py:22
constexpr double kKappa = 287.0 / 1004.0;     // constants.py:28
constexpr double kP0 = 100000.0;              // constants.py:31
constexpr double kMuAir = 18.5 * 1e-6;        // constants.py:51

// 1/x: v_rcp_f64 seed + two Newton-Raphson steps (no denormal/inf fix-ups: the
// operands on this path are pressures, densities and sigma thicknesses).
__device__ __forceinline__ double rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// ---- (p / P0) ** kappa by table + short series --------------------------------------
// p = 2^e m, m in [1,2); i = top 6 mantissa bits; rc_i ~ 1/c_i with c_i the midpoint of
// the i-th sixty-fourth of [1,2); t = m rc_i - 1 exactly (fma), |t| <= 2^-7;
//   (p/P0)^kappa = [2^(kappa e) P0^-kappa] * [(1/rc_i)^kappa] * (1 + t)^kappa
// with the two bracketed factors from a 2 KB table (built in extended precision on the
// host, gcm_build_exner_table) and the last a degree-7 binomial series (truncation
// < 1e-18).  Max relative error vs extended precision 3e-16..5.3e-16 (tests/test_math_cpu.py);
// numpy's pow, which the reference uses, is at 1.5e-16.  Valid for 2^-64 <= p < 2^64;
// outside (or NaN / negative) the result is NaN, as pow gives for a negative base.
constexpr double kBinom1 = kKappa;
constexpr double kBinom2 = kBinom1 * (kKappa - 1) / 2;
constexpr double kBinom3 = kBinom2 * (kKappa - 2) / 3;
constexpr double kBinom4 = kBinom3 * (kKappa - 3) / 4;
constexpr double kBinom5 = kBinom4 * (kKappa - 4) / 5;
constexpr double kBinom6 = kBinom5 * (kKappa - 5) / 6;
constexpr double kBinom7 = kBinom6 * (kKappa - 6) / 7;

__device__ __forceinline__ double exner(double p, const double *tab /* LDS */) {
    const int hi = __double2hiint(p);
    const int e = ((hi >> 20) & 0x7ff) - 1023;
    const int idx = (hi >> 14) & 63;
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(p));
    const int ec = min(max(e, -64), 63);
    const double E = tab[ec + 64];
    const double rc = tab[128 + 2 * idx], ck = tab[129 + 2 * idx];
    const double t = __builtin_fma(m, rc, -1.0);
    // Estrin: 1 + t (b1 + t b2) + t^3 [(b3 + t b4) + t^2 (b5 + t b6) + t^4 b7]
    const double t2 = t * t, t4 = t2 * t2;
    const double q12 = __builtin_fma(t, kBinom2, kBinom1);
    const double q34 = __builtin_fma(t, kBinom4, kBinom3);
    const double q56 = __builtin_fma(t, kBinom6, kBinom5);
    const double hi3 = __builtin_fma(t4, kBinom7, __builtin_fma(t2, q56, q34));
    const double poly = __builtin_fma(t * t2, hi3, __builtin_fma(t, q12, 1.0));
    const double r = E * ck * poly;
    const bool ok = (hi >= 0) && (e >= -64) && (e <= 63);   // hi >= 0: sign bit clear
    return ok ? r : __builtin_nan("");
}

// value of the wave's lane-1 / lane+1 (columns i-1 / i+1): wave64 DPP shifts,
// two v_mov_b32_dpp per double, no LDS.  Lane 0 / 63 read 0 (bound_ctrl; no tied
// `old` operand, so no register copy) -- those lanes are halo and never stored.
__device__ __forceinline__ double from_west(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_east(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ---- 2-D shallow water operators (matsuno_c_grid.py) -------------------------
// names: c=(j,i) w=(j,i-1) e=(j,i+1) n=(j-1,i) s=(j+1,i) sw=(j+1,i-1)

// advection_of_velocity_u, matsuno_c_grid.py:15-51
__device__ __forceinline__ double adv_vel_u(double uc, double uw, double ue, double un, double us,
                                            double vc, double vw, double vs, double vsw,
                                            double h_dx) {
    // the reference's (a + b) / 2 factors are exact, so they are folded into h_dx = 0.5/dx
    double u_ipj = ue + uc, u_imj = uw + uc, v_ijm = vw + vc, v_ijp = vsw + vs;
    double du_ipj = ue - uc, du_imj = uc - uw, du_ijp = us - uc, du_ijm = uc - un;
    return (u_ipj * du_ipj + u_imj * du_imj + v_ijp * du_ijp + v_ijm * du_ijm) * h_dx;
}

// advection_of_velocity_v, matsuno_c_grid.py:54-80
__device__ __forceinline__ double adv_vel_v(double vc, double vw, double ve, double vn, double vs,
                                            double uc, double un, double uw, double usw,
                                            double h_dx) {
    double v_ijp = vs + vc, v_ijm = vn + vc, u_ipj = uc + un, u_imj = uw + usw;
    double dv_ipj = ve - vc, dv_imj = vc - vw, dv_ijp = vs - vc, dv_ijm = vc - vn;
    return (u_ipj * dv_ipj + u_imj * dv_imj + v_ijp * dv_ijp + v_ijm * dv_ijm) * h_dx;
}

// geopotential_gradient_u / _v, matsuno_c_grid.py:97-106: (p[+1] - p) / dx * G
__device__ __forceinline__ double geo_grad(double p_next, double pc, double g_dx) {
    return (p_next - pc) * g_dx;          // g_dx = G / dx folded on the host
}

// advection_of_geopotential, matsuno_c_grid.py:109-118
__device__ __forceinline__ double adv_geo(double uc, double uw, double vc, double vn,
                                          double pc, double pw, double pe, double pn, double ps,
                                          double h_dx) {
    double up_imj = (pw + pc) * uw;
    double up_ipj = (pe + pc) * uc;
    double vp_ijm = (pn + pc) * vn;
    double vp_ijp = (ps + pc) * vc;
    return (up_ipj - up_imj) * h_dx + (vp_ijp - vp_ijm) * h_dx;
}

// finite_laplacian_2d * mu, viscosity.py:12-25
__device__ __forceinline__ double visc_u(double uc, double uw, double ue, double un, double us,
                                         double mu_dx2) {
    double top = us + un + ue + uw - 4.0 * uc;
    return top * mu_dx2;                   // mu_dx2 = mu_air / dx^2 folded on the host
}

// density_from + geopotential_from + scaling, matsumo_temp.py:13-19,28-30,45-47.
// Returns Rd T / p (= Rd / (rho Rd) ... i.e. 1/rho), geo = p/(G rho) and the mass-weighted
// theta p t.  Two constant factors are folded away: the reference's scaling/unscaling by
// dx*dx (matsumo_temp.py:28-35) cancels exactly in  t* = (p t dx^2 - dt adv(p t dx^2)) /
// (p* dx^2)  because adv() is linear, and Rd of 1/rho = Rd T / p is carried by the viscosity
// constant (mu Rd / dx^2).  Both move results by O(1 ulp).
struct Thermo { double t_over_p, geo, st; };
__device__ __forceinline__ Thermo thermo(double p, double t, const double *tab, double rcp_p) {
    double temp = t * exner(p, tab);           // t / (1e5/p)**kappa
    Thermo r;
    r.t_over_p = temp * rcp_p;                 // 1/rho = Rd * (T / p)
    r.geo = temp * (kRd / kG);                 // p / (G rho) = Rd T / G
    r.st = p * t;
    return r;
}
__device__ __forceinline__ Thermo thermo(double p, double t, const double *tab) {
    return thermo(p, t, tab, rcp(p));
}

// ---- tracer face flux (two_d.py:103-116,135-149; flux_limiter.py:10-27) -----------
// flux through the face between cells 0 and +1 along an axis, times dt/dx.
template <bool LIMIT>
__device__ __forceinline__ double face_flux(double vel, double qm1, double q0, double q1,
                                            double q2, double dtdx) {
    const bool pos = vel > 0.0;                       // strict >, as flux_limiter.py:24
    // (q0 max(vel,0) + q1 min(vel,0)) dt/dx: one of the two products is a zero
    const double vd = vel * dtdx;
    const double f_low = (pos ? q0 : q1) * vd;
    if (!LIMIT) return f_low;
    const double f_high = vd * ((q0 + q1) * 0.5);
    const double b = q1 - q0;
    const double num = pos ? q0 - qm1 : q2 - q1;
    // van_leer(r), r = num / b (0 where b == 0, flux_limiter.py:19):
    // (r + |r|) / (1 + |r|) = 2 |num| / (|b| + |num|) if num b > 0 else 0 -- one reciprocal
    const double an = fabs(num), ab = fabs(b);
    // r > 0 <=> num and b are nonzero with equal signs (sign bits compared: no underflow)
    const bool rpos = ((__double2hiint(num) ^ __double2hiint(b)) >= 0) && an > 0.0 && ab > 0.0;
    const double phi = rpos ? (an + an) * rcp(ab + an) : 0.0;
    return f_low + phi * (f_high - f_low);
}

}  // namespace gcm
