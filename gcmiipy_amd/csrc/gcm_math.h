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

namespace gcm {

constexpr double kG = 9.8;                    // constants.py:45
constexpr double kRd = 287.0;                 // constants.py:16
constexpr double kCp = 1004.0;                // constants.py:22
constexpr double kKappa = 287.0 / 1004.0;     // constants.py:28
constexpr double kP0 = 100000.0;              // constants.py:31
constexpr double kMuAir = 18.5 * 1e-6;        // constants.py:51
constexpr double kLogP0 = 11.512925464970229; // log(1e5)

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

// (P0 / p) ** kappa   (temperature.py:10,18; matsumo_temp.py:15-16)
__device__ __forceinline__ double exner_inv(double p) { return exp(kKappa * (kLogP0 - log(p))); }
// (p / P0) ** kappa   (dynamics.py:125)
__device__ __forceinline__ double exner(double p) { return exp(kKappa * (log(p) - kLogP0)); }

// value of the wave's lane-1 / lane+1 (columns i-1 / i+1): wave64 DPP shifts,
// two v_mov_b32_dpp per double, no LDS.  Lane 0 / 63 keep their own value
// (those lanes are halo and never stored).
__device__ __forceinline__ double from_west(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_east(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// ---- 2-D shallow water operators (matsuno_c_grid.py) -------------------------
// names: c=(j,i) w=(j,i-1) e=(j,i+1) n=(j-1,i) s=(j+1,i) sw=(j+1,i-1)

// advection_of_velocity_u, matsuno_c_grid.py:15-51
__device__ __forceinline__ double adv_vel_u(double uc, double uw, double ue, double un, double us,
                                            double vc, double vw, double vs, double vsw,
                                            double inv_dx) {
    double u_ipj = (ue + uc) * 0.5;
    double u_imj = (uw + uc) * 0.5;
    double v_ijm = (vw + vc) * 0.5;
    double v_ijp = (vsw + vs) * 0.5;
    double du_ipj = ue - uc, du_imj = uc - uw, du_ijp = us - uc, du_ijm = uc - un;
    return (u_ipj * du_ipj + u_imj * du_imj + v_ijp * du_ijp + v_ijm * du_ijm) * inv_dx;
}

// advection_of_velocity_v, matsuno_c_grid.py:54-80
__device__ __forceinline__ double adv_vel_v(double vc, double vw, double ve, double vn, double vs,
                                            double uc, double un, double uw, double usw,
                                            double inv_dx) {
    double v_ijp = (vs + vc) * 0.5;
    double v_ijm = (vn + vc) * 0.5;
    double u_ipj = (uc + un) * 0.5;
    double u_imj = (uw + usw) * 0.5;
    double dv_ipj = ve - vc, dv_imj = vc - vw, dv_ijp = vs - vc, dv_ijm = vc - vn;
    return (u_ipj * dv_ipj + u_imj * dv_imj + v_ijp * dv_ijp + v_ijm * dv_ijm) * inv_dx;
}

// geopotential_gradient_u / _v, matsuno_c_grid.py:97-106: (p[+1] - p) / dx * G
__device__ __forceinline__ double geo_grad(double p_next, double pc, double inv_dx) {
    return (p_next - pc) * inv_dx * kG;
}

// advection_of_geopotential, matsuno_c_grid.py:109-118
__device__ __forceinline__ double adv_geo(double uc, double uw, double vc, double vn,
                                          double pc, double pw, double pe, double pn, double ps,
                                          double inv_dx) {
    double up_imj = (pw + pc) * 0.5 * uw;
    double up_ipj = (pe + pc) * 0.5 * uc;
    double vp_ijm = (pn + pc) * 0.5 * vn;
    double vp_ijp = (ps + pc) * 0.5 * vc;
    return (up_ipj - up_imj) * inv_dx + (vp_ijp - vp_ijm) * inv_dx;
}

// finite_laplacian_2d * mu, viscosity.py:12-25
__device__ __forceinline__ double visc_u(double uc, double uw, double ue, double un, double us,
                                         double inv_dx2) {
    double top = us + un + ue + uw - 4.0 * uc;
    return kMuAir * (top * inv_dx2);
}

// density_from + geopotential_from + scaling, matsumo_temp.py:13-19,28-30,45-47.
// Returns 1/rho, geo = p/(G rho), scaled_t = p t dx dx.
struct Thermo { double inv_rho, geo, st; };
__device__ __forceinline__ Thermo thermo(double p, double t, double dx2) {
    double temp = t * exner(p);                // t / (1e5/p)**kappa
    Thermo r;
    r.inv_rho = kRd * temp * rcp(p);           // 1 / (p / (Rd T))
    r.geo = temp * (kRd / kG);                 // p / (G rho) = Rd T / G
    r.st = p * t * dx2;
    return r;
}

// ---- tracer face flux (two_d.py:103-116,135-149; flux_limiter.py:10-27) -----------
// flux through the face between cells 0 and +1 along an axis, times dt/dx.
template <bool LIMIT>
__device__ __forceinline__ double face_flux(double vel, double qm1, double q0, double q1,
                                            double q2, double dt, double inv_dx) {
    double a_plus = fmax(vel, 0.0), a_minus = fmin(vel, 0.0);
    double f_low = (q0 * a_plus + q1 * a_minus) * dt * inv_dx;
    if (!LIMIT) return f_low;
    double f_high = vel * ((q0 + q1) * 0.5) * dt * inv_dx;
    double a = q0 - qm1, b = q1 - q0, c = q2 - q1;
    double num = vel > 0.0 ? a : c;            // strict >, as flux_limiter.py:24
    double r = (b != 0.0) ? num / b : 0.0;     // calc_r's zero-denominator rule, :19
    double ar = fabs(r);
    double phi = (r + ar) / (1.0 + ar);        // van_leer, flux_limiter.py:10-11
    return f_low + phi * (f_high - f_low);
}

}  // namespace gcm
