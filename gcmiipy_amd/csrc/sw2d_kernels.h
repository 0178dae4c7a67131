// Kernel argument block + launch prototypes for the 2-D shallow-water family
// (GCM_SW2D, GCM_SW2D_TEMP [+ tracer]).
#pragma once
#include <hip/hip_runtime.h>

namespace gcm {

constexpr int kExnerTabDoubles = 256;  // [0,128): E_e, e = -64..63; then 64 x {rc_i, ck_i}
constexpr int kGhost = 2;        // ghost rows on each side of a latitude band
constexpr int kStripCols = 60;   // output columns per wave in the fused kernel (64 lanes - 2x2 halo)

// Pointers address interior row 0; with wrap_j == 0 rows -2,-1 and H,H+1 are ghost rows.
struct Sw2dArgs {
    const double *bu, *bv, *bp, *bt, *bq;   // base (time n) state
    const double *su, *sv, *sp, *st;        // stage state the tendencies are evaluated on
    const double *sgeo, *sirho, *sst;       // staged TEMP: geo, T/p, p*t of the stage state
    double *ou, *ov, *op, *ot, *oq;         // output
    double *dgeo, *dirho, *dst;             // derive kernel outputs
    const double *exner_tab;                // device copy of the 256-double exner table
    int W, H;                               // columns, rows owned
    int wrap_j;                             // 1: rows wrap modulo H (single band); 0: ghost rows
    int j0, j1;                             // row range [j0, j1) to produce
    int rows_per_band;                      // fused: output rows per wave
    double dt, dx, inv_dx, dx2, inv_dx2;
    double h_dx;                            // 0.5 / dx (exact halving folded in)
    double dtdx;                            // dt / dx
    double g_dx, mu_dx2, inv_dx2_;          // G / dx, mu_air Rd / dx^2, 1 / dx^2
};

// staged variant
void launch_sw2d_stage(const Sw2dArgs &a, bool temp, hipStream_t s);
void launch_sw2d_derive(const Sw2dArgs &a, hipStream_t s);            // p,t -> geo, 1/rho, scaled_t
void launch_tracer_axis(const Sw2dArgs &a, int axis, bool limit, const double *q_in,
                        double *q_out, hipStream_t s);
// fused variant: predictor + corrector (+ both tracer passes) in one launch
bool launch_sw2d_fused(const Sw2dArgs &a, bool temp, int tracer, hipStream_t s);   // false: the launch was refused
int sw2d_fused_rows_per_band(int W, int H, bool temp, int tracer, bool wrap);
// GCM_SW2D, single band, short bands (small grids): TWO steps in one launch; false if not applicable
bool launch_sw2d_fused2(const Sw2dArgs &a, hipStream_t s);

// GCM_PE2D: one Euler stage; base/stage/out are {p,u,v,t,q} interior pointers (wrap only)
void launch_pe2d_stage(const double *const base[5], const double *const stage[5], double *const out[5],
                       const double *exner_tab, int W, int H, double dt, double dx, hipStream_t s);

// fills tab[256] (host) for gcm_math.h's exner(); kappa and P0 as in constants.py:28,31
void build_exner_table(double *tab);

// ghost rows for a single band that is stepped with wrap_j == 0 (tests) and halo pack/unpack
void launch_copy_rows(double *dst, const double *src, int W, int nrows, hipStream_t s);

// up to 5 contiguous segments copied by ONE launch (ghost-row pack / unpack of all fields)
struct SegCopy {          // up to (5 fields + the ground temperature) x 2 sides in one launch
    double *dst[12];
    const double *src[12];
    long n[12];
    int nseg;
};
// stop: an event signalled by the copy kernel's own completion (hipExtLaunchKernelGGL), or null
void launch_seg_copy(const SegCopy &c, hipStream_t s, hipEvent_t stop = nullptr);

}  // namespace gcm
