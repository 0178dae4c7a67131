// Device-side definitions shared by the translation units of GCM_PE25D (pe25d_kernels.hip = host side
// and column kernels; pe25d_k1_*.hip, pe25d_k3_*.hip, pe25d_k4_*.hip = the filter and update kernels, one
// file per real type -- K4: per real type and group height -- so that they compile in parallel): kernel arguments, index helpers, the small
// arithmetic helpers that several kernels must round identically, and the kernel pickers' declarations.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fft_lds.h"
#include "gcm_math.h"
#include "sw2d_kernels.h"

namespace gcm {

constexpr int kMaxSeg = 4;      // level segments of the update kernel (short bands)
constexpr int kMaxEdgeCols = 96; // K3: columns that are multiples of 64 (W <= 5120 + rounding)
// real-type specific pieces: reciprocal and (p/P0)**kappa.  fp32: v_rcp_f32 is 1 ulp; the Exner function goes
// through the float64 table + series of gcm_math.h and is rounded to float once.  (__powf, which this used
// to be, expands to ~110 VALU instructions -- the fp32 update kernel executed TWICE the vector instructions
// of the fp64 one, profiles/r03 -- and is less accurate than one rounding of the float64 value.)
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exner(float p, const double *tab) { return (float)exner((double)p, tab); }

template <typename T>
struct PeArgsT {
    using T2 = typename Vec2<T>::type;
    // base (time n) and stage state, device layout, pointers at interior row 0
    const T *p, *u, *v, *t, *q;
    const T *sp, *su, *sv, *st, *sq;
    T *op, *ou, *ov, *ot, *oq;
    // intermediates
    T *spu, *phi, *pgfu;              // 3-D (phi: even levels only, see rho_of / phi_up)
    T *pit, *pn;                      // 2-D
    // column sums sum_k dsig[k] u[k], sum_k dsig[k] v[k] of the stage state (own rows; see pe_pit2d_kernel)
    // and where K4 leaves those of the state it writes (null: not kept)
    T *scs_u, *scs_v, *ocs_u, *ocs_v;
    T *part;                          // [nseg-1] 2-D slabs: conv summed from the top down to a segment boundary
    // tables (device)
    const T *inv_dxj, *inv_dxh;       // [Hg] reciprocals of geometry.py:136-137
    const T *sig, *dsig, *inv_dsig, *sigb, *sigt;  // [L]
    const T *heightmap;               // [Hg][W] (global rows) or null
    const T *cor_u, *cor_v;           // [Hg] Coriolis factors or null (dynamics.py:82-92)
    const T *smul;                    // [Hg][W/2+1] filter multiplier (low_pass.py:61-72)
    const T2 *tw;                     // [W] exp(-2 pi i n / W)
    const double *exner_tab;               // always float64 (gcm_math.h exner())
    FftPlan plan;                          // generic ping-pong passes (fallback)
    SuperPlan cplan;                       // composite-radix in-place passes
    int W, H, L, Hg, row0;                 // local rows, global rows, first global row
    int wrap;                              // 1: rows wrap modulo H (single band)
    int filter;
    int j0, j1;                            // rows to produce
    int jb0, jb1;                          // second row range of the same launch (K4 edge rows), or empty
    int nseg;                              // K4 marches the column in nseg level segments (1: whole column)
    int cs_rows, geo_j0, geo_j1;           // pe_geopot_kernel: also form the rows' column sums; rows to form phi for
    int spu_j0, spu_j1, pit_j0, pit_j1;    // K1 (looping form): of the launch's rows, those it forms spu / pit for
    long part_stride;                      // elements per slab of `part`
    T dt, inv_dy, ptop;
};

__device__ __forceinline__ int wrapi(int x, int n) {
    x %= n;
    return x < 0 ? x + n : x;
}

struct Idx {
    int W, H, L, wrap;
    __device__ __forceinline__ int jr(int j) const { return wrap ? wrapi(j, H) : j; }
    __device__ __forceinline__ long r3(int j) const { return (long)jr(j) * L * W; }   // row slab
    __device__ __forceinline__ long r2(int j) const { return (long)jr(j) * W; }
};

// The update kernel may march a column in several level segments (short latitude bands: more,
// shorter workgroups).  Segment s covers levels [seg_lo(s), seg_lo(s+1)); the running sum of conv
// from the top that sigma-dot needs (dynamics.py:42) then starts from a partial sum that
// pe_pit_kernel leaves at every segment boundary.  Both kernels accumulate through these two
// functions with explicit fma, so that the partial sums are bit-identical to what an unsplit march
// has at that level and the result does not depend on the number of segments.
__device__ __forceinline__ int seg_lo(int s, int nseg, int L) { return (int)((long)s * L / nseg); }
// acc + ((fx_hi - fx_lo) / dx + (sv_hi jph_hi - sv_lo jph_lo) / dy) dsig; the meridional flux
// products are formed in here: handed over as values, one kernel might fuse them into the
// difference and the other not
template <typename T>
__device__ __forceinline__ T conv_acc(T acc, T fx_hi, T fx_lo, T inv_dx, T sv_hi, T jph_hi, T sv_lo, T jph_lo,
                                      T inv_dy, T dsg) {
    const T dy = fma(sv_hi, jph_hi, -(sv_lo * jph_lo));
    return fma(fma(fx_hi - fx_lo, inv_dx, dy * inv_dy), dsg, acc);
}
template <typename T>
__device__ __forceinline__ T sd_of(T rc, T pit, T sgb) { return fma(-pit, sgb, rc); }
// kmh(q) sd: the flux of advec_sig (dynamics.py:50) through the face between two levels, a rounded
// product (no contraction), so that it can be carried from the level above instead of recomputed
template <typename T>
__device__ __forceinline__ T face_flux_v(T q_a, T q_b, T sd) {
#pragma clang fp contract(off)
    return ((q_a + q_b) * T(0.5)) * sd;
}

// Density and geopotential are NOT kept in HBM level by level.  pe_geopot_kernel stores phi on
// the even levels only (the anchors); the filter kernel K3 and the update kernel K4 rebuild rho on
// every level and phi on the odd levels from the stage theta they read anyway, through the two
// helpers below.  Contraction is off inside them, so that the three kernels round identically:
// phi is then one well-defined field, whichever kernel evaluates it and however K4's level march
// is segmented.
//   rho = tp / (Rd tt), tt = t (tp/P0)**kappa                       dynamics.py:122-126
//   phi[k] = phi[k-1] + Cp kph(t)[k-1] (pk[k-1] - pk[k])            dynamics.py:128-134 (cumsum)
template <typename T>
__device__ __forceinline__ T rho_of(T tp, T t, T ex) {
#pragma clang fp contract(off)
    const T tt = t * ex;
    return tp * rcp(T(kRd) * tt);
}
template <typename T>
__device__ __forceinline__ T stp_of(T t_lo, T t_hi, T ex_lo, T ex_hi) {
#pragma clang fp contract(off)
    return T(kCp) * ((t_lo + t_hi) * T(0.5)) * (ex_lo - ex_hi);
}
template <typename T>
__device__ __forceinline__ T add_rn(T a, T b) {
#pragma clang fp contract(off)
    return a + b;
}
template <typename T>
__device__ __forceinline__ T phi_up(T phi_lo, T t_lo, T t_hi, T ex_lo, T ex_hi) {
#pragma clang fp contract(off)
    const T stp = stp_of(t_lo, t_hi, ex_lo, ex_hi);
    return phi_lo + stp;
}
// value of the wave's lane+1 (column i+1), fp32 flavour of gcm_math.h's from_east
__device__ __forceinline__ float from_east(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}

// ---------------------------------------------------------------- K2b', the 2-D form of pit
// The filter is linear and iph(sp), jph(sp) do not depend on the level, so
//   pit = sum_k dsig[k] conv[k] = d_i( filter(iph(sp) U) ) / dx + d_j( jph(sp) V ) / dy,
//   U = sum_k dsig[k] su[k], V = sum_k dsig[k] sv[k]:
// one filtered ROW per latitude instead of a second pass over the 3-D spu and sv (the sum is
// reassociated: pit moves by a few ulp of its largest term, far inside the 1e-10 of the state).
// K4 leaves U and V of the state it writes (cs_acc per level, k = L-1 .. 0); rows it does not own --
// a band's ghost rows -- are summed by pe_colsum_kernel from the 3-D winds in the same order with the
// same fma, so a band and the single domain see the same bits.
template <typename T>
__device__ __forceinline__ T cs_acc(T acc, T x, T dsg) { return fma(x, dsg, acc); }

// ---------------------------------------------------------------- kernel pickers (defined next to the kernels)
// the filter kernels are instantiated per widest composite radix (12 / 16 / 25), so that a plan
// of small radices (1440 = 10.12.12) is not held to the register budget of a 25-point butterfly;
// 0 = generic ping-pong passes
template <typename T> using FilterKernel = void (*)(PeArgsT<T>);
template <typename T> using FilterLoopKernel = void (*)(PeArgsT<T>, int, int);   // (args, pairs per workgroup, y-block that forms pit or -1)
// plans with their own instantiation (only their passes compiled in): the row lengths of the
// BASELINE configs and the powers of 16
constexpr unsigned kMask1440 = pass_bit(5, 2) | pass_bit(4, 3);                    // 1440, 720, 360, 120 ...
constexpr unsigned kMask2880 = pass_bit(5, 3) | pass_bit(4, 3) | pass_bit(4, 4);   // 2880
constexpr unsigned kMask4096 = pass_bit(4, 4);                                     // 256, 4096
template <typename T> FilterKernel<T> spu_filter_kernel_for(const SuperPlan &P);            // pe25d_k1.h
template <typename T> FilterLoopKernel<T> spu_filter_loop_kernel_for(const SuperPlan &P);   // pe25d_k1.h (null: no looping form)
template <typename T> FilterKernel<T> pgf_filter_kernel_for(const SuperPlan &P);            // pe25d_k3.h
template <typename T> FilterKernel<T> pit2d_kernel_for(const SuperPlan &P);                 // pe25d_k3.h
template <typename T, int R> FilterKernel<T> update_rows_kernel_rt(bool same, bool oddtop);   // pe25d_k4.h, instantiated in pe25d_k4_f*_r*.hip
// oddtop: the march starts on an odd level (whole columns, even L): anchors requested with the even levels only
template <typename T> inline FilterKernel<T> update_rows_kernel_for(int rows_per_group, bool same, bool oddtop = false) {
    return rows_per_group == 7 ? update_rows_kernel_rt<T, 7>(same, oddtop) : update_rows_kernel_rt<T, 3>(same, oddtop);
}  // pe25d_k4.h (R = 3 or 7)
constexpr int kUpdCols = 62;      // row-group update kernel: columns a wave produces (lanes 0 and 63 carry the halo columns)
constexpr int kFftThreads = 256;  // generic filter path; the composite path sizes the workgroup from its plan

}  // namespace gcm
