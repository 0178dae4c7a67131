// HIP kernels for the 2-D shallow-water family on the doubly periodic C-grid:
//   GCM_SW2D       matsuno_c_grid.matsumo_scheme   (matsuno_c_grid.py:125-142)
//   GCM_SW2D_TEMP  matsumo_temp.matsumo_scheme     (matsumo_temp.py:66-99)
//   tracer         two_d.finite_volume_advection   (two_d.py:198-207) [+ van Leer]
//
// Two variants of the same arithmetic (gcm_math.h):
//   staged  one thread per cell, one launch per Euler stage, neighbours through L1/L2;
//           the predicted ("star") state is materialised in HBM.
//   fused   one wave marches down a 60-column strip keeping a 3-row window of the
//           base state and of the predicted state in registers; i+-1 neighbours come
//           from wave64 DPP shifts; predictor and corrector (and both tracer passes)
//           run in one launch, so every field is read once and written once per step.
#include "sw2d_kernels.h"

#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>

#include "gcm_math.h"

namespace gcm {

__device__ __forceinline__ long row_off(int j, int H, int W, bool wrap) {
    if (wrap) {
        j %= H;
        if (j < 0) j += H;
    }
    return (long)j * W;
}

void build_exner_table(double *tab) {
    const long double kappa = (long double)kKappa, ln2 = logl(2.0L);
    for (int e = -64; e <= 63; ++e)
        tab[e + 64] = (double)expl(kappa * ((long double)e * ln2 - logl((long double)kP0)));
    for (int i = 0; i < 64; ++i) {
        const double rc = (double)(1.0L / (1.0L + ((long double)i + 0.5L) / 64.0L));
        tab[128 + 2 * i] = rc;
        tab[129 + 2 * i] = (double)powl(1.0L / (long double)rc, kappa);
    }
}

// ------------------------------------------------------------------ staged
__global__ __launch_bounds__(256) void sw2d_derive_kernel(Sw2dArgs a) {
    __shared__ double tab[kExnerTabDoubles];
    tab[threadIdx.y * 64 + threadIdx.x] = a.exner_tab[threadIdx.y * 64 + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = a.j0 + blockIdx.y * 4 + threadIdx.y;
    if (i >= a.W || j >= a.j1) return;
    const long o = (long)j * a.W + i;
    Thermo th = thermo(a.sp[o], a.st[o], tab);
    a.dgeo[o] = th.geo;
    a.dirho[o] = th.t_over_p;
    a.dst[o] = th.st;
}

template <bool TEMP>
__global__ __launch_bounds__(256) void sw2d_stage_kernel(Sw2dArgs a) {
    const int W = a.W, H = a.H;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = a.j0 + blockIdx.y * 4 + threadIdx.y;
    if (i >= W || j >= a.j1) return;
    const bool wrap = a.wrap_j;
    const int iw = i == 0 ? W - 1 : i - 1, ie = i == W - 1 ? 0 : i + 1;
    const long rc = row_off(j, H, W, wrap), rn = row_off(j - 1, H, W, wrap),
               rs = row_off(j + 1, H, W, wrap);
    const double uc = a.su[rc + i], uw = a.su[rc + iw], ue = a.su[rc + ie], un = a.su[rn + i],
                 us = a.su[rs + i], usw = a.su[rs + iw];
    const double vc = a.sv[rc + i], vw = a.sv[rc + iw], ve = a.sv[rc + ie], vn = a.sv[rn + i],
                 vs = a.sv[rs + i], vsw = a.sv[rs + iw];
    const double pc = a.sp[rc + i], pw = a.sp[rc + iw], pe = a.sp[rc + ie], pn = a.sp[rn + i],
                 ps = a.sp[rs + i];
    double gc = pc, ge = pe, gs = ps;
    if (TEMP) {
        gc = a.sgeo[rc + i];
        ge = a.sgeo[rc + ie];
        gs = a.sgeo[rs + i];
    }
    double du = adv_vel_u(uc, uw, ue, un, us, vc, vw, vs, vsw, a.h_dx) + geo_grad(ge, gc, a.g_dx);
    double dv = adv_vel_v(vc, vw, ve, vn, vs, uc, un, uw, usw, a.h_dx) + geo_grad(gs, gc, a.g_dx);
    if (TEMP) {
        const double vis = visc_u(uc, uw, ue, un, us, a.mu_dx2) * a.sirho[rc + i];
        du -= vis;
        dv -= vis;  // the v equation uses the viscosity of u, matsumo_temp.py:75,91
    }
    const double dp = adv_geo(uc, uw, vc, vn, pc, pw, pe, pn, ps, a.h_dx);
    const long o = (long)j * W + i;
    const double bp = a.bp[o];
    const double pnew = bp - a.dt * dp;
    a.ou[o] = a.bu[o] - a.dt * du;
    a.ov[o] = a.bv[o] - a.dt * dv;
    a.op[o] = pnew;
    if (TEMP) {
        const double dst = adv_geo(uc, uw, vc, vn, a.sst[rc + i], a.sst[rc + iw], a.sst[rc + ie],
                                   a.sst[rn + i], a.sst[rs + i], a.h_dx);
        const double tt = bp * a.bt[o] - a.dt * dst;
        a.ot[o] = tt * rcp(pnew);  // unscaling, matsumo_temp.py:33-35 (dx*dx cancels)
    }
}

// one axis of the dimension-split tracer step (two_d.py:103-116), axis 0 = j with V[0] = v,
// axis 1 = i with V[1] = u
template <int AXIS, bool LIMIT>
__global__ __launch_bounds__(256) void tracer_axis_kernel(Sw2dArgs a, const double *qin,
                                                           double *qout) {
    const int W = a.W, H = a.H;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = a.j0 + blockIdx.y * 4 + threadIdx.y;
    if (i >= W || j >= a.j1) return;
    const bool wrap = a.wrap_j;
    double f, fm;
    const long rc = row_off(j, H, W, wrap);
    if (AXIS == 0) {
        const long r1 = row_off(j + 1, H, W, wrap), r2 = row_off(j + 2, H, W, wrap),
                   rm = row_off(j - 1, H, W, wrap), rmm = row_off(j - 2, H, W, wrap);
        const double qmm = qin[rmm + i], qm = qin[rm + i], q0 = qin[rc + i], q1 = qin[r1 + i],
                     q2 = qin[r2 + i];
        f = face_flux<LIMIT>(a.bv[rc + i], qm, q0, q1, q2, a.dtdx);
        fm = face_flux<LIMIT>(a.bv[rm + i], qmm, qm, q0, q1, a.dtdx);
        qout[(long)j * W + i] = q0 - f + fm;
    } else {
        auto wi = [W](int x) { x %= W; return x < 0 ? x + W : x; };
        const int i1 = wi(i + 1), i2 = wi(i + 2), im = wi(i - 1), imm = wi(i - 2);
        const double qmm = qin[rc + imm], qm = qin[rc + im], q0 = qin[rc + i], q1 = qin[rc + i1],
                     q2 = qin[rc + i2];
        f = face_flux<LIMIT>(a.bu[rc + i], qm, q0, q1, q2, a.dtdx);
        fm = face_flux<LIMIT>(a.bu[rc + im], qmm, qm, q0, q1, a.dtdx);
        qout[(long)j * W + i] = q0 - f + fm;
    }
}

static dim3 cell_grid(const Sw2dArgs &a) {
    return dim3((a.W + 63) / 64, (a.j1 - a.j0 + 3) / 4);
}

void launch_sw2d_derive(const Sw2dArgs &a, hipStream_t s) {
    if (a.j1 <= a.j0) return;
    hipLaunchKernelGGL(sw2d_derive_kernel, cell_grid(a), dim3(64, 4), 0, s, a);
}

void launch_sw2d_stage(const Sw2dArgs &a, bool temp, hipStream_t s) {
    if (a.j1 <= a.j0) return;
    if (temp)
        hipLaunchKernelGGL(sw2d_stage_kernel<true>, cell_grid(a), dim3(64, 4), 0, s, a);
    else
        hipLaunchKernelGGL(sw2d_stage_kernel<false>, cell_grid(a), dim3(64, 4), 0, s, a);
}

void launch_tracer_axis(const Sw2dArgs &a, int axis, bool limit, const double *q_in,
                        double *q_out, hipStream_t s) {
    if (a.j1 <= a.j0) return;
    dim3 g = cell_grid(a), b(64, 4);
    if (axis == 0) {
        if (limit) hipLaunchKernelGGL((tracer_axis_kernel<0, true>), g, b, 0, s, a, q_in, q_out);
        else hipLaunchKernelGGL((tracer_axis_kernel<0, false>), g, b, 0, s, a, q_in, q_out);
    } else {
        if (limit) hipLaunchKernelGGL((tracer_axis_kernel<1, true>), g, b, 0, s, a, q_in, q_out);
        else hipLaunchKernelGGL((tracer_axis_kernel<1, false>), g, b, 0, s, a, q_in, q_out);
    }
}

// ------------------------------------------------------------------ fused
// One row of a state (base or predicted) as a lane keeps it: own column and, for TEMP,
// the derived fields.
struct Row {
    double u, uw, v, vw, p, st, g, irho;   // uw, vw: west neighbours, shifted once per row
};

template <bool TEMP>
__device__ __forceinline__ void make_row_r(Row &r, double u, double v, double p, double t,
                                           const double *tab, double rcp_p) {
    r.u = u;
    r.v = v;
    r.p = p;
    r.uw = from_west(u);
    r.vw = from_west(v);
    if (TEMP) {
        Thermo th = thermo(p, t, tab, rcp_p);
        r.st = th.st;
        r.g = th.geo;
        r.irho = th.t_over_p;
    } else {
        r.st = 0.0;
        r.g = p;
        r.irho = 0.0;
    }
}

template <bool TEMP>
__device__ __forceinline__ void make_row(Row &r, double u, double v, double p, double t,
                                         const double *tab) {
    make_row_r<TEMP>(r, u, v, p, t, tab, TEMP ? rcp(p) : 0.0);
}

struct Tend {
    double du, dv, dp, dst;
};

// tendencies at the centre row R0 of a 3-row window (north RM, south RP)
template <bool TEMP>
__device__ __forceinline__ Tend tendencies(const Row &RM, const Row &R0, const Row &RP,
                                           double g_dx, double h_dx, double mu_dx2) {
    const double ue = from_east(R0.u), ve = from_east(R0.v);
    const double uw = R0.uw, vw = R0.vw, usw = RP.uw, vsw = RP.vw;
    const double pw = from_west(R0.p), pe = from_east(R0.p);
    const double ge = TEMP ? from_east(R0.g) : pe;
    Tend t;
    t.du = adv_vel_u(R0.u, uw, ue, RM.u, RP.u, R0.v, vw, RP.v, vsw, h_dx) +
           geo_grad(ge, R0.g, g_dx);
    t.dv = adv_vel_v(R0.v, vw, ve, RM.v, RP.v, R0.u, RM.u, uw, usw, h_dx) +
           geo_grad(RP.g, R0.g, g_dx);
    if (TEMP) {
        const double vis = visc_u(R0.u, uw, ue, RM.u, RP.u, mu_dx2) * R0.irho;
        t.du -= vis;
        t.dv -= vis;
    }
    t.dp = adv_geo(R0.u, uw, R0.v, RM.v, R0.p, pw, pe, RM.p, RP.p, h_dx);
    t.dst = 0.0;
    if (TEMP) {
        const double stw = from_west(R0.st), ste = from_east(R0.st);
        t.dst = adv_geo(R0.u, uw, R0.v, RM.v, R0.st, stw, ste, RM.st, RP.st, h_dx);
    }
    return t;
}

struct Raw {
    double u, v, p, t, q;
};

// STREAM: the base state is read with nontemporal loads -- a grid far larger than the caches is read once
// per step (plus halo columns / band-edge rows), and the streaming hint is worth 1.5-2 % there (A/B on C3,
// two boxes); nontemporal STORES cost 2-8 % (a strip's 480-byte rows share their edge lines with the
// neighbouring strips, which only the L2 merges), so the results are stored normally.
template <bool TEMP, int TRACER, bool WRAPJ, bool STREAM = false>
struct FusedCtx {
    const Sw2dArgs &a;
    const double *tab;
    int ci, col, ja, jb;
    bool store_lane;
    double f0_prev;

    __device__ __forceinline__ Raw load(int j) const {
        const long o = row_off(j, a.H, a.W, WRAPJ) + ci;
        Raw r;
        if (STREAM) {
            r.u = __builtin_nontemporal_load(&a.bu[o]);
            r.v = __builtin_nontemporal_load(&a.bv[o]);
            r.p = __builtin_nontemporal_load(&a.bp[o]);
            r.t = TEMP ? __builtin_nontemporal_load(&a.bt[o]) : 0.0;
            r.q = TRACER ? __builtin_nontemporal_load(&a.bq[o]) : 0.0;
        } else {
            r.u = a.bu[o];
            r.v = a.bv[o];
            r.p = a.bp[o];
            r.t = TEMP ? a.bt[o] : 0.0;
            r.q = TRACER ? a.bq[o] : 0.0;
        }
        return r;
    }

    // One row step.  On entry BM/B0/BP hold base rows r-1, r, r+1, SM/S0 the predicted rows
    // r-2, r-1; SN is a dead slot that receives predicted row r.  On exit BM's slot holds
    // base row r+2 (from the prefetched `nxt`) and `nxt` is row r+3: the caller rotates the
    // slot names instead of moving registers.
    template <bool FETCH = true>
    __device__ __forceinline__ void iter(int r, Row &BM, Row &B0, Row &BP, Row &SN, Row &SM,
                                         Row &S0, Raw &nxt, double &qmm, double &qm, double &q0,
                                         double &qp) {
        const double dt = a.dt, g_dx = a.g_dx, h_dx = a.h_dx, mu_dx2 = a.mu_dx2;
        // ---- predictor: predicted row r from base rows r-1, r, r+1
        {
            const Tend t = tendencies<TEMP>(BM, B0, BP, g_dx, h_dx, mu_dx2);
            const double us = B0.u - dt * t.du;
            const double vs = B0.v - dt * t.dv;
            const double ps = B0.p - dt * t.dp;
            double ts = 0.0, rps = 0.0;
            if (TEMP) {
                rps = rcp(ps);            // shared by the unscaling and by 1/rho of the predicted row
                ts = (B0.st - dt * t.dst) * rps;
            }
            make_row_r<TEMP>(SN, us, vs, ps, ts, tab, rps);
        }
        // ---- tracer, axis-0 flux through the face between rows r-1 and r
        double f0_cur = 0.0;
        if (TRACER && r >= ja) f0_cur = face_flux<TRACER == 2>(BM.v, qmm, qm, q0, qp, a.dtdx);
        // ---- corrector: output row r-1 from predicted rows r-2, r-1, r and base row r-1
        if (r >= ja + 1) {
            const Tend t = tendencies<TEMP>(SM, S0, SN, g_dx, h_dx, mu_dx2);
            const double un = BM.u - dt * t.du;
            const double vn = BM.v - dt * t.dv;
            const double pn = BM.p - dt * t.dp;
            double tn = 0.0, qn = 0.0;
            if (TEMP) tn = (BM.st - dt * t.dst) * rcp(pn);
            if (TRACER) {
                const double qs = qm - f0_cur + f0_prev;  // after the axis-0 pass
                const double qs_w = from_west(qs), qs_e = from_east(qs);
                const double qs_ee = from_east(qs_e);
                const double f1 = face_flux<TRACER == 2>(BM.u, qs_w, qs, qs_e, qs_ee, a.dtdx);
                qn = qs - f1 + from_west(f1);
            }
            if (store_lane) {
                const long o = (long)(r - 1) * a.W + col;
                a.ou[o] = un;
                a.ov[o] = vn;
                a.op[o] = pn;
                if (TEMP) a.ot[o] = tn;
                if (TRACER) a.oq[o] = qn;
            }
        }
        // ---- slide south: the oldest base slot takes row r+2, prefetch row r+3
        f0_prev = f0_cur;
        make_row<TEMP>(BM, nxt.u, nxt.v, nxt.p, nxt.t, tab);
        if (TRACER) {
            qmm = qm;
            qm = q0;
            q0 = qp;
            qp = nxt.q;
        }
        if (FETCH && r + 3 <= jb + 1) nxt = load(r + 3);   // !FETCH: the caller has the rows already
    }
};

// Short bands (small grids: a band is 2-4 rows): the rows are all loaded before the first one is
// used, and the iterations are unrolled with the window slots rotated by name.  A wave then waits
// for memory once instead of once per row, which is most of its life on a 720x360 grid.
template <int N, int PRE, class Ctx>
__device__ __forceinline__ void preloaded_iters(Ctx &c, const Raw (&pre)[PRE + 4], Row &A, Row &B, Row &C, Row &X,
                                                Row &Y, Row &Z, Raw &nxt, double &qmm, double &qm, double &q0,
                                                double &qp) {
    if constexpr (N < PRE + 2) {
        const int r = c.ja - 1 + N;
        if (r > c.jb) return;
        c.template iter<false>(r, A, B, C, X, Y, Z, nxt, qmm, qm, q0, qp);
        if constexpr (N < PRE) nxt = pre[N + 4];          // row r + 3
        preloaded_iters<N + 1, PRE>(c, pre, B, C, A, Y, Z, X, nxt, qmm, qm, q0, qp);
    }
}

template <bool TEMP, int TRACER, bool WRAPJ, int PRE = 0, bool STREAM = false>
__global__ __launch_bounds__(64) void sw2d_fused_kernel(Sw2dArgs a) {
    const int W = a.W;
    const int lane = threadIdx.x;
    // Tile = (strip, band).  Workgroups are dealt round-robin over the 8 XCDs (b % 8 names the
    // XCD group), each with its own L2: give every XCD a contiguous run of tiles, so that
    // neighbouring strips -- which share halo columns and the 128-B lines straddling a strip
    // edge that both write -- meet in one L2.  Speed only: any placement is correct.
    const int strips = (W + kStripCols - 1) / kStripCols;
    const int per_xcd = gridDim.x / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int band = tile / strips;
    const int i0 = (tile - band * strips) * kStripCols;
    __shared__ double tab[kExnerTabDoubles];
    if (TEMP) {
        for (int k = 0; k < kExnerTabDoubles / 64; ++k) tab[lane + 64 * k] = a.exner_tab[lane + 64 * k];
        __syncthreads();
    }
    FusedCtx<TEMP, TRACER, WRAPJ, STREAM> c{a, tab};
    c.ja = a.j0 + band * a.rows_per_band;
    c.jb = min(c.ja + a.rows_per_band, a.j1);
    if (c.ja >= c.jb) return;   // also the padding tiles beyond the last band
    c.col = i0 - 2 + lane;
    c.ci = c.col % W;
    if (c.ci < 0) c.ci += W;
    c.store_lane = lane >= 2 && lane < 62 && c.col < W;
    c.f0_prev = 0.0;
    const int ja = c.ja, jb = c.jb;

    Row A, B, C, X, Y, Z;
    if constexpr (PRE > 0) {               // rows_per_band <= PRE
        Raw pre[PRE + 4];
#pragma unroll
        for (int n = 0; n < PRE + 4; ++n) pre[n] = c.load(min(ja - 2 + n, jb + 1));
        make_row<TEMP>(A, pre[0].u, pre[0].v, pre[0].p, pre[0].t, tab);
        make_row<TEMP>(B, pre[1].u, pre[1].v, pre[1].p, pre[1].t, tab);
        make_row<TEMP>(C, pre[2].u, pre[2].v, pre[2].p, pre[2].t, tab);
        double qmm = 0.0, qm = pre[0].q, q0 = pre[1].q, qp = pre[2].q;
        Raw nxt = pre[3];
        X = Y = Z = A;
        preloaded_iters<0, PRE>(c, pre, A, B, C, X, Y, Z, nxt, qmm, qm, q0, qp);
        return;
    }
    Raw x = c.load(ja - 2);
    make_row<TEMP>(A, x.u, x.v, x.p, x.t, tab);
    double qmm = 0.0, qm = x.q;
    x = c.load(ja - 1);
    make_row<TEMP>(B, x.u, x.v, x.p, x.t, tab);
    double q0 = x.q;
    x = c.load(ja);
    make_row<TEMP>(C, x.u, x.v, x.p, x.t, tab);
    double qp = x.q;
    Raw nxt = c.load(ja + 1);
    X = Y = Z = A;  // overwritten before first use

    // rows r = ja-1 .. jb, three per trip so that the window slots rotate by name
    for (int r = ja - 1; r <= jb; r += 3) {
        c.iter(r, A, B, C, X, Y, Z, nxt, qmm, qm, q0, qp);
        if (r + 1 > jb) break;
        c.iter(r + 1, B, C, A, Y, Z, X, nxt, qmm, qm, q0, qp);
        if (r + 2 > jb) break;
        c.iter(r + 2, C, A, B, Z, X, Y, nxt, qmm, qm, q0, qp);
    }
}

// ------------------------------------------------------------------ two steps per launch
// Plain shallow water on a small grid (720x360) is bound by one dependent launch per step (about
// 2.3 of 5.7 us) plus one wait for memory.  This kernel does TWO Matsuno steps per launch: a second
// pipeline of the same row-march consumes the rows of the first as they leave its corrector, three
// rows behind, and only its results go to memory.  Per step the halo grows by two cells each way:
// 56 of the 64 lanes and RPB of the RPB + 4 first-step rows are output.  Rows of the band are all
// loaded up front (compile-time indexed), the iterations are unrolled with both windows rotated by
// name.  Periodic rows only (a single band).
constexpr int kStrip2Cols = 56;

struct Out3 {
    double u, v, p;
};

// predictor for row r from base rows (BM, B0, BP) into SN; corrector for row r - 1 from the
// predicted rows (SM, S0, SN) and base row BM if `corr`
__device__ __forceinline__ Out3 matsuno_row(bool corr, const Row &BM, const Row &B0, const Row &BP, Row &SN,
                                            const Row &SM, const Row &S0, double dt, double g_dx, double h_dx) {
    {
        const Tend t = tendencies<false>(BM, B0, BP, g_dx, h_dx, 0.0);
        make_row<false>(SN, B0.u - dt * t.du, B0.v - dt * t.dv, B0.p - dt * t.dp, 0.0, nullptr);
    }
    Out3 o{0.0, 0.0, 0.0};
    if (corr) {
        const Tend t = tendencies<false>(SM, S0, SN, g_dx, h_dx, 0.0);
        o.u = BM.u - dt * t.du;
        o.v = BM.v - dt * t.dv;
        o.p = BM.p - dt * t.dp;
    }
    return o;
}

struct Fused2Ctx {
    const Sw2dArgs &a;
    int ja, jb, col;
    bool store_lane;
};

// iteration N of RPB + 7: first-step row r1 = ja - 3 + N, second-step row r2 = r1 - 3.
// (A1..Z1) and (A2..Z2) arrive rotated: A* is the slot of the oldest base row.
template <int N, int RPB>
__device__ __forceinline__ void fused2_iters(const Fused2Ctx &c, const Raw (&pre)[RPB + 8], Row &A1, Row &B1, Row &C1,
                                             Row &X1, Row &Y1, Row &Z1, Row &A2, Row &B2, Row &C2, Row &X2, Row &Y2,
                                             Row &Z2) {
    if constexpr (N < RPB + 7) {
        const double dt = c.a.dt, g_dx = c.a.g_dx, h_dx = c.a.h_dx;
        const int r1 = c.ja - 3 + N;
        Out3 o1{0.0, 0.0, 0.0};
        if (r1 <= c.jb + 2) {
            // first step: predicted row r1; its output row r1 - 1 once the window is primed (N >= 2)
            o1 = matsuno_row(N >= 2, A1, B1, C1, X1, Y1, Z1, dt, g_dx, h_dx);
            if constexpr (N + 3 < RPB + 8) make_row<false>(A1, pre[N + 3].u, pre[N + 3].v, pre[N + 3].p, 0.0, nullptr);  // row r1 + 2
        }
        if constexpr (N >= 5) {
            // second step on the first step's rows: predicted row r2, output row r2 - 1 (N >= 7)
            const int r2 = r1 - 3;
            if (r2 <= c.jb) {
                const Out3 o2 = matsuno_row(N >= 7, A2, B2, C2, X2, Y2, Z2, dt, g_dx, h_dx);
                if (N >= 7 && r2 - 1 < c.jb && c.store_lane) {
                    const long o = (long)(r2 - 1) * c.a.W + c.col;
                    c.a.ou[o] = o2.u;
                    c.a.ov[o] = o2.v;
                    c.a.op[o] = o2.p;
                }
            }
        }
        // the first step's row r1 - 1 enters the second window: rows ja-2, ja-1, ja prime it
        // (N = 2, 3, 4), later ones replace its oldest row
        if constexpr (N >= 2) make_row<false>(A2, o1.u, o1.v, o1.p, 0.0, nullptr);
        // rotate: pipeline 1 by one slot; pipeline 2 likewise once it runs or is being primed
        if constexpr (N >= 2)
            fused2_iters<N + 1, RPB>(c, pre, B1, C1, A1, Y1, Z1, X1, B2, C2, A2, Y2, Z2, X2);
        else
            fused2_iters<N + 1, RPB>(c, pre, B1, C1, A1, Y1, Z1, X1, A2, B2, C2, X2, Y2, Z2);
    }
}

template <int RPB>
__global__ __launch_bounds__(64) void sw2d_fused2_kernel(Sw2dArgs a) {
    const int W = a.W, H = a.H;
    const int lane = threadIdx.x;
    const int strips = (W + kStrip2Cols - 1) / kStrip2Cols;
    const int per_xcd = gridDim.x / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int band = tile / strips;
    const int i0 = (tile - band * strips) * kStrip2Cols;
    Fused2Ctx c{a};
    c.ja = a.j0 + band * RPB;
    c.jb = min(c.ja + RPB, a.j1);
    if (c.ja >= c.jb) return;
    c.col = i0 - 4 + lane;
    int ci = c.col % W;
    if (ci < 0) ci += W;
    c.store_lane = lane >= 4 && lane < 60 && c.col < W;
    Raw pre[RPB + 8];                         // rows ja - 4 .. ja + RPB + 3, periodic in j
#pragma unroll
    for (int n = 0; n < RPB + 8; ++n) {
        int j = (c.ja - 4 + n) % H;
        if (j < 0) j += H;
        const long o = (long)j * W + ci;
        pre[n].u = a.bu[o];
        pre[n].v = a.bv[o];
        pre[n].p = a.bp[o];
        pre[n].t = 0.0;
        pre[n].q = 0.0;
    }
    Row A1, B1, C1, X1, Y1, Z1, A2, B2, C2, X2, Y2, Z2;
    make_row<false>(A1, pre[0].u, pre[0].v, pre[0].p, 0.0, nullptr);
    make_row<false>(B1, pre[1].u, pre[1].v, pre[1].p, 0.0, nullptr);
    make_row<false>(C1, pre[2].u, pre[2].v, pre[2].p, 0.0, nullptr);
    X1 = Y1 = Z1 = A2 = B2 = C2 = X2 = Y2 = Z2 = A1;       // overwritten before first use
    fused2_iters<0, RPB>(c, pre, A1, B1, C1, X1, Y1, Z1, A2, B2, C2, X2, Y2, Z2);
}

// two Matsuno steps of GCM_SW2D in one launch; needs wrap_j, rows_per_band in 2..4
bool launch_sw2d_fused2(const Sw2dArgs &a, hipStream_t s) {
    if (!a.wrap_j || a.j0 != 0 || a.j1 != a.H || a.rows_per_band < 2 || a.rows_per_band > 4) return false;
    const int strips = (a.W + kStrip2Cols - 1) / kStrip2Cols;
    const int bands = (a.H + a.rows_per_band - 1) / a.rows_per_band;
    dim3 g((unsigned)(((long)strips * bands + 7) / 8 * 8));
    Sw2dArgs arg = a;
    void *params[] = {&arg};
    const void *fn = a.rows_per_band == 2 ? (const void *)sw2d_fused2_kernel<2>
                   : a.rows_per_band == 3 ? (const void *)sw2d_fused2_kernel<3>
                                          : (const void *)sw2d_fused2_kernel<4>;
    return hipLaunchKernel(fn, g, dim3(64), params, 0, s) == hipSuccess;
}

constexpr int kPreloadRows = 4;     // bands of up to this many rows use the preloading variant (plain SW2D)

template <bool TEMP, int TRACER>
static const void *fused_fn(bool wrap, bool stream) {
    if (stream)
        return wrap ? (const void *)sw2d_fused_kernel<TEMP, TRACER, true, 0, true>
                    : (const void *)sw2d_fused_kernel<TEMP, TRACER, false, 0, true>;
    return wrap ? (const void *)sw2d_fused_kernel<TEMP, TRACER, true>
                : (const void *)sw2d_fused_kernel<TEMP, TRACER, false>;
}

// stream: the rows this launch reads are far more than the caches hold (see FusedCtx)
static const void *fused_kernel_ptr(bool temp, int tracer, bool wrap, int rows_per_band = 1 << 30, bool stream = false) {
    if (!temp) {
        if (rows_per_band <= kPreloadRows)
            return wrap ? (const void *)sw2d_fused_kernel<false, 0, true, kPreloadRows>
                        : (const void *)sw2d_fused_kernel<false, 0, false, kPreloadRows>;
        return fused_fn<false, 0>(wrap, stream);
    }
    if (tracer == 0) return fused_fn<true, 0>(wrap, stream);
    if (tracer == 1) return fused_fn<true, 1>(wrap, stream);
    return fused_fn<true, 2>(wrap, stream);
}

// Rows per wave.  Large grids: one resident round -- as many waves as the chip holds at
// this kernel's register footprint (a second, partly filled round would idle most SIMDs
// at the tail); small grids: short bands so that every SIMD gets a wave.
int sw2d_fused_rows_per_band(int W, int H, bool temp, int tracer, bool wrap) {
    if (const char *e = getenv("GCM_FUSED_ROWS")) {
        int v = atoi(e);
        if (v > 0) return v;
    }
    int waves_per_cu = 12, cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cus = prop.multiProcessorCount;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fused_kernel_ptr(temp, tracer, wrap), 64,
                                                     0) == hipSuccess && nb > 0)
        waves_per_cu = nb;
    const long slots = (long)waves_per_cu * cus;
    const long strips = (W + kStripCols - 1) / kStripCols;
    auto waves = [&](int rpb) { return strips * ((H + rpb - 1) / rpb); };
    if (waves(8) < slots) {  // small grid: aim at one wave per SIMD at least
        long rpb = (long)H * strips / (5L * cus);   // ~1.3 waves per SIMD (measured best on 720x360)
        return (int)(rpb < 2 ? 2 : rpb > 8 ? 8 : rpb);
    }
    long rounds = (waves(64) + slots - 1) / slots;
    long bands = rounds * slots / strips;  // floor: stay within `rounds` full rounds
    if (bands < 1) bands = 1;
    long rpb = (H + bands - 1) / bands;
    return (int)(rpb < 8 ? 8 : rpb);
}

bool launch_sw2d_fused(const Sw2dArgs &a, bool temp, int tracer, hipStream_t s) {
    if (a.j1 <= a.j0) return true;
    const int strips = (a.W + kStripCols - 1) / kStripCols;
    const int bands = (a.j1 - a.j0 + a.rows_per_band - 1) / a.rows_per_band;
    dim3 g((unsigned)(((long)strips * bands + 7) / 8 * 8));  // 1-D, padded to 8 XCD groups
    Sw2dArgs arg = a;
    void *params[] = {&arg};
    // fields x 8 bytes x the rows of this launch, read once: stream it when that is beyond the 256 MB Infinity Cache
    const int nfields = 3 + (temp ? 1 : 0) + (tracer ? 1 : 0);
    const bool stream = (long)a.W * (a.j1 - a.j0) * 8 * nfields > (256L << 20);
    return hipLaunchKernel(fused_kernel_ptr(temp, tracer, a.wrap_j != 0, a.rows_per_band, stream), g, dim3(64), params, 0, s) == hipSuccess;
}

__global__ void copy_rows_kernel(double *dst, const double *src, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

void launch_copy_rows(double *dst, const double *src, int W, int nrows, hipStream_t s) {
    const long n = (long)W * nrows;
    if (n <= 0) return;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, s, dst, src, n);
}

__global__ void seg_copy_kernel(SegCopy c) {
    const int seg = blockIdx.y;
    if (seg >= c.nseg) return;
    const long n = c.n[seg];
    const double *src = c.src[seg];
    double *dst = c.dst[seg];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

void launch_seg_copy(const SegCopy &c, hipStream_t s, hipEvent_t stop) {
    long mx = 0;
    for (int k = 0; k < c.nseg; ++k) mx = c.n[k] > mx ? c.n[k] : mx;
    if (mx <= 0 || c.nseg <= 0) {
        if (stop) (void)hipEventRecord(stop, s);
        return;
    }
    int blocks = (int)((mx + 255) / 256);
    if (blocks > 512) blocks = 512;
    if (stop) hipExtLaunchKernelGGL(seg_copy_kernel, dim3(blocks, c.nseg), dim3(256), 0, s, nullptr, stop, 0, c);
    else hipLaunchKernelGGL(seg_copy_kernel, dim3(blocks, c.nseg), dim3(256), 0, s, c);
}

}  // namespace gcm
