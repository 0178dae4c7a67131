// GCM_PE25D: 2.5-D sigma-level primitive equations, Matsuno on the lat-lon C-grid
// (reference dynamics.py:15-237, low_pass.py:41-78, temperature.py:7-19).
//
// Device layout: 3-D fields are [j][k][i] (i fastest, then the L levels, then the rows), so a
// latitude band and its ghost rows are contiguous slabs; p is [j][i].  The host-facing layout
// stays the reference's [k][j][i]; set/get transpose on the device.
//
// One half_timestep (dynamics.py:183-227) is five launches on two streams:
//   K1  spu_filter  spu = arakawa_1977(su * iph(sp))            one workgroup per (row, level pair)
//   K2b pit         conv, pit, p_n (sigma-dot is rebuilt in K4)  one thread per (j, i) column
//   K2a geopot      rho, phi                                     one thread per (j, i) column
//   K3  pgf_filter  pgfu = arakawa_1977(pgu + phiu)              one workgroup per (row, level pair)
//   K4  update      advec_m_pu, advec_sig, advec_t, un_pu/un_pv  one thread per (j, i) column
// K1 -> K2b and K2a -> K3 are independent chains (two streams), K4 needs both.
// The zonal filter is a complex Stockham FFT in LDS: two levels of one row are packed as
// real and imaginary part (the filter multiplier is real and symmetric in the wavenumber, so
// it acts on both parts independently), multiplied by S[j][n] and transformed back.
#include "pe25d_kernels.h"

#include <hip/hip_ext.h>

#include "pe25d_dev.h"

namespace gcm {

// ---------------------------------------------------------------- K2: column kernels
// K2a pe_geopot_kernel: rho, phi from the stage theta and surface pressure (compute_geopotential);
// K2b pe_pit_kernel: pit = sum_k conv and p_n from the filtered mass flux (aflux).  They are two
// kernels because they sit on two independent chains, K1 -> K2b and K2a -> K3 (see half_t).
// The per-level stp that phi needs after the column sum is parked in LDS, park[k][thread],
// instead of a round trip through HBM.
constexpr int kColThreads = 128;
// sum_k dsig[k] u[k] and sum_k dsig[k] v[k] of one column, k = L-1 .. 0 with cs_acc, as K4 accumulates them.  Eight levels of
// BOTH fields are requested at a time, then added in order: one memory latency per eight levels (a load per iteration
// waits one per level: 20 us for the two ghost rows of a band, on the edge rows' chain; round 4: the two fields together)
template <typename T>
__device__ __forceinline__ void column_sum2(const T *cu, const T *cv, const T *dsig, int L, int W, T *su, T *sv) {
    T au = T(0.0), av = T(0.0);
    int k = L - 1;
    for (; k >= 7; k -= 8) {
        T x[8], y[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) { x[n] = cu[(long)(k - n) * W]; y[n] = cv[(long)(k - n) * W]; }
#pragma unroll
        for (int n = 0; n < 8; ++n) { au = cs_acc(au, x[n], dsig[k - n]); av = cs_acc(av, y[n], dsig[k - n]); }
    }
    for (; k >= 0; --k) { au = cs_acc(au, cu[(long)k * W], dsig[k]); av = cs_acc(av, cv[(long)k * W], dsig[k]); }
    *su = au; *sv = av;
}
// LMAX > 0: L <= LMAX and the per-level stp stay in registers (loops unrolled over LMAX; no LDS
// park, so the occupancy is not limited by it); LMAX == 0: any L, stp parked in LDS
// CS: the launch also forms the column sums of its rows (a band's ghost rows; a template parameter, because the
// mere presence of that code in the kernel cost the plain instantiation 40 % of its speed)
// (Round 4, priced with a timing proxy and not built: this kernel also emitting pgu + phiu of every level, so that K3
// becomes a pure filter -- K2a 74.8 -> 116 us, K3 171 -> the 120 us K1 takes for the same traffic: -1.2 % of the step,
// profiles/r04/ab_k2a_emits_pgu_phiu_proxy.txt.)
template <typename T, int LMAX = 0, bool CS = false>
__global__ __launch_bounds__(kColThreads) void pe_geopot_kernel(PeArgsT<T> a) {
    __shared__ double tab[kExnerTabDoubles];
    extern __shared__ unsigned char park_raw[];
    T *park = (T *)park_raw;                    // [L][kColThreads] (LMAX == 0)
    T stp_reg[LMAX > 0 ? LMAX : 1];
    for (int n = threadIdx.x; n < kExnerTabDoubles; n += kColThreads) tab[n] = a.exner_tab[n];
    __syncthreads();
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W, L = a.L;
    // tiles (row, column block) in contiguous runs of rows per XCD, as pe_update_kernel
    const int iblocks = (W + kColThreads - 1) / kColThreads;
    const int per_xcd = gridDim.x / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int jrel = tile / iblocks;
    const int na = a.j1 - a.j0;
    if (jrel >= na + (a.jb1 - a.jb0)) return;
    const int i = (tile - jrel * iblocks) * kColThreads + threadIdx.x;
    const int j = jrel < na ? a.j0 + jrel : a.jb0 + (jrel - na);
    if (i >= W) return;
    if (CS) {
        // a band's ghost rows in one launch: their column sums too (see pe_colsum_kernel)
        T cs_u, cs_v;
        column_sum2(a.su + ix.r3(j) + i, a.sv + ix.r3(j) + i, a.dsig, L, W, &cs_u, &cs_v);
        a.scs_u[ix.r2(j) + i] = cs_u;
        a.scs_v[ix.r2(j) + i] = cs_v;
    }
    if (j < a.geo_j0 || j >= a.geo_j1) return;
    T *pk = park + threadIdx.x;
    const int jg = wrapi(a.row0 + j, a.Hg);
    const T spc = a.sp[ix.r2(j) + i];
    const long c3 = ix.r3(j);
    // ---- compute_geopotential, dynamics.py:111-143
    const T hmG = a.heightmap ? a.heightmap[(long)jg * W + i] * T(kG) : T(0.0) * T(kG);
    // LMAX > 0: the whole theta column is requested before any of it is used (one HBM latency per
    // column instead of one per level)
    T tcol[LMAX > 0 ? LMAX : 1];
    if (LMAX > 0) {
#pragma unroll
        for (int k = 0; k < LMAX; ++k) tcol[k] = k < L ? a.st[c3 + (long)k * W + i] : T(0.0);
    }
    T t_k = LMAX > 0 ? tcol[0] : a.st[c3 + i];
    T ex_k = exner(spc * a.sig[0] + a.ptop, tab);
    const T t0 = t_k, ex0 = ex_k;                       // level 0, for the k wrap at the top
    T acc = T(0.0);
    constexpr int kUnrollA = LMAX > 0 ? LMAX : 6, kUnrollB = LMAX > 0 ? LMAX : 1;
#pragma unroll kUnrollA
    for (int k = 0; k < (LMAX > 0 ? LMAX : L); ++k) {
        if (LMAX > 0 && k >= L) break;
        const T tp = spc * a.sig[k] + a.ptop;
        T t_n, ex_n;
        if (k + 1 < L) {
            t_n = LMAX > 0 ? tcol[k + 1 < LMAX ? k + 1 : 0] : a.st[c3 + (long)(k + 1) * W + i];
            ex_n = exner(spc * a.sig[k + 1] + a.ptop, tab);
        } else {
            t_n = t0;            // kp() wraps to the bottom layer, coordinates_3d.py:55-56
            ex_n = ex0;
        }
        const T rho = rho_of(tp, t_k, ex_k);            // tp / (Rd t / (P0/tp)**kappa)
        const T s1 = (a.sig[k] * spc * rcp(rho)) * a.dsig[k];
        const T stp = stp_of(t_k, t_n, ex_k, ex_n);
        const T s2 = a.sigt[k] * stp;
        acc += s1 - s2;
        if (LMAX > 0) stp_reg[k] = stp;
        else pk[k * kColThreads] = stp;
        t_k = t_n;
        ex_k = ex_n;
    }
    T run = acc + hmG;                                  // stp_n[0], dynamics.py:132
    a.phi[c3 + i] = run;
#pragma unroll kUnrollB
    for (int k = 1; k < (LMAX > 0 ? LMAX : L); ++k) {        // phi = cumsum(stp_n), stp_n = km(stp)
        if (LMAX > 0 && k >= L) break;
        run = add_rn(run, LMAX > 0 ? stp_reg[k - 1] : pk[(k - 1) * kColThreads]);
        if ((k & 1) == 0) a.phi[c3 + (long)k * W + i] = run;     // anchors: even levels only
    }
}

// aflux, dynamics.py:35-46: pit = sum_k conv (ascending, as np.sum over the outer axis) and
// p_n = p - pit dt (dynamics.py:194).  sigma-dot itself is not materialised: the update kernel
// rebuilds it on the fly from pit.
template <typename T>
__global__ __launch_bounds__(256) void pe_pit_kernel(PeArgsT<T> a) {
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W, L = a.L;
    const int iblocks = (W + 255) / 256;
    const int per_xcd = gridDim.x / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int jrel = tile / iblocks;
    if (jrel >= a.j1 - a.j0) return;
    const int i = (tile - jrel * iblocks) * 256 + threadIdx.x;
    const int j = a.j0 + jrel;
    if (i >= W) return;
    const int iw = i == 0 ? W - 1 : i - 1;
    const int jg = wrapi(a.row0 + j, a.Hg);
    const T inv_dxj = a.inv_dxj[jg], inv_dy = a.inv_dy;
    const T spc = a.sp[ix.r2(j) + i], spn = a.sp[ix.r2(j - 1) + i], sps = a.sp[ix.r2(j + 1) + i];
    const T jph_c = (spc + sps) * T(0.5), jph_n = (spn + spc) * T(0.5);  // jph(sp) at j, j-1
    const long c3 = ix.r3(j), n3 = ix.r3(j - 1);
    T pit = T(0.0);
#pragma unroll 12
    for (int k = 0; k < L; ++k) {
        const long o = c3 + (long)k * W;
        const T spv_c = a.sv[o + i] * jph_c;
        const T spv_n = a.sv[n3 + (long)k * W + i] * jph_n;
        pit += ((a.spu[o + i] - a.spu[o + iw]) * inv_dxj + (spv_c - spv_n) * inv_dy) * a.dsig[k];
    }
    a.pit[ix.r2(j) + i] = pit;
    a.pn[ix.r2(j) + i] = a.p[ix.r2(j) + i] - pit * a.dt;
    if (a.nseg > 1) {
        // top-down partial sums at the segment boundaries, exactly as pe_update_kernel accumulates
        T rc = T(0.0);
        int s = a.nseg - 2;
        int stop = seg_lo(s + 1, a.nseg, L);
#pragma unroll 4
        for (int k = L - 1; k >= 1 && s >= 0; --k) {
            const long o = c3 + (long)k * W;
            rc = conv_acc(rc, a.spu[o + i], a.spu[o + iw], inv_dxj, a.sv[o + i], jph_c, a.sv[n3 + (long)k * W + i], jph_n,
                          inv_dy, a.dsig[k]);
            if (k == stop) {
                a.part[(long)s * a.part_stride + ix.r2(j) + i] = rc;
                --s;
                if (s >= 0) stop = seg_lo(s + 1, a.nseg, L);
            }
        }
    }
}

// The partial sums of conv at the segment boundaries alone (pit itself comes from pe_pit2d_kernel),
// rows [j0, j1) and [jb0, jb1): a band's edge rows, which K4 marches in level segments so that they
// are done -- and on their way to the neighbours -- long before the interior rows (a K4 workgroup
// is a chain of L dependent levels, however few rows it has).  Same accumulation as pe_pit_kernel.
template <typename T>
__global__ __launch_bounds__(256) void pe_part_kernel(PeArgsT<T> a) {
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W, L = a.L;
    const int iblocks = (W + 255) / 256;
    const int jrel = blockIdx.x / iblocks;
    const int i = (blockIdx.x - jrel * iblocks) * 256 + threadIdx.x;
    const int na = a.j1 - a.j0;
    if (i >= W || jrel >= na + (a.jb1 - a.jb0)) return;
    const int j = jrel < na ? a.j0 + jrel : a.jb0 + (jrel - na);
    const int iw = i == 0 ? W - 1 : i - 1;
    const int jg = wrapi(a.row0 + j, a.Hg);
    const T inv_dxj = a.inv_dxj[jg], inv_dy = a.inv_dy;
    const T spc = a.sp[ix.r2(j) + i], spn = a.sp[ix.r2(j - 1) + i], sps = a.sp[ix.r2(j + 1) + i];
    const T jph_c = (spc + sps) * T(0.5), jph_n = (spn + spc) * T(0.5);
    const long c3 = ix.r3(j), n3 = ix.r3(j - 1);
    T rc = T(0.0);
    int s = a.nseg - 2;
    int stop = seg_lo(s + 1, a.nseg, L);
    // the levels of one segment are requested together, then accumulated in order
    for (int k = L - 1; s >= 0;) {
        constexpr int kB = 8;
        T xu[kB], xw[kB], xv[kB], xn[kB];
        const int n = min(kB, k - stop + 1);
#pragma unroll
        for (int m = 0; m < kB; ++m) {
            const long o = (long)max(k - m, stop) * W;
            xu[m] = a.spu[c3 + o + i]; xw[m] = a.spu[c3 + o + iw]; xv[m] = a.sv[c3 + o + i]; xn[m] = a.sv[n3 + o + i];
        }
#pragma unroll
        for (int m = 0; m < kB; ++m)
            if (m < n) rc = conv_acc(rc, xu[m], xw[m], inv_dxj, xv[m], jph_c, xn[m], jph_n, inv_dy, a.dsig[k - m]);
        k -= n;
        if (k < stop) {
            a.part[(long)s * a.part_stride + ix.r2(j) + i] = rc;
            --s;
            if (s >= 0) stop = seg_lo(s + 1, a.nseg, L);
        }
    }
}

// U, V of the stage state's rows [j0, j1) and [jb0, jb1): the rows no K4 has produced them for (a
// state that came through gcm_set_state; a band's ghost rows, which the exchange fills)
template <typename T>
__global__ __launch_bounds__(256) void pe_colsum_kernel(PeArgsT<T> a) {
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W;
    const int iblocks = (W + 255) / 256;
    const int jrel = blockIdx.x / iblocks;
    const int i = (blockIdx.x - jrel * iblocks) * 256 + threadIdx.x;
    const int na = a.j1 - a.j0;
    if (i >= W || jrel >= na + (a.jb1 - a.jb0)) return;
    const int j = jrel < na ? a.j0 + jrel : a.jb0 + (jrel - na);
    T cs_u, cs_v;
    column_sum2(a.su + ix.r3(j) + i, a.sv + ix.r3(j) + i, a.dsig, a.L, W, &cs_u, &cs_v);
    a.scs_u[ix.r2(j) + i] = cs_u;
    a.scs_v[ix.r2(j) + i] = cs_v;
}

using PeArgs = PeArgsT<double>;   // the diagnostics and the column physics below are fp64 only

// ---------------------------------------------------------------- calc_energy + STATS (no_limits_2_5d.py:35-60,85-91)
// thread per (j,i) column; out[kStatsWords*block + {0,1,2}] = partial sums of ke, ate, geo,
// {3,4,5,6} = max u, min u, max v, min v of the block's columns, {7} = NaNs seen in u and v
constexpr int kStatsWords = 8;
__global__ __launch_bounds__(256) void pe_energy_kernel(PeArgs a, const double *area, int area_by_i,
                                                        double *out) {
    __shared__ double tab[kExnerTabDoubles];
    __shared__ double red[kStatsWords][4];
    tab[threadIdx.x] = a.exner_tab[threadIdx.x];
    __syncthreads();
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W, L = a.L;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    double ke = 0.0, ate = 0.0, geo = 0.0;
    double umax = -INFINITY, umin = INFINITY, vmax = -INFINITY, vmin = INFINITY, nn = 0.0;
    if (i < W) {
        const int iw = i == 0 ? W - 1 : i - 1;
        const double pc = a.p[ix.r2(j) + i];
        const double ar = area[area_by_i ? i : 0];   // geom.area (H,) broadcasts along the LAST axis (:49)
        const long c3 = ix.r3(j), n3 = ix.r3(j - 1);
        double depth = 0.0;
        for (int k = 0; k < L; ++k) {
            const long o = c3 + (long)k * W;
            const double u_c = a.u[o + i], v_c = a.v[o + i];
            umax = fmax(umax, u_c); umin = fmin(umin, u_c);
            vmax = fmax(vmax, v_c); vmin = fmin(vmin, v_c);
            if (u_c != u_c || v_c != v_c) nn += 1.0;
            const double uc = (u_c + a.u[o + iw]) * 0.5;                          // imh(u)
            const double vc = (v_c + a.v[n3 + (long)k * W + i]) * 0.5;            // jmh(v)
            const double mag = sqrt(uc * uc + vc * vc);
            const double tp = pc * a.sig[k] + a.ptop;
            const double tt = a.t[o + i] * exner(tp, tab);
            const double rho = tp / (kRd * tt);
            const double gd = (pc * a.dsig[k]) / (rho * kG);
            const double airmass = rho * gd * ar;
            depth += gd;                                                          // cumsum over k
            geo += depth * airmass * kG;
            ke += mag * mag * .5 * airmass;
            ate += tt * kCp * airmass;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        ke += __shfl_down(ke, o);
        ate += __shfl_down(ate, o);
        geo += __shfl_down(geo, o);
        umax = fmax(umax, __shfl_down(umax, o)); umin = fmin(umin, __shfl_down(umin, o));
        vmax = fmax(vmax, __shfl_down(vmax, o)); vmin = fmin(vmin, __shfl_down(vmin, o));
        nn += __shfl_down(nn, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][w] = ke; red[1][w] = ate; red[2][w] = geo;
        red[3][w] = umax; red[4][w] = umin; red[5][w] = vmax; red[6][w] = vmin; red[7][w] = nn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = out + kStatsWords * ((long)blockIdx.y * gridDim.x + blockIdx.x);
        for (int q = 0; q < 3; ++q) o[q] = red[q][0] + red[q][1] + red[q][2] + red[q][3];
        o[3] = fmax(fmax(red[3][0], red[3][1]), fmax(red[3][2], red[3][3]));
        o[4] = fmin(fmin(red[4][0], red[4][1]), fmin(red[4][2], red[4][3]));
        o[5] = fmax(fmax(red[5][0], red[5][1]), fmax(red[5][2], red[5][3]));
        o[6] = fmin(fmin(red[6][0], red[6][1]), fmin(red[6][2], red[6][3]));
        o[7] = red[7][0] + red[7][1] + red[7][2] + red[7][3];
    }
}

// ---------------------------------------------------------------- grey radiation (column physics)
// basic_grey_radiation (grey_solar.py:358-563) + solar_timestep (no_limits_2_5d.py:66-75):
// one thread per (j,i) column, the upwelling scan bottom-up, the downwelling scan top-down.
template <typename T>
struct RadArgsT {
    const double *tlw, *tsw, *csw_top, *clw_b_div, *swfac;   // [L] level tables (host-built)
    const double *sigk;                                      // [L] sig^kappa (FACT, see pe_radiation_kernel)
    const double *coslat, *sinlat, *lon;                     // [Hg], [Hg], [W]
    double *gt;                                              // ground temperature [H][W]
    T *dTdt, *dtg;                                           // tendencies out of the diagnostic form (3-D, 2-D scratch)
    double hour_angle, albedo, dt;
    int apply;                                               // 1: t, gt updated in place
    int j0, n0, jb0;                                         // rows of the launch: [j0, j0 + n0), then from jb0 on (a band's ghost
                                                             // rows on either side in one launch: negative / >= H)
};

// The arithmetic is float64 for either storage type T: the column physics is a small share of a
// step, and the fp32 variant then differs from fp64 only by the rounding of what it stores.
// One thread per column.  The long-wave absorption needs the upwelling flux from BELOW a level and
// the downwelling flux from ABOVE it, two opposite scans: the bottom-up scan parks one value per
// level (the absorbed upwelling) and the top-down scan recomputes the level's emission from theta
// (LMAX > 0: kept in registers from the one up-front request of the column; LMAX == 0: read again).
// LMAX > 0: L <= LMAX and the parked column lives in registers (loops unrolled); LMAX == 0: any L,
// parked in LDS, park[L][threads].  The kernel reads theta and writes it (apply) or dTdt (diagnostic);
// nothing else goes through HBM.
// FACT (ptop == 0, the reference's geometry): the Exner factor of level k is (p sig_k / P0)^kappa =
// (p / P0)^kappa sig_k^kappa -- ONE table-and-series evaluation per column and a product per use instead of
// three evaluations per level (emission in either scan, to_potential_temp); sig^kappa comes from the host in
// extended precision.  The product differs from the direct evaluation by an ulp or two of a factor that enters
// theta -> T -> theta symmetrically, far inside the 1e-10 of the parity tests (golden g13, the 2880x1440x40 strips).
constexpr int kRadThreads = 128;
constexpr int kRadTabs = 7;      // per level: tlw, clw_b_div, swfac, sig, dsig, sig^kappa, (free)
template <typename T, int LMAX, bool FACT = false>
__global__ __launch_bounds__(kRadThreads) void pe_radiation_kernel(PeArgsT<T> a, RadArgsT<T> r, T *t_inout) {
    __shared__ double tab[kExnerTabDoubles];
    __shared__ double lev[kRadTabs][LMAX > 0 ? LMAX : 1];
    extern __shared__ unsigned char rad_park_raw[];
    for (int n = threadIdx.x; n < kExnerTabDoubles; n += kRadThreads) tab[n] = a.exner_tab[n];
    const int W = a.W, L = a.L;
    if (LMAX > 0) {
        // the level tables go to LDS (read from global memory inside the scans, every one of their
        // waits would also wait for the theta column still in flight)
        for (int k = threadIdx.x; k < LMAX; k += kRadThreads) {
            const int kk = min(k, L - 1);
            lev[0][k] = r.tlw[kk]; lev[1][k] = r.clw_b_div[kk]; lev[2][k] = r.swfac[kk];
            lev[3][k] = (double)a.sig[kk]; lev[4][k] = (double)a.dsig[kk];
            if (FACT) lev[5][k] = r.sigk[kk];
        }
    }
    __syncthreads();
    constexpr double kSolar = 1.3608 * 1000.0, kSb = 5.67e-8, kCg = 1.13e6;   // constants.py:59,71,25
    double *p_lwb = (double *)rad_park_raw + threadIdx.x;
    double lwb_reg[LMAX > 0 ? LMAX : 1];
    const int i = blockIdx.x * kRadThreads + threadIdx.x;
    const int j = (int)blockIdx.y < r.n0 ? r.j0 + (int)blockIdx.y : r.jb0 + ((int)blockIdx.y - r.n0);
    if (i >= W) return;
    const int jg = wrapi(a.row0 + j, a.Hg);
    const long c3 = (long)j * L * W + i, c2 = (long)j * W + i;
    // LMAX > 0: the whole theta column is requested before any of it is used (one memory latency per
    // column instead of one per level and scan) and kept: the top-down scan does not read it again
    T tcol[LMAX > 0 ? LMAX : 1];
    if (LMAX > 0) {
#pragma unroll
        for (int k = 0; k < LMAX; ++k) tcol[k] = t_inout[c3 + (long)min(k, L - 1) * W];
    }
    const double pc = (double)a.p[c2], gt = r.gt[c2], ptop = (double)a.ptop;
    // zenith_angle, grey_solar.py:49-65 (declination 0)
    const double pa = r.lon[i] + r.hour_angle;
    const double sza = fmax(r.sinlat[jg] * 0.0 + r.coslat[jg] * 1.0 * cos(pa), 0.0);
    const double Sc = kSolar * sza;
    const double S = (1 - r.albedo) * Sc * r.csw_top[0];
    const double g2 = gt * gt;
    const double U_s = 1 * kSb * (g2 * g2);
    const auto tlw = [&](int k) { return LMAX > 0 ? lev[0][k] : r.tlw[k]; };
    const auto clw = [&](int k) { return LMAX > 0 ? lev[1][k] : r.clw_b_div[k]; };
    const auto swf = [&](int k) { return LMAX > 0 ? lev[2][k] : r.swfac[k]; };
    const auto sig = [&](int k) { return LMAX > 0 ? lev[3][k] : (double)a.sig[k]; };
    const auto dsg = [&](int k) { return LMAX > 0 ? lev[4][k] : (double)a.dsig[k]; };
    const double ex_col = FACT ? exner(pc, tab) : 0.0;
    const auto exk = [&](int k) { return FACT ? ex_col * lev[5][LMAX > 0 ? k : 0] : exner(pc * sig(k) + ptop, tab); };
    // true temperature and emission of one level (to_true_temp; grey_solar.py emission)
    const auto emission = [&](int k, double *tt_out) {
        const double th = LMAX > 0 ? (double)tcol[LMAX > 0 ? k : 0] : (double)t_inout[c3 + (long)k * W];
        const double tt = th * exk(k);
        const double t2 = tt * tt;
        *tt_out = tt;
        return (1 - tlw(k)) * kSb * (t2 * t2);
    };
    double B = 0.0, up = 0.0;
    constexpr int kUnroll = LMAX > 0 ? LMAX : 2;
#pragma unroll kUnroll
    for (int k = 0; k < (LMAX > 0 ? LMAX : L); ++k) {       // bottom-up: emission, B, LWA_b
        if (LMAX > 0 && k >= L) break;
        double tt;
        const double em = emission(k, &tt);
        B += em * clw(k);
        const double lwb = up * (1 - tlw(k));
        if (LMAX > 0) lwb_reg[k] = lwb;
        else p_lwb[k * kRadThreads] = lwb;
        up = up * tlw(k) + em;
    }
    const double dtg = (B + S - U_s) / kCg / (.1);
    if (r.apply) r.gt[c2] = gt + dtg * r.dt;
    else r.dtg[c2] = (T)dtg;
    double down = 0.0;
#pragma unroll kUnroll
    for (int kk = 0; kk < (LMAX > 0 ? LMAX : L); ++kk) {     // top-down: LWA_a, then eq. 2.34
        const int k = (LMAX > 0 ? LMAX : L) - 1 - kk;
        if (LMAX > 0 && k >= L) continue;
        const long o = c3 + (long)k * W;
        double tt;
        const double em = emission(k, &tt);
        const double lwa = down * (1 - tlw(k));
        down = down * tlw(k) + em;
        const double U_n = clw(k) * U_s * (1 - tlw(k));
        const double S_n = swf(k) * Sc;
        const double lwb = LMAX > 0 ? lwb_reg[k] : p_lwb[k * kRadThreads];
        const double dTdt = (U_n + S_n - 2 * em + lwa + lwb) * (kG / (kCp * pc * dsg(k)));
        if (r.apply) {
            const double tt_n = tt + dTdt * r.dt;
            t_inout[o] = (T)(tt_n * rcp(exk(k)));                  // to_potential_temp
        } else {
            r.dTdt[o] = (T)dTdt;
        }
    }
}

// ---------------------------------------------------------------- layout transposes
// host layout [k][j][i] (rows of THIS band only, always float64) <-> device [j][k][i] in the
// handle's real type
template <typename T>
__global__ void pe_to_device_kernel(T *dst, const double *src, int W, int H, int L) {
    const long n = (long)W * H * L;
    for (long x = (long)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (long)gridDim.x * blockDim.x) {
        const int i = x % W;
        const long r = x / W;
        const int k = r % L, j = r / L;
        dst[x] = (T)src[((long)k * H + j) * W + i];
    }
}
template <typename T>
__global__ void pe_to_host_kernel(double *dst, const T *src, int W, int H, int L) {
    const long n = (long)W * H * L;
    for (long x = (long)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (long)gridDim.x * blockDim.x) {
        const int i = x % W;
        const long r = x / W;
        const int k = r % L, j = r / L;
        dst[((long)k * H + j) * W + i] = (double)src[x];
    }
}

// ================================================================== host side
// Device buffers in the handle's real type T (fp64, or fp32 for the tolerance sweep).
template <typename T>
struct PeBufs {
    using T2 = typename Vec2<T>::type;
    // state sets: 0/1 ping-pong (cur = set[cur_i]), 2 = star.  [f] p,u,v,t,q; interior pointers
    T *st[3][GCM_NFIELDS] = {};
    T *spu = nullptr, *phi = nullptr, *pgfu = nullptr, *pit = nullptr, *pn = nullptr;
    T *cs[3][2] = {};                           // per state set: sum_k dsig[k] u[k], sum_k dsig[k] v[k] (2-D)
    T *part = nullptr;                          // (kMaxSeg - 1) slabs like pit
    T *cor_u = nullptr, *cor_v = nullptr;
    T *inv_dxj = nullptr, *inv_dxh = nullptr, *sig = nullptr, *dsig = nullptr, *inv_dsig = nullptr,
      *sigb = nullptr, *sigt = nullptr, *heightmap = nullptr, *smul = nullptr;
    T2 *tw = nullptr;
};

struct Pe25d {
    gcm_config cfg{};
    int W = 0, H = 0, L = 0, Hg = 0;
    bool wrap = true, f32 = false;
    std::vector<void *> allocs;
    std::vector<double> dsig_host;              // geometry.py dsig, float64 (radiation level tables)
    std::vector<double> sig_host;               // geometry.py sig, float64
    PeBufs<double> d;
    PeBufs<float> f;
    int cur_i = 0;
    bool star_valid = false;
    int nseg = 1;                               // level segments of K4, chosen from the band's size
    int upd_rows = 7;                           // rows per workgroup of K4 (3 or 7)
    int cus = 256;
    int last_stage_set = -1;                    // state set the last half step took its stage state from (gcm_get_intermediate)
    int ghost_ready = -1;                       // state set whose ghost rows' column sums and anchors are queued already (pe25d_prep_ghost_rows)
    int last_unpack_set = -1;                   // state set whose ghost rows the last unpack filled (halo_t)
    bool pit2d = true;                          // pit from the column sums K4 leaves (nseg == 1, row-group K4)
    int nseg_edge = 1;                          // bands: level segments of the EDGE rows' K4 launch (see half_t)
    bool cs_valid[3] = {false, false, false};   // the state set's column sums belong to its winds
    int pack_set = -1;                          // >= 0: state set gcm_halo_pack reads (step_phase)
    double *stage3 = nullptr;                   // float64 transpose staging, host layout
    double *exner_tab = nullptr;
    FftPlan plan{};
    SuperPlan cplan{};
    double *gt = nullptr;                       // ground temperature [H + 2 ghost rows a side][W], interior row 0 (column physics)
    bool gt_set = false;                        // gcm_set_ground was called
    double *stats_dev = nullptr;                // gcm_stats: block partials, then the area table
    std::vector<double> stats_host, area_host;
    double *rad_tab = nullptr;                  // 5 x [L] level tables of the last radiation call
    double *rad_geo = nullptr;                  // coslat[Hg], sinlat[Hg], lon[W]
    double rad_key[2] = {-1.0, -1.0};           // (t_lw, t_sw) the level tables were built for
    std::vector<double> rad_tab_host, rad_geo_host, rad_latlon;   // host copies (upload sources, change detection)
    std::vector<hipEvent_t> *ev = nullptr;
    size_t *ev_used = nullptr;
    hipStream_t aux = nullptr;                  // second stream of a stage (K2a -> K3), see half_t
    hipStream_t aux2 = nullptr;                 // bands: third stream, K1 of the band's OWN rows (no ghost data: off the exchange chain)
    hipEvent_t ev_cs = nullptr;                 // aux2: the own edge rows' column sums of the state just produced are in place
    int edge_cs_set = -1;                       // state set whose own edge rows' column sums were queued on aux2 (nseg_edge > 1)
    bool k1_split = true;                       // GCM_PE_K1_SPLIT=0: K1 of all rows behind the exchange, as in round 3
    // The events a stage's chains hand each other are signalled by the producing kernel's OWN completion
    // (hipExtLaunchKernelGGL's stopEvent) where a kernel is what they follow: a hipEventRecord is a packet of
    // its own behind the kernel and costs the stream 3 us (tools/micro/sync_cost.hip: 8.9 vs 5.9 us per
    // kernel + record; with a stop event 5.95), four of them per stage on the band's long chain.
    bool stop_events = true;                    // GCM_PE_STOP_EVENTS=0: records, as in round 3
    hipEvent_t ev_k4 = nullptr;                 // completion of the last K4 launch on the caller's stream
    bool k4_fork_valid = false;                 // nothing the second stream must follow was queued on the caller's stream since
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // latitude band with registered send buffers: the edge rows of a stage are updated and packed
    // on `aux` while the interior rows run on the caller's stream (pe25d_step_phase)
    void *send_buf[2] = {nullptr, nullptr};
    hipEvent_t ev_a = nullptr, ev_edges = nullptr;
    bool edges_pending = false;
    bool edges_ev_valid = false;                // ev_edges has been recorded at least once (a wait for it means something)
    // gcm_set_band_overlap(1) on a GCM_PE25D band: the interior rows' K4 is held back until chain B has reached the
    // edge rows' K4 (an event recorded right in front of it), so that the edge rows' workgroups are dispatched
    // first: they then take 15-20 us instead of the 60-70 they take when both launches race for the chip, and the
    // pack and the exchange start that much earlier -- at the price of the ~8 us per stage the interior rows wait.
    // Worth it where an exchange takes longer than the ~20 us of slack the edge chain has otherwise; bench.py --gpus N
    // times both on the real ring and keeps the faster.
    bool edges_first = false;
    hipEvent_t ev_pre_edge = nullptr;
    bool pre_edge_pending = false;
};

template <typename T> static PeBufs<T> &bufs(Pe25d *m);
template <> PeBufs<double> &bufs<double>(Pe25d *m) { return m->d; }
template <> PeBufs<float> &bufs<float>(Pe25d *m) { return m->f; }

template <typename T>
static bool dev_upload(Pe25d *m, T **dst, const T *src, size_t count) {
    void *d = nullptr;
    if (hipMalloc(&d, count * sizeof(T)) != hipSuccess) return false;
    m->allocs.push_back(d);
    if (src && hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
    if (!src && hipMemset(d, 0, count * sizeof(T)) != hipSuccess) return false;
    *dst = (T *)d;
    return true;
}

// host float64 table -> device table in T
template <typename T>
static bool upload_as(Pe25d *m, T **dst, const double *src, size_t count) {
    std::vector<T> tmp(count);
    for (size_t i = 0; i < count; ++i) tmp[i] = (T)src[i];
    return dev_upload<T>(m, dst, tmp.data(), count);
}

static size_t rows_alloc(const Pe25d *m) { return (size_t)m->H + 2 * kGhost; }

template <typename T>
static size_t upd_lds_bytes(int R, int L) { return sizeof(T) * ((size_t)3 * (11 * R + 11) * 64 + 2 + 4 * (size_t)L); }
// looping filter kernels: the complex row + iph(sp) of the row + the row's multiplier
template <typename T>
static size_t filter_loop_lds_bytes(const Pe25d *m) {
    return (size_t)m->W * sizeof(typename Vec2<T>::type) + ((size_t)m->W + m->W / 2 + 1) * sizeof(T);
}
template <typename T>
static size_t filter_lds_bytes(const Pe25d *m) {
    return (size_t)(m->cplan.ok ? 1 : 2) * m->W * sizeof(typename Vec2<T>::type);
}

template <typename T>
static const char *alloc_all(Pe25d *m, const gcm_config &cfg) {
    PeBufs<T> &B = bufs<T>(m);
    const int W = m->W, L = m->L, Hg = m->Hg;
    const size_t n2 = rows_alloc(m) * W, n3 = n2 * L;
    for (int s = 0; s < 3; ++s)
        for (int f = 0; f < GCM_NFIELDS; ++f) {
            T *d = nullptr;
            if (!dev_upload<T>(m, &d, nullptr, f == GCM_P ? n2 : n3)) return "state";
            B.st[s][f] = d + (size_t)kGhost * W * (f == GCM_P ? 1 : L);
        }
    T **inter3[] = {&B.spu, &B.phi, &B.pgfu};
    for (T **pp : inter3) {
        T *d = nullptr;
        if (!dev_upload<T>(m, &d, nullptr, n3)) return "intermediate";
        *pp = d + (size_t)kGhost * W * L;
    }
    T **inter2[] = {&B.pit, &B.pn};
    for (T **pp : inter2) {
        T *d = nullptr;
        if (!dev_upload<T>(m, &d, nullptr, n2)) return "intermediate";
        *pp = d + (size_t)kGhost * W;
    }
    for (int st = 0; st < 3; ++st)
        for (int f = 0; f < 2; ++f) {
            T *d = nullptr;
            if (!dev_upload<T>(m, &d, nullptr, n2)) return "intermediate";
            B.cs[st][f] = d + (size_t)kGhost * W;
        }
    {
        T *d = nullptr;
        if (!dev_upload<T>(m, &d, nullptr, n2 * (kMaxSeg - 1))) return "intermediate";
        B.part = d + (size_t)kGhost * W;
    }
    // tables
    std::vector<double> idj(Hg), idh(Hg), ids(L);
    for (int j = 0; j < Hg; ++j) {
        idj[j] = 1.0 / cfg.dx_j[j];
        idh[j] = 1.0 / cfg.dx_h[j];
    }
    for (int k = 0; k < L; ++k) ids[k] = 1.0 / cfg.dsig[k];
    if (!upload_as<T>(m, &B.inv_dxj, idj.data(), Hg) || !upload_as<T>(m, &B.inv_dxh, idh.data(), Hg) ||
        !upload_as<T>(m, &B.sig, cfg.sig, L) || !upload_as<T>(m, &B.dsig, cfg.dsig, L) ||
        !upload_as<T>(m, &B.inv_dsig, ids.data(), L) || !upload_as<T>(m, &B.sigb, cfg.sigb, L) ||
        !upload_as<T>(m, &B.sigt, cfg.sigt, L))
        return "tables";
    if (cfg.heightmap && !upload_as<T>(m, &B.heightmap, cfg.heightmap, (size_t)Hg * W)) return "heightmap";
    if (cfg.cor_u && (!upload_as<T>(m, &B.cor_u, cfg.cor_u, Hg) || !upload_as<T>(m, &B.cor_v, cfg.cor_v, Hg)))
        return "coriolis tables";
    if (W > 1) {
        // filter multiplier, low_pass.py:61-72, same expression order as the reference
        const int nh = W / 2 + 1;
        std::vector<double> S((size_t)Hg * nh);
        for (int j = 0; j < Hg; ++j) {
            const double drat = cfg.dy / cfg.dx_j[j];
            S[(size_t)j * nh] = 1.0;
            for (int n = 1; n < nh; ++n) {
                const double bysn = 1.0 / std::sin(M_PI / W * (double)n);
                const double sm = 1.0 - bysn / drat;
                S[(size_t)j * nh + n] = 1.0 - std::fmax(sm, 0.0);
            }
        }
        if (!upload_as<T>(m, &B.smul, S.data(), S.size())) return "filter multiplier";
        using T2 = typename Vec2<T>::type;
        std::vector<T2> tw(W);
        for (int n = 0; n < W; ++n) {
            const long double ang = -2.0L * 3.14159265358979323846264338327950288L * n / W;
            tw[n].x = (T)cosl(ang);
            tw[n].y = (T)sinl(ang);
        }
        if (!dev_upload<T2>(m, &B.tw, tw.data(), W)) return "twiddles";
    }
    if ((spu_filter_loop_kernel_for<T>(m->cplan) &&
         hipFuncSetAttribute((const void *)spu_filter_loop_kernel_for<T>(m->cplan), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)filter_loop_lds_bytes<T>(m)) != hipSuccess) ||
        hipFuncSetAttribute((const void *)spu_filter_kernel_for<T>(m->cplan), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)filter_lds_bytes<T>(m)) != hipSuccess ||
        hipFuncSetAttribute((const void *)pgf_filter_kernel_for<T>(m->cplan), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)filter_lds_bytes<T>(m)) != hipSuccess ||
        hipFuncSetAttribute((const void *)pit2d_kernel_for<T>(m->cplan), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(filter_lds_bytes<T>(m) + sizeof(T) * (size_t)W)) != hipSuccess ||
        hipFuncSetAttribute((const void *)pe_geopot_kernel<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(L * kColThreads * sizeof(T))) != hipSuccess ||
        hipFuncSetAttribute((const void *)pe_geopot_kernel<T, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(L * kColThreads * sizeof(T))) != hipSuccess ||
        hipFuncSetAttribute((const void *)pe_radiation_kernel<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(double) * (size_t)L * kRadThreads)) != hipSuccess ||
        hipFuncSetAttribute((const void *)update_rows_kernel_for<T>(7, true), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(7, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)update_rows_kernel_for<T>(7, false), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(7, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)update_rows_kernel_for<T>(7, true, true), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(7, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)update_rows_kernel_for<T>(7, false, true), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(7, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)update_rows_kernel_for<T>(3, true, true), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(3, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)update_rows_kernel_for<T>(3, false, true), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(3, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)update_rows_kernel_for<T>(3, true), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(3, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)update_rows_kernel_for<T>(3, false), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(3, L)) != hipSuccess)
        return "dynamic LDS size";
    return nullptr;
}

// A second stream that really runs beside `main`.  HIP maps streams onto a few hardware queues
// round-robin; two streams on one queue execute in order, and which streams share depends on how
// many were created before (with RCCL initialised in the process the library's second stream
// landed on the compute stream's queue: every kernel of a stage serialised; the comm stream on it
// made the exchange wait for the interior rows).  So: create a few candidates, run a 100 us spin
// kernel on `main` (and `other`) and on the candidate at once, and keep the first candidate for
// which they all overlapped.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();          // 100 MHz
    while (wall_clock64() - t0 < ticks) {
    }
}

// a kernel that does nothing for `us` microseconds (the loopback exchange's stand-in for a transfer time)
void launch_spin(hipStream_t s, double us) {
    if (us > 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, (long long)(us * 100.0));
}

hipStream_t concurrent_stream(hipStream_t main, hipStream_t other) {
    constexpr int kCandidates = 6;
    constexpr long long kSpinTicks = 10000;       // 100 us
    hipStream_t cand[kCandidates] = {};
    hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr, ec = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess ||
        hipEventCreate(&ec) != hipSuccess)
        return nullptr;
    int pick = -1, made = 0;
    const char *vb = getenv("GCM_VERBOSE");
    const bool verbose = vb && vb[0] == '1';
    for (int c = 0; c < kCandidates && pick < 0; ++c) {
        if (hipStreamCreateWithFlags(&cand[c], hipStreamNonBlocking) != hipSuccess) break;
        ++made;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipStreamSynchronize(main);
            if (other) (void)hipStreamSynchronize(other);
            (void)hipStreamSynchronize(cand[c]);
            (void)hipEventRecord(e0, main);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, main, kSpinTicks);
            if (other) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, other, kSpinTicks);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cand[c], kSpinTicks);
            (void)hipEventRecord(ea, main);
            if (other) (void)hipEventRecord(ec, other);
            (void)hipEventRecord(eb, cand[c]);
            (void)hipStreamSynchronize(main);
            if (other) (void)hipStreamSynchronize(other);
            (void)hipStreamSynchronize(cand[c]);
            float ta = 0.f, tb = 0.f, tc = 0.f;
            (void)hipEventElapsedTime(&ta, e0, ea);
            (void)hipEventElapsedTime(&tb, e0, eb);
            if (other) (void)hipEventElapsedTime(&tc, e0, ec);
            best = std::min(best, std::max(ta, std::max(tb, tc)));
        }
        if (verbose) fprintf(stderr, "gcmcore: stream candidate %d: the 100 us spins took %.1f us\n", c, best * 1e3f);
        if (best < 0.16f) pick = c;               // all spins inside 160 us: they overlapped
    }
    if (pick < 0 && made > 0) pick = 0;           // none overlaps: still correct, only serialised
    for (int c = 0; c < made; ++c)
        if (c != pick) (void)hipStreamDestroy(cand[c]);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    (void)hipEventDestroy(ec);
    if (verbose) fprintf(stderr, "gcmcore: picked stream candidate %d of %d\n", pick, made);
    return pick >= 0 ? cand[pick] : nullptr;
}

hipStream_t pe25d_aux_stream(const Pe25d *m) { return m->aux; }
// gcm_band_run's one join: whatever follows on `s` also follows what the third stream still holds (the own edge
// rows' column sums of the last stage)
void pe25d_join_third_stream(Pe25d *m, hipStream_t s) {
    if (!m->aux2) return;
    (void)hipEventRecord(m->ev_cs, m->aux2);
    (void)hipStreamWaitEvent(s, m->ev_cs, 0);
}
// something chain B reads was queued on the caller's stream by somebody else (ghost rows unpacked there, the ground
// temperature uploaded): the next stage's chain B follows that stream's position, not just the last K4
void pe25d_fork_invalidate(Pe25d *m) { m->k4_fork_valid = false; }
void pe25d_set_edges_first(Pe25d *m, bool on) { m->edges_first = on; }
int pe25d_new_state_set(const Pe25d *m) { return (m->pack_set >= 0 && m->pack_set != 2) ? m->pack_set : m->cur_i; }

Pe25d *pe25d_create(const gcm_config &cfg, hipStream_t main_stream, std::string *err) {
    if (!cfg.dx_j || !cfg.dx_h || !cfg.sig || !cfg.dsig || !cfg.sigb || !cfg.sigt) {
        *err = "GCM_PE25D: geometry tables (dx_j, dx_h, sig, dsig, sigb, sigt) are required";
        return nullptr;
    }
    if (!(cfg.dy > 0)) { *err = "GCM_PE25D: dy must be > 0"; return nullptr; }
    if (cfg.global_height < cfg.height || cfg.row0 < 0 || cfg.row0 + cfg.height > cfg.global_height) {
        *err = "GCM_PE25D: band rows outside the global grid";
        return nullptr;
    }
    if (cfg.nranks == 1 && cfg.global_height != cfg.height) {
        *err = "GCM_PE25D: nranks == 1 needs height == global_height";
        return nullptr;
    }
    if (cfg.filter && cfg.width > 1 && cfg.width % 2) {
        *err = "GCM_PE25D: the zonal filter needs an even width (low_pass.py:57; numpy irfft)";
        return nullptr;
    }
    if ((cfg.cor_u != nullptr) != (cfg.cor_v != nullptr)) {
        *err = "GCM_PE25D: cor_u and cor_v must be given together";
        return nullptr;
    }
    if (cfg.dtype != GCM_F64 && cfg.dtype != GCM_F32) { *err = "GCM_PE25D: bad dtype"; return nullptr; }
    Pe25d *m = new Pe25d;
    m->cfg = cfg;
    m->W = cfg.width;
    m->H = cfg.height;
    m->L = cfg.layers;
    m->Hg = cfg.global_height;
    m->wrap = cfg.nranks == 1;
    m->f32 = cfg.dtype == GCM_F32;
    m->dsig_host.assign(cfg.dsig, cfg.dsig + cfg.layers);
    m->sig_host.assign(cfg.sig, cfg.sig + cfg.layers);
    const int W = m->W, L = m->L;
    auto bad = [&](const char *what) {
        *err = std::string("hip: GCM_PE25D allocation/upload failed: ") + what;
        pe25d_destroy(m);
        return (Pe25d *)nullptr;
    };
    if (W > 1) make_super_plan(W, &m->cplan);
    if (W > 1 && (!make_plan(W, &m->plan) || (size_t)W * 32 + 8192 > 160 * 1024)) {
        *err = "GCM_PE25D: width not supported by the in-LDS FFT (too many factors or > 4864)";
        pe25d_destroy(m);
        return nullptr;
    }
    if ((size_t)L * kColThreads * sizeof(double) + kExnerTabDoubles * sizeof(double) > 160 * 1024 ||
        sizeof(double) * (size_t)L * kRadThreads + 4096 > 160 * 1024 ||
        upd_lds_bytes<double>(3, L) + 4096 > 160 * 1024) {
        *err = "GCM_PE25D: too many layers for the column kernels' LDS (max 100)";
        pe25d_destroy(m);
        return nullptr;
    }
    {
        // K4 keeps 8 waves per CU resident (2 per SIMD), one row x 62 columns each.  A band with less than
        // about a round and a half of them splits the level march, so that the launch is several short
        // rounds instead of one long one (results do not depend on the split).  Measured on 1440 columns
        // x 24 levels: 90 rows best with 2 segments, 180 and more with 1.  Short bands also take the
        // 3-row workgroups (two per CU, out of step with each other: 3-5 % faster up to ~200 rows; the
        // 7-row form reads the halo rows 9/7 instead of 5/3 times and is kept where bytes matter).
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        m->cus = cus;
        const double rounds = (double)((W + kUpdCols - 1) / kUpdCols) * m->H / (8.0 * cus);
        long want = (long)std::ceil(1.5 / std::max(rounds, 1e-3));
        // fp32: 153 VGPRs and 34 KB of tiles per 3-row workgroup let THREE of them share a CU (three waves per SIMD; the
        // 7-row form holds one 8-wave workgroup, two waves per SIMD): c4_f32 1.214 -> 1.126 ms per step (round 4, A/B
        // on one box; 5-row groups, 6 waves, place only one workgroup per CU and take 1.44).  fp64 (215 / 231 VGPRs)
        // stays at two waves per SIMD either way: 3-row groups only where the band is short -- up to 360 rows they win
        // (round 4, one box: a 360-row band of C4 1.050 -> 1.008 ms, of the 2880x1440x40 grid 3.461 -> 3.396), at 720
        // rows the 7-row form does (C4 1.874 vs 1.907, the 2880-column grid 6.597 vs 6.627).
        m->upd_rows = (m->H <= 400 || m->f32) ? 3 : 7;
        bool forced = false;
        if (const char *e = getenv("GCM_PE_LEVEL_SEGMENTS")) { want = atoi(e); forced = true; }
        if (const char *e = getenv("GCM_PE_PIT2D")) m->pit2d = atoi(e) != 0;      // 0: pit from the 3-D fields (pe_pit_kernel)
        if (const char *e = getenv("GCM_PE_UPDATE_ROWS")) m->upd_rows = atoi(e) == 3 ? 3 : 7;      // rows per workgroup
        const int cap = std::min(kMaxSeg, std::max(1, L / 4));
        m->nseg = (int)std::max(1L, std::min((long)cap, want));
        // K4 fills the chip with whole columns (a 90-row band: 2 % slower than in two segments) and then
        // leaves the column sums pit needs: segments only on request
        if (!forced) m->nseg = 1;
        if (!m->wrap && m->nseg == 1 && m->pit2d && m->H > 2 * kGhost) {
            m->nseg_edge = std::min(kMaxSeg, std::max(1, L / 6));
            if (const char *e = getenv("GCM_PE_EDGE_SEGMENTS")) m->nseg_edge = std::max(1, std::min(kMaxSeg, atoi(e)));
            if (L / m->nseg_edge < 2) m->nseg_edge = 1;
        }
    }
    if (const char *what = m->f32 ? alloc_all<float>(m, cfg) : alloc_all<double>(m, cfg)) return bad(what);
    if (!dev_upload<double>(m, &m->stage3, nullptr, (size_t)m->H * W * L)) return bad("staging");
    double tab[kExnerTabDoubles];
    build_exner_table(tab);
    if (!dev_upload(m, &m->exner_tab, tab, kExnerTabDoubles)) return bad("exner table");
    {
        // ground temperature (column physics), with a band's ghost rows: they travel with every ghost-row message
        double *d = nullptr;
        if (!dev_upload<double>(m, &d, nullptr, rows_alloc(m) * (size_t)W)) return bad("ground temperature");
        m->gt = d + (size_t)kGhost * W;
    }
    const char *no_aux = getenv("GCM_PE_SINGLE_STREAM");      // diagnostic: one chain, one stream
    // a plain stream: a high-priority one finished the edge rows earlier, but in some processes
    // (depending on how many streams existed before) the whole step then ran at half speed
    if (!(no_aux && no_aux[0] == '1')) {
        m->aux = concurrent_stream(main_stream, nullptr);
        if (!m->aux) return bad("second stream");
        if (const char *e = getenv("GCM_PE_K1_SPLIT")) m->k1_split = atoi(e) != 0;
        if (!m->wrap && m->k1_split) {
            m->aux2 = concurrent_stream(main_stream, m->aux);
            if (!m->aux2) return bad("third stream");
        }
    }
    if (const char *e = getenv("GCM_PE_STOP_EVENTS")) m->stop_events = atoi(e) != 0;
    if (hipEventCreateWithFlags(&m->ev_pre_edge, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_k4, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_cs, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_edges, hipEventDisableTiming) != hipSuccess)
        return bad("events");
    return m;
}

void pe25d_destroy(Pe25d *m) {
    if (!m) return;
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->ev_a) (void)hipEventDestroy(m->ev_a);
    if (m->ev_edges) (void)hipEventDestroy(m->ev_edges);
    if (m->ev_cs) (void)hipEventDestroy(m->ev_cs);
    if (m->ev_k4) (void)hipEventDestroy(m->ev_k4);
    if (m->ev_pre_edge) (void)hipEventDestroy(m->ev_pre_edge);
    if (m->aux2) {
        (void)hipStreamSynchronize(m->aux2);
        (void)hipStreamDestroy(m->aux2);
    }
    if (m->aux) {
        (void)hipStreamSynchronize(m->aux);      // (a band's last exchange may still be unpacking)
        (void)hipStreamDestroy(m->aux);
    }
    for (void *p : m->allocs) (void)hipFree(p);
    delete m;
}

// State transfers run on the handle's stream `s` and synchronise only that stream: other handles
// and streams of the process are not stalled.  The float64 staging buffer is reused field by field,
// which the stream order makes safe.
template <typename T>
static int xfer_t(Pe25d *m, int set, bool to_dev, const double *const in[GCM_NFIELDS],
                  double *const out[GCM_NFIELDS], hipStream_t s, std::string *err) {
    PeBufs<T> &B = bufs<T>(m);
    const int W = m->W, H = m->H;
    hipError_t e = hipSuccess;
    for (int f = 0; f < GCM_NFIELDS && e == hipSuccess; ++f) {
        const void *hp = to_dev ? (const void *)in[f] : (const void *)out[f];
        if (!hp) continue;
        const int L = f == GCM_P ? 1 : m->L;
        const size_t bytes = sizeof(double) * (size_t)H * W * L;
        if (to_dev) {
            e = hipMemcpyAsync(m->stage3, in[f], bytes, hipMemcpyHostToDevice, s);
            if (e == hipSuccess)
                hipLaunchKernelGGL(pe_to_device_kernel<T>, dim3(1024), dim3(256), 0, s, B.st[set][f], m->stage3, W, H, L);
        } else {
            hipLaunchKernelGGL(pe_to_host_kernel<T>, dim3(1024), dim3(256), 0, s, m->stage3, B.st[set][f], W, H, L);
            e = hipMemcpyAsync(out[f], m->stage3, bytes, hipMemcpyDeviceToHost, s);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        *err = std::string("pe25d state transfer: ") + hipGetErrorString(e);
        return GCM_ERR_HIP;
    }
    return GCM_OK;
}

static int xfer(Pe25d *m, int set, bool to_dev, const double *const in[GCM_NFIELDS],
                double *const out[GCM_NFIELDS], hipStream_t s, std::string *err) {
    return m->f32 ? xfer_t<float>(m, set, to_dev, in, out, s, err) : xfer_t<double>(m, set, to_dev, in, out, s, err);
}

int pe25d_set(Pe25d *m, bool star, const double *p, const double *u, const double *v,
              const double *t, const double *q, hipStream_t s, std::string *err) {
    const double *in[GCM_NFIELDS] = {p, u, v, t, q};
    int rc = xfer(m, star ? 2 : m->cur_i, true, in, nullptr, s, err);
    if (u || v) m->cs_valid[star ? 2 : m->cur_i] = false;
    m->ghost_ready = -1;
    if (u || v) m->edge_cs_set = -1;
    m->k4_fork_valid = false;                    // the transposes on the caller's stream: the second stream follows them
    m->last_stage_set = -1;                      // gcm_get_intermediate: the stage state the anchors belong to is gone
    if (rc == GCM_OK) m->star_valid = star;
    return rc;
}

int pe25d_get(Pe25d *m, bool star, double *p, double *u, double *v, double *t, double *q,
              hipStream_t s, std::string *err) {
    if (star && !m->star_valid) {
        *err = "get_star: no predicted state yet";
        return GCM_ERR_STATE;
    }
    double *out[GCM_NFIELDS] = {p, u, v, t, q};
    return xfer(m, star ? 2 : m->cur_i, false, nullptr, out, s, err);
}

template <typename T>
static PeArgsT<T> make_args(Pe25d *m, int stage_set, int out_set, double dt) {
    PeBufs<T> &Bf = bufs<T>(m);
    PeArgsT<T> a{};
    T *const *B = Bf.st[m->cur_i];
    T *const *S = Bf.st[stage_set];
    T *const *O = Bf.st[out_set];
    a.p = B[GCM_P]; a.u = B[GCM_U]; a.v = B[GCM_V]; a.t = B[GCM_T]; a.q = B[GCM_Q];
    a.sp = S[GCM_P]; a.su = S[GCM_U]; a.sv = S[GCM_V]; a.st = S[GCM_T]; a.sq = S[GCM_Q];
    a.op = O[GCM_P]; a.ou = O[GCM_U]; a.ov = O[GCM_V]; a.ot = O[GCM_T]; a.oq = O[GCM_Q];
    a.spu = Bf.spu; a.phi = Bf.phi; a.pgfu = Bf.pgfu;
    a.pit = Bf.pit; a.pn = Bf.pn;
    a.scs_u = Bf.cs[stage_set][0]; a.scs_v = Bf.cs[stage_set][1];
    a.ocs_u = a.ocs_v = nullptr;
    a.part = Bf.part;
    a.part_stride = (long)rows_alloc(m) * m->W;
    a.nseg = m->nseg;
    a.spu_j0 = a.pit_j0 = -(1 << 30);            // K1: every row of the launch
    a.spu_j1 = a.pit_j1 = 1 << 30;
    a.inv_dxj = Bf.inv_dxj; a.inv_dxh = Bf.inv_dxh;
    a.sig = Bf.sig; a.dsig = Bf.dsig; a.inv_dsig = Bf.inv_dsig; a.sigb = Bf.sigb; a.sigt = Bf.sigt;
    a.heightmap = Bf.heightmap; a.cor_u = Bf.cor_u; a.cor_v = Bf.cor_v; a.smul = Bf.smul; a.tw = Bf.tw;
    a.exner_tab = m->exner_tab;
    a.plan = m->plan;
    a.cplan = m->cplan;
    a.W = m->W; a.H = m->H; a.L = m->L; a.Hg = m->Hg; a.row0 = m->cfg.row0;
    a.wrap = m->wrap ? 1 : 0;
    a.filter = m->cfg.filter;
    a.dt = (T)dt;
    a.inv_dy = (T)(1.0 / m->cfg.dy);
    a.ptop = (T)m->cfg.ptop;
    return a;
}

static bool async_edges(const Pe25d *m) { return m->send_buf[0] && m->send_buf[1]; }

static void tick(Pe25d *m, hipStream_t s) {
    if (m->ev && m->ev_used && *m->ev_used < m->ev->size()) (void)hipEventRecord((*m->ev)[(*m->ev_used)++], s);
}

// Column sums and geopotential anchors of the stage state's rows that no kernel of the previous stage left:
// all rows' sums of a freshly set state, else a band's two ghost rows next to its own (pit of row j takes V of
// row j - 1; the intermediates extend to row j1) -- and, in the same launch, the geopotential of a band's south
// ghost row (K4 of row j1 - 1 takes phi of row j1).  `a`: the stage's arguments with j0 / j1 set.
template <typename T>
static void prep_rows(Pe25d *m, const PeArgsT<T> &a, int stage_set, bool p2, int j1, int ext, hipStream_t sb) {
    const int W = m->W, L = m->L;
    const auto geopot = [&](const PeArgsT<T> &c, hipStream_t st) {
        const int rows = (c.j1 - c.j0) + (c.jb1 - c.jb0);
        if (rows <= 0) return;
        const long tiles = (long)((W + kColThreads - 1) / kColThreads) * rows;
        const dim3 gg((unsigned)((tiles + 7) / 8 * 8));
        const size_t park = sizeof(T) * (size_t)L * kColThreads;
        if (c.cs_rows) {
            if (L <= 24) hipLaunchKernelGGL((pe_geopot_kernel<T, 24, true>), gg, dim3(kColThreads), 0, st, c);
            else if (L <= 40) hipLaunchKernelGGL((pe_geopot_kernel<T, 40, true>), gg, dim3(kColThreads), 0, st, c);
            else hipLaunchKernelGGL((pe_geopot_kernel<T, 0, true>), gg, dim3(kColThreads), park, st, c);
        } else {
            if (L <= 24) hipLaunchKernelGGL((pe_geopot_kernel<T, 24>), gg, dim3(kColThreads), 0, st, c);
            else if (L <= 40) hipLaunchKernelGGL((pe_geopot_kernel<T, 40>), gg, dim3(kColThreads), 0, st, c);
            else hipLaunchKernelGGL((pe_geopot_kernel<T, 0>), gg, dim3(kColThreads), park, st, c);
        }
    };
    PeArgsT<T> c = a;
    c.jb0 = c.jb1 = 0;
    bool fresh = false;
    if (!p2) {
        c.j0 = j1; c.j1 = j1 + ext;
    } else if (!m->cs_valid[stage_set]) {
        c.j0 = m->wrap ? 0 : -1;
        c.j1 = m->H + ext;
        m->cs_valid[stage_set] = true;
        fresh = true;
    } else if (m->nseg_edge > 1 && m->edge_cs_set != stage_set) {   // + the own edge rows (marched in segments: no sums from K4)
        c.j0 = -1; c.j1 = kGhost;
        c.jb0 = m->H - kGhost; c.jb1 = m->H + 1;
    } else {
        c.j0 = -1; c.j1 = 0;
        c.jb0 = m->H; c.jb1 = m->H + 1;
    }
    c.cs_rows = p2 ? 1 : 0;
    c.geo_j0 = j1; c.geo_j1 = j1 + ext;
    if (fresh && (c.j1 - c.j0) > 8) {
        // a whole state's sums: the plain column-sum kernel (no thermodynamics compiled in), then the ghost row
        hipLaunchKernelGGL(pe_colsum_kernel<T>, dim3((unsigned)((W + 255) / 256) * (c.j1 - c.j0)), dim3(256), 0, sb, c);
        c.cs_rows = 0;
        c.j0 = j1; c.j1 = j1 + ext;
    }
    if (!m->wrap || (fresh && c.cs_rows)) geopot(c, sb);
}

// one Euler stage over rows [j0, j1): state `stage_set` -> `out_set`, base = current.
// mode 0: everything; mode 1: K1-K3 on all rows + K4 on the two edge rows of either side (the rows
// a neighbouring band needs); mode 2: K4 on the remaining interior rows.  Modes 1 + 2 == mode 0.
// chained: the call comes from gcm_band_run's own sequence (nothing but the stage's kernels between two stages on `s`)
template <typename T>
static void half_t(Pe25d *m, int stage_set, int out_set, double dt, int j0, int j1, hipStream_t s, int mode, bool chained) {
    if (j1 <= j0) return;
    PeArgsT<T> a = make_args<T>(m, stage_set, out_set, dt);
    m->last_stage_set = stage_set;
    const int W = m->W, L = m->L;
    const int ext = m->wrap ? 0 : 1;             // intermediates are also needed on row j1 (south)
    const size_t lds = filter_lds_bytes<T>(m);
    const int fft_threads = m->cplan.ok ? m->cplan.threads : kFftThreads;
    const int pairs = (L + 1) / 2;
    // pit from the 2-D column sums (pe_pit2d_kernel) where K4 marches whole columns and can leave them
    const bool p2 = m->pit2d && a.nseg == 1;
    if (p2) {
        a.ocs_u = bufs<T>(m).cs[out_set][0];
        a.ocs_v = bufs<T>(m).cs[out_set][1];
    }
    // Two chains on two streams (a cross-queue dependency costs ~10 us on this chip when the waiting queue
    // is already idle, ~3 us when the event completed earlier: tools/micro/sync_cost.hip):
    //   A, the caller's stream:  K2a (own rows) -> K3 -> K4 (all rows, or the interior rows of a band)
    //   B, the second stream:    column sums and anchors of the ghost rows -> K1 (+ pit) [-> a band's edge
    //                            rows: their partial sums, K4, pack; the exchange and the unpack follow]
    // A is the long chain and runs without waiting for anything that has not long finished: K4 waits for
    // B's K1 (done while K3 runs), the next stage's K2a for B's edge rows (done while the interior rows
    // run).  A never touches ghost rows, so it never waits for an exchange; B does, in stream order.
    hipStream_t sb = m->aux ? m->aux : s;
    // pe_geopot_kernel over the rows of `c` ([j0, j1) and [jb0, jb1)); c.geo_j0 / geo_j1: the rows it forms phi
    // for, c.cs_rows: it also forms the column sums of all its rows
    const auto geopot = [&](const PeArgsT<T> &c, hipStream_t st, hipEvent_t stop = nullptr) {
        const int rows = (c.j1 - c.j0) + (c.jb1 - c.jb0);
        if (rows <= 0) return;
        const long tiles = (long)((W + kColThreads - 1) / kColThreads) * rows;
        const dim3 gg((unsigned)((tiles + 7) / 8 * 8));
        const size_t park = sizeof(T) * (size_t)L * kColThreads;
        if (c.cs_rows) {
            if (L <= 24) hipLaunchKernelGGL((pe_geopot_kernel<T, 24, true>), gg, dim3(kColThreads), 0, st, c);
            else if (L <= 40) hipLaunchKernelGGL((pe_geopot_kernel<T, 40, true>), gg, dim3(kColThreads), 0, st, c);
            else hipLaunchKernelGGL((pe_geopot_kernel<T, 0, true>), gg, dim3(kColThreads), park, st, c);
        } else {
            void (*kern)(PeArgsT<T>) = L <= 24 ? pe_geopot_kernel<T, 24> : L <= 40 ? pe_geopot_kernel<T, 40> : pe_geopot_kernel<T, 0>;
            const unsigned dyn = L <= 40 ? 0u : (unsigned)park;
            if (stop) hipExtLaunchKernelGGL(kern, gg, dim3(kColThreads), dyn, st, nullptr, stop, 0, c);
            else hipLaunchKernelGGL(kern, gg, dim3(kColThreads), dyn, st, c);
        }
    };
    const bool split = mode != 0 && (j1 - j0) > 2 * kGhost;
    // (Round 4, built and rejected: the edge rows' K3 as a launch of its own on chain B right behind K2a, so that their
    // K4, the pack and the exchange start before the interior rows' K4 takes the chip.  Four rows are 48 workgroups of
    // five dependent LDS passes: 30-60 us on a chain that already holds K1's ghost-row launch and the partial sums, and
    // the N = 8 band got 4 % slower with no exchange time and 10 % slower with 40 us of it
    // (profiles/r04/ab_band_edge_k3_on_chain_b.txt: `base` = with it).)
    bool split_k1 = false;
    if (mode != 2) {
        a.j0 = j0;
        a.j1 = j1 + ext;
        // ---- chain B: everything that reads the whole stage state, ghost rows included
        // what chain B follows on the caller's stream: the previous stage's K4 (its own completion, ev_k4, when
        // nothing else that B reads or overwrites was queued since), else the stream's position now
        hipEvent_t fork = m->ev_fork;
        if (m->aux) {
            if (m->stop_events && m->k4_fork_valid && chained) fork = m->ev_k4;
            else (void)hipEventRecord(m->ev_fork, s);
            (void)hipStreamWaitEvent(m->aux, fork, 0);
        }
        const bool ghosts_queued = m->ghost_ready == stage_set && (!p2 || m->cs_valid[stage_set]);
        if (ghosts_queued) m->ghost_ready = -1;                    // queued behind the unpack already
        else prep_rows<T>(m, a, stage_set, p2, j1, ext, sb);
        static const bool no_loop = getenv("GCM_PE_FILTER_NO_LOOP") != nullptr;        // diagnostic: one workgroup per pair
        const FilterLoopKernel<T> k1 = (m->cfg.filter && W > 1 && !no_loop) ? spu_filter_loop_kernel_for<T>(m->cplan) : nullptr;
        bool pit_done = false, ev_a_done = false;
        // (the own edge rows' column sums of this state were queued on the third stream behind the edge rows' K4
        // of the stage that produced it: everything on the second stream that reads them waits for that)
        if (m->aux2 && m->edge_cs_set == stage_set) (void)hipStreamWaitEvent(sb, m->ev_cs, 0);
        // A band inside gcm_band_run (the ghost rows' column sums and anchors are queued behind the unpack already):
        // K1 is row-local -- spu of row j takes su and sp of row j only, pit of row j the column sums of rows j - 1, j
        // and sp of rows j - 1 .. j + 1 -- so the band's OWN rows need nothing from the exchange.  They go to a third
        // stream that waits for the previous stage's K4 only (spu of rows [0, H), pit of rows [2, H - 2]: what the
        // interior rows' K4 reads), and the second stream keeps the launch for the rows that do need ghost data (spu of
        // the south ghost row, pit of rows 0, 1, H - 1, H).  The interior rows' K4 then waits for the own-row launch
        // alone, not for the exchange chain (round 3: 32 us per corrector stage of the N = 8 band of C4).
        split_k1 = k1 && mode == 1 && m->aux2 && p2 && ghosts_queued && (j1 - j0) >= 2 * kGhost + 3 &&
                              (m->nseg_edge == 1 || m->edge_cs_set == stage_set);
        if (split_k1) {
            const auto launch_k1 = [&](const PeArgsT<T> &c, int rows, hipStream_t st, hipEvent_t stop) {
                const int groups = std::min(pairs, std::max(1, (3 * m->cus + rows - 1) / rows));
                const int ppw = (pairs + groups - 1) / groups;
                const int ny = (pairs + ppw - 1) / ppw;
                if (stop && m->stop_events)
                    hipExtLaunchKernelGGL(k1, dim3(rows, ny + 1), dim3(fft_threads), (unsigned)filter_loop_lds_bytes<T>(m), st, nullptr, stop, 0, c, ppw, ny);
                else
                    hipLaunchKernelGGL(k1, dim3(rows, ny + 1), dim3(fft_threads), filter_loop_lds_bytes<T>(m), st, c, ppw, ny);
                if (stop && !m->stop_events) (void)hipEventRecord(stop, st);
            };
            (void)hipStreamWaitEvent(m->aux2, fork, 0);
            if (m->edges_ev_valid) (void)hipStreamWaitEvent(m->aux2, m->ev_edges, 0);   // (the previous stage's edge rows: su, sp of rows 0, 1, H - 2, H - 1)
            PeArgsT<T> c = a;
            c.j0 = j0; c.j1 = j1; c.jb0 = c.jb1 = 0;
            c.pit_j0 = j0 + kGhost; c.pit_j1 = j1 - kGhost + 1;
            launch_k1(c, j1 - j0, m->aux2, m->ev_a);             // (ev_a: what K4 of the interior rows takes)
            c = a;
            c.j0 = j0; c.j1 = j0 + kGhost; c.jb0 = j1 - 1; c.jb1 = j1 + ext;
            c.spu_j0 = j1; c.spu_j1 = j1 + ext;
            launch_k1(c, kGhost + 1 + ext, sb, nullptr);
            pit_done = true;                                     // (chain B waits for ev_a below, ahead of the edge rows' partial sums)
        } else if (k1) {
            // all pairs of a row in one workgroup when there are rows enough to fill the chip, else groups;
            // with the 2-D form of pit one more workgroup per row forms pit and p_n (pe_pit2d_row)
            const int rows = a.j1 - a.j0;
            const int groups = std::min(pairs, std::max(1, (3 * m->cus + rows - 1) / rows));
            const int ppw = (pairs + groups - 1) / groups;
            const int ny = (pairs + ppw - 1) / ppw;
            if (p2 && m->aux && m->stop_events) {
                hipExtLaunchKernelGGL(k1, dim3(rows, ny + 1), dim3(fft_threads), (unsigned)filter_loop_lds_bytes<T>(m), sb, nullptr, m->ev_a, 0, a, ppw, ny);
                ev_a_done = true;
            } else {
                hipLaunchKernelGGL(k1, dim3(rows, ny + (p2 ? 1 : 0)), dim3(fft_threads), filter_loop_lds_bytes<T>(m), sb, a, ppw, p2 ? ny : -1);
            }
            pit_done = p2;
        } else {
            hipLaunchKernelGGL(spu_filter_kernel_for<T>(m->cplan), dim3(a.j1 - a.j0, pairs), dim3(fft_threads), lds, sb, a);
        }
        if (p2 && !pit_done) {
            hipLaunchKernelGGL(pit2d_kernel_for<T>(m->cplan), dim3(a.j1 - a.j0), dim3(fft_threads), lds + sizeof(T) * (size_t)W, sb, a);
        } else if (!p2) {
            const long tiles = (long)((W + 255) / 256) * (a.j1 - a.j0);
            hipLaunchKernelGGL(pe_pit_kernel<T>, dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(256), 0, sb, a);
        }
        if (m->aux && !split_k1 && !ev_a_done) (void)hipEventRecord(m->ev_a, m->aux);      // (what K4 of the interior rows takes from this chain)
        // ---- chain A: geopotential of the own rows, then the filtered pressure-gradient force
        a.j0 = j0;
        a.j1 = j1;
        {
            PeArgsT<T> c = a;
            c.jb0 = c.jb1 = 0;
            c.cs_rows = 0;
            c.geo_j0 = j0; c.geo_j1 = j1;
            geopot(c, s);
        }
        // (a looping form of this filter, as K1's, was built and is 25 % SLOWER: its requests and the
        // per-column thermodynamics push it to 187 VGPRs, two waves per SIMD instead of four)
        // (launched with whole waves -- 192 threads for the 144 butterflies of a 1440 row, so that the
        // per-column thermodynamics ahead of the transform fills its lanes -- it takes the same time)
        const bool join = m->aux && mode == 1 && async_edges(m);                     // the edge rows' K4 on B takes pgfu
        if (join && m->stop_events)
            hipExtLaunchKernelGGL(pgf_filter_kernel_for<T>(m->cplan), dim3((unsigned)(8 * ((a.j1 - a.j0 + 7) / 8) * pairs)), dim3(fft_threads),
                                  (unsigned)lds, s, nullptr, m->ev_join, 0, a);
        else
            hipLaunchKernelGGL(pgf_filter_kernel_for<T>(m->cplan), dim3((unsigned)(8 * ((a.j1 - a.j0 + 7) / 8) * pairs)), dim3(fft_threads), lds, s, a);
        if (m->aux) {
            if (join && !m->stop_events) (void)hipEventRecord(m->ev_join, s);
            (void)hipStreamWaitEvent(s, m->ev_a, 0);                                   // K4 on A takes spu, pit (and a band's ghost anchors)
        }
    }
    auto update_rows = [&](int r0, int r1, int rb0, int rb1, hipStream_t st, hipEvent_t stop = nullptr) {   // rows [r0, r1) and [rb0, rb1), one launch
        const int rows = std::max(0, r1 - r0) + std::max(0, rb1 - rb0);
        if (rows <= 0) return;
        a.j0 = r0;
        a.j1 = std::max(r0, r1);
        a.jb0 = rb0;
        a.jb1 = std::max(rb0, rb1);
        const int Rg = m->upd_rows;
        const long groups = (std::max(0, r1 - r0) + Rg - 1) / Rg + (std::max(0, rb1 - rb0) + Rg - 1) / Rg;
        // 8 XCDs x (row group, segment) pairs per XCD x column tiles (see the kernel's index map)
        const long rs_per_xcd = (groups * a.nseg + 7) / 8;
        const dim3 gg((unsigned)(8 * rs_per_xcd * ((W + kUpdCols - 1) / kUpdCols)));
        const size_t lds = upd_lds_bytes<T>(Rg, L);
        const bool same = a.u == a.su;
        // whole columns of an even number of levels start on an odd level: the geopotential anchor is requested with the
        // even levels only (an odd level steps up from the anchor in the tile below and never reads its own): one request
        // in eleven (seven) less every other level, C4 1.881 -> 1.857 ms per step (round 4, A/B on one box)
        static const bool oddtop_env = !(getenv("GCM_PE_K4_ODDTOP") && getenv("GCM_PE_K4_ODDTOP")[0] == '0');
        const bool oddtop = oddtop_env && a.nseg == 1 && L % 2 == 0;
        if (stop && m->stop_events) hipExtLaunchKernelGGL(update_rows_kernel_for<T>(Rg, same, oddtop), gg, dim3(64 * (Rg + 1)), (unsigned)lds, st, nullptr, stop, 0, a);
        else hipLaunchKernelGGL(update_rows_kernel_for<T>(Rg, same, oddtop), gg, dim3(64 * (Rg + 1)), lds, st, a);
    };
    m->cs_valid[out_set] = p2;                   // (modes 1 + 2 together cover the rows)
    if (mode == 0) {
        tick(m, s);
        update_rows(j0, j1, 0, 0, s, m->ev_k4);
        m->k4_fork_valid = m->stop_events && !(m->ev && m->ev_used);       // (timing runs put records behind the kernel)
        tick(m, s);
    } else if (mode == 1) {
        // the rows the neighbours wait for (an unsplittable, tiny band: all of them).  With send
        // buffers registered they are updated and packed on the second stream (chain B), which waits
        // for K3 here; the caller's stream goes straight on to the interior rows (mode 2), so the
        // two launches share the chip.
        const bool as = async_edges(m);
        hipStream_t se = as && m->aux ? m->aux : s;
        if (split_k1) (void)hipStreamWaitEvent(se, m->ev_a, 0);      // the edge rows' partial sums and K4 take spu of own rows
        if (split && p2 && m->nseg_edge > 1) {
            // a band's edge rows are marched in level segments (below): the partial sums of conv they start from,
            // behind K1 on chain B and ahead of its wait for K3 (beside the interior rows' K4 this kernel took 50 us
            // instead of 14)
            PeArgsT<T> c = a;
            c.nseg = m->nseg_edge;
            c.j0 = j0; c.j1 = j0 + kGhost + 1;            // (K4 of row j also takes the sums of row j + 1)
            c.jb0 = j1 - kGhost; c.jb1 = j1 + 1;
            hipLaunchKernelGGL(pe_part_kernel<T>, dim3((unsigned)((W + 255) / 256) * (2 * kGhost + 2)), dim3(256), 0, se, c);
        }
        if (as && m->aux) (void)hipStreamWaitEvent(m->aux, m->ev_join, 0);      // the edge rows' K4 takes pgfu
        if (as && m->aux && m->edges_first && split) {
            (void)hipEventRecord(m->ev_pre_edge, se);          // chain B is about to launch the edge rows' K4
            m->pre_edge_pending = true;
        }
        if (split && p2 && m->nseg_edge > 1) {
            // the edge rows in level segments: a quarter of the chain of dependent levels, so the pack
            // and the exchange start while the interior rows are still at work
            PeArgsT<T> keep = a;
            a.nseg = m->nseg_edge;
            a.ocs_u = a.ocs_v = nullptr;
            // (the partial sums of conv they start from: pe_part_kernel, queued behind K1 above)
            update_rows(j0, j0 + kGhost, j1 - kGhost, j1, se);
            a = keep;
        } else if (split) {
            update_rows(j0, j0 + kGhost, j1 - kGhost, j1, se);
        } else {
            update_rows(j0, j1, 0, 0, se);
        }
        if (as) {
            SegCopy c{};
            std::string err;
            (void)pe25d_halo_segments(m, true, 0, m->send_buf[0], &c, &err);
            (void)pe25d_halo_segments(m, true, 1, m->send_buf[1], &c, &err);
            if (m->stop_events) launch_seg_copy(c, se, m->ev_edges);
            else {
                launch_seg_copy(c, se);
                (void)hipEventRecord(m->ev_edges, se);
            }
            m->edges_pending = true;
            m->edges_ev_valid = true;
            if (m->aux2 && split && p2 && m->nseg_edge > 1) {
                // the edge rows were marched in level segments and left no column sums: formed here, on the third
                // stream, as soon as the rows exist -- beside the interior rows still at work, off every chain of the
                // next stage (which read them: pit of rows 0 .. 2 and H - 2 .. H)
                (void)hipStreamWaitEvent(m->aux2, m->ev_edges, 0);
                PeArgsT<T> cc = make_args<T>(m, out_set, out_set, dt);
                cc.j0 = j0; cc.j1 = j0 + kGhost; cc.jb0 = j1 - kGhost; cc.jb1 = j1;
                if (m->stop_events)
                    hipExtLaunchKernelGGL(pe_colsum_kernel<T>, dim3((unsigned)((W + 255) / 256) * (2 * kGhost)), dim3(256), 0, m->aux2, nullptr, m->ev_cs, 0, cc);
                else {
                    hipLaunchKernelGGL(pe_colsum_kernel<T>, dim3((unsigned)((W + 255) / 256) * (2 * kGhost)), dim3(256), 0, m->aux2, cc);
                    (void)hipEventRecord(m->ev_cs, m->aux2);
                }
                m->edge_cs_set = out_set;
            }
        }
    } else {
        if (split) {
            if (m->pre_edge_pending) (void)hipStreamWaitEvent(s, m->ev_pre_edge, 0);
            m->pre_edge_pending = false;
            update_rows(j0 + kGhost, j1 - kGhost, 0, 0, s, m->ev_k4);
            m->k4_fork_valid = m->stop_events;
        }
        // whatever follows on the caller's stream also follows the edge rows
        if (async_edges(m) && m->edges_pending) (void)hipStreamWaitEvent(s, m->ev_edges, 0);
        m->edges_pending = false;
    }
}

static void half(Pe25d *m, int stage_set, int out_set, double dt, int j0, int j1, hipStream_t s, int mode = 0, bool chained = false) {
    // (a single domain's stages follow one another on `s` with nothing between them that chain B must wait for)
    const bool ch = chained || (mode == 0 && m->wrap);
    if (m->f32) half_t<float>(m, stage_set, out_set, dt, j0, j1, s, mode, ch);
    else half_t<double>(m, stage_set, out_set, dt, j0, j1, s, mode, ch);
}

// gcm_band_run, right behind the unpack on the second stream: the ghost rows that have just arrived belong to
// the state the NEXT stage reads; their column sums and the south ghost row's geopotential depend on nothing
// else, so they are queued here -- beside the interior rows' K4 of the stage still running -- instead of at the
// head of the next stage's chain B, where they were 17 us in front of K1.
int pe25d_prep_ghost_rows(Pe25d *m, std::string *err) {
    if (m->wrap || !m->aux) return GCM_OK;
    int set = m->star_valid ? 2 : m->cur_i;                      // the set the unpack has just filled (halo_t)
    if (m->pack_set >= 0 && m->pack_set != 2) set = m->pack_set;
    if (set != m->last_unpack_set) {
        // the two functions pick the set by the same rule; if they ever disagree the next stage would take its
        // ghost rows' column sums and anchors from rows nobody filled
        *err = "gcm_band_run: the ghost rows just unpacked are not those of the state the next stage reads";
        return GCM_ERR_STATE;
    }
    const bool p2 = m->pit2d && m->nseg == 1;
    if (p2 && !m->cs_valid[set]) return GCM_OK;                  // (a fresh state: the stage does all rows itself)
    if (m->f32) {
        PeArgsT<float> a = make_args<float>(m, set, set, 0.0);
        a.j0 = 0; a.j1 = m->H + 1;
        prep_rows<float>(m, a, set, p2, m->H, 1, m->aux);
    } else {
        PeArgsT<double> a = make_args<double>(m, set, set, 0.0);
        a.j0 = 0; a.j1 = m->H + 1;
        prep_rows<double>(m, a, set, p2, m->H, 1, m->aux);
    }
    m->ghost_ready = set;
    return GCM_OK;
}

int pe25d_half(Pe25d *m, int stage, double dt, hipStream_t s, std::string *err) {
    if (!m->wrap) {
        *err = "half_step on a latitude band: use step_part";
        return GCM_ERR_UNSUPPORTED;
    }
    if (stage == 0) {
        half(m, m->cur_i, 2, dt, 0, m->H, s);
        m->star_valid = true;
    } else {
        if (!m->star_valid) {
            *err = "half_step(1) before half_step(0)";
            return GCM_ERR_STATE;
        }
        half(m, 2, 1 - m->cur_i, dt, 0, m->H, s);
        m->cur_i = 1 - m->cur_i;
        m->star_valid = false;
    }
    if (hipGetLastError() != hipSuccess) {
        *err = "hip: pe25d kernel launch failed";
        return GCM_ERR_HIP;
    }
    return GCM_OK;
}

int pe25d_step(Pe25d *m, double dt, hipStream_t s, std::string *err) {
    if (!m->wrap) {
        *err = "gcm_step: a latitude band needs ghost-row exchanges inside the step (use step_part)";
        return GCM_ERR_STATE;
    }
    half(m, m->cur_i, 2, dt, 0, m->H, s);
    half(m, 2, 1 - m->cur_i, dt, 0, m->H, s);
    m->cur_i = 1 - m->cur_i;
    m->star_valid = false;
    if (hipGetLastError() != hipSuccess) {
        *err = "hip: pe25d kernel launch failed";
        return GCM_ERR_HIP;
    }
    return GCM_OK;
}

// Latitude band: part 0 = predictor (needs ghost rows of the current state),
// part 1 = corrector (needs ghost rows of the predicted state), then the swap.
int pe25d_step_part(Pe25d *m, int part, double dt, hipStream_t s, std::string *err) {
    if (m->wrap) {
        *err = "step_part: handle is not a latitude band";
        return GCM_ERR_STATE;
    }
    if (part == 0) {
        half(m, m->cur_i, 2, dt, 0, m->H, s);
        m->star_valid = true;
    } else {
        half(m, 2, 1 - m->cur_i, dt, 0, m->H, s);
        m->cur_i = 1 - m->cur_i;
        m->star_valid = false;
    }
    if (hipGetLastError() != hipSuccess) {
        *err = "hip: pe25d kernel launch failed";
        return GCM_ERR_HIP;
    }
    return GCM_OK;
}

// Latitude band with the exchange hidden behind the interior rows of K4 (gcm_step_phase):
//   phase 0  predictor K1-K3 + K4 edge rows   -> the predicted edge rows can be sent
//   phase 1  predictor K4 interior rows
//   phase 2  corrector K1-K3 + K4 edge rows   -> the new state's edge rows can be sent
//   phase 3  corrector K4 interior rows, then the swap
// gcm_halo_pack after phase 0 / 2 packs the rows just produced; gcm_halo_unpack after phase 1 / 3
// fills the ghost rows of the predicted / the new current state.
int pe25d_step_phase(Pe25d *m, int phase, double dt, hipStream_t s, std::string *err, bool chained) {
    if (m->wrap) {
        *err = "step_phase: handle is not a latitude band";
        return GCM_ERR_STATE;
    }
    switch (phase) {
        case 0:
            m->star_valid = true;
            m->pack_set = 2;
            half(m, m->cur_i, 2, dt, 0, m->H, s, 1, chained);
            break;
        case 1:
            half(m, m->cur_i, 2, dt, 0, m->H, s, 2, chained);
            break;
        case 2:
            m->pack_set = 1 - m->cur_i;
            half(m, 2, 1 - m->cur_i, dt, 0, m->H, s, 1, chained);
            break;
        case 3:
            half(m, 2, 1 - m->cur_i, dt, 0, m->H, s, 2, chained);
            m->cur_i = 1 - m->cur_i;
            m->star_valid = false;
            m->pack_set = -1;
            break;
        default:
            *err = "step_phase: phase must be 0..3";
            return GCM_ERR_ARG;
    }
    if (hipGetLastError() != hipSuccess) {
        *err = "hip: pe25d kernel launch failed";
        return GCM_ERR_HIP;
    }
    return GCM_OK;
}

int pe25d_set_halo_buffers(Pe25d *m, void *north, void *south, hipStream_t s, std::string *err) {
    if (m->wrap) {
        *err = "set_halo_buffers: handle is not a latitude band";
        return GCM_ERR_STATE;
    }
    if ((north == nullptr) != (south == nullptr)) {
        *err = "set_halo_buffers: give both buffers, or neither to unregister";
        return GCM_ERR_ARG;
    }
    (void)hipStreamSynchronize(s);
    if (m->aux) (void)hipStreamSynchronize(m->aux);
    if (m->aux2) (void)hipStreamSynchronize(m->aux2);
    m->send_buf[0] = north;
    m->send_buf[1] = south;
    m->edges_pending = false;
    return GCM_OK;
}

int pe25d_wait_edges(Pe25d *m, hipStream_t s, std::string *err) {
    if (!async_edges(m)) {
        *err = "wait_edges: no send buffers registered (gcm_set_halo_buffers)";
        return GCM_ERR_STATE;
    }
    if (hipStreamWaitEvent(s, m->ev_edges, 0) != hipSuccess) {
        *err = "hip: wait_edges failed";
        return GCM_ERR_HIP;
    }
    return GCM_OK;
}

// ghost rows: [p: 2 rows][u,v,t,q: 2 rows x L levels]; contiguous in the device layout.
// Which state is exchanged follows the step phase: the predicted state once it exists.
size_t pe25d_halo_bytes(const Pe25d *m) {
    // (+ the ground temperature's two rows, float64 for either storage type: gcm_set_physics)
    return (m->f32 ? sizeof(float) : sizeof(double)) * (size_t)kGhost * m->W * (1 + 4 * (size_t)m->L) +
           sizeof(double) * (size_t)kGhost * m->W;
}

template <typename T>
static void halo_t(Pe25d *m, bool pack, int side, void *dev_buf, SegCopy *c) {
    PeBufs<T> &Bf = bufs<T>(m);
    // unpack: ghosts of the predicted state once it exists, else of the current state;
    // pack: the same, unless a step_phase call named the set whose edge rows were just produced
    int set = m->star_valid ? 2 : m->cur_i;
    if (pack && m->pack_set >= 0) set = m->pack_set;
    if (!pack && m->pack_set >= 0 && m->pack_set != 2) set = m->pack_set;   // new-state ghosts arrive before the swap
    if (!pack) m->last_unpack_set = set;
    T *b = (T *)dev_buf;
    for (int f = 0; f < GCM_NFIELDS; ++f) {
        const size_t per_row = (size_t)m->W * (f == GCM_P ? 1 : m->L);
        const long n = (long)(kGhost * per_row);
        T *base = Bf.st[set][f];
        T *edge = side == 0 ? base : base + (size_t)(m->H - kGhost) * per_row;
        T *ghost = side == 0 ? base - (size_t)kGhost * per_row : base + (size_t)m->H * per_row;
        // SegCopy moves 8-byte words: 2 rows x (even W) floats is a whole number of them
        c->src[c->nseg] = (const double *)(pack ? edge : b);
        c->dst[c->nseg] = (double *)(pack ? b : ghost);
        c->n[c->nseg++] = n * (long)sizeof(T) / 8;
        b += n;
    }
    // the ground temperature: one array for all state sets, advanced by the column physics only.  A band's
    // ghost rows of it are radiated locally (pe25d_solar_rows), so what a message carries equals what the
    // ghost rows hold already -- except in the first exchange after gcm_set_ground, which is what it is for.
    {
        double *gb = (double *)b;                 // (2 rows x even W floats: a whole number of 8-byte words)
        const size_t n2 = (size_t)kGhost * m->W;
        double *edge = side == 0 ? m->gt : m->gt + (size_t)(m->H - kGhost) * m->W;
        double *ghost = side == 0 ? m->gt - n2 : m->gt + (size_t)m->H * m->W;
        c->src[c->nseg] = pack ? edge : gb;
        c->dst[c->nseg] = pack ? gb : ghost;
        c->n[c->nseg++] = (long)n2;
    }
}

// appends the copies of one side to *c (the caller launches them: one side or both in one launch)
int pe25d_halo_segments(Pe25d *m, bool pack, int side, void *dev_buf, SegCopy *c, std::string *err) {
    if (m->f32 && (m->W % 2)) {
        *err = "pe25d halo: fp32 bands need an even width";
        return GCM_ERR_UNSUPPORTED;
    }
    if (m->f32) halo_t<float>(m, pack, side, dev_buf, c);
    else halo_t<double>(m, pack, side, dev_buf, c);
    return GCM_OK;
}

// low_pass.arakawa_1977 on a field of the handle's grid, nlev <= L levels, host [nlev][H][W] in and
// out: the spu filter kernel with iph(sp) = 1 (su * 1 is exact).  spu, pgfu and pit serve as scratch
// -- every stage rewrites them before it reads them.
template <typename T>
__global__ void pe_fill_kernel(T *dst, long n, T x) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) dst[e] = x;
}
template <typename T>
static int filter_field_t(Pe25d *m, int nlev, const double *in, double *out, hipStream_t s, std::string *err) {
    PeBufs<T> &B = bufs<T>(m);
    const int W = m->W, H = m->H;
    const size_t bytes = sizeof(double) * (size_t)nlev * H * W;
    m->last_stage_set = -1;                      // spu, pgfu and pit are scratch here: the parity tap has nothing to return
    m->k4_fork_valid = false;
    hipError_t e = hipMemcpyAsync(m->stage3, in, bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pe_to_device_kernel<T>, dim3(1024), dim3(256), 0, s, B.pgfu, m->stage3, W, H, nlev);
        hipLaunchKernelGGL(pe_fill_kernel<T>, dim3(256), dim3(256), 0, s, B.pit, (long)H * W, T(1.0));
        PeArgsT<T> a = make_args<T>(m, m->cur_i, m->cur_i, 0.0);
        a.L = nlev;
        a.sp = B.pit;
        a.su = B.pgfu;
        a.spu = B.spu;
        a.filter = 1;
        a.j0 = 0;
        a.j1 = H;
        const int fft_threads = m->cplan.ok ? m->cplan.threads : kFftThreads;
        hipLaunchKernelGGL(spu_filter_kernel_for<T>(m->cplan), dim3(H, (nlev + 1) / 2), dim3(fft_threads),
                           filter_lds_bytes<T>(m), s, a);
        hipLaunchKernelGGL(pe_to_host_kernel<T>, dim3(1024), dim3(256), 0, s, m->stage3, B.spu, W, H, nlev);
        e = hipMemcpyAsync(out, m->stage3, bytes, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        *err = std::string("pe25d polar filter: ") + hipGetErrorString(e);
        return GCM_ERR_HIP;
    }
    return GCM_OK;
}

// ---------------------------------------------------------------- parity tap: the stage's intermediates
// phi on every level in the host layout [k][j][i], float64: the stored anchors on the even levels, the odd
// levels stepped up from them with phi_up -- the expression K3 and K4 evaluate
template <typename T>
__global__ __launch_bounds__(256) void pe_phi_full_kernel(PeArgsT<T> a, double *out) {
    __shared__ double tab[kExnerTabDoubles];
    for (int n = threadIdx.x; n < kExnerTabDoubles; n += 256) tab[n] = a.exner_tab[n];
    __syncthreads();
    const int W = a.W, H = a.H, L = a.L;
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= W) return;
    const long c3 = (long)j * L * W + i;
    const T spc = a.sp[(long)j * W + i];
    T phi_lo = T(0.0), t_lo = T(0.0), ex_lo = T(0.0);
    for (int k = 0; k < L; ++k) {
        const T t = a.st[c3 + (long)k * W];
        const T ex = exner(spc * a.sig[k] + a.ptop, tab);
        const T phi = (k & 1) ? phi_up(phi_lo, t_lo, t, ex_lo, ex) : a.phi[c3 + (long)k * W];
        out[((long)k * H + j) * W + i] = (double)phi;
        phi_lo = phi; t_lo = t; ex_lo = ex;
    }
}

template <typename T>
static int intermediate_t(Pe25d *m, int kind, double *out, hipStream_t s, std::string *err) {
    PeBufs<T> &B = bufs<T>(m);
    const int W = m->W, H = m->H, L = m->L;
    const T *src = nullptr;
    int lev = L;
    switch (kind) {
        case GCM_INT_SPU: src = B.spu; break;
        case GCM_INT_PGFU: src = B.pgfu; break;
        case GCM_INT_PIT: src = B.pit; lev = 1; break;
        case GCM_INT_PN: src = B.pn; lev = 1; break;
        case GCM_INT_PHI: break;
        default: *err = "gcm_get_intermediate: unknown kind"; return GCM_ERR_ARG;
    }
    if (kind == GCM_INT_PHI) {
        PeArgsT<T> a = make_args<T>(m, m->last_stage_set, m->last_stage_set, 0.0);
        hipLaunchKernelGGL(pe_phi_full_kernel<T>, dim3((W + 255) / 256, H), dim3(256), 0, s, a, m->stage3);
    } else {
        hipLaunchKernelGGL(pe_to_host_kernel<T>, dim3(1024), dim3(256), 0, s, m->stage3, src, W, H, lev);
    }
    hipError_t e = hipMemcpyAsync(out, m->stage3, sizeof(double) * (size_t)lev * H * W, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        *err = std::string("gcm_get_intermediate: ") + hipGetErrorString(e);
        return GCM_ERR_HIP;
    }
    return GCM_OK;
}

int pe25d_intermediate(Pe25d *m, int kind, double *out, hipStream_t s, std::string *err) {
    if (!m->wrap) { *err = "gcm_get_intermediate: single band only"; return GCM_ERR_UNSUPPORTED; }
    if (m->last_stage_set < 0) { *err = "gcm_get_intermediate: no half step taken yet"; return GCM_ERR_STATE; }
    if (m->aux) (void)hipStreamSynchronize(m->aux);
    return m->f32 ? intermediate_t<float>(m, kind, out, s, err) : intermediate_t<double>(m, kind, out, s, err);
}

int pe25d_filter_field(Pe25d *m, int nlev, const double *in, double *out, hipStream_t s, std::string *err) {
    if (nlev < 1 || nlev > m->L) {
        *err = "polar filter: 1 <= levels <= the handle's layers";
        return GCM_ERR_ARG;
    }
    if (!m->cfg.filter && m->W > 1) {
        *err = "polar filter: the handle was created with filter = 0";
        return GCM_ERR_UNSUPPORTED;
    }
    if (m->W == 1) {                                    // low_pass.py:58-59: identity
        if (out != in) memcpy(out, in, sizeof(double) * (size_t)nlev * m->H);
        return GCM_OK;
    }
    return m->f32 ? filter_field_t<float>(m, nlev, in, out, s, err) : filter_field_t<double>(m, nlev, in, out, s, err);
}

int pe25d_ground(Pe25d *m, bool set, const double *in, double *out, hipStream_t s, std::string *err) {
    const size_t bytes = sizeof(double) * (size_t)m->H * m->W;
    hipError_t e = set ? hipMemcpyAsync(m->gt, in, bytes, hipMemcpyHostToDevice, s)
                       : hipMemcpyAsync(out, m->gt, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { *err = "hip: ground temperature transfer failed"; return GCM_ERR_HIP; }
    if (set) {
        m->gt_set = true;
        m->k4_fork_valid = false;
    }
    return GCM_OK;
}

// rows [j0, j1) and [jb0, jb1) of state set `set` (a band's ghost rows: negative, or >= H); keep_ghosts: the caller
// radiates the ghost rows itself before their column sums and anchors are queued (gcm_band_run), so what
// pe25d_prep_ghost_rows left stays valid
template <typename T>
static int radiation_launch(Pe25d *m, int set, int j0, int j1, int jb0, int jb1, bool keep_ghosts, bool apply, double dt,
                            double hour_angle, double albedo, double *dTdt_host, double *dtg_host, hipStream_t s,
                            std::string *err) {
    PeBufs<T> &B = bufs<T>(m);
    const int W = m->W, H = m->H, L = m->L, Hg = m->Hg;
    const int nrows = std::max(0, j1 - j0) + std::max(0, jb1 - jb0);
    if (nrows <= 0) return GCM_OK;
    PeArgsT<T> a = make_args<T>(m, set, set, dt);
    a.p = B.st[set][GCM_P];                               // (make_args takes the base state from the current set)
    RadArgsT<T> r{};
    r.j0 = j0; r.n0 = std::max(0, j1 - j0); r.jb0 = jb0;
    r.tlw = m->rad_tab; r.tsw = r.tlw + L; r.csw_top = r.tsw + L; r.clw_b_div = r.csw_top + L; r.swfac = r.clw_b_div + L;
    r.sigk = r.swfac + L;
    r.coslat = m->rad_geo; r.sinlat = m->rad_geo + Hg; r.lon = m->rad_geo + 2 * Hg;
    r.gt = m->gt;
    r.dTdt = B.pgfu; r.dtg = B.pit;
    r.hour_angle = hour_angle;
    r.albedo = albedo; r.dt = dt; r.apply = apply ? 1 : 0;
    if (apply && !keep_ghosts) m->ghost_ready = -1;        // theta changes in place
    m->last_stage_set = -1;                                // gcm_get_intermediate: theta changed, or pgfu / pit hold the tendencies
    // the diagnostic form writes pgfu / pit on `s`, and a band's explicit solar_timestep changes the ghost rows' theta
    // there: the next stage's chain B (K1 + pit, the ghost rows' anchors) follows the stream's position, not just the last K4
    if (!(apply && (keep_ghosts || m->wrap))) m->k4_fork_valid = false;
    {
        const dim3 gg((W + kRadThreads - 1) / kRadThreads, nrows);
        T *th = B.st[set][GCM_T];
        static const bool generic = getenv("GCM_PE_RAD_GENERIC") != nullptr;     // diagnostic: the LDS-parked form
        const bool fact = m->cfg.ptop == 0.0 && r.sigk != nullptr;
        if (L <= 24 && !generic && fact) hipLaunchKernelGGL((pe_radiation_kernel<T, 24, true>), gg, dim3(kRadThreads), 0, s, a, r, th);
        else if (L <= 40 && !generic && fact) hipLaunchKernelGGL((pe_radiation_kernel<T, 40, true>), gg, dim3(kRadThreads), 0, s, a, r, th);
        else if (L <= 24 && !generic) hipLaunchKernelGGL((pe_radiation_kernel<T, 24>), gg, dim3(kRadThreads), 0, s, a, r, th);
        else if (L <= 40 && !generic) hipLaunchKernelGGL((pe_radiation_kernel<T, 40>), gg, dim3(kRadThreads), 0, s, a, r, th);
        else hipLaunchKernelGGL((pe_radiation_kernel<T, 0>), gg, dim3(kRadThreads), sizeof(double) * (size_t)L * kRadThreads, s, a, r, th);
    }
    if (hipGetLastError() != hipSuccess) { *err = "hip: radiation kernel launch failed"; return GCM_ERR_HIP; }
    // solar_timestep (apply) stays asynchronous on `s`; the diagnostics form copies its results back
    if (dtg_host) {
        hipLaunchKernelGGL(pe_to_host_kernel<T>, dim3(64), dim3(256), 0, s, m->stage3, B.pit, W, H, 1);
        if (hipMemcpyAsync(dtg_host, m->stage3, sizeof(double) * (size_t)H * W, hipMemcpyDeviceToHost, s) != hipSuccess) {
            *err = "hip: dt_ground copy-back failed"; return GCM_ERR_HIP;
        }
    }
    if (dTdt_host) {
        hipLaunchKernelGGL(pe_to_host_kernel<T>, dim3(1024), dim3(256), 0, s, m->stage3, B.pgfu, W, H, L);
        if (hipMemcpyAsync(dTdt_host, m->stage3, sizeof(double) * (size_t)H * W * L, hipMemcpyDeviceToHost, s) != hipSuccess) {
            *err = "hip: dTdt copy-back failed"; return GCM_ERR_HIP;
        }
    }
    if ((dtg_host || dTdt_host) && hipStreamSynchronize(s) != hipSuccess) {
        *err = "hip: radiation kernel failed"; return GCM_ERR_HIP;
    }
    return GCM_OK;
}

// the level tables (t_lw, t_sw) and the lat / lon tables of the radiation kernel, uploaded when they change
int pe25d_physics_tables(Pe25d *m, double t_lw, double t_sw, const double *lat, const double *lon, hipStream_t s,
                         std::string *err) {
    if (!m->gt_set) { *err = "radiation: set the ground temperature first (gcm_set_ground)"; return GCM_ERR_STATE; }
    if (!lat || !lon) { *err = "radiation: lat and lon tables are required"; return GCM_ERR_ARG; }
    const int W = m->W, L = m->L, Hg = m->Hg;
    bool uploaded = false;
    if (m->rad_key[0] != t_lw || m->rad_key[1] != t_sw || !m->rad_tab) {
        // level tables, same expression order as grey_solar.py:323-333,377-385,541
        std::vector<double> &T = m->rad_tab_host;
        T.assign((size_t)6 * L, 0.0);
        const std::vector<double> &dsig = m->dsig_host;
        double *tlw = T.data(), *tsw = tlw + L, *csw = tsw + L, *cdiv = csw + L, *swf = cdiv + L;
        for (int k = 0; k < L; ++k) {
            tlw[k] = 1 - (1 - std::pow(t_lw, dsig[k]));
            tsw[k] = 1 - (1 - std::pow(t_sw, dsig[k]));
        }
        double c = 1.0;
        for (int k = L - 1; k >= 0; --k) { c = k == L - 1 ? tsw[k] : c * tsw[k]; csw[k] = c; }
        for (int k = 0; k < L; ++k) { c = k == 0 ? tlw[k] : c * tlw[k]; cdiv[k] = c / tlw[k]; }
        for (int k = 0; k < L; ++k) swf[k] = (1 - tsw[k]) * csw[k] / tsw[k];
        for (int k = 0; k < L; ++k) swf[L + k] = (double)powl((long double)m->sig_host[k], (long double)kKappa);   // sig^kappa (FACT)
        if (!m->rad_tab && !dev_upload<double>(m, &m->rad_tab, nullptr, (size_t)6 * L)) {
            *err = "hip: radiation table allocation failed"; return GCM_ERR_HIP;
        }
        // the host copy lives in the handle until the next change, so the asynchronous upload may
        // read it after this call returns; a change waits for the previous upload first
        if (hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpyAsync(m->rad_tab, T.data(), sizeof(double) * 6 * L, hipMemcpyHostToDevice, s) != hipSuccess) {
            *err = "hip: radiation table upload failed"; return GCM_ERR_HIP;
        }
        m->rad_key[0] = t_lw; m->rad_key[1] = t_sw;
        uploaded = true;
    }
    // lat / lon tables: uploaded when their content changes (normally once)
    if (m->rad_latlon.size() != (size_t)Hg + W || memcmp(m->rad_latlon.data(), lat, sizeof(double) * Hg) ||
        memcmp(m->rad_latlon.data() + Hg, lon, sizeof(double) * W)) {
        if (hipStreamSynchronize(s) != hipSuccess) { *err = "hip: radiation geometry upload failed"; return GCM_ERR_HIP; }
        m->rad_latlon.assign(lat, lat + Hg);
        m->rad_latlon.insert(m->rad_latlon.end(), lon, lon + W);
        std::vector<double> &Gt = m->rad_geo_host;
        Gt.assign((size_t)2 * Hg + W, 0.0);
        for (int j = 0; j < Hg; ++j) { Gt[j] = std::cos(lat[j]); Gt[Hg + j] = std::sin(lat[j]); }
        for (int i = 0; i < W; ++i) Gt[2 * Hg + i] = lon[i];
        if (!m->rad_geo && !dev_upload<double>(m, &m->rad_geo, nullptr, Gt.size())) {
            *err = "hip: radiation geometry allocation failed"; return GCM_ERR_HIP;
        }
        if (hipMemcpyAsync(m->rad_geo, Gt.data(), sizeof(double) * Gt.size(), hipMemcpyHostToDevice, s) != hipSuccess) {
            *err = "hip: radiation geometry upload failed"; return GCM_ERR_HIP;
        }
        uploaded = true;
    }
    // (a band radiates its ghost rows on the second stream: the tables are in place before anything is queued there)
    if (uploaded && hipStreamSynchronize(s) != hipSuccess) { *err = "hip: radiation table upload failed"; return GCM_ERR_HIP; }
    return GCM_OK;
}

// solar_timestep (no_limits_2_5d.py:66-75) of rows [j0, j1) and [jb0, jb1) of state set `set` (-1: the current one) on stream `s`;
// the tables must be in place (pe25d_physics_tables)
int pe25d_solar_rows(Pe25d *m, int set, int j0, int j1, int jb0, int jb1, bool keep_ghosts, double dt, double utc, double albedo,
                     hipStream_t s, std::string *err) {
    if (set < 0) set = m->cur_i;
    const double hour_angle = utc / (-24 * 3600.0) * 360 * (M_PI / 180);      // grey_solar.py:51
    return m->f32 ? radiation_launch<float>(m, set, j0, j1, jb0, jb1, keep_ghosts, true, dt, hour_angle, albedo, nullptr, nullptr, s, err)
                  : radiation_launch<double>(m, set, j0, j1, jb0, jb1, keep_ghosts, true, dt, hour_angle, albedo, nullptr, nullptr, s, err);
}

// basic_grey_radiation (+ optional in-place solar_timestep).  dTdt_host / dtg_host may be null.  On a latitude
// band the in-place form advances the ghost rows too (their theta as the post-corrector exchange delivered it,
// their ground temperature as the last message delivered it): the neighbour's own inputs, the neighbour's own bits.
int pe25d_radiation(Pe25d *m, bool apply, double dt, double utc, double t_lw, double t_sw, double albedo,
                    const double *lat, const double *lon, double *dTdt_host, double *dtg_host,
                    hipStream_t s, std::string *err) {
    int rc = pe25d_physics_tables(m, t_lw, t_sw, lat, lon, s, err);
    if (rc) return rc;
    const double hour_angle = utc / (-24 * 3600.0) * 360 * (M_PI / 180);      // grey_solar.py:51
    const int g = (apply && !m->wrap) ? kGhost : 0;
    return m->f32 ? radiation_launch<float>(m, m->cur_i, -g, m->H + g, 0, 0, false, apply, dt, hour_angle, albedo, dTdt_host, dtg_host, s, err)
                  : radiation_launch<double>(m, m->cur_i, -g, m->H + g, 0, 0, false, apply, dt, hour_angle, albedo, dTdt_host, dtg_host, s, err);
}

// calc_energy + STATS in one launch and one synchronisation of `s`: out9 = u_max, u_min, v_max,
// v_min, ke, ate, geo, total, NaN count.  The area table and the partials live in the handle.
int pe25d_stats(Pe25d *m, const double *area_host, int area_len, double out[9], hipStream_t s, std::string *err) {
    if (m->f32) { *err = "gcm_energy / gcm_stats: fp64 handles only"; return GCM_ERR_UNSUPPORTED; }
    if (!m->wrap) { *err = "gcm_energy / gcm_stats: single band only"; return GCM_ERR_UNSUPPORTED; }
    if (!(area_len == 1 || area_len == m->W)) {
        *err = "gcm_energy: geom.area (H,) must broadcast against the last axis W (no_limits_2_5d.py:49): "
               "needs H == W or H == 1";
        return GCM_ERR_ARG;
    }
    const int gx = (m->W + 255) / 256, nb = gx * m->H;
    if (!m->stats_dev) {
        if (!dev_upload<double>(m, &m->stats_dev, nullptr, (size_t)kStatsWords * nb + m->W)) {
            *err = "hip: gcm_stats allocation failed";
            return GCM_ERR_HIP;
        }
        m->stats_host.resize((size_t)kStatsWords * nb);
    }
    double *d_area = m->stats_dev + (size_t)kStatsWords * nb;
    if (m->area_host.size() != (size_t)area_len || memcmp(m->area_host.data(), area_host, sizeof(double) * area_len)) {
        m->area_host.assign(area_host, area_host + area_len);
        if (hipMemcpyAsync(d_area, m->area_host.data(), sizeof(double) * area_len, hipMemcpyHostToDevice, s) != hipSuccess) {
            *err = "hip: gcm_stats area upload failed";
            return GCM_ERR_HIP;
        }
    }
    PeArgs a = make_args<double>(m, m->cur_i, m->cur_i, 0.0);
    hipLaunchKernelGGL(pe_energy_kernel, dim3(gx, m->H), dim3(256), 0, s, a, d_area, area_len > 1 ? 1 : 0, m->stats_dev);
    double *part = m->stats_host.data();
    if (hipMemcpyAsync(part, m->stats_dev, sizeof(double) * kStatsWords * nb, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        *err = "hip: gcm_stats kernel failed";
        return GCM_ERR_HIP;
    }
    double ke = 0, ate = 0, geo = 0, nn = 0, umax = -INFINITY, umin = INFINITY, vmax = -INFINITY, vmin = INFINITY;
    for (int b = 0; b < nb; ++b) {
        const double *o = part + (size_t)kStatsWords * b;
        ke += o[0]; ate += o[1]; geo += o[2]; nn += o[7];
        umax = std::fmax(umax, o[3]); umin = std::fmin(umin, o[4]);
        vmax = std::fmax(vmax, o[5]); vmin = std::fmin(vmin, o[6]);
    }
    // np.max / np.min propagate NaN
    out[0] = nn > 0 ? NAN : umax; out[1] = nn > 0 ? NAN : umin; out[2] = nn > 0 ? NAN : vmax; out[3] = nn > 0 ? NAN : vmin;
    out[4] = ke; out[5] = ate; out[6] = geo; out[7] = ke + ate + geo; out[8] = nn;
    return GCM_OK;
}

// field geometry for get_total_variation (axis 0 of the reference layout): 2-D p differences rows,
// the 3-D fields difference levels inside a row slab
int pe25d_filter_plan(int n, unsigned *out, int cap) {
    if (!out || n < 2 || cap < 3 + 4 * kMaxSuper) return GCM_ERR_ARG;
    SuperPlan P;
    make_super_plan(n, &P);
    out[0] = (unsigned)P.ok;
    out[1] = (unsigned)P.npass;
    out[2] = (unsigned)P.threads;
    for (int p = 0; p < kMaxSuper; ++p) {
        out[3 + 4 * p] = (unsigned)P.r1[p];
        out[4 + 4 * p] = (unsigned)P.r2[p];
        out[5 + 4 * p] = P.magic[p];
        out[6 + 4 * p] = P.imagic[p];
    }
    return GCM_OK;
}

void pe25d_tv_shape(const Pe25d *m, int field, long *n_outer, long *n_axis, long *n_inner, int *wrap) {
    if (field == GCM_P) { *n_outer = 1; *n_axis = m->H; *n_inner = m->W; *wrap = m->wrap ? 1 : 0; }
    else { *n_outer = m->H; *n_axis = m->L; *n_inner = m->W; *wrap = 1; }
}

// current-state field for the diagnostics reductions; *f32 tells the element type
const void *pe25d_field(Pe25d *m, int field, long *n, int *f32) {
    *n = (long)m->H * m->W * (field == GCM_P ? 1 : m->L);
    *f32 = m->f32 ? 1 : 0;
    return m->f32 ? (const void *)m->f.st[m->cur_i][field] : (const void *)m->d.st[m->cur_i][field];
}

void pe25d_timing(Pe25d *m, std::vector<hipEvent_t> *ev, size_t *used) {
    m->ev = ev;
    m->ev_used = used;
}

}  // namespace gcm

