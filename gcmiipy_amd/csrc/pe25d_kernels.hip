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
// real-type specific pieces: reciprocal and (p/P0)**kappa (fp32: v_rcp_f32 is 1 ulp; powf)
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exner(float p, const double *) { return __powf(p * 1e-5f, (float)kKappa); }

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

// ---------------------------------------------------------------- K1: spu = filter(su * iph(sp))
constexpr int kFftThreads = 256;    // generic path; the composite path sizes the workgroup from its plan
template <typename T, int MAXR, unsigned MASK = 0>
__global__ __launch_bounds__(512) void pe_spu_filter_kernel(PeArgsT<T> a) {
    using V = typename Vec2<T>::type;
    extern __shared__ unsigned char lds_raw[];
    V *x = (V *)lds_raw;
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int j = a.j0 + blockIdx.x;
    const int k0 = 2 * blockIdx.y, k1 = k0 + 1;
    const bool two = k1 < a.L;
    const int W = a.W;
    const T *sp = a.sp + ix.r2(j);
    const T *su0 = a.su + ix.r3(j) + (long)k0 * W;
    const T *su1 = su0 + W;
    T *o0 = a.spu + ix.r3(j) + (long)k0 * W;
    const auto load = [=](int i, int = 0) {
        const int ie = i + 1 == W ? 0 : i + 1;
        const T pe = (sp[i] + sp[ie]) * T(0.5);      // iph(p), dynamics.py:15-17
        return mkv<V>(su0[i] * pe, two ? su1[i] * pe : T(0.0));
    };
    const auto store = [=](int i, V v) {
        o0[i] = v.x;
        if (two) o0[W + i] = v.y;
    };
    if (a.filter && W > 1) {
        if (MAXR > 0) {
            const int jg = wrapi(a.row0 + j, a.Hg);
            filter_rows_composite<MAXR, MASK, T>(x, load, store, a.tw, a.cplan, W, a.smul + (long)jg * (W / 2 + 1));
        } else {
            for (int i = threadIdx.x; i < W; i += blockDim.x) x[i] = load(i);
            __syncthreads();
            const V *res = filter_rows<T>(x, x + W, a.tw, a.plan, a.smul + (long)wrapi(a.row0 + j, a.Hg) * (W / 2 + 1));
            for (int i = threadIdx.x; i < W; i += blockDim.x) store(i, res[i]);
        }
    } else {
        for (int i = threadIdx.x; i < W; i += blockDim.x) store(i, load(i));
    }
}

// K1, looping form: the workgroup of (row, group of level pairs) filters its pairs one after the
// other.  What the passes fetch from tables (their twiddles, the filter multiplier) and iph(sp) depend
// on the thread and the row only and are fetched once; the su values of the NEXT pair are requested
// before the current pair is transformed (two register sets that swap by name), so the only waits
// left inside the loop are LDS round trips and barriers.
// NIN: radix of the plan's first pass (inputs per thread) where the instantiation knows it, else MAXR
template <typename T, int MAXR, unsigned MASK = 0, int NIN = MAXR>
__global__ __launch_bounds__(512) void pe_spu_filter_loop_kernel(PeArgsT<T> a, int pairs_per_wg) {
    using V = typename Vec2<T>::type;
    extern __shared__ unsigned char lds_raw[];
    V *x = (V *)lds_raw;
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W, L = a.L;
    const int j = a.j0 + blockIdx.x;
    const int npairs = (L + 1) / 2;
    const int pb0 = blockIdx.y * pairs_per_wg, pb1 = min(pb0 + pairs_per_wg, npairs);
    if (pb0 >= pb1) return;                                      // uniform
    const int jg = wrapi(a.row0 + j, a.Hg);
    // LDS: the complex row, then iph(sp) of the row and the row's filter multiplier / W
    T *pe = (T *)(x + W), *sl = pe + W;
    const int nb0 = W / (a.cplan.r1[0] * a.cplan.r2[0]);
    const T *su_row = a.su + ix.r3(j);
    T *out_row = a.spu + ix.r3(j);
    V in[NIN];
    // unconditional requests (an odd L's last pair reads its single level twice; the copy is not stored)
    const auto request = [&](int pair, int tid) {
        const int k0 = 2 * pair;
        const T *s0 = su_row + (long)k0 * W, *s1 = s0 + (k0 + 1 < L ? W : 0);
#pragma unroll
        for (int m = 0; m < NIN; ++m) {
            const int i = min(tid + m * nb0, W - 1);
            in[m] = mkv<V>(s0[i], s1[i]);
        }
    };
    request(pb0, threadIdx.x);                                   // travels while the row's tables are made
    {
        // (four columns of a thread requested at a time: one memory latency per batch, not per column)
        const T *sp = a.sp + ix.r2(j), *S = a.smul + (long)jg * (W / 2 + 1);
        const T inv_n = T(1.0) / (T)W;
        constexpr int kB = 4;
        for (int base = threadIdx.x; base < W; base += kB * (int)blockDim.x) {
            T pc[kB], pn[kB], sm[kB];
#pragma unroll
            for (int m = 0; m < kB; ++m) {
                const int i = min(base + m * (int)blockDim.x, W - 1);
                pc[m] = sp[i];
                pn[m] = sp[i + 1 == W ? 0 : i + 1];
                sm[m] = S[min(i, W / 2)];
            }
#pragma unroll
            for (int m = 0; m < kB; ++m) {
                const int i = base + m * (int)blockDim.x;
                if (i < W) {
                    pe[i] = (pc[m] + pn[m]) * T(0.5);            // dynamics.py:15-17
                    if (i <= W / 2) sl[i] = sm[m] * inv_n;
                }
            }
        }
    }
    FilterConsts<T> c;
    filter_consts<T>(c, a.tw, a.cplan, W);
    c.s = sl;
    __syncthreads();
    for (int pair = pb0; pair < pb1; ++pair) {
        const int k0 = 2 * pair;
        const bool two = k0 + 1 < L;
        T *o0 = out_row + (long)k0 * W;
        // The thread index and the base twiddles are made opaque per iteration: otherwise the index
        // arithmetic of all passes and every power of the twiddles (loop invariants now) would be
        // hoisted out of the loop and held in hundreds of registers.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        FilterConsts<T> cc = c;
#pragma unroll
        for (int n = 0; n < 3; ++n) asm volatile("" : "+v"(cc.w[n].x), "+v"(cc.w[n].y));
        const auto first = [&](int i, int m) {
            const T p = pe[i];
            return mkv<V>(in[m < NIN ? m : 0].x * p, in[m < NIN ? m : 0].y * p);
        };
        // the next pair's su goes into the same registers as soon as the first pass has read them,
        // and is in flight during the other passes
        const auto after_first = [&]() { request(min(pair + 1, pb1 - 1), tid); };
        const auto store = [=](int i, V v) {
            o0[i] = v.x;
            if (two) o0[W + i] = v.y;
        };
        filter_rows_hoisted<MAXR, MASK, T>(x, first, after_first, store, a.tw, a.cplan, W, cc, tid);
    }
}

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

// ---------------------------------------------------------------- K2: column kernels
// K2a pe_geopot_kernel: rho, phi from the stage theta and surface pressure (compute_geopotential);
// K2b pe_pit_kernel: pit = sum_k conv and p_n from the filtered mass flux (aflux).  They are two
// kernels because they sit on two independent chains, K1 -> K2b and K2a -> K3 (see half_t).
// The per-level stp that phi needs after the column sum is parked in LDS, park[k][thread],
// instead of a round trip through HBM.
constexpr int kColThreads = 128;
// LMAX > 0: L <= LMAX and the per-level stp stay in registers (loops unrolled over LMAX; no LDS
// park, so the occupancy is not limited by it); LMAX == 0: any L, stp parked in LDS
template <typename T, int LMAX = 0>
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
    if (jrel >= a.j1 - a.j0) return;
    const int i = (tile - jrel * iblocks) * kColThreads + threadIdx.x;
    const int j = a.j0 + jrel;
    if (i >= W) return;
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
#pragma unroll(LMAX > 0 ? LMAX : 6)
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
#pragma unroll(LMAX > 0 ? LMAX : 1)
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
template <typename T>
__device__ __forceinline__ T column_sum(const T *col, const T *dsig, int L, int W) {
    // eight levels requested at a time, then added in order (a load per iteration waits a memory
    // latency per level: 20 us for the two ghost rows of a band, on the stage's critical path)
    T acc = T(0.0);
    int k = L - 1;
    for (; k >= 7; k -= 8) {
        T x[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) x[n] = col[(long)(k - n) * W];
#pragma unroll
        for (int n = 0; n < 8; ++n) acc = cs_acc(acc, x[n], dsig[k - n]);
    }
    for (; k >= 0; --k) acc = cs_acc(acc, col[(long)k * W], dsig[k]);
    return acc;
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
    a.scs_u[ix.r2(j) + i] = column_sum(a.su + ix.r3(j) + i, a.dsig, a.L, W);
    a.scs_v[ix.r2(j) + i] = column_sum(a.sv + ix.r3(j) + i, a.dsig, a.L, W);
}
// one workgroup per row j of [j0, j1): pit and p_n = p - pit dt (dynamics.py:38-40,194)
template <typename T, int MAXR, unsigned MASK = 0>
__global__ __launch_bounds__(512) void pe_pit2d_kernel(PeArgsT<T> a) {
    using V = typename Vec2<T>::type;
    extern __shared__ unsigned char lds_raw[];
    V *x = (V *)lds_raw;
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W;
    T *fx = (T *)(x + (MAXR > 0 ? 1 : 2) * W);                  // the filtered row, after the transform's workspace
    const int j = a.j0 + blockIdx.x;
    const int jg = wrapi(a.row0 + j, a.Hg);
    const T *sp = a.sp + ix.r2(j);
    const T *cu = a.scs_u + ix.r2(j);
    const auto load = [=](int i, int = 0) {
        const int ie = i + 1 == W ? 0 : i + 1;
        const T pe = (sp[i] + sp[ie]) * T(0.5);                  // iph(p), dynamics.py:15-17
        return mkv<V>(cu[i] * pe, T(0.0));
    };
    const auto store = [=](int i, V v) { fx[i] = v.x; };
    if (a.filter && W > 1) {
        if (MAXR > 0) {
            filter_rows_composite<MAXR, MASK, T>(x, load, store, a.tw, a.cplan, W, a.smul + (long)jg * (W / 2 + 1));
        } else {
            for (int i = threadIdx.x; i < W; i += blockDim.x) x[i] = load(i);
            __syncthreads();
            const V *res = filter_rows<T>(x, x + W, a.tw, a.plan, a.smul + (long)jg * (W / 2 + 1));
            for (int i = threadIdx.x; i < W; i += blockDim.x) store(i, res[i]);
        }
    } else {
        for (int i = threadIdx.x; i < W; i += blockDim.x) store(i, load(i));
    }
    __syncthreads();
    const T inv_dxj = a.inv_dxj[jg], inv_dy = a.inv_dy;
    const T *spn = a.sp + ix.r2(j - 1), *sps = a.sp + ix.r2(j + 1);
    const T *cvc = a.scs_v + ix.r2(j), *cvn = a.scs_v + ix.r2(j - 1);
    // (four columns of a thread requested at a time: one memory latency per batch, not per column --
    // on a band this workgroup's chain is on the stage's critical path)
    constexpr int kB = 4;
    const T *pb = a.p + ix.r2(j);
    for (int base = threadIdx.x; base < W; base += kB * (int)blockDim.x) {
        T xc[kB], xs[kB], xn[kB], vc[kB], vn[kB], pp[kB];
#pragma unroll
        for (int m = 0; m < kB; ++m) {
            const int i = min(base + m * (int)blockDim.x, W - 1);
            xc[m] = sp[i]; xs[m] = sps[i]; xn[m] = spn[i]; vc[m] = cvc[i]; vn[m] = cvn[i]; pp[m] = pb[i];
        }
#pragma unroll
        for (int m = 0; m < kB; ++m) {
            const int i = base + m * (int)blockDim.x;
            if (i < W) {
                const int iw = i == 0 ? W - 1 : i - 1;
                const T jph_c = (xc[m] + xs[m]) * T(0.5), jph_n = (xn[m] + xc[m]) * T(0.5);  // jph(sp) at j, j-1
                const T pit = (fx[i] - fx[iw]) * inv_dxj + (vc[m] * jph_c - vn[m] * jph_n) * inv_dy;
                a.pit[ix.r2(j) + i] = pit;
                a.pn[ix.r2(j) + i] = pp[m] - pit * a.dt;
            }
        }
    }
}

// ---------------------------------------------------------------- K3: pgfu = filter(pgu + phiu)
// The workgroup of (row, level pair k0 = 2 b, k1 = k0 + 1) rebuilds rho on both levels and phi on
// the odd one from theta (see rho_of / phi_up); phi[k0] is the anchor pe_geopot_kernel stored.
// Column i + 1 comes from the next lane (DPP); where the next column belongs to another wave
// (lane 63, and the row's last column, which wraps to 0) it comes from a small LDS table of the
// columns that are multiples of 64, filled before the main loop.
template <typename T>
struct PgfCol { T rho0, rho1, phi0, phi1; };
constexpr int kPgfBatch = 4;

template <typename T, int MAXR, unsigned MASK = 0>
__global__ __launch_bounds__(512) void pe_pgf_filter_kernel(PeArgsT<T> a) {
    using V = typename Vec2<T>::type;
    extern __shared__ unsigned char lds_raw[];
    __shared__ double tab[kExnerTabDoubles];
    __shared__ PgfCol<T> edge[kMaxEdgeCols];
    V *x = (V *)lds_raw;
    for (int n = threadIdx.x; n < kExnerTabDoubles; n += blockDim.x) tab[n] = a.exner_tab[n];
    const Idx ix{a.W, a.H, a.L, a.wrap};
    // 1-D grid of 8 x ceil(rows / 8) x pairs workgroups.  Consecutive workgroup ids go to the 8 XCDs
    // in turn; within an XCD the level pairs of a row follow one another, so that the row's sp and
    // filter multiplier (read by every pair) come from that XCD's L2 after the first.
    const int npairs = (a.L + 1) / 2;
    const int rows_per_xcd = gridDim.x / (8 * npairs);
    const int l = blockIdx.x / 8;
    const int jrel = (blockIdx.x % 8) * rows_per_xcd + l / npairs;
    if (jrel >= a.j1 - a.j0) return;                             // padding (uniform)
    const int j = a.j0 + jrel;
    const int k0 = 2 * (l % npairs), k1 = k0 + 1;
    const bool two = k1 < a.L;
    const int W = a.W;
    const int jg = wrapi(a.row0 + j, a.Hg);
    const T inv_dxj = a.inv_dxj[jg];
    const T *sp = a.sp + ix.r2(j);
    const long o0 = ix.r3(j) + (long)k0 * W;
    const T *phi0 = a.phi + o0;
    const T *st0 = a.st + o0, *st1 = st0 + (two ? W : 0);
    const T sg0 = a.sig[k0], sg1 = two ? a.sig[k1] : sg0;
    const T ptop = a.ptop;
    T *out = a.pgfu + o0;
    const int wpad = (W + 63) / 64 * 64;
    __syncthreads();
    // what a column needs from memory, and what is made of it
    struct Raw { T pc, pe, t0, t1, ph; };
    const auto request = [=](int i_raw) {
        const int i = i_raw < W ? i_raw : W - 1;
        const int ie = i + 1 == W ? 0 : i + 1;
        return Raw{sp[i], sp[ie], st0[i], st1[i], phi0[i]};
    };
    const auto column_of = [=](const Raw &r) {
        const T tp0 = r.pc * sg0 + ptop, tp1 = r.pc * sg1 + ptop;
        const T ex0 = exner(tp0, tab), ex1 = exner(tp1, tab);
        PgfCol<T> c;
        c.rho0 = rho_of(tp0, r.t0, ex0);
        c.rho1 = rho_of(tp1, r.t1, ex1);
        c.phi0 = r.ph;
        c.phi1 = phi_up(c.phi0, r.t0, r.t1, ex0, ex1);
        return c;
    };
    for (int e = threadIdx.x; e * 64 < W; e += blockDim.x) edge[e] = column_of(request(e * 64));
    __syncthreads();
    // every lane of a wave goes through the loop body (DPP reads its neighbour lane): columns past
    // the end are clamped and not stored
    const int lane = threadIdx.x & 63;
    const auto value = [&](int i_raw, const Raw &r) {
        const int i = i_raw < W ? i_raw : W - 1;
        const int ie = i + 1 == W ? 0 : i + 1;
        const PgfCol<T> c = column_of(r);
        PgfCol<T> e;
        e.rho0 = from_east(c.rho0); e.rho1 = from_east(c.rho1);
        e.phi0 = from_east(c.phi0); e.phi1 = from_east(c.phi1);
        if (lane == 63 || ie == 0) e = edge[ie >> 6];
        const T pc = r.pc, pe = r.pe;
        const T iphp = (pc + pe) * T(0.5);
        const T gradp = (pe - pc) * inv_dxj;
        const T phiu0 = iphp * ((e.phi0 - c.phi0) * inv_dxj);                      // dynamics.py:159
        const T pgu0 = ((sg0 * pc + sg0 * pe) * T(0.5)) * rcp((c.rho0 + e.rho0) * T(0.5)) * gradp;   // dynamics.py:162-165
        const T phiu1 = iphp * ((e.phi1 - c.phi1) * inv_dxj);
        const T pgu1 = ((sg1 * pc + sg1 * pe) * T(0.5)) * rcp((c.rho1 + e.rho1) * T(0.5)) * gradp;
        return mkv<V>(pgu0 + phiu0, two ? pgu1 + phiu1 : T(0.0));
    };
    // the columns of a thread are requested kPgfBatch at a time (one memory latency per batch instead
    // of one per column), then worked off
    const auto sweep = [&](const auto &sink) {
        for (int base = threadIdx.x; base < wpad; base += kPgfBatch * (int)blockDim.x) {
            Raw r[kPgfBatch];
#pragma unroll
            for (int m = 0; m < kPgfBatch; ++m) r[m] = request(min(base + m * (int)blockDim.x, wpad - 1));
#pragma unroll
            for (int m = 0; m < kPgfBatch; ++m) {
                const int i = base + m * (int)blockDim.x;
                const V v = value(min(i, wpad - 1), r[m]);
                if (i < W) sink(i, v);
            }
        }
    };
    const auto store = [=](int i, V v) {
        out[i] = v.x;
        if (two) out[W + i] = v.y;
    };
    if (a.filter && W > 1) {
        sweep([x](int i, V v) { x[i] = v; });
        __syncthreads();
        if (MAXR > 0) {
            const auto from_x = [x](int i, int) { return x[i]; };
            filter_rows_composite<MAXR, MASK, T>(x, from_x, store, a.tw, a.cplan, W, a.smul + (long)jg * (W / 2 + 1), true);
        } else {
            const V *res = filter_rows<T>(x, x + W, a.tw, a.plan, a.smul + (long)jg * (W / 2 + 1));
            for (int i = threadIdx.x; i < W; i += blockDim.x) store(i, res[i]);
        }
    } else {
        sweep(store);
    }
}

// ---------------------------------------------------------------- K4: update
// One thread per (j, i) column marching up the levels: the k-1 / k / k+1 values of the stage
// winds, theta, q and sigma-dot rotate through registers, so only the horizontal neighbours
// are loaded per level.  Tiles (row, 64-column block) are dealt to the 8 XCDs in contiguous
// runs of rows (as sw2d_fused_kernel does): the blocks resident on one XCD work on adjacent
// rows at about the same level, so the j+-1 re-reads hit that XCD's L2.
constexpr int kUpdThreads = 64;   // one wave per workgroup: packs the rounds of a short band best (256: +2.5 %)
template <typename T>
__global__ __launch_bounds__(kUpdThreads) void pe_update_kernel(PeArgsT<T> a) {
    __shared__ double tab[kExnerTabDoubles];
    for (int n = threadIdx.x; n < kExnerTabDoubles; n += kUpdThreads) tab[n] = a.exner_tab[n];
    __syncthreads();
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W, L = a.L;
    const int iblocks = (W + kUpdThreads - 1) / kUpdThreads;
    const int per_xcd = gridDim.x / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    // tile = ((row, level segment), column block); rows come from [j0, j1) then [jb0, jb1)
    const int nseg = a.nseg;
    const int rowseg = tile / iblocks;
    const int jrel = rowseg / nseg, seg = rowseg - jrel * nseg;
    const int na = a.j1 - a.j0;
    if (jrel >= na + (a.jb1 - a.jb0)) return;                // padding tiles
    const int j = jrel < na ? a.j0 + jrel : a.jb0 + (jrel - na);
    const int i = (tile - rowseg * iblocks) * kUpdThreads + threadIdx.x;
    if (i >= W) return;
    const int iw = i == 0 ? W - 1 : i - 1, ie = i + 1 == W ? 0 : i + 1;
    const int jg = wrapi(a.row0 + j, a.Hg);
    const T inv_dxj = a.inv_dxj[jg], inv_dxh = a.inv_dxh[jg], inv_dy = a.inv_dy, dt = a.dt;
    const long rc = ix.r3(j), rn = ix.r3(j - 1), rs = ix.r3(j + 1);
    // surface pressure of the stage state on rows j-1 .. j+2 (level independent)
    const T *spr = a.sp;
    const long p_n = ix.r2(j - 1), p_c = ix.r2(j), p_s = ix.r2(j + 1), p_ss = ix.r2(j + 2);
    const T sp_c = spr[p_c + i], sp_e = spr[p_c + ie];
    const T sp_s = spr[p_s + i], sp_se = spr[p_s + ie], sp_ss = spr[p_ss + i];
    const T sp_n = spr[p_n + i], sp_ne = spr[p_n + ie];
    const T jph_c = (sp_c + sp_s) * T(0.5), jph_ce = (sp_e + sp_se) * T(0.5);     // jph(sp) at (j,i),(j,i+1)
    const T jph_n = (sp_n + sp_c) * T(0.5), jph_ne = (sp_ne + sp_e) * T(0.5);     // at (j-1,i),(j-1,i+1)
    const T jph_s = (sp_s + sp_ss) * T(0.5);                                   // at (j+1,i)
    const T pb_c = a.p[p_c + i], pb_e = a.p[p_c + ie], pb_s = a.p[p_s + i];
    const T iph_pb = (pb_c + pb_e) * T(0.5), jph_pb = (pb_c + pb_s) * T(0.5);
    const T pn_c = a.pn[p_c + i], pn_e = a.pn[p_c + ie], pn_s = a.pn[p_s + i];
    const T inv_pnu = rcp((pn_c + pn_e) * T(0.5)), inv_pnv = rcp((pn_c + pn_s) * T(0.5)), inv_pn = rcp(pn_c);
    const bool pole_edge = jg == a.Hg - 1;
    const bool coriolis = a.cor_u != nullptr;
    const bool same = a.u == a.su;
    const T cp_u = coriolis ? a.cor_u[jg] : T(0.0), cp_v = coriolis ? a.cor_v[jg] : T(0.0);
    if (seg == 0) a.op[(long)j * W + i] = pn_c;

    // The levels are marched from the top down, because sigma-dot is the top-down running sum of
    // conv (dynamics.py:42: cumsum(conv[::-1])[::-1] - pit sigb, sd[0] = 0): it is rebuilt here,
    // for this column and its east and south neighbours (iph(sd), jph(sd)), from the mass fluxes
    // the momentum advection loads anyway, instead of being read back from HBM.
    // Vertical window: level k+1 (p), k (c), k-1 (m).  kp()/km() wrap (coordinates_3d.py:55-60):
    // level L is 0 and level -1 is L-1; both only ever meet sd[0] = T(0.)
    // This workgroup marches levels [k_lo, k_hi) of its columns (the whole column if nseg == 1).
    const T inv_dxj_s = a.inv_dxj[wrapi(a.row0 + j + 1, a.Hg)];
    const T pit_c = a.pit[p_c + i], pit_e = a.pit[p_c + ie], pit_s = a.pit[p_s + i];
    const int k_lo = seg_lo(seg, nseg, L), k_hi = seg_lo(seg + 1, nseg, L);
    if (k_hi <= k_lo) return;
    const long top = (long)(L - 1) * W;
    const long kp0 = k_hi == L ? 0 : (long)k_hi * W;            // level k_hi; level L wraps to 0
    const long kc0 = (long)(k_hi - 1) * W;
    T su_p = a.su[rc + kp0 + i], sv_p = a.sv[rc + kp0 + i], st_p = a.st[rc + kp0 + i], sq_p = a.sq[rc + kp0 + i];
    T su_c = a.su[rc + kc0 + i], sv_c = a.sv[rc + kc0 + i], st_c = a.st[rc + kc0 + i], sq_c = a.sq[rc + kc0 + i];
    // running sums of conv from the top and sd at level k_hi: zero above the top level, else from
    // the partial sums pe_pit_kernel left at this segment boundary
    T rc_c = T(0.0), rc_e = T(0.0), rc_s = T(0.0);
    T sd_cp = T(0.0), sd_ep = T(0.0), sd_sp = T(0.0);
    if (k_hi < L) {
        const T *part = a.part + (long)seg * a.part_stride;
        const T sgb_hi = a.sigb[k_hi];
        rc_c = part[p_c + i]; rc_e = part[p_c + ie]; rc_s = part[p_s + i];
        sd_cp = sd_of(rc_c, pit_c, sgb_hi);
        sd_ep = sd_of(rc_e, pit_e, sgb_hi);
        sd_sp = sd_of(rc_s, pit_s, sgb_hi);
    }
    // rho and phi of this column and its south neighbour are rebuilt per level (rho_of / phi_up):
    // an odd level k takes the anchor phi[k-1] that pe_geopot_kernel stored and steps up from it,
    // and leaves the level k-1 values it needed (anchor, exner factors, the south theta) for the
    // next, even, iteration -- two exner evaluations per level and column on average
    const T ptop = a.ptop;
    bool have_lo = false;
    T lo_ex_c = T(0.0), lo_ex_s = T(0.0), lo_phi_c = T(0.0), lo_phi_s = T(0.0), lo_st_s = T(0.0);
    for (int k = k_hi - 1; k >= k_lo; --k) {
        const long kc = (long)k * W;
        T su_m, sv_m, st_m, sq_m;
        if (k > 0) {
            const long kmo = kc - W;
            su_m = a.su[rc + kmo + i]; sv_m = a.sv[rc + kmo + i]; st_m = a.st[rc + kmo + i]; sq_m = a.sq[rc + kmo + i];
        } else {
            su_m = a.su[rc + top + i]; sv_m = a.sv[rc + top + i]; st_m = a.st[rc + top + i]; sq_m = a.sq[rc + top + i];
        }
        // stage winds, horizontal neighbours
        const T su_w = a.su[rc + kc + iw], su_e = a.su[rc + kc + ie];
        const T su_n = a.su[rn + kc + i], su_s = a.su[rs + kc + i];
        const T sv_w = a.sv[rc + kc + iw], sv_e = a.sv[rc + kc + ie];
        const T sv_n = a.sv[rn + kc + i], sv_ne = a.sv[rn + kc + ie], sv_s = a.sv[rs + kc + i];
        // mass fluxes: spu filtered (K1); spv = sv * jph(sp), dynamics.py:20-22
        const T spu_c = a.spu[rc + kc + i], spu_w = a.spu[rc + kc + iw], spu_e = a.spu[rc + kc + ie];
        const T spu_s = a.spu[rs + kc + i], spu_sw = a.spu[rs + kc + iw];
        const T spv_c = sv_c * jph_c, spv_e = sv_e * jph_ce;
        const T spv_n = sv_n * jph_n, spv_ne = sv_ne * jph_ne;
        const T spv_s = sv_s * jph_s;
        // ---- aflux, dynamics.py:35-46, at (j,i), (j,i+1), (j+1,i)
        const T dsg = a.dsig[k], sgb = a.sigb[k];
        T sd_c = T(0.0), sd_e = T(0.0), sd_s = T(0.0);           // sd[0] = 0, dynamics.py:44
        if (k > 0) {
            rc_c = conv_acc(rc_c, spu_c, spu_w, inv_dxj, sv_c, jph_c, sv_n, jph_n, inv_dy, dsg);
            rc_e = conv_acc(rc_e, spu_e, spu_c, inv_dxj, sv_e, jph_ce, sv_ne, jph_ne, inv_dy, dsg);
            rc_s = conv_acc(rc_s, spu_s, spu_sw, inv_dxj_s, sv_s, jph_s, sv_c, jph_c, inv_dy, dsg);
            sd_c = sd_of(rc_c, pit_c, sgb);
            sd_e = sd_of(rc_e, pit_e, sgb);
            sd_s = sd_of(rc_s, pit_s, sgb);
        }
        // ---- advec_m_pu, dynamics.py:55-108
        const T puum = ((su_c + su_w) * T(0.5)) * ((spu_c + spu_w) * T(0.5));
        const T puup = ((su_e + su_c) * T(0.5)) * ((spu_e + spu_c) * T(0.5));
        const T puvp = ((spv_c + spv_e) * T(0.5)) * ((su_c + su_s) * T(0.5));
        const T puvm = ((spv_n + spv_ne) * T(0.5)) * ((su_n + su_c) * T(0.5));
        const T pvvm = ((sv_c + sv_n) * T(0.5)) * ((spv_c + spv_n) * T(0.5));
        const T pvvp = ((sv_s + sv_c) * T(0.5)) * ((spv_s + spv_c) * T(0.5));
        const T pvup = ((sv_c + sv_e) * T(0.5)) * ((spu_c + spu_s) * T(0.5));
        const T pvum = ((sv_w + sv_c) * T(0.5)) * ((spu_w + spu_sw) * T(0.5));
        T cor_u = T(0.0), cor_v = T(0.0);                         // the reference adds a literal 0
        if (coriolis) {                                          // dynamics.py:83-92
            const T pu_at_pv = (((spu_c + spu_s) * T(0.5)) + ((spu_w + spu_sw) * T(0.5))) * T(0.5);    // imh(jph(pu))
            const T pv_at_pu = (((spv_c + spv_n) * T(0.5)) + ((spv_e + spv_ne) * T(0.5))) * T(0.5);    // iph(jmh(pv))
            cor_u = cp_u * -pv_at_pu;
            cor_v = cp_v * pu_at_pv;
        }
        const T dut = (puum - puup) * inv_dxj + (puvm - puvp) * inv_dy + cor_u;
        const T dvt = (pvvm - pvvp) * inv_dy + (pvum - pvup) * inv_dxh + cor_v;
        // ---- pgf v-part, dynamics.py:160,167-169 (the u-part went through K3)
        const T sg = a.sig[k];
        T ex_c, ex_s, phi_c, phi_s, st_s;
        if (have_lo) {
            ex_c = lo_ex_c; ex_s = lo_ex_s; phi_c = lo_phi_c; phi_s = lo_phi_s; st_s = lo_st_s;
        } else {
            ex_c = exner(sp_c * sg + ptop, tab);
            ex_s = exner(sp_s * sg + ptop, tab);
            st_s = a.st[rs + kc + i];
            phi_c = phi_s = T(0.0);
            if ((k & 1) == 0) { phi_c = a.phi[rc + kc + i]; phi_s = a.phi[rs + kc + i]; }
        }
        have_lo = (k & 1) != 0;
        if (have_lo) {                                            // k odd: k - 1 >= 0 is an anchor level
            const T sg_lo = a.sig[k - 1];
            lo_ex_c = exner(sp_c * sg_lo + ptop, tab);
            lo_ex_s = exner(sp_s * sg_lo + ptop, tab);
            lo_st_s = a.st[rs + kc - W + i];
            lo_phi_c = a.phi[rc + kc - W + i];
            lo_phi_s = a.phi[rs + kc - W + i];
            phi_c = phi_up(lo_phi_c, st_m, st_c, lo_ex_c, ex_c);
            phi_s = phi_up(lo_phi_s, lo_st_s, st_s, lo_ex_s, ex_s);
        }
        const T rho_c = rho_of(sp_c * sg + ptop, st_c, ex_c), rho_s = rho_of(sp_s * sg + ptop, st_s, ex_s);
        const T phiv = jph_c * ((phi_s - phi_c) * inv_dy);
        const T pgv = ((sg * sp_c + sg * sp_s) * T(0.5)) * rcp((rho_c + rho_s) * T(0.5)) * ((sp_s - sp_c) * inv_dy);
        // ---- vertical advection, dynamics.py:49-52 with iph(sd), jph(sd), sd
        const T inv_ds = a.inv_dsig[k];
        const T sdi = (sd_c + sd_e) * T(0.5), sdi_p = (sd_cp + sd_ep) * T(0.5);
        const T sdj = (sd_c + sd_s) * T(0.5), sdj_p = (sd_cp + sd_sp) * T(0.5);
        const T dus = -((((su_c + su_m) * T(0.5)) * sdi - ((su_p + su_c) * T(0.5)) * sdi_p) * inv_ds);
        const T dvs = -((((sv_c + sv_m) * T(0.5)) * sdj - ((sv_p + sv_c) * T(0.5)) * sdj_p) * inv_ds);
        const T dts = -((((st_c + st_m) * T(0.5)) * sd_c - ((st_p + st_c) * T(0.5)) * sd_cp) * inv_ds);
        const T dqs = -((((sq_c + sq_m) * T(0.5)) * sd_c - ((sq_p + sq_c) * T(0.5)) * sd_cp) * inv_ds);
        // ---- momentum update, dynamics.py:186-212
        // predictor: the stage state IS the base state, its values are in the window already
        const T pu = (same ? su_c : a.u[rc + kc + i]) * iph_pb;
        const T pv = (same ? sv_c : a.v[rc + kc + i]) * jph_pb;
        const T pgfu = a.pgfu[rc + kc + i];
        const T pu_n = pu - (dut + dus + pgfu) * dt;
        const T pv_n = pv - (dvt + dvs + phiv + pgv) * dt;
        T u_n = pu_n * inv_pnu;
        T v_n = pv_n * inv_pnv;
        if (pole_edge) v_n *= T(0.0);                               // v_n[:, -1, :] *= 0, dynamics.py:222
        // ---- advec_t for t and q, dynamics.py:174-181,214-219
        const T st_e = a.st[rc + kc + ie], st_w = a.st[rc + kc + iw];
        const T st_n = a.st[rn + kc + i];
        const T sq_e = a.sq[rc + kc + ie], sq_w = a.sq[rc + kc + iw];
        const T sq_s = a.sq[rs + kc + i], sq_n = a.sq[rn + kc + i];
        const T adt = (spu_c * ((st_c + st_e) * T(0.5)) - spu_w * ((st_w + st_c) * T(0.5))) * inv_dxj +
                           (spv_c * ((st_c + st_s) * T(0.5)) - spv_n * ((st_n + st_c) * T(0.5))) * inv_dy;
        const T adq = (spu_c * ((sq_c + sq_e) * T(0.5)) - spu_w * ((sq_w + sq_c) * T(0.5))) * inv_dxj +
                           (spv_c * ((sq_c + sq_s) * T(0.5)) - spv_n * ((sq_n + sq_c) * T(0.5))) * inv_dy;
        const T t_n = ((same ? st_c : a.t[rc + kc + i]) * pb_c - (adt + dts) * dt) * inv_pn;
        const T q_n = ((same ? sq_c : a.q[rc + kc + i]) * pb_c - (adq + dqs) * dt) * inv_pn;
        const long o = (long)j * L * W + kc + i;                 // interior rows: no wrap needed
        a.ou[o] = u_n;
        a.ov[o] = v_n;
        a.ot[o] = t_n;
        a.oq[o] = q_n;
        // rotate the vertical window downwards
        su_p = su_c; sv_p = sv_c; st_p = st_c; sq_p = sq_c;
        su_c = su_m; sv_c = sv_m; st_c = st_m; sq_c = sq_m;
        sd_cp = sd_c; sd_ep = sd_e; sd_sp = sd_s;
    }
}

// ---------------------------------------------------------------- K4, row-group form
// A workgroup is R compute waves = R consecutive rows x 62 columns (lanes 1..62; lanes 0 and 63
// carry the halo columns i-1 / i+1 and are not stored) plus ONE loader wave, marching the levels
// top-down in lockstep.  Everything the march reads from global memory goes through LDS tiles, one
// per level, [field][row slot][lane]:
//   * at the top of the iteration of level k every compute wave REQUESTS its own row of level k-3
//     (su, sv, st, sq, spu, the phi anchor, pgfu, the base state) and the loader wave the two halo
//     rows (above and below the group); the requests of the previous iteration (level k-2) are
//     written to their tile at the END of the iteration, one barrier per level.  A request thus has
//     two levels of arithmetic to arrive, and the only reader of a requested register is that tile
//     write, so the compiler's in-order vmcnt wait leaves the newest requests in flight;
//   * the iteration reads the tiles of level k (own row, rows j-1 / j+1) and k-1 (the level below:
//     vertical fluxes, the south theta of the geopotential anchor); three tiles rotate;
//   * columns i-1 / i+1 of the own row come from the neighbouring lanes (DPP);
//   * sigma-dot at (j, i+1) is the east lane's value (DPP); at (j+1, i) it is rebuilt from the tile;
//   * the fluxes through the upper faces are the lower-face fluxes of the level above, carried.
// A row of the stage state leaves HBM (R+2)/R times instead of up to three times.
constexpr int kUpdCols = 62;
__device__ __forceinline__ float from_west(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
// slots of one level tile, in units of 64 lanes
template <int R> struct UpdTile {
    static constexpr int kMain = 0;                      // su, sv, st, sq, spu: R+2 slots each (0 = row above the group)
    static constexpr int kPhi = 5 * (R + 2);             // phi anchor: R+1 slots (own rows, then the row below the group)
    static constexpr int kOwn = kPhi + R + 1;            // pgfu, u, v, t, q of the base state: R slots each
    static constexpr int kSlots = kOwn + 5 * R;
};
// SAME: the stage state is the base state (predictor): no base-state requests
template <typename T, int R, bool SAME>
__global__ __launch_bounds__(64 * (R + 1)) void pe_update_rows_kernel(PeArgsT<T> a) {
    using TL = UpdTile<R>;
    extern __shared__ unsigned char upd_lds_raw[];
    __shared__ double tab[kExnerTabDoubles];
    // the level tables go to LDS too: read from global memory inside the march, their waits (vmcnt
    // counts in order) would also wait for every request still in flight
    T *lv_sig = (T *)upd_lds_raw, *lv_dsig = lv_sig + a.L, *lv_inv_dsig = lv_dsig + a.L, *lv_sigb = lv_inv_dsig + a.L;
    T *tile = lv_sigb + a.L + 1;                     // one word of slack on either side: lane -1 / 64 reads
    for (int n = threadIdx.x; n < kExnerTabDoubles; n += 64 * (R + 1)) tab[n] = a.exner_tab[n];
    for (int n = threadIdx.x; n < a.L; n += 64 * (R + 1)) {
        lv_sig[n] = a.sig[n]; lv_dsig[n] = a.dsig[n]; lv_inv_dsig[n] = a.inv_dsig[n]; lv_sigb[n] = a.sigb[n];
    }
    __syncthreads();
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W, L = a.L;
    const int ncol = (W + kUpdCols - 1) / kUpdCols;
    // Workgroup = ((row group, level segment), column tile); groups come from [j0, j1) then [jb0, jb1).
    // Each XCD (workgroups b, b+8, ... share one) takes a contiguous run of (group, segment) pairs and
    // walks it with the GROUP index fastest: the workgroups resident on an XCD at one time are
    // vertically adjacent groups of a few column tiles, marching in step, so the halo row one of them
    // requests is the own row its neighbour requests at about the same time -- it comes from that
    // XCD's L2 instead of HBM.
    const int nseg = a.nseg;
    const int na = a.j1 - a.j0, nb = a.jb1 - a.jb0;
    const int ga = (na + R - 1) / R, gb = (nb + R - 1) / R;
    const int per_xcd = gridDim.x / 8;                           // = rs_per_xcd * ncol (launch)
    const int rs_per_xcd = per_xcd / ncol;
    const int l = blockIdx.x / 8;
    const int ct = l / rs_per_xcd;
    const int rowseg = (blockIdx.x % 8) * rs_per_xcd + (l - ct * rs_per_xcd);
    const int grp = rowseg / nseg, seg = rowseg - grp * nseg;
    if (grp >= ga + gb) return;                                  // padding workgroups (uniform)
    const int jg = grp < ga ? a.j0 + grp * R : a.jb0 + (grp - ga) * R;
    const int jend = min(jg + R, grp < ga ? a.j1 : a.jb1);
    const int nact = jend - jg;
    const int k_lo = seg_lo(seg, nseg, L), k_hi = seg_lo(seg + 1, nseg, L);
    if (k_hi <= k_lo) return;                                    // uniform
    const int r = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    constexpr int kBuf = TL::kSlots * 64;
    const int iraw = ct * kUpdCols + lane - 1;
    const int i = wrapi(iraw, W), ie = i + 1 == W ? 0 : i + 1;
    const int k0 = k_hi - 1;
    const int kmin = k_lo > 0 ? k_lo - 1 : 0;                    // tiles exist for levels k0 .. kmin
    constexpr bool same = SAME;
    T *const t0 = tile + lane;
    // tile of the level at distance d below k0: buffers rotate 0, 1, 2
    int bc = 0, bm = 1, bf = 2;                                  // level k, k-1, k-2 (being filled)

    if (r == R) {
        // ---- loader wave: the halo rows jg-1 (slot 0) and jend (slot nact+1; its phi anchor: slot nact)
        const long rn = ix.r3(jg - 1), rs = ix.r3(jend);
        const int ss = nact + 1;
        T h[2][10];
        const auto load = [&](T (&d)[10], int k) {
            const long kl = (long)k * W;
            d[0] = a.su[rn + kl + i]; d[1] = a.sv[rn + kl + i]; d[2] = a.st[rn + kl + i]; d[3] = a.sq[rn + kl + i];
            d[4] = a.su[rs + kl + i]; d[5] = a.sv[rs + kl + i]; d[6] = a.st[rs + kl + i]; d[7] = a.sq[rs + kl + i];
            d[8] = a.spu[rs + kl + i]; d[9] = a.phi[rs + (long)(k & ~1) * W + i];
        };
        const auto put = [&](const T (&d)[10], int buf) {
            T *t = t0 + buf * kBuf;
#pragma unroll
            for (int f = 0; f < 4; ++f) t[(TL::kMain + f * (R + 2)) * 64] = d[f];
#pragma unroll
            for (int f = 0; f < 5; ++f) t[(TL::kMain + f * (R + 2) + ss) * 64] = d[4 + f];
            t[(TL::kPhi + nact) * 64] = d[9];
        };
        load(h[0], k0);
        put(h[0], 0);
        if (k0 - 1 >= kmin) { load(h[1], k0 - 1); put(h[1], 1); }
        // requests are unconditional (level clamped to kmin): a conditional one would make the compiler
        // size its in-order vmcnt waits for the path without it, i.e. wait for the newest requests too
        load(h[0], max(k0 - 2, kmin));
        __syncthreads();
        for (int k = k0;;) {
            load(h[1], max(k - 3, kmin));
            if (k - 2 >= kmin) put(h[0], bf);
            if (k == k_lo) break;
            __syncthreads();
            { const int t = bc; bc = bm; bm = bf; bf = t; }
            --k;
            load(h[0], max(k - 3, kmin));
            if (k - 2 >= kmin) put(h[1], bf);
            if (k == k_lo) break;
            __syncthreads();
            { const int t = bc; bc = bm; bm = bf; bf = t; }
            --k;
        }
        return;
    }
    if (r >= nact) {                                             // rows past the range: keep the barriers
        for (int k = k0; k >= k_lo; --k) __syncthreads();
        return;
    }
    const int j = jg + r;
    const bool store = lane >= 1 && lane <= kUpdCols && iraw < W;
    const int jg_row = wrapi(a.row0 + j, a.Hg);
    const T inv_dxj = a.inv_dxj[jg_row], inv_dxh = a.inv_dxh[jg_row], inv_dy = a.inv_dy, dt = a.dt;
    const T inv_dxj_s = a.inv_dxj[wrapi(a.row0 + j + 1, a.Hg)];
    const T q_dxj = T(0.25) * inv_dxj, q_dxh = T(0.25) * inv_dxh, q_dy = T(0.25) * inv_dy;
    const T h_dxj = T(0.5) * inv_dxj, h_dy = T(0.5) * inv_dy;
    const long rc = ix.r3(j);
    const T *spr = a.sp;
    const long p_n = ix.r2(j - 1), p_c = ix.r2(j), p_s = ix.r2(j + 1), p_ss = ix.r2(j + 2);
    const T sp_c = spr[p_c + i], sp_e = spr[p_c + ie];
    const T sp_s = spr[p_s + i], sp_se = spr[p_s + ie], sp_ss = spr[p_ss + i];
    const T sp_n = spr[p_n + i], sp_ne = spr[p_n + ie];
    const T jph_c = (sp_c + sp_s) * T(0.5), jph_ce = (sp_e + sp_se) * T(0.5);
    const T jph_n = (sp_n + sp_c) * T(0.5), jph_ne = (sp_ne + sp_e) * T(0.5);
    const T jph_s = (sp_s + sp_ss) * T(0.5);
    const T pb_c = a.p[p_c + i], pb_e = a.p[p_c + ie], pb_s = a.p[p_s + i];
    const T iph_pb = (pb_c + pb_e) * T(0.5), jph_pb = (pb_c + pb_s) * T(0.5);
    const T pn_c = a.pn[p_c + i], pn_e = a.pn[p_c + ie], pn_s = a.pn[p_s + i];
    const T inv_pnu = rcp((pn_c + pn_e) * T(0.5)), inv_pnv = rcp((pn_c + pn_s) * T(0.5)), inv_pn = rcp(pn_c);
    const bool pole_edge = jg_row == a.Hg - 1;
    const bool coriolis = a.cor_u != nullptr;
    const T cp_u = coriolis ? a.cor_u[jg_row] : T(0.0), cp_v = coriolis ? a.cor_v[jg_row] : T(0.0);
    if (seg == 0 && store) a.op[(long)j * W + i] = pn_c;
    const T pit_c = a.pit[p_c + i], pit_s = a.pit[p_s + i];
    const T ptop = a.ptop;

    // own-row requests of one level: su, sv, st, sq, spu, phi anchor, pgfu, base u, v, t, q
    T q[2][11];
    // (the level's offset is wave-uniform: the request is scalar base + ONE 32-bit byte offset per lane,
    // the addressing mode that needs no vector arithmetic)
    const unsigned ob = (unsigned)i * (unsigned)sizeof(T);
    // (the base goes through an opaque scalar register pair: left visible, the compiler reassociates
    // it into eleven loop-invariant per-lane 64-bit addresses plus a scalar level offset -- 22 VGPRs
    // and a 64-bit vector add per request)
    const auto sbase = [](const T *p) {
        unsigned long long v = (unsigned long long)p;
        unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        asm volatile("" : "+s"(lo), "+s"(hi));
        return (__attribute__((address_space(1))) char *)(((unsigned long long)hi << 32) | lo);      // global, not flat
    };
    const auto load = [&](T (&d)[11], int k) {
        // (the lane offset is made opaque once per level, so that its zero extension stays next to the
        // requests and they take the scalar-base + 32-bit-offset form)
        unsigned ol = ob;
        asm volatile("" : "+v"(ol));
        const auto at = [ol, sbase](const T *base) { return *(const __attribute__((address_space(1))) T *)(sbase(base) + ol); };
        const long o = rc + (long)k * W;
        d[0] = at(a.su + o); d[1] = at(a.sv + o); d[2] = at(a.st + o); d[3] = at(a.sq + o);
        d[4] = at(a.spu + o); d[6] = at(a.pgfu + o);
        d[5] = at(a.phi + (rc + (long)(k & ~1) * W));        // the anchor at or below k (odd k: unused, and a cache hit)
        if (!same) { d[7] = at(a.u + o); d[8] = at(a.v + o); d[9] = at(a.t + o); d[10] = at(a.q + o); }
    };
    const auto put = [&](const T (&d)[11], int buf) {
        T *t = t0 + buf * kBuf;
#pragma unroll
        for (int f = 0; f < 5; ++f) t[(TL::kMain + f * (R + 2) + 1 + r) * 64] = d[f];
        t[(TL::kPhi + r) * 64] = d[5];
        t[(TL::kOwn + r) * 64] = d[6];
        if (!same) {
#pragma unroll
            for (int f = 1; f < 5; ++f) t[(TL::kOwn + f * R + r) * 64] = d[6 + f];
        }
    };
    // running sums of conv from the top; fluxes kmh(q) sd through the upper face of level k0
    // (advec_sig, dynamics.py:49-52): zero at the top of the column (sd wraps to sd[0] = 0), else
    // rebuilt from level k_hi exactly as the iteration of level k_hi forms its lower-face fluxes
    // (face_flux_v), so a segmented march gives the same bits
    T rc_c = T(0.0), rc_s = T(0.0);
    T fu_up = T(0.0), fv_up = T(0.0), ft_up = T(0.0), fq_up = T(0.0);
    T cs_u = T(0.0), cs_v = T(0.0);                              // sum_k dsig[k] u_n[k], v_n[k] (pe_pit2d_kernel)
    load(q[0], k0);
    if (k_hi < L) {
        const T *part = a.part + (long)seg * a.part_stride;
        const T sgb_hi = lv_sigb[k_hi];
        rc_c = part[p_c + i]; rc_s = part[p_s + i];
        const T sd_cp = sd_of(rc_c, pit_c, sgb_hi), sd_sp = sd_of(rc_s, pit_s, sgb_hi);
        const T sd_ep = from_east(sd_cp);
        const long kp0 = (long)k_hi * W;
        fu_up = face_flux_v(a.su[rc + kp0 + i], q[0][0], (sd_cp + sd_ep) * T(0.5));
        fv_up = face_flux_v(a.sv[rc + kp0 + i], q[0][1], (sd_cp + sd_sp) * T(0.5));
        ft_up = face_flux_v(a.st[rc + kp0 + i], q[0][2], sd_cp);
        fq_up = face_flux_v(a.sq[rc + kp0 + i], q[0][3], sd_cp);
    }
    put(q[0], 0);
    if (k0 - 1 >= kmin) { load(q[1], k0 - 1); put(q[1], 1); }
    load(q[0], max(k0 - 2, kmin));                               // unconditional, clamped: see the loader wave
    // geopotential anchors (see phi_up): an odd level k steps up from the anchor phi[k-1] (tile of
    // level k-1) and leaves the level k-1 exner factors, anchors and south theta to the even level below
    bool have_lo = false;
    T lo_ex_c = T(0.0), lo_ex_s = T(0.0), lo_phi_c = T(0.0), lo_phi_s = T(0.0), lo_st_s = T(0.0);
    __syncthreads();
    const auto level = [&](const int k) __attribute__((always_inline)) {
        const long kc = (long)k * W;
        const T *tc = t0 + bc * kBuf, *tm = t0 + bm * kBuf;
        const auto own = [&](const T *t, int f) { return t[(TL::kMain + f * (R + 2) + 1 + r) * 64]; };
        const auto nrow = [&](const T *t, int f) { return t + (TL::kMain + f * (R + 2) + r) * 64; };
        const auto srow = [&](const T *t, int f) { return t + (TL::kMain + f * (R + 2) + 2 + r) * 64; };
        const auto orow = [&](const T *t, int f) { return t + (TL::kMain + f * (R + 2) + 1 + r) * 64; };
        const T su_c = own(tc, 0), sv_c = own(tc, 1), st_c = own(tc, 2), sq_c = own(tc, 3), spu_c = own(tc, 4);
        // ---- neighbours: columns i-1 / i+1 of the own row and rows j-1 / j+1 from the tile (the edge
        //      lanes read a word of the neighbouring slot: they feed nothing that is stored)
        const T su_w = orow(tc, 0)[-1], su_e = orow(tc, 0)[1];
        const T sv_w = orow(tc, 1)[-1], sv_e = orow(tc, 1)[1];
        const T spu_w = orow(tc, 4)[-1], spu_e = orow(tc, 4)[1];
        const T st_w = orow(tc, 2)[-1], st_e = orow(tc, 2)[1];
        const T sq_w = orow(tc, 3)[-1], sq_e = orow(tc, 3)[1];
        const T su_n = *nrow(tc, 0), sv_n = *nrow(tc, 1), sv_ne = nrow(tc, 1)[1], st_n = *nrow(tc, 2), sq_n = *nrow(tc, 3);
        const T su_s = *srow(tc, 0), sv_s = *srow(tc, 1), st_sl = *srow(tc, 2), sq_s = *srow(tc, 3);
        const T spu_s = *srow(tc, 4), spu_sw = srow(tc, 4)[-1];
        const T spv_c = sv_c * jph_c, spv_e = sv_e * jph_ce;
        const T spv_n = sv_n * jph_n, spv_ne = sv_ne * jph_ne;
        const T spv_s = sv_s * jph_s;
        // ---- aflux, dynamics.py:35-46, at (j,i) and (j+1,i); (j,i+1) is the east lane's
        const T dsg = lv_dsig[k], sgb = lv_sigb[k];
        T sd_c = T(0.0), sd_s = T(0.0);                           // sd[0] = 0, dynamics.py:44
        if (k > 0) {
            rc_c = conv_acc(rc_c, spu_c, spu_w, inv_dxj, sv_c, jph_c, sv_n, jph_n, inv_dy, dsg);
            rc_s = conv_acc(rc_s, spu_s, spu_sw, inv_dxj_s, sv_s, jph_s, sv_c, jph_c, inv_dy, dsg);
            sd_c = sd_of(rc_c, pit_c, sgb);
            sd_s = sd_of(rc_s, pit_s, sgb);
        }
        const T sd_e = from_east(sd_c);
        // ---- advec_m_pu, dynamics.py:55-108
        // ---- advec_m_pu, dynamics.py:55-108.  Every product there is a product of two averages,
        //      ((a + b)/2) ((c + d)/2): the quarters are taken out of the sums and folded into the grid
        //      factors (q_dx = 1/(4 dx)) -- scaling by a power of two commutes with every rounding
        const T puum = (su_c + su_w) * (spu_c + spu_w);
        const T puup = (su_e + su_c) * (spu_e + spu_c);
        const T puvp = (spv_c + spv_e) * (su_c + su_s);
        const T puvm = (spv_n + spv_ne) * (su_n + su_c);
        const T pvvm = (sv_c + sv_n) * (spv_c + spv_n);
        const T pvvp = (sv_s + sv_c) * (spv_s + spv_c);
        const T pvup = (sv_c + sv_e) * (spu_c + spu_s);
        const T pvum = (sv_w + sv_c) * (spu_w + spu_sw);
        T cor_u = T(0.0), cor_v = T(0.0);                         // the reference adds a literal 0
        if (coriolis) {                                          // dynamics.py:83-92
            const T pu_at_pv = (((spu_c + spu_s) * T(0.5)) + ((spu_w + spu_sw) * T(0.5))) * T(0.5);    // imh(jph(pu))
            const T pv_at_pu = (((spv_c + spv_n) * T(0.5)) + ((spv_e + spv_ne) * T(0.5))) * T(0.5);    // iph(jmh(pv))
            cor_u = cp_u * -pv_at_pu;
            cor_v = cp_v * pu_at_pv;
        }
        const T dut = (puum - puup) * q_dxj + (puvm - puvp) * q_dy + cor_u;
        const T dvt = (pvvm - pvvp) * q_dy + (pvum - pvup) * q_dxh + cor_v;
        // ---- advec_t for t and q, dynamics.py:174-181 (flux times ONE average: halves folded, h_dx = 1/(2 dx))
        const T adt = (spu_c * (st_c + st_e) - spu_w * (st_w + st_c)) * h_dxj +
                      (spv_c * (st_c + st_sl) - spv_n * (st_n + st_c)) * h_dy;
        const T adq = (spu_c * (sq_c + sq_e) - spu_w * (sq_w + sq_c)) * h_dxj +
                      (spv_c * (sq_c + sq_s) - spv_n * (sq_n + sq_c)) * h_dy;
        // ---- the level below (k-1), from its tile: vertical fluxes and the anchor step.  At k == 0 the
        //      lower face carries sd[0] = 0: any finite value serves
        T su_m = su_c, sv_m = sv_c, st_m = st_c, sq_m = sq_c;
        if (k > 0) { su_m = own(tm, 0); sv_m = own(tm, 1); st_m = own(tm, 2); sq_m = own(tm, 3); }
        // ---- pgf v-part, dynamics.py:160,167-169: rho and phi of (j,i) and (j+1,i) rebuilt (rho_of / phi_up)
        const T sg = lv_sig[k];
        T ex_c, ex_s, phi_c, phi_s, st_s;
        if (have_lo) {
            ex_c = lo_ex_c; ex_s = lo_ex_s; phi_c = lo_phi_c; phi_s = lo_phi_s; st_s = lo_st_s;
        } else {
            ex_c = exner(sp_c * sg + ptop, tab);
            ex_s = exner(sp_s * sg + ptop, tab);
            st_s = st_sl;
            phi_c = tc[(TL::kPhi + r) * 64]; phi_s = tc[(TL::kPhi + r + 1) * 64];   // an even level's own anchor (unused when k is odd)
        }
        have_lo = (k & 1) != 0;
        if (have_lo) {                                            // k odd: level k-1 >= 0 is the anchor
            const T sg_lo = lv_sig[k - 1];
            lo_ex_c = exner(sp_c * sg_lo + ptop, tab);
            lo_ex_s = exner(sp_s * sg_lo + ptop, tab);
            lo_st_s = *srow(tm, 2); lo_phi_c = tm[(TL::kPhi + r) * 64]; lo_phi_s = tm[(TL::kPhi + r + 1) * 64];
            phi_c = phi_up(lo_phi_c, st_m, st_c, lo_ex_c, ex_c);
            phi_s = phi_up(lo_phi_s, lo_st_s, st_s, lo_ex_s, ex_s);
        }
        const T rho_c = rho_of(sp_c * sg + ptop, st_c, ex_c), rho_s = rho_of(sp_s * sg + ptop, st_s, ex_s);
        const T phiv = jph_c * ((phi_s - phi_c) * inv_dy);
        // jph(sig p) / jph(rho): the two halves cancel exactly
        const T pgv = (sg * sp_c + sg * sp_s) * rcp(rho_c + rho_s) * ((sp_s - sp_c) * inv_dy);
        // ---- vertical advection, dynamics.py:49-52 with iph(sd), jph(sd), sd
        const T inv_ds = lv_inv_dsig[k];
        const T fu = face_flux_v(su_c, su_m, (sd_c + sd_e) * T(0.5)), fv = face_flux_v(sv_c, sv_m, (sd_c + sd_s) * T(0.5));
        const T ft = face_flux_v(st_c, st_m, sd_c), fq = face_flux_v(sq_c, sq_m, sd_c);
        const T dus = -((fu - fu_up) * inv_ds);
        const T dvs = -((fv - fv_up) * inv_ds);
        const T dts = -((ft - ft_up) * inv_ds);
        const T dqs = -((fq - fq_up) * inv_ds);
        fu_up = fu; fv_up = fv; ft_up = ft; fq_up = fq;
        // ---- momentum, theta and q update, dynamics.py:186-219 (predictor: the stage state IS the base state)
        const T pgfu_c = tc[(TL::kOwn + r) * 64];
        T bu_c = su_c, bv_c = sv_c, bt_c = st_c, bq_c = sq_c;
        if (!same) {
            bu_c = tc[(TL::kOwn + R + r) * 64]; bv_c = tc[(TL::kOwn + 2 * R + r) * 64];
            bt_c = tc[(TL::kOwn + 3 * R + r) * 64]; bq_c = tc[(TL::kOwn + 4 * R + r) * 64];
        }
        const T pu = bu_c * iph_pb;
        const T pv = bv_c * jph_pb;
        const T pu_n = pu - (dut + dus + pgfu_c) * dt;
        const T pv_n = pv - (dvt + dvs + phiv + pgv) * dt;
        T u_n = pu_n * inv_pnu;
        T v_n = pv_n * inv_pnv;
        if (pole_edge) v_n *= T(0.0);                               // v_n[:, -1, :] *= 0, dynamics.py:222
        const T t_n = (bt_c * pb_c - (adt + dts) * dt) * inv_pn;
        const T q_n = (bq_c * pb_c - (adq + dqs) * dt) * inv_pn;
        cs_u = cs_acc(cs_u, u_n, dsg);
        cs_v = cs_acc(cs_v, v_n, dsg);
        if (store) {
            const long o = (long)j * L * W + kc;                 // rows to produce are interior: no wrap
            unsigned ol = ob;
            asm volatile("" : "+v"(ol));
            *(__attribute__((address_space(1))) T *)(sbase(a.ou + o) + ol) = u_n;
            *(__attribute__((address_space(1))) T *)(sbase(a.ov + o) + ol) = v_n;
            *(__attribute__((address_space(1))) T *)(sbase(a.ot + o) + ol) = t_n;
            *(__attribute__((address_space(1))) T *)(sbase(a.oq + o) + ol) = q_n;
        }
    };
    // two request sets that swap BY NAME (loop unrolled by two): a register copy of a value still in
    // flight would make the wave wait for it at once
    for (int k = k0;;) {
        load(q[1], max(k - 3, kmin));
        level(k);
        if (k - 2 >= kmin) put(q[0], bf);
        if (k == k_lo) break;
        __syncthreads();
        { const int t = bc; bc = bm; bm = bf; bf = t; }
        --k;
        load(q[0], max(k - 3, kmin));
        level(k);
        if (k - 2 >= kmin) put(q[1], bf);
        if (k == k_lo) break;
        __syncthreads();
        { const int t = bc; bc = bm; bm = bf; bf = t; }
        --k;
    }
    if (store && a.ocs_u) {                                      // whole column marched (nseg == 1)
        a.ocs_u[(long)j * W + i] = cs_u;
        a.ocs_v[(long)j * W + i] = cs_v;
    }
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
    const double *coslat, *sinlat, *lon;                     // [Hg], [Hg], [W]
    double *gt;                                              // ground temperature [H][W]
    T *dTdt, *dtg;                                           // tendencies out of the diagnostic form (3-D, 2-D scratch)
    double hour_angle, albedo, dt;
    int apply;                                               // 1: t, gt updated in place
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
constexpr int kRadThreads = 128;
constexpr int kRadTabs = 7;      // per level: tlw, clw_b_div, swfac, sig, dsig, 1 - tlw, G / (Cp dsig)
template <typename T, int LMAX>
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
        }
    }
    __syncthreads();
    constexpr double kSolar = 1.3608 * 1000.0, kSb = 5.67e-8, kCg = 1.13e6;   // constants.py:59,71,25
    double *p_lwb = (double *)rad_park_raw + threadIdx.x;
    double lwb_reg[LMAX > 0 ? LMAX : 1];
    const int i = blockIdx.x * kRadThreads + threadIdx.x;
    const int j = blockIdx.y;
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
    // true temperature and emission of one level (to_true_temp; grey_solar.py emission)
    const auto emission = [&](int k, double *tt_out) {
        const double tp = pc * sig(k) + ptop;
        const double th = LMAX > 0 ? (double)tcol[LMAX > 0 ? k : 0] : (double)t_inout[c3 + (long)k * W];
        const double tt = th * exner(tp, tab);
        const double t2 = tt * tt;
        *tt_out = tt;
        return (1 - tlw(k)) * kSb * (t2 * t2);
    };
    double B = 0.0, up = 0.0;
#pragma unroll(LMAX > 0 ? LMAX : 2)
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
#pragma unroll(LMAX > 0 ? LMAX : 2)
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
            const double tp = pc * sig(k) + ptop;
            const double tt_n = tt + dTdt * r.dt;
            t_inout[o] = (T)(tt_n * rcp(exner(tp, tab)));          // to_potential_temp
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
    PeBufs<double> d;
    PeBufs<float> f;
    int cur_i = 0;
    bool star_valid = false;
    int nseg = 1;                               // level segments of K4, chosen from the band's size
    int upd_rows = 7;                           // rows per workgroup of the row-group K4 (0: one-wave form)
    int cus = 256;
    bool pit2d = true;                          // pit from the column sums K4 leaves (nseg == 1, row-group K4)
    int nseg_edge = 1;                          // bands: level segments of the EDGE rows' K4 launch (see half_t)
    bool cs_valid[3] = {false, false, false};   // the state set's column sums belong to its winds
    int pack_set = -1;                          // >= 0: state set gcm_halo_pack reads (step_phase)
    double *stage3 = nullptr;                   // float64 transpose staging, host layout
    double *exner_tab = nullptr;
    FftPlan plan{};
    SuperPlan cplan{};
    double *gt = nullptr;                       // ground temperature [H][W] (column physics)
    double *stats_dev = nullptr;                // gcm_stats: block partials, then the area table
    std::vector<double> stats_host, area_host;
    double *rad_tab = nullptr;                  // 5 x [L] level tables of the last radiation call
    double *rad_geo = nullptr;                  // coslat[Hg], sinlat[Hg], lon[W]
    double rad_key[2] = {-1.0, -1.0};           // (t_lw, t_sw) the level tables were built for
    std::vector<double> rad_tab_host, rad_geo_host, rad_latlon;   // host copies (upload sources, change detection)
    std::vector<hipEvent_t> *ev = nullptr;
    size_t *ev_used = nullptr;
    hipStream_t aux = nullptr;                  // second stream of a stage (K2a -> K3), see half_t
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // latitude band with registered send buffers: the edge rows of a stage are updated and packed
    // on `aux` while the interior rows run on the caller's stream (pe25d_step_phase)
    void *send_buf[2] = {nullptr, nullptr};
    hipEvent_t ev_a = nullptr, ev_edges = nullptr;
    bool edges_pending = false;
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

// the filter kernels are instantiated per widest composite radix (12 / 16 / 25), so that a plan
// of small radices (1440 = 10.12.12) is not held to the register budget of a 25-point butterfly;
// 0 = generic ping-pong passes
template <typename T> using FilterKernel = void (*)(PeArgsT<T>);
// plans with their own instantiation (only their passes compiled in): the row lengths of the
// BASELINE configs and the powers of 16
constexpr unsigned kMask1440 = pass_bit(5, 2) | pass_bit(4, 3);                    // 1440, 720, 360, 120 ...
constexpr unsigned kMask2880 = pass_bit(5, 3) | pass_bit(4, 3) | pass_bit(4, 4);   // 2880
constexpr unsigned kMask4096 = pass_bit(4, 4);                                     // 256, 4096
template <typename T>
static FilterKernel<T> spu_filter_kernel_for(const SuperPlan &P) {
    if (!P.ok) return pe_spu_filter_kernel<T, 0>;
    if (P.mask == kMask1440) return pe_spu_filter_kernel<T, 12, kMask1440>;
    if (P.mask == kMask2880) return pe_spu_filter_kernel<T, 16, kMask2880>;
    if (P.mask == kMask4096) return pe_spu_filter_kernel<T, 16, kMask4096>;
    if (P.maxr <= 12) return pe_spu_filter_kernel<T, 12>;
    if (P.maxr <= 16) return pe_spu_filter_kernel<T, 16>;
    return pe_spu_filter_kernel<T, 25>;
}
template <typename T> using FilterLoopKernel = void (*)(PeArgsT<T>, int);
template <typename T>
static FilterLoopKernel<T> spu_filter_loop_kernel_for(const SuperPlan &P) {
    if (!P.ok || P.npass > 4) return nullptr;
    // the plans these masks stand for start with a pass of radix 5.2 / 5.3 / 4.4 (make_super_plan)
    if (P.mask == kMask1440 && P.r1[0] * P.r2[0] == 10) return pe_spu_filter_loop_kernel<T, 12, kMask1440, 10>;
    if (P.mask == kMask2880 && P.r1[0] * P.r2[0] == 15) return pe_spu_filter_loop_kernel<T, 16, kMask2880, 15>;
    if (P.mask == kMask1440) return pe_spu_filter_loop_kernel<T, 12, kMask1440>;
    if (P.mask == kMask2880) return pe_spu_filter_loop_kernel<T, 16, kMask2880>;
    if (P.mask == kMask4096) return pe_spu_filter_loop_kernel<T, 16, kMask4096>;
    if (P.maxr <= 12) return pe_spu_filter_loop_kernel<T, 12>;
    if (P.maxr <= 16) return pe_spu_filter_loop_kernel<T, 16>;
    return pe_spu_filter_loop_kernel<T, 25>;
}
template <typename T>
static FilterKernel<T> pgf_filter_kernel_for(const SuperPlan &P) {
    if (!P.ok) return pe_pgf_filter_kernel<T, 0>;
    if (P.mask == kMask1440) return pe_pgf_filter_kernel<T, 12, kMask1440>;
    if (P.mask == kMask2880) return pe_pgf_filter_kernel<T, 16, kMask2880>;
    if (P.mask == kMask4096) return pe_pgf_filter_kernel<T, 16, kMask4096>;
    if (P.maxr <= 12) return pe_pgf_filter_kernel<T, 12>;
    if (P.maxr <= 16) return pe_pgf_filter_kernel<T, 16>;
    return pe_pgf_filter_kernel<T, 25>;
}
template <typename T>
static FilterKernel<T> pit2d_kernel_for(const SuperPlan &P) {
    if (!P.ok) return pe_pit2d_kernel<T, 0>;
    if (P.mask == kMask1440) return pe_pit2d_kernel<T, 12, kMask1440>;
    if (P.mask == kMask2880) return pe_pit2d_kernel<T, 16, kMask2880>;
    if (P.mask == kMask4096) return pe_pit2d_kernel<T, 16, kMask4096>;
    if (P.maxr <= 12) return pe_pit2d_kernel<T, 12>;
    if (P.maxr <= 16) return pe_pit2d_kernel<T, 16>;
    return pe_pit2d_kernel<T, 25>;
}
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
        hipFuncSetAttribute((const void *)pe_radiation_kernel<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(double) * (size_t)L * kRadThreads)) != hipSuccess ||
        hipFuncSetAttribute((const void *)pe_update_rows_kernel<T, 7, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(7, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)pe_update_rows_kernel<T, 7, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(7, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)pe_update_rows_kernel<T, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)upd_lds_bytes<T>(3, L)) != hipSuccess ||
        hipFuncSetAttribute((const void *)pe_update_rows_kernel<T, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
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
        m->upd_rows = m->H <= 256 ? 3 : 7;
        bool forced = false;
        if (const char *e = getenv("GCM_PE_LEVEL_SEGMENTS")) { want = atoi(e); forced = true; }
        if (const char *e = getenv("GCM_PE_PIT2D")) m->pit2d = atoi(e) != 0;      // 0: pit from the 3-D fields (pe_pit_kernel)
        if (const char *e = getenv("GCM_PE_UPDATE_ROWS")) {      // 0: the one-wave update kernel; 3, 7: rows per group
            const int v = atoi(e);
            m->upd_rows = (v == 0 || v == 3 || v == 7) ? v : 7;
        }
        const int cap = std::min(kMaxSeg, std::max(1, L / 4));
        m->nseg = (int)std::max(1L, std::min((long)cap, want));
        // the row-group K4 fills the chip with whole columns (a 90-row band: 2 % slower than in two
        // segments) and then leaves the column sums pit needs: segments only on request, or for the
        // one-wave kernel
        if (m->upd_rows > 0 && !forced) m->nseg = 1;
        if (!m->wrap && m->upd_rows > 0 && m->nseg == 1 && m->pit2d && m->H > 2 * kGhost) {
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
    const char *no_aux = getenv("GCM_PE_SINGLE_STREAM");      // diagnostic: one chain, one stream
    // a plain stream: a high-priority one finished the edge rows earlier, but in some processes
    // (depending on how many streams existed before) the whole step then ran at half speed
    if (!(no_aux && no_aux[0] == '1')) {
        m->aux = concurrent_stream(main_stream, nullptr);
        if (!m->aux) return bad("second stream");
    }
    if (hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
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
    if (m->aux) (void)hipStreamDestroy(m->aux);
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

// one Euler stage over rows [j0, j1): state `stage_set` -> `out_set`, base = current.
// mode 0: everything; mode 1: K1-K3 on all rows + K4 on the two edge rows of either side (the rows
// a neighbouring band needs); mode 2: K4 on the remaining interior rows.  Modes 1 + 2 == mode 0.
template <typename T>
static void half_t(Pe25d *m, int stage_set, int out_set, double dt, int j0, int j1, hipStream_t s, int mode) {
    if (j1 <= j0) return;
    PeArgsT<T> a = make_args<T>(m, stage_set, out_set, dt);
    const int W = m->W, L = m->L;
    const int ext = m->wrap ? 0 : 1;             // intermediates are also needed on row j1 (south)
    const size_t lds = filter_lds_bytes<T>(m);
    const int fft_threads = m->cplan.ok ? m->cplan.threads : kFftThreads;
    const int pairs = (L + 1) / 2;
    // pit from the 2-D column sums (pe_pit2d_kernel) where K4 marches whole columns and can leave them
    const bool p2 = m->pit2d && a.nseg == 1 && m->upd_rows > 0;
    if (p2) {
        a.ocs_u = bufs<T>(m).cs[out_set][0];
        a.ocs_v = bufs<T>(m).cs[out_set][1];
    }
    if (mode != 2) {
        // two independent chains: K1 -> K2b (mass flux, pit) on the caller's stream, K2a -> K3
        // (geopotential, filtered pressure-gradient force) on the handle's second stream.  The FFT
        // kernels are latency bound and the column kernels bandwidth bound, so they share the chip.
        // (Which chain sits on which stream makes no difference: they take about equally long.)
        hipStream_t s2 = m->aux ? m->aux : s;
        if (m->aux) {
            (void)hipEventRecord(m->ev_fork, s);
            (void)hipStreamWaitEvent(m->aux, m->ev_fork, 0);
        }
        a.j0 = j0;
        a.j1 = j1 + ext;
        {
            const long tiles = (long)((W + kColThreads - 1) / kColThreads) * (a.j1 - a.j0);
            const dim3 gg((unsigned)((tiles + 7) / 8 * 8));
            if (L <= 24) hipLaunchKernelGGL((pe_geopot_kernel<T, 24>), gg, dim3(kColThreads), 0, s2, a);
            else if (L <= 40) hipLaunchKernelGGL((pe_geopot_kernel<T, 40>), gg, dim3(kColThreads), 0, s2, a);
            else hipLaunchKernelGGL((pe_geopot_kernel<T, 0>), gg, dim3(kColThreads), sizeof(T) * (size_t)L * kColThreads, s2, a);
        }
        static const bool no_loop = getenv("GCM_PE_FILTER_NO_LOOP") != nullptr;        // diagnostic: one workgroup per pair
        const FilterLoopKernel<T> k1 = (m->cfg.filter && W > 1 && !no_loop) ? spu_filter_loop_kernel_for<T>(m->cplan) : nullptr;
        if (k1) {
            // all pairs of a row in one workgroup when there are rows enough to fill the chip, else groups
            const int rows = a.j1 - a.j0;
            const int groups = std::min(pairs, std::max(1, (3 * m->cus + rows - 1) / rows));
            const int ppw = (pairs + groups - 1) / groups;
            hipLaunchKernelGGL(k1, dim3(rows, (pairs + ppw - 1) / ppw), dim3(fft_threads), filter_loop_lds_bytes<T>(m), s, a, ppw);
        } else {
            hipLaunchKernelGGL(spu_filter_kernel_for<T>(m->cplan), dim3(a.j1 - a.j0, pairs), dim3(fft_threads), lds, s, a);
        }
        if (p2) {
            // rows whose sums no K4 left: all of a freshly set state, else a band's two ghost rows
            // next to its own (pit of row j takes V of row j - 1; the intermediates extend to row j1)
            PeArgsT<T> c = a;
            if (!m->cs_valid[stage_set]) {
                c.j0 = m->wrap ? 0 : -1;
                c.j1 = m->H + ext;
                c.jb0 = c.jb1 = 0;
                m->cs_valid[stage_set] = true;
            } else if (m->nseg_edge > 1) {                   // + the own edge rows (marched in segments: no sums from K4)
                c.j0 = -1; c.j1 = kGhost;
                c.jb0 = m->H - kGhost; c.jb1 = m->H + 1;
            } else {
                c.j0 = -1; c.j1 = 0;
                c.jb0 = m->H; c.jb1 = m->H + 1;
            }
            if (!m->wrap || c.j1 - c.j0 > 1) {
                const int rows = (c.j1 - c.j0) + (c.jb1 - c.jb0);
                hipLaunchKernelGGL(pe_colsum_kernel<T>, dim3((unsigned)((W + 255) / 256) * rows), dim3(256), 0, s, c);
            }
            hipLaunchKernelGGL(pit2d_kernel_for<T>(m->cplan), dim3(a.j1 - a.j0), dim3(fft_threads), lds + sizeof(T) * (size_t)W, s, a);
        } else {
            const long tiles = (long)((W + 255) / 256) * (a.j1 - a.j0);
            hipLaunchKernelGGL(pe_pit_kernel<T>, dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(256), 0, s, a);
        }
        if (mode == 1 && async_edges(m) && m->aux) (void)hipEventRecord(m->ev_a, s);
        a.j1 = j1;
        // (a looping form of this filter, as K1's, was built and is 25 % SLOWER: its requests and the
        // per-column thermodynamics push it to 187 VGPRs, two waves per SIMD instead of four)
        // (launched with whole waves -- 192 threads for the 144 butterflies of a 1440 row, so that the
        // per-column thermodynamics ahead of the transform fills its lanes -- it takes the same time)
        hipLaunchKernelGGL(pgf_filter_kernel_for<T>(m->cplan), dim3((unsigned)(8 * ((a.j1 - a.j0 + 7) / 8) * pairs)), dim3(fft_threads), lds, s2, a);
        if (m->aux) {
            (void)hipEventRecord(m->ev_join, m->aux);
            (void)hipStreamWaitEvent(s, m->ev_join, 0);
        }
    }
    auto update_rows = [&](int r0, int r1, int rb0, int rb1, hipStream_t st) {   // rows [r0, r1) and [rb0, rb1), one launch
        const int rows = std::max(0, r1 - r0) + std::max(0, rb1 - rb0);
        if (rows <= 0) return;
        a.j0 = r0;
        a.j1 = std::max(r0, r1);
        a.jb0 = rb0;
        a.jb1 = std::max(rb0, rb1);
        if (m->upd_rows > 0) {
            const int Rg = m->upd_rows;
            const long groups = (std::max(0, r1 - r0) + Rg - 1) / Rg + (std::max(0, rb1 - rb0) + Rg - 1) / Rg;
            // 8 XCDs x (row group, segment) pairs per XCD x column tiles (see the kernel's index map)
            const long rs_per_xcd = (groups * a.nseg + 7) / 8;
            const dim3 gg((unsigned)(8 * rs_per_xcd * ((W + kUpdCols - 1) / kUpdCols)));
            const size_t lds = upd_lds_bytes<T>(Rg, L);
            const bool same = a.u == a.su;
            if (Rg == 7 && same) hipLaunchKernelGGL((pe_update_rows_kernel<T, 7, true>), gg, dim3(64 * 8), lds, st, a);
            else if (Rg == 7) hipLaunchKernelGGL((pe_update_rows_kernel<T, 7, false>), gg, dim3(64 * 8), lds, st, a);
            else if (same) hipLaunchKernelGGL((pe_update_rows_kernel<T, 3, true>), gg, dim3(64 * 4), lds, st, a);
            else hipLaunchKernelGGL((pe_update_rows_kernel<T, 3, false>), gg, dim3(64 * 4), lds, st, a);
            return;
        }
        const long tiles = (long)((W + kUpdThreads - 1) / kUpdThreads) * rows * a.nseg;
        hipLaunchKernelGGL(pe_update_kernel<T>, dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(kUpdThreads), 0, st, a);
    };
    m->cs_valid[out_set] = p2;                   // (modes 1 + 2 together cover the rows)
    const bool split = mode != 0 && (j1 - j0) > 2 * kGhost;
    if (mode == 0) {
        tick(m, s);
        update_rows(j0, j1, 0, 0, s);
        tick(m, s);
    } else if (mode == 1) {
        // the rows the neighbours wait for (an unsplittable, tiny band: all of them).  With send
        // buffers registered they are updated and packed on the second stream, which has the
        // K2a -> K3 chain already and waits for K1 -> K2b here; the caller's stream goes straight
        // on to the interior rows (mode 2), so the two launches share the chip.
        const bool as = async_edges(m);
        hipStream_t se = as && m->aux ? m->aux : s;
        if (as && m->aux) (void)hipStreamWaitEvent(m->aux, m->ev_a, 0);
        if (split && p2 && m->nseg_edge > 1) {
            // the edge rows in level segments: a quarter of the chain of dependent levels, so the pack
            // and the exchange start while the interior rows are still at work
            PeArgsT<T> keep = a;
            a.nseg = m->nseg_edge;
            a.ocs_u = a.ocs_v = nullptr;
            // the partial sums of conv they start from (launched behind K1 instead, beside K3, this
            // kernel takes 14 us instead of 50 -- but on the stage's critical chain, a net loss)
            a.j0 = j0; a.j1 = j0 + kGhost + 1;            // (K4 of row j also takes the sums of row j + 1)
            a.jb0 = j1 - kGhost; a.jb1 = j1 + 1;
            hipLaunchKernelGGL(pe_part_kernel<T>, dim3((unsigned)((W + 255) / 256) * (2 * kGhost + 2)), dim3(256), 0, se, a);
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
            launch_seg_copy(c, se);
            (void)hipEventRecord(m->ev_edges, se);
            m->edges_pending = true;
        }
    } else {
        if (split) update_rows(j0 + kGhost, j1 - kGhost, 0, 0, s);
        // whatever follows on the caller's stream also follows the edge rows
        if (async_edges(m) && m->edges_pending) (void)hipStreamWaitEvent(s, m->ev_edges, 0);
        m->edges_pending = false;
    }
}

static void half(Pe25d *m, int stage_set, int out_set, double dt, int j0, int j1, hipStream_t s, int mode = 0) {
    if (m->f32) half_t<float>(m, stage_set, out_set, dt, j0, j1, s, mode);
    else half_t<double>(m, stage_set, out_set, dt, j0, j1, s, mode);
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
int pe25d_step_phase(Pe25d *m, int phase, double dt, hipStream_t s, std::string *err) {
    if (m->wrap) {
        *err = "step_phase: handle is not a latitude band";
        return GCM_ERR_STATE;
    }
    switch (phase) {
        case 0:
            m->star_valid = true;
            m->pack_set = 2;
            half(m, m->cur_i, 2, dt, 0, m->H, s, 1);
            break;
        case 1:
            half(m, m->cur_i, 2, dt, 0, m->H, s, 2);
            break;
        case 2:
            m->pack_set = 1 - m->cur_i;
            half(m, 2, 1 - m->cur_i, dt, 0, m->H, s, 1);
            break;
        case 3:
            half(m, 2, 1 - m->cur_i, dt, 0, m->H, s, 2);
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
    m->send_buf[0] = north;
    m->send_buf[1] = south;
    m->edges_pending = false;
    return GCM_OK;
}

// host-side step state (which state set is current, ...): gcm_band_run replays a captured step as a
// hipGraph, which runs none of the host code that advances this state
void pe25d_host_state(Pe25d *m, bool save, int st[4]) {
    if (save) {
        st[0] = m->cur_i; st[1] = m->star_valid; st[2] = m->pack_set;
        st[3] = (m->edges_pending ? 1 : 0) | (m->cs_valid[0] ? 2 : 0) | (m->cs_valid[1] ? 4 : 0) | (m->cs_valid[2] ? 8 : 0);
    } else {
        m->cur_i = st[0]; m->star_valid = st[1] != 0; m->pack_set = st[2]; m->edges_pending = (st[3] & 1) != 0;
        for (int n = 0; n < 3; ++n) m->cs_valid[n] = (st[3] & (2 << n)) != 0;
    }
}
// a step recorded now replays correctly later only if it does not contain the one-off column sums
// of a freshly set state (see half_t)
bool pe25d_step_is_steady(const Pe25d *m) { return !m->pit2d || m->nseg != 1 || m->upd_rows == 0 || m->cs_valid[m->cur_i]; }
int pe25d_parity(const Pe25d *m) { return m->cur_i; }
void pe25d_advance_step(Pe25d *m) {          // what one full band step (phases 0..3) leaves behind
    m->cur_i = 1 - m->cur_i;
    m->star_valid = false;
    m->pack_set = -1;
    m->edges_pending = false;
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
    return (m->f32 ? sizeof(float) : sizeof(double)) * (size_t)kGhost * m->W * (1 + 4 * (size_t)m->L);
}

template <typename T>
static void halo_t(Pe25d *m, bool pack, int side, void *dev_buf, SegCopy *c) {
    PeBufs<T> &Bf = bufs<T>(m);
    // unpack: ghosts of the predicted state once it exists, else of the current state;
    // pack: the same, unless a step_phase call named the set whose edge rows were just produced
    int set = m->star_valid ? 2 : m->cur_i;
    if (pack && m->pack_set >= 0) set = m->pack_set;
    if (!pack && m->pack_set >= 0 && m->pack_set != 2) set = m->pack_set;   // new-state ghosts arrive before the swap
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
    if (!m->gt) {
        void *d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess || hipMemsetAsync(d, 0, bytes, s) != hipSuccess) {
            *err = "hip: ground temperature allocation failed";
            return GCM_ERR_HIP;
        }
        m->allocs.push_back(d);
        m->gt = (double *)d;
    }
    hipError_t e = set ? hipMemcpyAsync(m->gt, in, bytes, hipMemcpyHostToDevice, s)
                       : hipMemcpyAsync(out, m->gt, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { *err = "hip: ground temperature transfer failed"; return GCM_ERR_HIP; }
    return GCM_OK;
}

template <typename T>
static int radiation_launch(Pe25d *m, bool apply, double dt, double hour_angle, double albedo,
                            double *dTdt_host, double *dtg_host, hipStream_t s, std::string *err) {
    PeBufs<T> &B = bufs<T>(m);
    const int W = m->W, H = m->H, L = m->L, Hg = m->Hg;
    PeArgsT<T> a = make_args<T>(m, m->cur_i, m->cur_i, dt);
    RadArgsT<T> r{};
    r.tlw = m->rad_tab; r.tsw = r.tlw + L; r.csw_top = r.tsw + L; r.clw_b_div = r.csw_top + L; r.swfac = r.clw_b_div + L;
    r.coslat = m->rad_geo; r.sinlat = m->rad_geo + Hg; r.lon = m->rad_geo + 2 * Hg;
    r.gt = m->gt;
    r.dTdt = B.pgfu; r.dtg = B.pit;
    r.hour_angle = hour_angle;
    r.albedo = albedo; r.dt = dt; r.apply = apply ? 1 : 0;
    {
        const dim3 gg((W + kRadThreads - 1) / kRadThreads, H);
        T *th = B.st[m->cur_i][GCM_T];
        static const bool generic = getenv("GCM_PE_RAD_GENERIC") != nullptr;     // diagnostic: the LDS-parked form
        if (L <= 24 && !generic) hipLaunchKernelGGL((pe_radiation_kernel<T, 24>), gg, dim3(kRadThreads), 0, s, a, r, th);
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

// basic_grey_radiation (+ optional in-place solar_timestep).  dTdt_host / dtg_host may be null.
int pe25d_radiation(Pe25d *m, bool apply, double dt, double utc, double t_lw, double t_sw, double albedo,
                    const double *lat, const double *lon, double *dTdt_host, double *dtg_host,
                    hipStream_t s, std::string *err) {
    if (!m->gt) { *err = "radiation: set the ground temperature first (gcm_set_ground)"; return GCM_ERR_STATE; }
    if (!lat || !lon) { *err = "radiation: lat and lon tables are required"; return GCM_ERR_ARG; }
    const int W = m->W, L = m->L, Hg = m->Hg;
    if (m->rad_key[0] != t_lw || m->rad_key[1] != t_sw || !m->rad_tab) {
        // level tables, same expression order as grey_solar.py:323-333,377-385,541
        std::vector<double> &T = m->rad_tab_host;
        T.assign((size_t)5 * L, 0.0);
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
        if (!m->rad_tab && !dev_upload<double>(m, &m->rad_tab, nullptr, (size_t)5 * L)) {
            *err = "hip: radiation table allocation failed"; return GCM_ERR_HIP;
        }
        // the host copy lives in the handle until the next change, so the asynchronous upload may
        // read it after this call returns; a change waits for the previous upload first
        if (hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpyAsync(m->rad_tab, T.data(), sizeof(double) * 5 * L, hipMemcpyHostToDevice, s) != hipSuccess) {
            *err = "hip: radiation table upload failed"; return GCM_ERR_HIP;
        }
        m->rad_key[0] = t_lw; m->rad_key[1] = t_sw;
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
    }
    const double hour_angle = utc / (-24 * 3600.0) * 360 * (M_PI / 180);      // grey_solar.py:51
    return m->f32 ? radiation_launch<float>(m, apply, dt, hour_angle, albedo, dTdt_host, dtg_host, s, err)
                  : radiation_launch<double>(m, apply, dt, hour_angle, albedo, dTdt_host, dtg_host, s, err);
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
