// K3 of GCM_PE25D (see pe25d_kernels.hip): pgfu = arakawa_1977(pgu + phiu); and the 2-D form of pit
// (one filtered row per latitude), with their pickers.  Included by pe25d_k3_f64.hip / pe25d_k3_f32.hip only.
#pragma once
#include "pe25d_dev.h"
#include "pe25d_pit2d.h"

namespace gcm {

// pit and p_n of row j0 + blockIdx.x as a kernel of its own (plans without a looping K1)
template <typename T, int MAXR, unsigned MASK = 0>
__global__ __launch_bounds__(512) void pe_pit2d_kernel(PeArgsT<T> a) {
    extern __shared__ unsigned char lds_raw[];
    pe_pit2d_row<T, MAXR, MASK>(a, (typename Vec2<T>::type *)lds_raw, a.j0 + (int)blockIdx.x);
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
    __syncthreads();
    // what a column needs from memory, and what is made of it
    struct Raw { T pc, pe, t0, t1, ph; };
    const auto request = [=](int i) {
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
    // A wave covers 63 columns; its lane 63 computes the column after them only to hand it to lane 62 (the east
    // neighbour of every column comes from the next lane, DPP), so consecutive waves overlap by one column and
    // nothing has to cross a wave.  (Rounds 1-2 filled a table of the columns that are multiples of 64 first:
    // one more memory latency and a barrier at the head of every workgroup.)  The row's last column takes
    // column 0 as its east neighbour the same way: column indices wrap.
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6, nwav = blockDim.x >> 6;
    const int nchunks = (W + 62) / 63;
    const auto value = [&](const Raw &r) {
        const PgfCol<T> c = column_of(r);
        PgfCol<T> e;
        e.rho0 = from_east(c.rho0); e.rho1 = from_east(c.rho1);
        e.phi0 = from_east(c.phi0); e.phi1 = from_east(c.phi1);
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
    // of one per column), then worked off; every lane of a wave goes through the loop body
    const auto sweep = [&](const auto &sink) {
        for (int ch0 = wv; ch0 < nchunks; ch0 += kPgfBatch * nwav) {
            Raw r[kPgfBatch];
#pragma unroll
            for (int m = 0; m < kPgfBatch; ++m) r[m] = request((min(ch0 + m * nwav, nchunks - 1) * 63 + lane) % W);
#pragma unroll
            for (int m = 0; m < kPgfBatch; ++m) {
                const int ch = ch0 + m * nwav;
                const int i = ch * 63 + lane;
                const V v = value(r[m]);
                if (ch < nchunks && lane < 63 && i < W) sink(i, v);
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

template <typename T>
FilterKernel<T> pgf_filter_kernel_for(const SuperPlan &P) {
    if (!P.ok) return pe_pgf_filter_kernel<T, 0>;
    if (P.mask == kMask1440) return pe_pgf_filter_kernel<T, 12, kMask1440>;
    if (P.mask == kMask2880) return pe_pgf_filter_kernel<T, 16, kMask2880>;
    if (P.mask == kMask4096) return pe_pgf_filter_kernel<T, 16, kMask4096>;
    if (P.maxr <= 12) return pe_pgf_filter_kernel<T, 12>;
    if (P.maxr <= 16) return pe_pgf_filter_kernel<T, 16>;
    return pe_pgf_filter_kernel<T, 25>;
}
template <typename T>
FilterKernel<T> pit2d_kernel_for(const SuperPlan &P) {
    if (!P.ok) return pe_pit2d_kernel<T, 0>;
    if (P.mask == kMask1440) return pe_pit2d_kernel<T, 12, kMask1440>;
    if (P.mask == kMask2880) return pe_pit2d_kernel<T, 16, kMask2880>;
    if (P.mask == kMask4096) return pe_pit2d_kernel<T, 16, kMask4096>;
    if (P.maxr <= 12) return pe_pit2d_kernel<T, 12>;
    if (P.maxr <= 16) return pe_pit2d_kernel<T, 16>;
    return pe_pit2d_kernel<T, 25>;
}

}  // namespace gcm
