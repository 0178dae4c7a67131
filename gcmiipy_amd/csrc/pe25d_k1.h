// K1 of GCM_PE25D (see pe25d_kernels.hip): spu = arakawa_1977(su * iph(sp)), and the picker that
// chooses the instantiation for a plan.  Included by pe25d_k1_f64.hip / pe25d_k1_f32.hip only.
#pragma once
#include "pe25d_dev.h"
#include "pe25d_pit2d.h"

namespace gcm {

// ---------------------------------------------------------------- K1: spu = filter(su * iph(sp))
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
// NW: base twiddles kept in registers (FilterConsts): 3 for plans of up to three passes, 5 for four
template <typename T, int MAXR, unsigned MASK = 0, int NIN = MAXR, int NW = 5>
__global__ __launch_bounds__(512)
void pe_spu_filter_loop_kernel(PeArgsT<T> a, int pairs_per_wg, int pit_block) {
    using V = typename Vec2<T>::type;
    extern __shared__ unsigned char lds_raw[];
    V *x = (V *)lds_raw;
    // rows of the launch: [j0, j1), then [jb0, jb1) (a band's launch for the rows that need ghost data: rows 0, 1 and
    // H - 1, H); of those, spu is formed for [spu_j0, spu_j1) and pit for [pit_j0, pit_j1) (uniform early exits)
    const int nr0 = a.j1 - a.j0;
    const int j = (int)blockIdx.x < nr0 ? a.j0 + (int)blockIdx.x : a.jb0 + ((int)blockIdx.x - nr0);
    if ((int)blockIdx.y == pit_block) {
        // one more workgroup per row: pit and p_n from the 2-D column sums (pe_pit2d_row) -- independent
        // of the pairs' transforms, and a launch less on the stage's dependency chain
        if (j < a.pit_j0 || j >= a.pit_j1) return;
        pe_pit2d_row<T, MAXR, MASK>(a, x, j);
        return;
    }
    if (j < a.spu_j0 || j >= a.spu_j1) return;
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W, L = a.L;
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
    FilterConsts<T, NW> c;
    filter_consts<T, NW>(c, a.tw, a.cplan, W);
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
        FilterConsts<T, NW> cc = c;
#pragma unroll
        for (int n = 0; n < NW; ++n) asm volatile("" : "+v"(cc.w[n].x), "+v"(cc.w[n].y));
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
        filter_rows_hoisted<MAXR, MASK, T, NW>(x, first, after_first, store, a.tw, a.cplan, W, cc, tid);
    }
}

template <typename T>
FilterKernel<T> spu_filter_kernel_for(const SuperPlan &P) {
    if (!P.ok) return pe_spu_filter_kernel<T, 0>;
    if (P.mask == kMask1440) return pe_spu_filter_kernel<T, 12, kMask1440>;
    if (P.mask == kMask2880) return pe_spu_filter_kernel<T, 16, kMask2880>;
    if (P.mask == kMask4096) return pe_spu_filter_kernel<T, 16, kMask4096>;
    if (P.maxr <= 12) return pe_spu_filter_kernel<T, 12>;
    if (P.maxr <= 16) return pe_spu_filter_kernel<T, 16>;
    return pe_spu_filter_kernel<T, 25>;
}
template <typename T>
FilterLoopKernel<T> spu_filter_loop_kernel_for(const SuperPlan &P) {
    if (!P.ok || P.npass > 4 || P.npass < 2) return nullptr;
    // the plans these masks stand for start with a pass of radix 5.2 / 5.3 / 4.4 (make_super_plan)
    if (P.mask == kMask1440 && P.r1[0] * P.r2[0] == 10 && P.npass <= 3) return pe_spu_filter_loop_kernel<T, 12, kMask1440, 10, 3>;
    if (P.mask == kMask2880 && P.r1[0] * P.r2[0] == 15 && P.npass <= 3) return pe_spu_filter_loop_kernel<T, 16, kMask2880, 15, 3>;
    if (P.mask == kMask1440) return pe_spu_filter_loop_kernel<T, 12, kMask1440>;
    if (P.mask == kMask2880) return pe_spu_filter_loop_kernel<T, 16, kMask2880>;
    if (P.mask == kMask4096) return pe_spu_filter_loop_kernel<T, 16, kMask4096>;
    if (P.maxr <= 12) return pe_spu_filter_loop_kernel<T, 12>;
    if (P.maxr <= 16) return pe_spu_filter_loop_kernel<T, 16>;
    return pe_spu_filter_loop_kernel<T, 25>;
}

}  // namespace gcm
