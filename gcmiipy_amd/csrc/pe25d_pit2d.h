// K2b', the 2-D form of pit (see pe25d_kernels.hip), as a device function: one workgroup filters ONE row
// per latitude and differences it.  It runs as extra workgroups of K1's launch (pe_spu_filter_loop_kernel)
// or, for plans without a looping K1, as a kernel of its own (pe_pit2d_kernel, pe25d_k3.h).
#pragma once
#include "pe25d_dev.h"

namespace gcm {

// one workgroup per row j: pit and p_n = p - pit dt (dynamics.py:38-40,194)
template <typename T, int MAXR, unsigned MASK = 0>
__device__ __forceinline__ void pe_pit2d_row(const PeArgsT<T> &a, typename Vec2<T>::type *x, const int j) {
    using V = typename Vec2<T>::type;
    const Idx ix{a.W, a.H, a.L, a.wrap};
    const int W = a.W;
    T *fx = (T *)(x + (MAXR > 0 ? 1 : 2) * W);                  // the filtered row, after the transform's workspace
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


}  // namespace gcm
