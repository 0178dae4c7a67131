// Zonal Fourier filter building blocks for GCM_PE25D (low_pass.py:41-78): complex FFTs of one
// grid row held in LDS, used by pe_spu_filter_kernel / pe_pgf_filter_kernel (pe25d_kernels.hip).
//   * composite-radix in-place Stockham passes for {2,3,5}-smooth lengths (the product path)
//   * a generic mixed-radix ping-pong Stockham FFT for every other even length
//   * the host-side planners for both
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

namespace gcm {

constexpr int kMaxRadices = 24;

struct FftPlan {
    int n, nrad;
    int rad[kMaxRadices];
    unsigned magic[kMaxRadices];   // ceil(2^32 / Ns) per pass: b / Ns == umulhi(b, magic) for b*Ns < 2^32
};

constexpr int kMaxSuper = 8;
struct SuperPlan {
    int ok;                        // 0: use the generic ping-pong path
    int npass, threads, maxr;      // workgroup size = widest pass rounded up to whole waves
    unsigned mask;                 // pass_bit() of every pass
    int r1[kMaxSuper], r2[kMaxSuper];
    unsigned magic[kMaxSuper];     // ceil(2^32 / Ns) of the pass
    unsigned imagic[kMaxSuper];    // the same for the inverse transform, whose passes run in REVERSED radix order
};

template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };
template <typename V> using Sc = decltype(V().x);                    // scalar type of a 2-vector
template <typename V> __device__ __forceinline__ V mkv(Sc<V> a, Sc<V> b) { V r; r.x = a; r.y = b; return r; }

// ---------------------------------------------------------------- device side
// Stockham autosort, mixed radix.  x -> result returned in x or y (pointer returned).
template <typename V>
__device__ __forceinline__ V cmul(V a, V b) {
    return mkv<V>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <typename V>
__device__ __forceinline__ V cadd(V a, V b) { return mkv<V>(a.x + b.x, a.y + b.y); }
template <typename V>
__device__ __forceinline__ V csub(V a, V b) { return mkv<V>(a.x - b.x, a.y - b.y); }
// -i z (forward transforms) or +i z (inverse)
template <bool INV, typename V>
__device__ __forceinline__ V rot(V z) {
    return INV ? mkv<V>(-z.y, z.x) : mkv<V>(z.y, -z.x);
}
template <bool INV, typename V>
__device__ __forceinline__ V twid(const V *tw, int idx) {
    V w = tw[idx];
    if (INV) w.y = -w.y;
    return w;
}

template <bool INV, typename V>
__device__ V *fft_lds(V *x, V *y, const V *tw, const FftPlan &P) {
    using T = Sc<V>;
    const int N = P.n;
    int Ns = 1;
    for (int pass = 0; pass < P.nrad; ++pass) {
        const int r = P.rad[pass];
        const int nb = N / r;
        const int tstep = N / (Ns * r);  // twiddle index step
        const unsigned magic = P.magic[pass];
        for (int b = threadIdx.x; b < nb; b += blockDim.x) {
            const int blk = Ns == 1 ? b : (int)__umulhi((unsigned)b, magic);
            const int k = b - blk * Ns;
            const int j0 = blk * Ns * r + k;
            const int t1 = k * tstep;
            if (r == 2) {
                const V a0 = x[b], a1 = cmul(x[b + nb], twid<INV>(tw, t1));
                y[j0] = cadd(a0, a1);
                y[j0 + Ns] = csub(a0, a1);
            } else if (r == 4) {
                const V a0 = x[b];
                const V a1 = cmul(x[b + nb], twid<INV>(tw, t1));
                const V a2 = cmul(x[b + 2 * nb], twid<INV>(tw, 2 * t1));
                const V a3 = cmul(x[b + 3 * nb], twid<INV>(tw, 3 * t1));
                const V s02 = cadd(a0, a2), d02 = csub(a0, a2);
                const V s13 = cadd(a1, a3), jd = rot<INV>(csub(a1, a3));
                y[j0] = cadd(s02, s13);
                y[j0 + Ns] = cadd(d02, jd);
                y[j0 + 2 * Ns] = csub(s02, s13);
                y[j0 + 3 * Ns] = csub(d02, jd);
            } else if (r == 3) {
                const V a0 = x[b];
                const V a1 = cmul(x[b + nb], twid<INV>(tw, t1));
                const V a2 = cmul(x[b + 2 * nb], twid<INV>(tw, 2 * t1));
                const V t = cadd(a1, a2);
                const V m = mkv<V>(a0.x - T(0.5) * t.x, a0.y - T(0.5) * t.y);
                V sv = rot<INV>(csub(a1, a2));
                sv.x *= T(0.86602540378443864676);  // sin(2 pi / 3)
                sv.y *= T(0.86602540378443864676);
                y[j0] = cadd(a0, t);
                y[j0 + Ns] = cadd(m, sv);
                y[j0 + 2 * Ns] = csub(m, sv);
            } else if (r == 5) {
                constexpr T c1 = T(0.30901699437494742410), c2 = -T(0.80901699437494742410);
                constexpr T s1 = T(0.95105651629515357212), s2 = T(0.58778525229247312917);
                const V a0 = x[b];
                const V a1 = cmul(x[b + nb], twid<INV>(tw, t1));
                const V a2 = cmul(x[b + 2 * nb], twid<INV>(tw, 2 * t1));
                const V a3 = cmul(x[b + 3 * nb], twid<INV>(tw, 3 * t1));
                const V a4 = cmul(x[b + 4 * nb], twid<INV>(tw, 4 * t1));
                const V p1 = cadd(a1, a4), p2 = cadd(a2, a3), q1 = csub(a1, a4), q2 = csub(a2, a3);
                const V m1 = mkv<V>(a0.x + c1 * p1.x + c2 * p2.x, a0.y + c1 * p1.y + c2 * p2.y);
                const V m2 = mkv<V>(a0.x + c2 * p1.x + c1 * p2.x, a0.y + c2 * p1.y + c1 * p2.y);
                const V n1 = rot<INV>(mkv<V>(s1 * q1.x + s2 * q2.x, s1 * q1.y + s2 * q2.y));
                const V n2 = rot<INV>(mkv<V>(s2 * q1.x - s1 * q2.x, s2 * q1.y - s1 * q2.y));
                y[j0] = cadd(a0, cadd(p1, p2));
                y[j0 + Ns] = cadd(m1, n1);
                y[j0 + 2 * Ns] = cadd(m2, n2);
                y[j0 + 3 * Ns] = csub(m2, n2);
                y[j0 + 4 * Ns] = csub(m1, n1);
            } else {
                // generic radix: out[q] = sum_m (x_m w^(m k)) W_r^(q m), W_r^t = tw[(t mod r) N/r]
                const int rstep = N / r;
                for (int qq = 0; qq < r; ++qq) {
                    V acc = mkv<V>(T(0.0), T(0.0));
                    for (int m = 0; m < r; ++m) {
                        const V t = cmul(x[b + m * nb], twid<INV>(tw, (m * t1 + ((qq * m) % r) * rstep) % N));
                        acc.x += t.x;
                        acc.y += t.y;
                    }
                    y[j0 + qq * Ns] = acc;
                }
            }
        }
        __syncthreads();
        V *tmp = x;
        x = y;
        y = tmp;
        Ns *= r;
    }
    return x;
}

// ---- composite-radix in-place passes for {2,3,5}-smooth lengths (the product path).
// A pass has radix R = R1 * R2 (R1, R2 in {2,3,4,5}, R2 may be 1): 1440 = 10.12.12 is three
// passes per direction instead of six, each thread does ONE R-point butterfly per pass entirely in
// registers (Cooley-Tukey split into R2 butterflies of radix R1, the W_R twiddles, R1 butterflies
// of radix R2), and a workgroup has only as many threads as the widest pass has butterflies.
// The first forward pass reads its inputs straight from global memory (a functor), the last forward
// pass, the filter multiplier and the first inverse pass are one step in registers (merged_pass: the
// inverse runs the radices in reversed order), the last inverse pass stores to global memory: LDS
// holds one row of complex values and is touched once between two passes.
template <int R, bool INV, typename V>
__device__ __forceinline__ void butterfly(V (&v)[R]) {
    using T = Sc<V>;
    if (R == 1) {
    } else if (R == 2) {
        const V a0 = v[0], a1 = v[1];
        v[0] = cadd(a0, a1);
        v[1] = csub(a0, a1);
    } else if (R == 4) {
        const V s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
        const V s13 = cadd(v[1], v[3]), jd = rot<INV>(csub(v[1], v[3]));
        v[0] = cadd(s02, s13);
        v[1] = cadd(d02, jd);
        v[2] = csub(s02, s13);
        v[3] = csub(d02, jd);
    } else if (R == 3) {
        const V t = cadd(v[1], v[2]);
        const V m = mkv<V>(v[0].x - T(0.5) * t.x, v[0].y - T(0.5) * t.y);
        V sv = rot<INV>(csub(v[1], v[2]));
        sv.x *= T(0.86602540378443864676);
        sv.y *= T(0.86602540378443864676);
        v[0] = cadd(v[0], t);
        v[1] = cadd(m, sv);
        v[2] = csub(m, sv);
    } else {   // R == 5
        constexpr T c1 = T(0.30901699437494742410), c2 = -T(0.80901699437494742410);
        constexpr T s1 = T(0.95105651629515357212), s2 = T(0.58778525229247312917);
        const V a0 = v[0];
        const V p1 = cadd(v[1], v[4]), p2 = cadd(v[2], v[3]), q1 = csub(v[1], v[4]), q2 = csub(v[2], v[3]);
        const V m1 = mkv<V>(a0.x + c1 * p1.x + c2 * p2.x, a0.y + c1 * p1.y + c2 * p2.y);
        const V m2 = mkv<V>(a0.x + c2 * p1.x + c1 * p2.x, a0.y + c2 * p1.y + c1 * p2.y);
        const V n1 = rot<INV>(mkv<V>(s1 * q1.x + s2 * q2.x, s1 * q1.y + s2 * q2.y));
        const V n2 = rot<INV>(mkv<V>(s2 * q1.x - s1 * q2.x, s2 * q1.y - s1 * q2.y));
        v[0] = cadd(a0, cadd(p1, p2));
        v[1] = cadd(m1, n1);
        v[2] = cadd(m2, n2);
        v[3] = csub(m2, n2);
        v[4] = csub(m1, n1);
    }
}

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int inv_mod(int a, int n) {               // a^-1 mod n (n small), 0 if none
    for (int t = 1; t < n; ++t)
        if ((a * t) % n == 1) return t;
    return n == 1 ? 0 : 0;
}

// R-point DFT of v[m] in place, R = R1 R2.
//  * R1, R2 coprime (10 = 5.2, 12 = 4.3, 15, 20, 6): Good-Thomas.  With m = (R2 m1 + R1 m2) mod R
//    and q = (c1 q1 + c2 q2) mod R, c1 = R2 (R2^-1 mod R1), c2 = R1 (R1^-1 mod R2), the kernel
//    factors exactly, W_R^(m q) = W_R1^(m1 q1) W_R2^(m2 q2): no twiddles between the two stages.
//  * otherwise (8, 9, 16, 25): Cooley-Tukey, m = R2 m1 + m2, q = q1 + R1 q2,
//    W_R^(m q) = W_R1^(m1 q1) . W_R^(m2 q1) . W_R2^(m2 q2);   W_R^t = tw[t * wstep], wstep = N / R
template <int R1, int R2, bool INV, typename V>
__device__ __forceinline__ void dft_composite(V (&v)[R1 * R2], const V *tw, int wstep) {
    constexpr int R = R1 * R2;
    if constexpr (R2 == 1) {
        butterfly<R1, INV>(v);
    } else if constexpr (cgcd(R1, R2) == 1) {
        constexpr int c1 = R2 * inv_mod(R2 % R1, R1), c2 = R1 * inv_mod(R1 % R2, R2);
        V a[R];                                        // a[q1 R2 + m2]
#pragma unroll
        for (int m2 = 0; m2 < R2; ++m2) {
            V t[R1];
#pragma unroll
            for (int m1 = 0; m1 < R1; ++m1) t[m1] = v[(R2 * m1 + R1 * m2) % R];
            butterfly<R1, INV>(t);
#pragma unroll
            for (int q1 = 0; q1 < R1; ++q1) a[q1 * R2 + m2] = t[q1];
        }
#pragma unroll
        for (int q1 = 0; q1 < R1; ++q1) {
            V t[R2];
#pragma unroll
            for (int m2 = 0; m2 < R2; ++m2) t[m2] = a[q1 * R2 + m2];
            butterfly<R2, INV>(t);
#pragma unroll
            for (int q2 = 0; q2 < R2; ++q2) v[(c1 * q1 + c2 * q2) % R] = t[q2];
        }
    } else {
        V a[R];                                        // a[q1 R2 + m2]
#pragma unroll
        for (int m2 = 0; m2 < R2; ++m2) {
            V t[R1];
#pragma unroll
            for (int m1 = 0; m1 < R1; ++m1) t[m1] = v[R2 * m1 + m2];
            butterfly<R1, INV>(t);
#pragma unroll
            for (int q1 = 0; q1 < R1; ++q1) a[q1 * R2 + m2] = t[q1];
        }
#pragma unroll
        for (int q1 = 1; q1 < R1; ++q1)
#pragma unroll
            for (int m2 = 1; m2 < R2; ++m2) a[q1 * R2 + m2] = cmul(a[q1 * R2 + m2], twid<INV>(tw, (m2 * q1) * wstep));
#pragma unroll
        for (int q1 = 0; q1 < R1; ++q1) {
            V t[R2];
#pragma unroll
            for (int m2 = 0; m2 < R2; ++m2) t[m2] = a[q1 * R2 + m2];
            butterfly<R2, INV>(t);
#pragma unroll
            for (int q2 = 0; q2 < R2; ++q2) v[q1 + R1 * q2] = t[q2];
        }
    }
}

// one Stockham pass, one butterfly per thread.  src(i, m) -> V reads element i of the pass input
// (m = its position in this thread's butterfly), dst(i, V) writes element i of its output; `fence` =
// both are the same LDS buffer.  wpre, if given, is this thread's base twiddle of the pass
// (tw[k N / (Ns R)], forward sign), fetched once by the caller instead of once per call.
template <int R1, int R2, bool INV, typename V, typename Src, typename Dst>
__device__ __forceinline__ void composite_pass(const Src &src, const Dst &dst, bool fence, const V *tw, int N, int Ns,
                                               unsigned magic, const V *wpre = nullptr, int tid = -1) {
    constexpr int R = R1 * R2;
    const int nb = N / R;
    // tid: the thread index as the caller holds it.  A caller that runs the passes inside a loop hands
    // in a copy it has made opaque to the optimiser, so that the index arithmetic of all passes is not
    // hoisted out of that loop (it would be held in hundreds of registers).
    const int b = tid >= 0 ? tid : (int)threadIdx.x;
    const bool act = b < nb;
    V v[R];
    int j0 = 0;
    if (act) {
        const int blk = Ns == 1 ? b : (int)__umulhi((unsigned)b, magic);
        const int k = b - blk * Ns;
        j0 = blk * Ns * R + k;
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = src(b + m * nb, m);
        if (Ns > 1) {
            // pass twiddles w^(m t1): ONE gathered table entry per butterfly (the lanes' indices are
            // strided, so a gather costs a cache line per lane), the powers by squaring / one product
            const int t1 = k * (nb / Ns);          // k * N / (Ns R)
            V wp[R];
            if (wpre) {
                wp[1] = *wpre;
                if (INV) wp[1].y = -wp[1].y;
            } else {
                wp[1] = twid<INV>(tw, t1);
            }
#pragma unroll
            for (int m = 2; m < R; ++m) wp[m] = (m % 2 == 0) ? cmul(wp[m / 2], wp[m / 2]) : cmul(wp[m - 1], wp[1]);
#pragma unroll
            for (int m = 1; m < R; ++m) v[m] = cmul(v[m], wp[m]);
        }
        dft_composite<R1, R2, INV>(v, tw, nb);
    }
    if (fence) __syncthreads();
    if (act) {
#pragma unroll
        for (int q = 0; q < R; ++q) dst(j0 + q * Ns, v[q]);
    }
    __syncthreads();
}

// MAXR: widest radix compiled in.  MASK != 0: only the passes whose bit is set (bit = position in
// the list below) -- a kernel instantiated for one plan's own radices; the register allocation
// of a kernel is that of its hungriest compiled-in pass, used or not.
constexpr unsigned pass_bit(int a, int b) {
    return a == 2 ? 1u << 0 : (a == 3 && b == 1) ? 1u << 1 : (a == 4 && b == 1) ? 1u << 2 : (a == 5 && b == 1) ? 1u << 3
         : (a == 3 && b == 2) ? 1u << 4 : (a == 4 && b == 2) ? 1u << 5 : (a == 3 && b == 3) ? 1u << 6
         : (a == 5 && b == 2) ? 1u << 7 : (a == 4 && b == 3) ? 1u << 8 : (a == 5 && b == 3) ? 1u << 9
         : (a == 4 && b == 4) ? 1u << 10 : (a == 5 && b == 4) ? 1u << 11 : 1u << 12;
}

template <int MAXR, unsigned MASK, bool INV, typename V, typename Src, typename Dst>
__device__ __forceinline__ void pass_dispatch(int r1, int r2, const Src &src, const Dst &dst, bool fence, const V *tw,
                                              int N, int Ns, unsigned magic, const V *wpre = nullptr, int tid = -1) {
#define GCM_PASS(A, B)                                                                      \
    case (A) * 8 + (B):                                                                     \
        if constexpr ((A) * (B) <= MAXR && (MASK == 0 || (MASK & pass_bit(A, B)) != 0))     \
            composite_pass<A, B, INV>(src, dst, fence, tw, N, Ns, magic, wpre, tid);        \
        break;
    switch (r1 * 8 + r2) {
        GCM_PASS(2, 1) GCM_PASS(3, 1) GCM_PASS(4, 1) GCM_PASS(5, 1)
        GCM_PASS(3, 2) GCM_PASS(4, 2) GCM_PASS(3, 3) GCM_PASS(5, 2) GCM_PASS(4, 3)
        GCM_PASS(5, 3) GCM_PASS(4, 4) GCM_PASS(5, 4) GCM_PASS(5, 5)
        default: break;
    }
#undef GCM_PASS
}

// The LAST forward pass, the filter multiplier and the FIRST inverse pass as one register-resident step.
// The inverse transform takes the plan's radices in reversed order, so its first pass (Ns = 1, no pass
// twiddles) has the radix R of the last forward pass and reads x[b + m N/R], m = 0..R-1 -- exactly the
// elements X[b + q N/R] thread b holds when its last forward butterfly is done (Stockham leaves natural
// order).  The spectrum therefore never goes to LDS: one round trip and one barrier pair less per
// filtered row than forward passes, multiply-on-read, inverse passes.  mult(n) = S[n folded] / N.
// Thread b writes x[b R + q] (the Ns = 1 output positions of the inverse pass).
template <int R1, int R2, typename V, typename Src, typename Dst, typename Mult>
__device__ __forceinline__ void merged_pass(const Src &src, const Dst &dst, bool fence, bool twiddle, const Mult &mult,
                                            const V *tw, int N, const V *wpre = nullptr, int tid = -1) {
    constexpr int R = R1 * R2;
    const int nb = N / R;
    const int b = tid >= 0 ? tid : (int)threadIdx.x;
    const bool act = b < nb;
    V v[R];
    if (act) {
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = src(b + m * nb, m);
        if (twiddle) {                                // not the only pass: Ns = N / R = nb, k = b, t1 = b
            V wp[R];
            wp[1] = wpre ? *wpre : tw[b];
#pragma unroll
            for (int m = 2; m < R; ++m) wp[m] = (m % 2 == 0) ? cmul(wp[m / 2], wp[m / 2]) : cmul(wp[m - 1], wp[1]);
#pragma unroll
            for (int m = 1; m < R; ++m) v[m] = cmul(v[m], wp[m]);
        }
        dft_composite<R1, R2, false>(v, tw, nb);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const Sc<V> sc = mult(b + q * nb);
            v[q].x *= sc;
            v[q].y *= sc;
        }
        dft_composite<R1, R2, true>(v, tw, nb);
    }
    if (fence) __syncthreads();
    if (act) {
#pragma unroll
        for (int q = 0; q < R; ++q) dst(b * R + q, v[q]);
    }
    __syncthreads();
}

template <int MAXR, unsigned MASK, typename V, typename Src, typename Dst, typename Mult>
__device__ __forceinline__ void merged_dispatch(int r1, int r2, const Src &src, const Dst &dst, bool fence, bool twiddle,
                                                const Mult &mult, const V *tw, int N, const V *wpre = nullptr, int tid = -1) {
#define GCM_PASS(A, B)                                                                      \
    case (A) * 8 + (B):                                                                     \
        if constexpr ((A) * (B) <= MAXR && (MASK == 0 || (MASK & pass_bit(A, B)) != 0))     \
            merged_pass<A, B>(src, dst, fence, twiddle, mult, tw, N, wpre, tid);            \
        break;
    switch (r1 * 8 + r2) {
        GCM_PASS(2, 1) GCM_PASS(3, 1) GCM_PASS(4, 1) GCM_PASS(5, 1)
        GCM_PASS(3, 2) GCM_PASS(4, 2) GCM_PASS(3, 3) GCM_PASS(5, 2) GCM_PASS(4, 3)
        GCM_PASS(5, 3) GCM_PASS(4, 4) GCM_PASS(5, 4) GCM_PASS(5, 5)
        default: break;
    }
#undef GCM_PASS
}

// Filter the two real rows that `load(i)` delivers as re/im: forward FFT, multiply by S[n]/N
// (n folded, low_pass.py:61-72; numpy's irfft scales by 1/N), inverse FFT, `store(i, V)`.
// Forward passes 0 .. np-2, the merged pass (last forward + multiplier + first inverse), inverse passes
// 1 .. np-1 with the radices reversed: 2 np - 1 passes, 2 np - 2 LDS round trips.
template <int MAXR, unsigned MASK, typename T, typename Load, typename Store>
__device__ __forceinline__ void filter_rows_composite(typename Vec2<T>::type *x, const Load &load, const Store &store,
                                                      const typename Vec2<T>::type *tw, const SuperPlan &P, int N,
                                                      const T *S, bool load_reads_x = false) {
    using V = typename Vec2<T>::type;
    const T inv_n = T(1.0) / (T)N;
    const auto lds_src = [x](int i, int) { return x[i]; };
    const auto lds_dst = [x](int i, V v) { x[i] = v; };
    const auto mult = [S, N, inv_n](int n) { return S[n <= N / 2 ? n : N - n] * inv_n; };
    const int np = P.npass;
    if (np == 1) {
        merged_dispatch<MAXR, MASK, V>(P.r1[0], P.r2[0], load, store, false, false, mult, tw, N);
        return;
    }
    // the first and the last pass of either direction are written out rather than selected inside
    // one loop: their global addresses would otherwise be hoisted out of it and pinned in registers
    int Ns = P.r1[0] * P.r2[0];
    pass_dispatch<MAXR, MASK, false>(P.r1[0], P.r2[0], load, lds_dst, load_reads_x, tw, N, 1, P.magic[0]);
    for (int pass = 1; pass < np - 1; ++pass) {
        pass_dispatch<MAXR, MASK, false>(P.r1[pass], P.r2[pass], lds_src, lds_dst, true, tw, N, Ns, P.magic[pass]);
        Ns *= P.r1[pass] * P.r2[pass];
    }
    merged_dispatch<MAXR, MASK, V>(P.r1[np - 1], P.r2[np - 1], lds_src, lds_dst, true, true, mult, tw, N);
    Ns = P.r1[np - 1] * P.r2[np - 1];
    for (int q = 1; q < np - 1; ++q) {
        const int pass = np - 1 - q;
        pass_dispatch<MAXR, MASK, true>(P.r1[pass], P.r2[pass], lds_src, lds_dst, true, tw, N, Ns, P.imagic[q]);
        Ns *= P.r1[pass] * P.r2[pass];
    }
    pass_dispatch<MAXR, MASK, true>(P.r1[0], P.r2[0], lds_src, store, false, tw, N, Ns, P.imagic[np - 1]);
}

// The same filter for a workgroup that loops over several row pairs of ONE latitude (the filter
// multiplier and the pass twiddles depend on the thread and the latitude only): what the passes
// fetch from tables is fetched once, before the loop, into FilterConsts, and the first forward pass
// takes its inputs from registers (`first(i, m)`: the m-th input of this thread's butterfly, element
// i = threadIdx.x + m N / R0), so that a caller can request the next pair's inputs while this pair is
// transformed.  The multiplier row (already divided by N) is read from LDS.  Plans of two to four passes.
// Base twiddles (forward sign): w[0] = tw[b], shared by the merged pass (last forward, Ns = N / R) and the
// last inverse pass (Ns = N / R0); w[p], p = 1 .. np-2: forward pass p; w[np-2+q], q = 1 .. np-2: inverse
// pass q.  NW = 3 serves plans of up to three passes, 5 those of four.
template <typename T, int NW = 5>
struct FilterConsts {
    typename Vec2<T>::type w[NW];
    const T *s;                    // LDS: S[n] / N, n = 0 .. N/2 (the caller fills it)
};
template <typename T, int NW>
__device__ __forceinline__ void filter_consts(FilterConsts<T, NW> &c, const typename Vec2<T>::type *tw, const SuperPlan &P, int N) {
    using V = typename Vec2<T>::type;
    const int b = threadIdx.x;
    const int np = P.npass;
#pragma unroll
    for (int n = 0; n < NW; ++n) c.w[n] = mkv<V>(T(1.0), T(0.0));
    c.w[0] = tw[min(b, N - 1)];
    int Ns = P.r1[0] * P.r2[0];
#pragma unroll
    for (int pass = 1; pass < (NW + 1) / 2; ++pass) {            // forward passes 1 .. np-2
        if (pass < np - 1) {
            const int R = P.r1[pass] * P.r2[pass], nb = N / R;
            const int bb = min(b, nb - 1);
            const int blk = (int)__umulhi((unsigned)bb, P.magic[pass]);
            const int k = bb - blk * Ns;
            c.w[pass] = tw[k * (nb / Ns)];
            Ns *= R;
        }
    }
    Ns = P.r1[np - 1] * P.r2[np - 1];
#pragma unroll
    for (int q = 1; q < (NW + 1) / 2; ++q) {                     // inverse passes 1 .. np-2
        if (q < np - 1) {
            const int pass = np - 1 - q;
            const int R = P.r1[pass] * P.r2[pass], nb = N / R;
            const int bb = min(b, nb - 1);
            const int blk = (int)__umulhi((unsigned)bb, P.imagic[q]);
            const int k = bb - blk * Ns;
            c.w[(NW - 1) / 2 + q] = tw[k * (nb / Ns)];
            Ns *= R;
        }
    }
}
// after_first() runs when the first forward pass has consumed its inputs: the caller re-uses their
// registers for the next request there.
template <int MAXR, unsigned MASK, typename T, int NW, typename First, typename After, typename Store>
__device__ __forceinline__ void filter_rows_hoisted(typename Vec2<T>::type *x, const First &first, const After &after_first,
                                                    const Store &store, const typename Vec2<T>::type *tw, const SuperPlan &P,
                                                    int N, const FilterConsts<T, NW> &c, int tid) {
    using V = typename Vec2<T>::type;
    constexpr int kInv = (NW - 1) / 2;                            // w[kInv + q]: inverse pass q
    const auto reg_src = [&first](int i, int m) { return first(i, m); };
    const auto lds_src = [x](int i, int) { return x[i]; };
    const auto lds_dst = [x](int i, V v) { x[i] = v; };
    const T *sl = c.s;
    const auto mult = [sl, N](int n) { return sl[n <= N / 2 ? n : N - n]; };
    const int np = P.npass;
    // (np >= 2: single-pass plans take the one-pair kernels -- the merged pass from registers to global
    // memory, compiled in here, cost the looping kernel 40 registers whether it ran or not)
    // forward: pass 0 from registers, then the LDS passes up to the merged one
    pass_dispatch<MAXR, MASK, false>(P.r1[0], P.r2[0], reg_src, lds_dst, false, tw, N, 1, P.magic[0], (const V *)nullptr, tid);
    after_first();
    int Ns = P.r1[0] * P.r2[0];
    if (np > 2) { pass_dispatch<MAXR, MASK, false>(P.r1[1], P.r2[1], lds_src, lds_dst, true, tw, N, Ns, P.magic[1], &c.w[1], tid); Ns *= P.r1[1] * P.r2[1]; }
    if constexpr (NW > 3)
        if (np > 3) { pass_dispatch<MAXR, MASK, false>(P.r1[2], P.r2[2], lds_src, lds_dst, true, tw, N, Ns, P.magic[2], &c.w[2], tid); }
    merged_dispatch<MAXR, MASK, V>(P.r1[np - 1], P.r2[np - 1], lds_src, lds_dst, true, true, mult, tw, N, &c.w[0], tid);
    // inverse passes 1 .. np-1 (radices reversed); the last one stores
    Ns = P.r1[np - 1] * P.r2[np - 1];
    if (np > 2) { pass_dispatch<MAXR, MASK, true>(P.r1[np - 2], P.r2[np - 2], lds_src, lds_dst, true, tw, N, Ns, P.imagic[1], &c.w[kInv + 1], tid); Ns *= P.r1[np - 2] * P.r2[np - 2]; }
    if constexpr (NW > 3)
        if (np > 3) { pass_dispatch<MAXR, MASK, true>(P.r1[1], P.r2[1], lds_src, lds_dst, true, tw, N, Ns, P.imagic[2], &c.w[kInv + 2], tid); Ns *= P.r1[1] * P.r2[1]; }
    pass_dispatch<MAXR, MASK, true>(P.r1[0], P.r2[0], lds_src, store, false, tw, N, Ns, P.imagic[np - 1], &c.w[0], tid);
}

// generic path: filter two real rows held as re/im of x[0..N): FFT, multiply by S[n] (n folded),
// inverse FFT.  Returns the buffer (x or y) holding the result, already scaled by 1/N.
template <typename T>
__device__ typename Vec2<T>::type *filter_rows(typename Vec2<T>::type *x, typename Vec2<T>::type *y,
                                               const typename Vec2<T>::type *tw, const FftPlan &plan, const T *S) {
    using T2 = typename Vec2<T>::type;
    const int N = plan.n;
    T2 *z = fft_lds<false>(x, y, tw, plan);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const T s = S[n <= N / 2 ? n : N - n];
        z[n].x *= s;
        z[n].y *= s;
    }
    __syncthreads();
    T2 *o = (z == x) ? y : x;
    T2 *res = fft_lds<true>(z, o, tw, plan);
    const T inv_n = T(1.0) / (T)N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        res[n].x *= inv_n;
        res[n].y *= inv_n;
    }
    __syncthreads();
    return res;
}

// ---------------------------------------------------------------- host side: plans
inline bool make_plan(int n, FftPlan *P) {
    P->n = n;
    P->nrad = 0;
    int m = n;
    auto push = [&](int r) { if (P->nrad < kMaxRadices) P->rad[P->nrad++] = r; };
    while (m % 4 == 0) { push(4); m /= 4; }
    while (m % 2 == 0) { push(2); m /= 2; }
    for (int r = 3; r <= m && m > 1; r += 2)
        while (m % r == 0) { push(r); m /= r; }
    long Ns = 1;
    for (int i = 0; i < P->nrad; ++i) {
        P->magic[i] = (unsigned)(((1ULL << 32) + (unsigned long long)Ns - 1) / (unsigned long long)Ns);
        Ns *= P->rad[i];
    }
    return m == 1 && P->nrad <= kMaxRadices;
}

// composite-radix plan: n = product of base radices {5,4,3,2} (pairs of 2s become 4s), sorted
// descending and paired largest-with-smallest into passes of radix r1 * r2 (1440 -> 5.2, 4.3, 4.3;
// 2880 -> 5.3, 4.3, 4.4).  ok = 0 when n has a prime factor > 5 or a pass is wider than 512
// butterflies (the workgroup has one thread per butterfly).
inline void make_super_plan(int n, SuperPlan *P) {
    *P = SuperPlan{};
    std::vector<int> base;
    int m = n;
    while (m % 5 == 0) { base.push_back(5); m /= 5; }
    int twos = 0;
    while (m % 2 == 0) { ++twos; m /= 2; }
    for (int i = 0; i < twos / 2; ++i) base.push_back(4);
    while (m % 3 == 0) { base.push_back(3); m /= 3; }
    if (twos % 2) base.push_back(2);
    if (m != 1 || base.empty()) return;
    std::sort(base.begin(), base.end(), [](int x, int y) { return x > y; });
    int lo = 0, hi = (int)base.size() - 1;
    long Ns = 1;
    int widest = 1;
    while (lo <= hi) {
        if (P->npass == kMaxSuper) return;
        const int r1 = base[lo], r2 = lo < hi ? base[hi] : 1;
        // 2 x 2 never occurs (pairs of 2s are 4s); r1 >= r2 by the sort
        P->r1[P->npass] = r1;
        P->r2[P->npass] = r2;
        P->magic[P->npass] = (unsigned)(((1ULL << 32) + (unsigned long long)Ns - 1) / (unsigned long long)Ns);
        Ns *= (long)r1 * r2;
        if (r1 * r2 > P->maxr) P->maxr = r1 * r2;
        P->mask |= pass_bit(r1, r2);
        if (n / (r1 * r2) > widest) widest = n / (r1 * r2);
        ++P->npass;
        ++lo;
        --hi;
    }
    if (widest > 512) return;
    Ns = 1;
    for (int q = 0; q < P->npass; ++q) {                        // the inverse runs the radices in reversed order
        P->imagic[q] = (unsigned)(((1ULL << 32) + (unsigned long long)Ns - 1) / (unsigned long long)Ns);
        Ns *= (long)P->r1[P->npass - 1 - q] * P->r2[P->npass - 1 - q];
    }
    P->threads = (widest + 63) / 64 * 64;
    P->ok = 1;
}

}  // namespace gcm
