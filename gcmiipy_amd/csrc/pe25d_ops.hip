// The operators of dynamics.py one by one (dynamics.py:15-181), host float64 arrays in the
// reference's layout [k][j][i] (p: [j][i]) in and out, periodic in i and j as the reference's rolls
// (and in k where it rolls k).  The step kernels of pe25d_kernels.hip evaluate the same expressions
// fused; these are the reference's call surface for them and the per-operator parity anchors
// (golden g7).  One thread per cell, or per column for the two operators that scan the levels.
#include "dev_arena.h"
#include "../../include/gcmcore.h"
#include "gcm_math.h"
#include "sw2d_kernels.h"

#include <string>
#include <vector>

namespace gcm {

struct PeOpArgs {
    int kind, W, H, L;
    const double *x[5];
    double *o[4];
    const double *dx_j, *dx_h, *dsig, *sig, *sigb, *sigt, *heightmap, *etab;
    double dy, ptop;
};

struct Ix3 {
    int W, H, L;
    __device__ __forceinline__ static int w(int x, int n) { return x < 0 ? x + n : x >= n ? x - n : x; }
    __device__ __forceinline__ long c(int k, int j, int i) const { return ((long)w(k, L) * H + w(j, H)) * W + w(i, W); }
    __device__ __forceinline__ long s(int j, int i) const { return (long)w(j, H) * W + w(i, W); }
};

__global__ __launch_bounds__(256) void pe_op_cell_kernel(PeOpArgs a) {
    __shared__ double tab[kExnerTabDoubles];
    tab[threadIdx.x] = a.etab[threadIdx.x];
    __syncthreads();
    const Ix3 ix{a.W, a.H, a.L};
    const long n = (long)a.W * a.H * a.L, e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int i = (int)(e % a.W), j = (int)((e / a.W) % a.H), k = (int)(e / ((long)a.W * a.H));
    const double *x0 = a.x[0], *x1 = a.x[1], *x2 = a.x[2], *x3 = a.x[3], *x4 = a.x[4];
    const auto iph2 = [&](const double *p, int jj, int ii) { return (p[ix.s(jj, ii)] + p[ix.s(jj, ii + 1)]) / 2; };
    const auto jph2 = [&](const double *p, int jj, int ii) { return (p[ix.s(jj, ii)] + p[ix.s(jj + 1, ii)]) / 2; };
    switch (a.kind) {
        case GCM_PEOP_CALC_PU: a.o[0][e] = x1[e] * iph2(x0, j, i); break;                     // dynamics.py:15-17
        case GCM_PEOP_CALC_PV: a.o[0][e] = x1[e] * jph2(x0, j, i); break;                     // :20-22
        case GCM_PEOP_UN_PU: a.o[0][e] = x0[e] / iph2(x1, j, i); break;                       // :25-27
        case GCM_PEOP_UN_PV: a.o[0][e] = x0[e] / jph2(x1, j, i); break;                       // :30-32
        case GCM_PEOP_ADVEC_SIG: {                                                            // :49-52 (sd, q)
            const auto flux = [&](int kk) { return ((x1[ix.c(kk, j, i)] + x1[ix.c(kk - 1, j, i)]) / 2) * x0[ix.c(kk, j, i)]; };
            a.o[0][e] = -((flux(k) - flux(k + 1)) / a.dsig[k]);
            break;
        }
        case GCM_PEOP_ADVEC_M_PU: {                                                           // :55-108 (p, u, v, pu, pv)
            const double *u = x1, *v = x2, *pu = x3, *pv = x4;
            const auto U = [&](const double *f, int jj, int ii) { return f[ix.c(k, jj, ii)]; };
            const auto puum = [&](int jj, int ii) { return ((U(u, jj, ii) + U(u, jj, ii - 1)) / 2) * ((U(pu, jj, ii) + U(pu, jj, ii - 1)) / 2); };
            const auto puvp = [&](int jj, int ii) { return ((U(pv, jj, ii) + U(pv, jj, ii + 1)) / 2) * ((U(u, jj, ii) + U(u, jj + 1, ii)) / 2); };
            const auto pvvm = [&](int jj, int ii) { return ((U(v, jj, ii) + U(v, jj - 1, ii)) / 2) * ((U(pv, jj, ii) + U(pv, jj - 1, ii)) / 2); };
            const auto pvup = [&](int jj, int ii) { return ((U(v, jj, ii) + U(v, jj, ii + 1)) / 2) * ((U(pu, jj, ii) + U(pu, jj + 1, ii)) / 2); };
            a.o[0][e] = (puum(j, i) - puum(j, i + 1)) / a.dx_j[j] + (puvp(j - 1, i) - puvp(j, i)) / a.dy + 0.0;
            a.o[1][e] = (pvvm(j, i) - pvvm(j + 1, i)) / a.dy + (pvup(j, i - 1) - pvup(j, i)) / a.dx_h[j] + 0.0;
            break;
        }
        case GCM_PEOP_ADVEC_T: {                                                              // :174-181 (pu, pv, t)
            const auto U = [&](const double *f, int jj, int ii) { return f[ix.c(k, jj, ii)]; };
            const auto tpu = [&](int jj, int ii) { return U(x0, jj, ii) * ((U(x2, jj, ii) + U(x2, jj, ii + 1)) / 2); };
            const auto tpv = [&](int jj, int ii) { return U(x1, jj, ii) * ((U(x2, jj, ii) + U(x2, jj + 1, ii)) / 2); };
            a.o[0][e] = (tpu(j, i) - tpu(j, i - 1)) / a.dx_j[j] + (tpv(j, i) - tpv(j - 1, i)) / a.dy;
            break;
        }
        case GCM_PEOP_PGF: {                                                                  // :147-171 (p, t, phi)
            const double *p = x0, *t = x1, *phi = x2;
            const double sg = a.sig[k];
            const auto rho = [&](int jj, int ii) {
                const double tp = p[ix.s(jj, ii)] * sg + a.ptop;
                const double tt = t[ix.c(k, jj, ii)] * exner(tp, tab);                       // to_true_temp, temperature.py:7-12
                return tp / (kRd * tt);
            };
            const double pc = p[ix.s(j, i)], pe = p[ix.s(j, i + 1)], ps = p[ix.s(j + 1, i)];
            const double r_c = rho(j, i);
            const double gi = (pe - pc) / a.dx_j[j], gj = (ps - pc) / a.dy;
            a.o[0][e] = ((sg * pc + sg * pe) / 2) / ((r_c + rho(j, i + 1)) / 2) * gi;        // pgfu
            a.o[1][e] = ((sg * pc + sg * ps) / 2) / ((r_c + rho(j + 1, i)) / 2) * gj;        // pgfv
            a.o[2][e] = ((pc + pe) / 2) * ((phi[ix.c(k, j, i + 1)] - phi[e]) / a.dx_j[j]);    // phiu
            a.o[3][e] = ((pc + ps) / 2) * ((phi[ix.c(k, j + 1, i)] - phi[e]) / a.dy);         // phiv
            break;
        }
        default: break;
    }
}

__global__ __launch_bounds__(256) void pe_op_col_kernel(PeOpArgs a) {
    __shared__ double tab[kExnerTabDoubles];
    tab[threadIdx.x] = a.etab[threadIdx.x];
    __syncthreads();
    const Ix3 ix{a.W, a.H, a.L};
    const long n = (long)a.W * a.H, c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const int i = (int)(c % a.W), j = (int)(c / a.W), L = a.L;
    if (a.kind == GCM_PEOP_AFLUX) {                                                           // :35-46 (pu, pv) -> pit, sd
        const auto conv = [&](int k) {
            return ((a.x[0][ix.c(k, j, i)] - a.x[0][ix.c(k, j, i - 1)]) / a.dx_j[j] +
                    (a.x[1][ix.c(k, j, i)] - a.x[1][ix.c(k, j - 1, i)]) / a.dy) * a.dsig[k];
        };
        double pit = 0.0;
        for (int k = 0; k < L; ++k) pit += conv(k);                   // np.sum over the outer axis: ascending
        a.o[0][c] = pit;
        double run = 0.0;
        for (int k = L - 1; k >= 0; --k) {                            // cumsum of the reversed column
            run += conv(k);
            a.o[1][ix.c(k, j, i)] = k == 0 ? 0.0 : run - pit * a.sigb[k];
        }
    } else {                                                                                  // compute_geopotential :111-143 (p, t)
        const double p = a.x[0][c];
        const auto T = [&](int k) { return a.x[1][ix.c(k, j, i)]; };
        const auto pk = [&](int k) { return exner(a.sig[Ix3::w(k, L)] * p + a.ptop, tab); };   // ((sig p + ptop)/P0)**kappa
        const auto stp = [&](int k) { return kCp * ((T(k) + T(k + 1)) / 2) * (pk(k) - pk(k + 1)); };
        double acc = 0.0;
        for (int k = 0; k < L; ++k) {
            const double tp = p * a.sig[k] + a.ptop;
            const double rho = tp / (kRd * (T(k) * exner(tp, tab)));
            acc += (a.sig[k] * p / rho) * a.dsig[k] - a.sigt[k] * stp(k);
        }
        double run = acc + (a.heightmap ? a.heightmap[c] : 0.0) * kG;
        a.o[0][ix.c(0, j, i)] = run;
        for (int k = 1; k < L; ++k) {
            run += stp(k - 1);
            a.o[0][ix.c(k, j, i)] = run;
        }
    }
}

}  // namespace gcm

using namespace gcm;

namespace {
thread_local std::string g_peop_error;
using Bufs = gcm::DevScratch;      // operands from the calling thread's grow-only arena (dev_arena.h)
int peop_fail(int code, const char *msg) {
    g_peop_error = msg;
    return code;
}
}  // namespace

extern "C" {

const char *gcm_pe25d_op_last_error(void) { return g_peop_error.c_str(); }

int gcm_pe25d_op(int kind, int width, int height, int layers, const gcm_pe_geom *g, const double *const in[5],
                 double *const out[4]) {
    // per operator: number of inputs, which of them are 2-D (bit mask), number of outputs, which are 2-D
    static const struct { int nin, in2d, nout, out2d; } sig[] = {
        {2, 1, 1, 0},   // CALC_PU (p, u)
        {2, 1, 1, 0},   // CALC_PV (p, v)
        {2, 2, 1, 0},   // UN_PU (pu, p)
        {2, 2, 1, 0},   // UN_PV (pv, p)
        {2, 0, 2, 1},   // AFLUX (pu, pv) -> pit, sd
        {2, 0, 1, 0},   // ADVEC_SIG (sd, q)
        {5, 1, 2, 0},   // ADVEC_M_PU (p, u, v, pu, pv) -> dut, dvt
        {2, 1, 1, 0},   // GEOPOTENTIAL (p, t) -> phi
        {2, 1, 4, 0},   // PGF (p, t) -> pgfu, pgfv, phiu, phiv
        {3, 0, 1, 0},   // ADVEC_T (pu, pv, t)
    };
    if (kind < 0 || kind > GCM_PEOP_ADVEC_T || width < 1 || height < 1 || layers < 1 || !g || !in || !out)
        return peop_fail(GCM_ERR_ARG, "gcm_pe25d_op: bad argument");
    if (!g->dx_j || !g->dx_h || !g->dsig || !g->sig || !g->sigb || !g->sigt || !(g->dy != 0.0))
        return peop_fail(GCM_ERR_ARG, "gcm_pe25d_op: geometry tables missing");
    for (int n = 0; n < sig[kind].nin; ++n)
        if (!in[n]) return peop_fail(GCM_ERR_ARG, "gcm_pe25d_op: null input");
    for (int n = 0; n < sig[kind].nout; ++n)
        if (!out[n]) return peop_fail(GCM_ERR_ARG, "gcm_pe25d_op: null output");
    if (gcm_device_count() < 1) return peop_fail(GCM_ERR_NODEVICE, "gcm_pe25d_op: no HIP device; no CPU fallback");
    const size_t n2 = (size_t)width * height, n3 = n2 * layers;
    Bufs mem;
    PeOpArgs a{};
    a.kind = kind; a.W = width; a.H = height; a.L = layers;
    a.dy = g->dy; a.ptop = g->ptop;
    double tab[kExnerTabDoubles];
    build_exner_table(tab);
    a.etab = mem.get(kExnerTabDoubles, tab);
    a.dx_j = mem.get(height, g->dx_j); a.dx_h = mem.get(height, g->dx_h);
    a.dsig = mem.get(layers, g->dsig); a.sig = mem.get(layers, g->sig);
    a.sigb = mem.get(layers, g->sigb); a.sigt = mem.get(layers, g->sigt);
    a.heightmap = g->heightmap ? mem.get(n2, g->heightmap) : nullptr;
    bool ok = a.etab && a.dx_j && a.dx_h && a.dsig && a.sig && a.sigb && a.sigt && (!g->heightmap || a.heightmap);
    for (int n = 0; n < sig[kind].nin && ok; ++n) ok = (a.x[n] = mem.get((sig[kind].in2d >> n) & 1 ? n2 : n3, in[n])) != nullptr;
    for (int n = 0; n < sig[kind].nout && ok; ++n) ok = (a.o[n] = mem.get((sig[kind].out2d >> n) & 1 ? n2 : n3)) != nullptr;
    if (!ok) return peop_fail(GCM_ERR_HIP, "gcm_pe25d_op: device allocation or upload failed");
    const dim3 gcell((unsigned)((n3 + 255) / 256)), gcol((unsigned)((n2 + 255) / 256));
    if (kind == GCM_PEOP_AFLUX || kind == GCM_PEOP_GEOPOTENTIAL) {
        hipLaunchKernelGGL(pe_op_col_kernel, gcol, dim3(256), 0, nullptr, a);
    } else if (kind == GCM_PEOP_PGF) {
        PeOpArgs c = a;                                   // phi first (a column scan), then the gradients
        c.kind = GCM_PEOP_GEOPOTENTIAL;
        double *phi = mem.get(n3);
        if (!phi) return peop_fail(GCM_ERR_HIP, "gcm_pe25d_op: device allocation failed");
        c.o[0] = phi;
        hipLaunchKernelGGL(pe_op_col_kernel, gcol, dim3(256), 0, nullptr, c);
        a.x[2] = phi;
        hipLaunchKernelGGL(pe_op_cell_kernel, gcell, dim3(256), 0, nullptr, a);
    } else {
        hipLaunchKernelGGL(pe_op_cell_kernel, gcell, dim3(256), 0, nullptr, a);
    }
    for (int n = 0; n < sig[kind].nout; ++n)
        if (hipMemcpy(out[n], a.o[n], ((sig[kind].out2d >> n) & 1 ? n2 : n3) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            return peop_fail(GCM_ERR_HIP, "gcm_pe25d_op: kernel or copy-back failed");
    return GCM_OK;
}

}  // extern "C"
