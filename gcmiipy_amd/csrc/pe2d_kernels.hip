// GCM_PE2D: 2-D single-layer primitive equations in momentum form, scalar dx, doubly
// periodic (reference no_limits_2d.py:21-131).  One launch per Euler stage, one thread per
// cell; neighbours are read through L1/L2.  q passes through unchanged (no_limits_2d.py:126).
#include "gcm_math.h"
#include "sw2d_kernels.h"

namespace gcm {

struct Pe2dView {
    const double *p, *u, *v, *t;
    int W, H;
    __device__ __forceinline__ long at(int j, int i) const {
        j %= H; if (j < 0) j += H;
        i %= W; if (i < 0) i += W;
        return (long)j * W + i;
    }
    __device__ __forceinline__ double P(int j, int i) const { return p[at(j, i)]; }
    __device__ __forceinline__ double U(int j, int i) const { return u[at(j, i)]; }
    __device__ __forceinline__ double V(int j, int i) const { return v[at(j, i)]; }
    __device__ __forceinline__ double T(int j, int i) const { return t[at(j, i)]; }
    // calc_pu / calc_pv, no_limits_2d.py:21-28
    __device__ __forceinline__ double PU(int j, int i) const { return U(j, i) * ((P(j, i) + P(j, i + 1)) * 0.5); }
    __device__ __forceinline__ double PV(int j, int i) const { return V(j, i) * ((P(j, i) + P(j + 1, i)) * 0.5); }
    // advec_p, no_limits_2d.py:41-44
    __device__ __forceinline__ double advec_p(int j, int i, double inv_dx) const {
        return (PU(j, i) - PU(j, i - 1)) * inv_dx + (PV(j, i) - PV(j - 1, i)) * inv_dx;
    }
    // p_mid = iph(jph(p)), vph = iph(v), no_limits_2d.py:54-55
    __device__ __forceinline__ double pmid(int j, int i) const {
        return (((P(j, i) + P(j + 1, i)) * 0.5) + ((P(j, i + 1) + P(j + 1, i + 1)) * 0.5)) * 0.5;
    }
    __device__ __forceinline__ double vph(int j, int i) const { return (V(j, i) + V(j, i + 1)) * 0.5; }
    __device__ __forceinline__ double jphu(int j, int i) const { return (U(j, i) + U(j + 1, i)) * 0.5; }
    // puum = imh(u)**2 * p; puvm = jmh(u) * ijm(vph) * ijm(p_mid)   (:57-61)
    __device__ __forceinline__ double puum(int j, int i) const {
        const double m = (U(j, i) + U(j, i - 1)) * 0.5;
        return m * m * P(j, i);
    }
    __device__ __forceinline__ double puvm(int j, int i) const {
        return ((U(j, i) + U(j - 1, i)) * 0.5) * vph(j - 1, i) * pmid(j - 1, i);
    }
    // pvvm = jmh(v)**2 * p; pvum = imj(p_mid) * imh(v) * imj(jph(u))   (:65-69)
    __device__ __forceinline__ double pvvm(int j, int i) const {
        const double m = (V(j, i) + V(j - 1, i)) * 0.5;
        return m * m * P(j, i);
    }
    __device__ __forceinline__ double pvum(int j, int i) const {
        return pmid(j, i - 1) * ((V(j, i) + V(j, i - 1)) * 0.5) * jphu(j, i - 1);
    }
};

struct Pe2dArgs {
    const double *bp, *bu, *bv, *bt, *bq;
    const double *sp, *su, *sv, *st;
    double *op, *ou, *ov, *ot, *oq;
    const double *exner_tab;
    int W, H;
    double dt, inv_dx;
};

__global__ __launch_bounds__(256) void pe2d_stage_kernel(Pe2dArgs a) {
    __shared__ double tab[kExnerTabDoubles];
    tab[threadIdx.y * 64 + threadIdx.x] = a.exner_tab[threadIdx.y * 64 + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= a.W || j >= a.H) return;
    const Pe2dView B{a.bp, a.bu, a.bv, a.bt, a.W, a.H}, S{a.sp, a.su, a.sv, a.st, a.W, a.H};
    const double inv_dx = a.inv_dx, dt = a.dt;
    // p_n = p - advec_p(spu, spv) dt at (j,i), (j,i+1), (j+1,i)      (:110)
    const double pn_c = B.P(j, i) - S.advec_p(j, i, inv_dx) * dt;
    const double pn_e = B.P(j, i + 1) - S.advec_p(j, i + 1, inv_dx) * dt;
    const double pn_s = B.P(j + 1, i) - S.advec_p(j + 1, i, inv_dx) * dt;
    // advec_m, :47-73 (puvp = ipj(puvm) and pvup = ipj(pvum) exactly as the reference writes them)
    const double dut = (S.puum(j, i) - S.puum(j, i + 1)) * inv_dx + (S.puvm(j, i) - S.puvm(j, i + 1)) * inv_dx;
    const double dvt = (S.pvvm(j, i) - S.pvvm(j + 1, i)) * inv_dx + (S.pvum(j, i) - S.pvum(j, i + 1)) * inv_dx;
    // pgf, :76-89
    const double pc = S.P(j, i), pe = S.P(j, i + 1), ps = S.P(j + 1, i);
    const double tc = S.T(j, i);
    const double ppih = (pc + pe) * 0.5, ppjh = (pc + ps) * 0.5;
    const double ttu = ((tc + S.T(j, i + 1)) * 0.5) * exner(ppih, tab);
    const double ttv = ((tc + S.T(j + 1, i)) * 0.5) * exner(ppjh, tab);
    const double rhou = ppih * rcp(kRd * ttu), rhov = ppjh * rcp(kRd * ttv);
    const double pgu = ppih * rcp(rhou) * ((pe - pc) * inv_dx);
    const double pgv = ppjh * rcp(rhov) * ((ps - pc) * inv_dx);
    // momentum and temperature updates, :112-121
    const double pu_n = B.PU(j, i) - (dut + pgu) * dt;
    const double pv_n = B.PV(j, i) - (dvt + pgv) * dt;
    const long o = (long)j * a.W + i;
    a.op[o] = pn_c;
    a.ou[o] = pu_n * rcp((pn_c + pn_e) * 0.5);
    a.ov[o] = pv_n * rcp((pn_c + pn_s) * 0.5);
    // advec_t, :92-99
    const double tpu_c = S.PU(j, i) * ((tc + S.T(j, i + 1)) * 0.5);
    const double tpu_w = S.PU(j, i - 1) * ((S.T(j, i - 1) + tc) * 0.5);
    const double tpv_c = S.PV(j, i) * ((tc + S.T(j + 1, i)) * 0.5);
    const double tpv_n = S.PV(j - 1, i) * ((S.T(j - 1, i) + tc) * 0.5);
    const double adt = (tpu_c - tpu_w) * inv_dx + (tpv_c - tpv_n) * inv_dx;
    a.ot[o] = a.bt[o] - (adt * rcp(pn_c)) * dt;
    a.oq[o] = a.bq[o];
}

void launch_pe2d_stage(const double *const base[5], const double *const stage[5], double *const out[5],
                       const double *exner_tab, int W, int H, double dt, double dx, hipStream_t s) {
    Pe2dArgs a{};
    a.bp = base[0]; a.bu = base[1]; a.bv = base[2]; a.bt = base[3]; a.bq = base[4];
    a.sp = stage[0]; a.su = stage[1]; a.sv = stage[2]; a.st = stage[3];
    a.op = out[0]; a.ou = out[1]; a.ov = out[2]; a.ot = out[3]; a.oq = out[4];
    a.exner_tab = exner_tab;
    a.W = W; a.H = H; a.dt = dt; a.inv_dx = 1.0 / dx;
    hipLaunchKernelGGL(pe2d_stage_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, a);
}

}  // namespace gcm
