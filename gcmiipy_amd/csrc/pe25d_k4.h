// K4 of GCM_PE25D (see pe25d_kernels.hip): the update kernel and its picker.  Included by pe25d_k4_f64.hip / pe25d_k4_f32.hip only.
#pragma once
#include "pe25d_dev.h"

namespace gcm {

// ---------------------------------------------------------------- K4: update
// The levels are marched from the top down, because sigma-dot is the top-down running sum of conv
// (dynamics.py:42: cumsum(conv[::-1])[::-1] - pit sigb, sd[0] = 0): it is rebuilt here from the mass
// fluxes the momentum advection loads anyway, instead of being read back from HBM.  kp()/km() wrap
// (coordinates_3d.py:55-60): level L is 0 and level -1 is L-1; both only ever meet sd[0] = 0.
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
// ODDTOP: the march starts on an odd level (L even, whole columns): the geopotential anchor is then requested with the
// even levels only -- an odd level takes its anchor from the tile of the level below and never reads its own
template <typename T, int R, bool SAME, bool ODDTOP = false>
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
    // Each XCD (workgroups b, b+8, ... share one) takes every eighth (group, segment) pair and walks its
    // share with the COLUMN TILE fastest: the workgroups resident on an XCD at one time are the 24 column
    // tiles of one or two row groups, marching in step, and the eight XCDs work on eight ADJACENT groups --
    // a level's rows leave HBM as whole 11.5 KB rows within a short time and the chip's traffic stays in one
    // compact region of each field, instead of 512-byte pieces from eight distant regions (round 3: -5.5 %
    // on this kernel against the walk of round 2 -- contiguous groups per XCD, group fastest -- which bought
    // L2 hits on the halo rows with exactly that scatter).
    const int nseg = a.nseg;
    const int na = a.j1 - a.j0, nb = a.jb1 - a.jb0;
    const int ga = (na + R - 1) / R, gb = (nb + R - 1) / R;
    const int l = blockIdx.x / 8;
    const int rsl = l / ncol;
    const int ct = l - rsl * ncol;
    const int rowseg = rsl * 8 + (blockIdx.x % 8);
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
        const auto load = [&](T (&d)[10], int k, bool anch = true) {
            const long kl = (long)k * W;
            d[0] = a.su[rn + kl + i]; d[1] = a.sv[rn + kl + i]; d[2] = a.st[rn + kl + i]; d[3] = a.sq[rn + kl + i];
            d[4] = a.su[rs + kl + i]; d[5] = a.sv[rs + kl + i]; d[6] = a.st[rs + kl + i]; d[7] = a.sq[rs + kl + i];
            d[8] = a.spu[rs + kl + i];
            if (anch || !ODDTOP) d[9] = a.phi[rs + (long)(k & ~1) * W + i];
        };
        const auto put = [&](const T (&d)[10], int buf) {
            T *t = t0 + buf * kBuf;
#pragma unroll
            for (int f = 0; f < 4; ++f) t[(TL::kMain + f * (R + 2)) * 64] = d[f];
#pragma unroll
            for (int f = 0; f < 5; ++f) t[(TL::kMain + f * (R + 2) + ss) * 64] = d[4 + f];
            t[(TL::kPhi + nact) * 64] = d[9];
        };
        load(h[0], k0, false);
        put(h[0], 0);
        if (k0 - 1 >= kmin) { load(h[1], k0 - 1, true); put(h[1], 1); }
        // requests are unconditional (level clamped to kmin): a conditional one would make the compiler
        // size its in-order vmcnt waits for the path without it, i.e. wait for the newest requests too
        load(h[0], max(k0 - 2, kmin), false);
        __syncthreads();
        for (int k = k0;;) {
            load(h[1], max(k - 3, kmin), true);
            if (k - 2 >= kmin) put(h[0], bf);
            if (k == k_lo) break;
            __syncthreads();
            { const int t = bc; bc = bm; bm = bf; bf = t; }
            --k;
            load(h[0], max(k - 3, kmin), false);
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
    const auto load = [&](T (&d)[11], int k, bool anch = true) {
        // (the lane offset is made opaque once per level, so that its zero extension stays next to the
        // requests and they take the scalar-base + 32-bit-offset form)
        unsigned ol = ob;
        asm volatile("" : "+v"(ol));
        const auto at = [ol, sbase](const T *base) { return *(const __attribute__((address_space(1))) T *)(sbase(base) + ol); };
        const long o = rc + (long)k * W;
        d[0] = at(a.su + o); d[1] = at(a.sv + o); d[2] = at(a.st + o); d[3] = at(a.sq + o);
        d[4] = at(a.spu + o); d[6] = at(a.pgfu + o);
        if (anch || !ODDTOP) d[5] = at(a.phi + (rc + (long)(k & ~1) * W));        // the anchor at or below k (odd k: unused, and a cache hit)
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
    load(q[0], k0, false);
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
    if (k0 - 1 >= kmin) { load(q[1], k0 - 1, true); put(q[1], 1); }
    load(q[0], max(k0 - 2, kmin), false);                        // unconditional, clamped: see the loader wave
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
        load(q[1], max(k - 3, kmin), true);
        level(k);
        if (k - 2 >= kmin) put(q[0], bf);
        if (k == k_lo) break;
        __syncthreads();
        { const int t = bc; bc = bm; bm = bf; bf = t; }
        --k;
        load(q[0], max(k - 3, kmin), false);
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

// the four instantiations of one (real type, rows per workgroup): a translation unit each (pe25d_k4_f{64,32}_r{3,7}.hip),
// so that they compile in parallel
template <typename T, int R>
FilterKernel<T> update_rows_kernel_rt(bool same, bool oddtop) {
    if (oddtop) return same ? pe_update_rows_kernel<T, R, true, true> : pe_update_rows_kernel<T, R, false, true>;
    return same ? pe_update_rows_kernel<T, R, true> : pe_update_rows_kernel<T, R, false>;
}

}  // namespace gcm
