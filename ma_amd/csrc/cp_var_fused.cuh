// Variance-TV (NNE/var_tv_registration_no_ref_ml.m): one field kernel per Chambolle-Pock iteration.
//
// k_var_lazy_tv(n) = primal step n followed by dual step n+1 of PDO/chambolle_pock.m:162-165, per pixel, with the state
// read and written once:   reads  u_{n-1} (2), y1_n (1), y2_n (4), grad T (2), B*b (1)     writes u_n (2), y1 (1), y2 (4)
// = 17 elements per pixel-image (+ 4/K for two per-pixel fields) instead of the 28 of the split passes (k_var_dual 17 +
// k_var_primal 11). Same organisation as k_mf_fused_tv (cp_mfnn_fused.cuh): a wavefront marches along one image column in
// aligned segments of 64 rows, (i+1, j) from lane+1 / the LDS hand-over of the previous step, (i, j+1) recomputed from y_n
// (read-only here), four adjacent columns per workgroup, no workgroup barrier in the loop.
// A = [B*diag(grad T); D]: B (the mean over the K images of a pixel) enters twice, and a first version (round 2,
// k_var_fused_tv) therefore walked the images of a pixel three times (means of y1; primal step into LDS tiles; dual step):
// 13.2-13.5 ms per iteration at 4096^2 x 16 against 14.6 ms of the split passes -- the second and third walk re-read what
// the first had streamed out of L2 long before. The lazy form below takes both means out of the loop: 8.3 ms.
// k_var_efused: reduction of the scalars, history rows, mean for the next pass, stop rule (no Gram, no eigensolve).
#pragma once
#include "cp_mfnn_fused.cuh"
#include "cp_mfnn_curv.cuh"
#include "cp_var.cuh"

enum { VS_SGX = DS_F2S, VS_SGY = DS_SGY };   // sum of grad T * (B*y1_new) per component (field kernels of the fused paths)
enum { VSC_SGX = 11 };                       // the same, x component, from k_var_dual<., 1> (there DS_F2S holds F2*(y2))

// ---- k_var_lazy_tv ------------------------------------------------------------------------------------------------------
// B (the mean over the images) sits on both sides of y1: (A'y)_u,k = grad T_k * (y1_k - mean_k y1) and
// (A xbar)_1,k = s_k - mean_k s, s_k = grad T_k . ubar_k. Both means can be taken out of the loop over the images:
//   * the SSD prox of y1 is affine (NNE/SSD.m:60-61): y1+_k = c*(y1_k + sigma*(s_k - mean s)) - e*gg_k with c = lt/(1+lt),
//     e = sigma*c, lt = mu1/sigma (c = e = sigma/(1+sigma) for the swapped call of var_tv_registration_no_ref.m:173). Store
//     y1_k = alpha_k - beta with a per-pixel beta: alpha+_k = c*(alpha_k + sigma*s_k) - e*gg_k needs no mean, and
//     beta+ = c*(beta + sigma*mean s) is one number per pixel, known after the loop;
//   * B*y1 does not see beta: (B*y1)_k = alpha_k - abar, abar = mean_k alpha_k, one number per pixel written by the
//     previous pass (read at the own pixel and at (i, j+1)).
// The monitors of y1+ follow from per-pixel sums (sum alpha+, sum alpha+^2, sum alpha+*gg, sum gg, ...). State: alpha in the
// y1 fields, abar and beta in two per-pixel fields, ping-ponged with the iteration parity like x and y; k_var_lazy_init
// starts them (abar = mean y1, beta = 0), k_var_lazy_finish turns alpha back into y1. 17 elements + 4/K per pixel-image,
// every image of a pixel loaded once. Addressing through buffer descriptors, range-dropped stores (cf. k_mf_fused_tv).
struct VarRsrc { cp_rsrc y1, y2a, y2b, g, u; };
struct VarLoads { double a, gx, gy, v0, v1, v2, v3, w0m, w1m, w2m, w3m, uxo, uyo, l0, l1, l2, l3; };
template <typename S, bool LINEAR>
__device__ __forceinline__ void var_load(const MfParams &P, int i, int jg, unsigned oc, unsigned fsb, const VarRsrc &R,
                                         VarLoads &q) {
    const int m = P.m;
    constexpr unsigned ES = (unsigned)sizeof(S);
    const unsigned oim = oc - ES, ojm = oc - ES * (unsigned)m;
    q.a = bld<S>(R.y1, oc);
    q.gx = bld<S>(R.g, oc); q.gy = bld<S>(R.g, oc, fsb);
    q.v0 = bld<S>(R.y2a, oc); q.v1 = bld<S>(R.y2a, oc, fsb); q.v2 = bld<S>(R.y2b, oc); q.v3 = bld<S>(R.y2b, oc, fsb);
    q.w0m = bld<S>(R.y2a, oim); q.w2m = bld<S>(R.y2b, oim);
    q.w1m = bld<S>(R.y2a, ojm, fsb); q.w3m = bld<S>(R.y2b, ojm, fsb);
    q.uxo = bld<S>(R.u, oc); q.uyo = bld<S>(R.u, oc, fsb);
    q.l0 = q.l1 = q.l2 = q.l3 = 0.0;
    if (LINEAR) {
        const unsigned oli = oc + ES * (unsigned)((m - 1) - i);
        const int dj = (jg >= P.n_global - 2) ? (P.n_global - 1) - jg : 0;
        const unsigned olj = oc + ES * (unsigned)(dj * m);
        q.l0 = bld<S>(R.y2a, oli); q.l2 = bld<S>(R.y2b, oli);
        q.l1 = bld<S>(R.y2a, olj, fsb); q.l3 = bld<S>(R.y2b, olj, fsb);
    }
}
// u_n = (u_{n-1} - tau*(A'y_n)_u) - mean at the position; abar = mean_k alpha_k there
template <typename S, bool LINEAR>
__device__ __forceinline__ void var_eval(const MfParams &P, int i, int jg, const VarLoads &q, double abar, double cnx, double cny,
                                         double &ux, double &uy, double &wx, double &wy) {
    const int m = P.m;
    const double y1c = q.a - abar;                                             // (B*y1)_k
    const double a0 = fwd_diff_adj(q.v0, (i >= 1) ? q.w0m : 0.0, q.l0, i, m, LINEAR, P.inv_h1);
    const double a1 = fwd_diff_adj(q.v1, (jg >= 1) ? q.w1m : 0.0, q.l1, jg, P.n_global, LINEAR, P.inv_h2);
    const double a2 = fwd_diff_adj(q.v2, (i >= 1) ? q.w2m : 0.0, q.l2, i, m, LINEAR, P.inv_h1);
    const double a3 = fwd_diff_adj(q.v3, (jg >= 1) ? q.w3m : 0.0, q.l3, jg, P.n_global, LINEAR, P.inv_h2);
    const double vux = q.gx * y1c + (a0 + a1);
    const double vuy = q.gy * y1c + (a2 + a3);
    ux = rnd<S>((q.uxo - P.tau * vux) - cnx);                                  // NNE/mean_zero_indicator.m:50 (the stored value)
    uy = rnd<S>((q.uyo - P.tau * vuy) - cny);                                  // :53
    wx = -vux;
    wy = -vuy;
}

// NWV = waves (adjacent columns) per workgroup; 8: with the LDS hand-over of the own-pixel results between neighbouring waves
// (hs_give / hs_take, cp_mfnn_fused.cuh) instead of the (i, j+1) evaluation from memory
template <typename S, bool LINEAR, int NWV = CP_WAVES>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void k_var_lazy_tv(MfParams P, const S *__restrict__ xo, S *__restrict__ xn,
                                                              const S *__restrict__ pin, S *__restrict__ pout,
                                                              const S *__restrict__ g, const S *__restrict__ bbar,
                                                              const double *__restrict__ lz_in, double *__restrict__ lz_out,
                                                              const CpCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double carry_all[NWV * CP_VAR_KMAX * 4];   // [wave][image][ux, uy, uxbar, uybar] of lane 0
    __shared__ double stage_all[NWV * CP_NSCAL];
    constexpr bool HS = (CP_FUSED_HS != 0) && !LINEAR && NWV == 8;
    __shared__ double xch_all[HS ? (NWV - 1) * 256 : 1];
    __shared__ int hs_flags[3 * NWV];
    const int stopped = ctrl->stop;
    const double cnx = ctrl->cn[0], cny = ctrl->cn[1];
    if (stopped) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 3 * NWV) hs_flags[tid] = 0;
    __syncthreads();
    lds_int *fl = (lds_int *)hs_flags;
    int hs_fb = 0;
    lds_double *xl = (lds_double *)xch_all + (HS && wave > 0 ? (wave - 1) * 256 : 0);
    const lds_double *xr = (lds_double *)xch_all + (HS && wave < NWV - 1 ? wave * 256 : 0);
    int pseq = 0, cseq = 0;
    double *carry = carry_all + wave * CP_VAR_KMAX * 4;
    const int K = P.K, m = P.m;
    const size_t fs = (size_t)P.fs;
    const bool linear = LINEAR;
    const double sigma = P.sigma, theta = P.theta;
    const double lt = P.mu1 / P.sigma;
    const double cc = P.ssd_swap ? sigma / (1.0 + sigma) : lt / (1.0 + lt);   // y1+ = cc*yt - ee*gg
    const double ee = P.ssd_swap ? sigma / (1.0 + sigma) : sigma * (lt / (1.0 + lt));
    constexpr unsigned ES = (unsigned)sizeof(S);
    const unsigned mb = ES * (unsigned)m, fsb = ES * (unsigned)P.fs;
    const cp_rsrc Rab = mk_rsrc(lz_in), Rbe = mk_rsrc(lz_in + fs), Rabo = mk_rsrc(lz_out), Rbeo = mk_rsrc(lz_out + fs);

    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = -1.0e300;
    maxs[DM_S1 - CP_NSUM] = 0.0;

    const Fused2Geom GM = fused2_geom(m, 64, fused2_cap((P.nl + NWV - 1) / NWV, NWV, P.parts8));
    const int ntasks = GM.parts * ((P.nl + NWV - 1) / NWV);
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int cg = task / GM.parts, pt = task - cg * GM.parts;
        const int j = cg * NWV + wave;
        if (j >= P.nl) continue;                                   // wave-uniform
        const int seg_lo = pt * GM.steps, seg_hi = min(seg_lo + GM.steps, GM.nseg) - 1;
        const int jg = P.j_off + j;
        const bool last_j = (jg == P.n_global - 1);
        const int oj = (last_j && LINEAR) ? -1 : 1;
        const bool hs_p = HS && wave > 0, hs_c = HS && wave < NWV - 1 && j + 1 < P.nl;   // wave-uniform roles (cf. k_mf_fused_tv)
        for (int sg = seg_hi; sg >= seg_lo; --sg) {
            const int ii = sg * 64 + lane;
            const bool valid = ii < m;
            const int i = min(ii, m - 1);
            const unsigned off = ES * (unsigned)((j + 1) * m + i);
            const unsigned off8 = 8u * (unsigned)((j + 1) * m + i);            // the per-pixel fields abar, beta are doubles
            const bool last_i = (i == m - 1);
            const int oi = (last_i && LINEAR) ? -1 : 1;
            const int nb_lane = lane + oi;
            const bool want = valid && (LINEAR || !last_i);
            const bool from_carry = want && nb_lane == 64 && sg < seg_hi;
            const bool recompute = want && (nb_lane < 0 || nb_lane > 63) && !from_carry;
            const bool any_rc = __any(recompute);
            const unsigned offj = oj > 0 ? off + mb : off - mb;
            const unsigned offi = oi > 0 ? off + ES : off - ES;
            const unsigned offj8 = oj > 0 ? off8 + 8u * (unsigned)m : off8 - 8u * (unsigned)m, offi8 = oi > 0 ? off8 + 8u : off8 - 8u;
            const unsigned off_st = valid ? off : 0xFFFFFFFFu, off_st8 = valid ? off8 : 0xFFFFFFFFu;
            const double vmask = valid ? 1.0 : 0.0;
            // per-pixel state of the lazy y1: abar at the own pixel, at (i, j +- 1) (and at (i +- 1, j) where that is recomputed), beta
            const double ab = bld<double>(Rab, off8), abj = bld<double>(Rab, offj8), beta = bld<double>(Rbe, off8);
            double abi = 0.0;
            if (any_rc) abi = bld<double>(Rab, offi8);
            double S_sb = 0.0, S_sc = 0.0, D1 = 0.0, D2 = 0.0, A1 = 0.0, A2 = 0.0, AG = 0.0, G1 = 0.0, GX = 0.0, GY = 0.0, GXA = 0.0,
                   GYA = 0.0;
            for (int k = 0; k < K; ++k) {
                const S *y2k = pin + ((size_t)K + 4 * (size_t)k) * fs;
                S *o2k = pout + ((size_t)K + 4 * (size_t)k) * fs;
                const VarRsrc R = {mk_rsrc(pin + (size_t)k * fs), mk_rsrc(y2k), mk_rsrc(y2k + 2 * fs), mk_rsrc(g + 2 * (size_t)k * fs),
                                   mk_rsrc(xo + 2 * (size_t)k * fs)};
                const cp_rsrc Run = mk_rsrc(xn + 2 * (size_t)k * fs), Ro1 = mk_rsrc(pout + (size_t)k * fs), Ro2a = mk_rsrc(o2k),
                              Ro2b = mk_rsrc(o2k + 2 * fs);
                VarLoads q0, qj;
                var_load<S, LINEAR>(P, i, jg, off, fsb, R, q0);
                if (!hs_c) var_load<S, LINEAR>(P, i, jg + oj, offj, fsb, R, qj);
                const double gg = -bld_stream<S>(mk_rsrc(bbar + (size_t)k * fs), off);   // F(y, (-b), ...)   NNE/var_tv...:172
                double ux, uy, wx, wy, uxj, uyj, wd0, wd1;
                var_eval<S, LINEAR>(P, i, jg, q0, ab, cnx, cny, ux, uy, wx, wy);
                double uxoj = 0.0, uyoj = 0.0;
                if (HS) {
                    if (hs_p) hs_fb += hs_give(xl, fl + NWV + (wave - 1), fl + wave, fl + 2 * NWV + wave, ++pseq, lane, ux, uy, q0.uxo, q0.uyo);
                    if (hs_c) {
                        if (!hs_take(xr, fl + (wave + 1), fl + NWV + wave, fl + 2 * NWV + (wave + 1), ++cseq, lane, uxj, uyj, uxoj, uyoj)) {
                            ++hs_fb;
                            var_load<S, LINEAR>(P, i, jg + oj, offj, fsb, R, qj);         // never in practice: see hs_take
                            var_eval<S, LINEAR>(P, i, jg + oj, qj, abj, cnx, cny, uxj, uyj, wd0, wd1);
                            uxoj = qj.uxo; uyoj = qj.uyo;
                        }
                    } else {
                        var_eval<S, LINEAR>(P, i, jg + oj, qj, abj, cnx, cny, uxj, uyj, wd0, wd1);
                        uxoj = qj.uxo; uyoj = qj.uyo;
                    }
                } else {
                    var_eval<S, LINEAR>(P, i, jg + oj, qj, abj, cnx, cny, uxj, uyj, wd0, wd1);
                    uxoj = qj.uxo; uyoj = qj.uyo;
                }
                const double gx = q0.gx, gy = q0.gy;
                const double uxb = ux + theta * (ux - q0.uxo), uyb = uy + theta * (uy - q0.uyo);
                const double uxbj = uxj + theta * (uxj - uxoj), uybj = uyj + theta * (uyj - uyoj);
                // (i +- 1, j): lane +- 1's own result; lane 63: the hand-over of the previous step; recomputed where neither exists
                double uxi = __shfl(ux, nb_lane), uyi = __shfl(uy, nb_lane);
                double uxbi = __shfl(uxb, nb_lane), uybi = __shfl(uyb, nb_lane);
                if (from_carry) { uxi = carry[4 * k]; uyi = carry[4 * k + 1]; uxbi = carry[4 * k + 2]; uybi = carry[4 * k + 3]; }
                if (any_rc) {
                    if (recompute) {
                        VarLoads qi;
                        var_load<S, LINEAR>(P, i + oi, jg, offi, fsb, R, qi);
                        var_eval<S, LINEAR>(P, i + oi, jg, qi, abi, cnx, cny, uxi, uyi, wd0, wd1);
                        uxbi = uxi + theta * (uxi - qi.uxo);
                        uybi = uyi + theta * (uyi - qi.uyo);
                    }
                }
                if (lane == 0) { carry[4 * k] = ux; carry[4 * k + 1] = uy; carry[4 * k + 2] = uxb; carry[4 * k + 3] = uyb; }
                bst<S>(Run, off_st, 0u, ux);
                bst<S>(Run, off_st, fsb, uy);
                maxs[PM_WXMAX - CP_NSUM] = fmax(maxs[PM_WXMAX - CP_NSUM], valid ? wx : -1.0e300);
                maxs[PM_WXMIN - CP_NSUM] = fmax(maxs[PM_WXMIN - CP_NSUM], valid ? -wx : -1.0e300);
                maxs[PM_WYMAX - CP_NSUM] = fmax(maxs[PM_WYMAX - CP_NSUM], valid ? wy : -1.0e300);
                maxs[PM_WYMIN - CP_NSUM] = fmax(maxs[PM_WYMIN - CP_NSUM], valid ? -wy : -1.0e300);
                sums[PS_WX] += vmask * wx;
                sums[DS_UX] += vmask * ux;
                sums[DS_UY] += vmask * uy;
                // ---- y1 (lazy): alpha+ = cc*(alpha + sigma*s_bar) - ee*gg; the mean of s_bar goes into beta+ after the loop ------
                const double s_cur = gx * ux + gy * uy, s_bar = gx * uxb + gy * uyb;
                const double an = rnd<S>(cc * (q0.a + sigma * s_bar) - ee * gg);
                bst<S>(Ro1, off_st, 0u, an);
                const double dk = s_cur - gg;
                S_sb += s_bar; S_sc += s_cur; D1 += dk; D2 += dk * dk;
                A1 += an; A2 += an * an; AG += an * gg; G1 += gg; GX += gx; GY += gy; GXA += gx * an; GYA += gy * an;
                // ---- y2: norm21* ---------------------------------------------------------------------------------------------
                const double d0 = fwd_diff(ux, uxi, uxi, last_i, linear, P.inv_h1);
                const double d1 = fwd_diff(ux, uxj, uxj, last_j, linear, P.inv_h2);
                const double d2 = fwd_diff(uy, uyi, uyi, last_i, linear, P.inv_h1);
                const double d3 = fwd_diff(uy, uyj, uyj, last_j, linear, P.inv_h2);
                const double e0 = fwd_diff(uxb, uxbi, uxbi, last_i, linear, P.inv_h1);
                const double e1 = fwd_diff(uxb, uxbj, uxbj, last_j, linear, P.inv_h2);
                const double e2 = fwd_diff(uyb, uybi, uybi, last_i, linear, P.inv_h1);
                const double e3 = fwd_diff(uyb, uybj, uybj, last_j, linear, P.inv_h2);
                sums[DS_F2] += vmask * sqrt(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);      // NNE/norm21.m:56
                const double t0 = q0.v0 + sigma * e0, t1 = q0.v1 + sigma * e1, t2 = q0.v2 + sigma * e2, t3 = q0.v3 + sigma * e3;
                const double nn = fmax(sqrt(t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3), P.mu2);   // :70
                const double n0 = rnd<S>((P.mu2 * t0) / nn), n1 = rnd<S>((P.mu2 * t1) / nn), n2 = rnd<S>((P.mu2 * t2) / nn), n3 = rnd<S>((P.mu2 * t3) / nn);
                bst<S>(Ro2a, off_st, 0u, n0); bst<S>(Ro2a, off_st, fsb, n1);
                bst<S>(Ro2b, off_st, 0u, n2); bst<S>(Ro2b, off_st, fsb, n3);
                maxs[DM_S1 - CP_NSUM] = fmax(maxs[DM_S1 - CP_NSUM], valid ? sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3) : 0.0);   // :81
            }
            // ---- per pixel: the means that B takes, the lazy offset, the monitors of y1+ = alpha+ - beta+ -------------------------
            const double Kd = (double)K;
            const double mrb = S_sb * P.invK, mrc = S_sc * P.invK;
            const double bn = cc * (beta + sigma * mrb);
            const double abn = A1 * P.invK;
            bst<double>(Rabo, off_st8, 0u, abn);
            bst<double>(Rbeo, off_st8, 0u, bn);
            sums[DS_F1] += vmask * ((D2 - 2.0 * mrc * D1) + Kd * mrc * mrc);             // sum_k (s_k - mrc - gg_k)^2   NNE/SSD.m:48
            const double im = 1.0 / P.mu1;
            sums[VS_Y1SQ] += vmask * (((A2 - 2.0 * bn * A1) + Kd * bn * bn) * (im * im));   // sum_k (y1+_k/mu1)^2      NNE/SSD.m:66-67
            sums[VS_Y1G] += vmask * ((AG - bn * G1) * im);
            sums[VS_SGX] += vmask * (GXA - abn * GX);                                    // sum_k grad T_k * (B*y1+)_k
            sums[VS_SGY] += vmask * (GYA - abn * GY);
            wave_lds_sync();   // hand-over of this step is complete before the next step reads it
        }
    }
    if (HS) hs_report(ctrl, hs_fb, lane);
    {
        double *st = stage_all + wave * CP_NSCAL;
#pragma unroll
        for (int a = 0; a < CP_NSUM; ++a) {
            const double v = wave_sum(sums[a]);
            if (lane == 0) st[a] = v;
        }
#pragma unroll
        for (int a = 0; a < CP_NMAX; ++a) {
            const double v = wave_max(maxs[a]);
            if (lane == 0) st[CP_NSUM + a] = v;
        }
        __syncthreads();
        if (tid < CP_NSCAL) {
            const bool is_max = tid >= CP_NSUM;
            double v = stage_all[tid];
#pragma unroll
            for (int w = 1; w < NWV; ++w) {
                const double o = stage_all[w * CP_NSCAL + tid];
                v = is_max ? fmax(v, o) : v + o;
            }
            part[(size_t)blockIdx.x * CP_PSTRIDE + tid] = v;
        }
    }
}

// ---- variance-curvature (NNE/var_curvature_registration_no_ref_ml.m): two field passes per iteration -------------------------
// A = [B*diag(grad T); Lap], F2 = SSD(y2, 0). As for NN-curvature (cp_mfnn_curv.cuh) the iteration is split where the stencils
// are: k_var_lazy_curv(n) = primal step n (u_n from Lap y2_n, read-only here) + the lazy y1 update of k_var_lazy_tv
// (reads alpha u(2) y2(2) grad T(2) b, writes u(2) alpha = 11 elements), k_var_efused(n) (reg = 1), then k_curv_y2(n) --
// the very kernel of the NN-curvature model: u is the first 2K fields of x and y2 follows the K fields of y1 in both --
// (8 elements) and k_var_curv_fin(n). 19 elements instead of the 17 + 9 of the split passes, no position recomputed.
template <typename S, bool NEUMANN>
__global__ __launch_bounds__(CP_BLOCK, 2) void k_var_lazy_curv(MfParams P, const S *__restrict__ xo, S *__restrict__ xn,
                                                                const S *__restrict__ pin, S *__restrict__ pout,
                                                                const S *__restrict__ g, const S *__restrict__ bbar,
                                                                const double *__restrict__ lz_in, double *__restrict__ lz_out,
                                                                const CpCtrl *__restrict__ ctrl, double *__restrict__ part) {
    constexpr int NWV = CP_WAVES;
    __shared__ double carry_all[NWV * 2 * CP_VAR_KMAX * 2];   // [wave][step parity][image][y2 component] of lane 0
    __shared__ double stage_all[NWV * CP_NSCAL];
    const int stopped = ctrl->stop;
    const double cnx = ctrl->cn[0], cny = ctrl->cn[1];
    if (stopped) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *carry = carry_all + wave * 2 * CP_VAR_KMAX * 2;
    const int K = P.K, m = P.m;
    const size_t fs = (size_t)P.fs;
    const double sigma = P.sigma, theta = P.theta, tau = P.tau;
    const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
    const double lt = P.mu1 / P.sigma;
    const double cc = lt / (1.0 + lt), ee = sigma * (lt / (1.0 + lt));   // y1+ = cc*yt - ee*gg  (NNE/SSD.m:60-61)
    constexpr unsigned ES = (unsigned)sizeof(S);
    const unsigned mb = ES * (unsigned)m, fsb = ES * (unsigned)P.fs;
    const cp_rsrc Rab = mk_rsrc(lz_in), Rbe = mk_rsrc(lz_in + fs), Rabo = mk_rsrc(lz_out), Rbeo = mk_rsrc(lz_out + fs);

    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = -1.0e300;
    maxs[DM_S1 - CP_NSUM] = 0.0;

    const Fused2Geom GM = fused2_geom(m);
    const int ntasks = GM.parts * ((P.nl + NWV - 1) / NWV);
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int cg = task / GM.parts, pt = task - cg * GM.parts;
        const int j = cg * NWV + wave;
        if (j >= P.nl) continue;                                   // wave-uniform
        const int seg_lo = pt * GM.steps, seg_hi = min(seg_lo + GM.steps, GM.nseg) - 1;
        const int jg = P.j_off + j;
        const bool first_j = (jg == 0), last_j = (jg == P.n_global - 1);
        for (int sg = seg_hi; sg >= seg_lo; --sg) {
            const int par = (seg_hi - sg) & 1;
            const int ii = sg * 64 + lane;
            const bool valid = ii < m;
            const int i = min(ii, m - 1);
            const unsigned off = ES * (unsigned)((j + 1) * m + i);
            const unsigned off8 = 8u * (unsigned)((j + 1) * m + i);            // the per-pixel fields abar, beta are doubles
            const unsigned off_st = valid ? off : 0xFFFFFFFFu, off_st8 = valid ? off8 : 0xFFFFFFFFu;
            const bool first_i = (i == 0), last_i = (i == m - 1);
            const bool want = valid && !last_i;                    // y2_n at (i+1, j) is used
            const bool from_carry = want && lane == 63 && sg < seg_hi;
            const bool reload = want && lane == 63 && sg == seg_hi;
            const bool any_reload = __any(reload);                 // wave-uniform: first step of a part below the image's top
            const double *cin = carry + (par ^ 1) * CP_VAR_KMAX * 2;
            double *cout_ = carry + par * CP_VAR_KMAX * 2;
            const double vmask = valid ? 1.0 : 0.0;
            const double ab = bld<double>(Rab, off8), beta = bld<double>(Rbe, off8);
            double S_sb = 0.0, S_sc = 0.0, D1 = 0.0, D2 = 0.0, A1 = 0.0, A2 = 0.0, AG = 0.0, G1 = 0.0, GX = 0.0, GY = 0.0, GXA = 0.0,
                   GYA = 0.0;
            for (int k = 0; k < K; ++k) {
                const cp_rsrc Ry1 = mk_rsrc(pin + (size_t)k * fs), Ry2 = mk_rsrc(pin + ((size_t)K + 2 * (size_t)k) * fs),
                              Rg = mk_rsrc(g + 2 * (size_t)k * fs), Ru = mk_rsrc(xo + 2 * (size_t)k * fs),
                              Run = mk_rsrc(xn + 2 * (size_t)k * fs), Ro1 = mk_rsrc(pout + (size_t)k * fs);
                const double al = bld<S>(Ry1, off);
                const double gx = bld<S>(Rg, off), gy = bld<S>(Rg, off, fsb);
                const double gg = -bld_stream<S>(mk_rsrc(bbar + (size_t)k * fs), off);   // F(y, (-b), ...)   NNE/var_curvature...:131
                const double uxo_ = bld_stream<S>(Ru, off), uyo_ = bld_stream<S>(Ru, off, fsb);
                double lap[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned so = c ? fsb : 0u;
                    const double w0 = bld<S>(Ry2, off, so);
                    const double wim = bld<S>(Ry2, off - ES, so);
                    const double wjm = bld<S>(Ry2, off - mb, so), wjp = bld<S>(Ry2, off + mb, so);
                    double wip = __shfl(w0, lane + 1);
                    if (from_carry) wip = cin[2 * k + c];
                    if (any_reload) {
                        if (reload) wip = bld<S>(Ry2, off + ES, so);
                    }
                    if (lane == 0) cout_[2 * k + c] = w0;
                    // NNE/discrete_laplacian.m:23-46: a missing neighbour is the centre value (neumann) or zero (dirichlet)
                    const double a0 = !first_i ? wim : (NEUMANN ? w0 : 0.0);
                    const double a1 = !last_i ? wip : (NEUMANN ? w0 : 0.0);
                    const double a2 = !first_j ? wjm : (NEUMANN ? w0 : 0.0);
                    const double a3 = !last_j ? wjp : (NEUMANN ? w0 : 0.0);
                    lap[c] = (a0 + a1 - 2.0 * w0) * ih1 + (a2 + a3 - 2.0 * w0) * ih2;
                }
                const double y1c = al - ab;                                        // (B*y1)_k
                const double vux = gx * y1c + lap[0];                              // (A'y)_u = grad T * (B*y1) + Lap y2
                const double vuy = gy * y1c + lap[1];
                const double ux = rnd<S>((uxo_ - tau * vux) - cnx);                // NNE/mean_zero_indicator.m:50,53
                const double uy = rnd<S>((uyo_ - tau * vuy) - cny);
                const double wx = -vux, wy = -vuy;
                bst<S>(Run, off_st, 0u, ux);
                bst<S>(Run, off_st, fsb, uy);
                maxs[PM_WXMAX - CP_NSUM] = fmax(maxs[PM_WXMAX - CP_NSUM], valid ? wx : -1.0e300);
                maxs[PM_WXMIN - CP_NSUM] = fmax(maxs[PM_WXMIN - CP_NSUM], valid ? -wx : -1.0e300);
                maxs[PM_WYMAX - CP_NSUM] = fmax(maxs[PM_WYMAX - CP_NSUM], valid ? wy : -1.0e300);
                maxs[PM_WYMIN - CP_NSUM] = fmax(maxs[PM_WYMIN - CP_NSUM], valid ? -wy : -1.0e300);
                sums[PS_WX] += vmask * wx;
                sums[DS_UX] += vmask * ux;
                sums[DS_UY] += vmask * uy;
                const double uxb = ux + theta * (ux - uxo_), uyb = uy + theta * (uy - uyo_);
                const double s_cur = gx * ux + gy * uy, s_bar = gx * uxb + gy * uyb;
                const double an = rnd<S>(cc * (al + sigma * s_bar) - ee * gg);
                bst<S>(Ro1, off_st, 0u, an);
                const double dk = s_cur - gg;
                S_sb += s_bar; S_sc += s_cur; D1 += dk; D2 += dk * dk;
                A1 += an; A2 += an * an; AG += an * gg; G1 += gg; GX += gx; GY += gy; GXA += gx * an; GYA += gy * an;
            }
            const double Kd = (double)K;
            const double mrb = S_sb * P.invK, mrc = S_sc * P.invK;
            const double bn = cc * (beta + sigma * mrb);
            const double abn = A1 * P.invK;
            bst<double>(Rabo, off_st8, 0u, abn);
            bst<double>(Rbeo, off_st8, 0u, bn);
            sums[DS_F1] += vmask * ((D2 - 2.0 * mrc * D1) + Kd * mrc * mrc);             // sum_k (s_k - mrc - gg_k)^2   NNE/SSD.m:48
            const double im = 1.0 / P.mu1;
            sums[VS_Y1SQ] += vmask * (((A2 - 2.0 * bn * A1) + Kd * bn * bn) * (im * im));   // NNE/SSD.m:66-67
            sums[VS_Y1G] += vmask * ((AG - bn * G1) * im);
            sums[VS_SGX] += vmask * (GXA - abn * GX);                                    // sum_k grad T_k * (B*y1+)_k
            sums[VS_SGY] += vmask * (GYA - abn * GY);
            wave_lds_sync();
        }
    }
    {
        double *st = stage_all + wave * CP_NSCAL;
#pragma unroll
        for (int a = 0; a < CP_NSUM; ++a) {
            const double v = wave_sum(sums[a]);
            if (lane == 0) st[a] = v;
        }
#pragma unroll
        for (int a = 0; a < CP_NMAX; ++a) {
            const double v = wave_max(maxs[a]);
            if (lane == 0) st[CP_NSUM + a] = v;
        }
        __syncthreads();
        if (tid < CP_NSCAL) {
            const bool is_max = tid >= CP_NSUM;
            double v = stage_all[tid];
#pragma unroll
            for (int w = 1; w < NWV; ++w) {
                const double o = stage_all[w * CP_NSCAL + tid];
                v = is_max ? fmax(v, o) : v + o;
            }
            part[(size_t)blockIdx.x * CP_PSTRIDE + tid] = v;
        }
    }
}

// lazy state from a plain y1 (abar = mean_k y1_k, beta = 0) and back (y1_k = alpha_k - beta); lz = [abar field | beta field]
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_var_lazy_init(MfParams P, const S *__restrict__ p, double *__restrict__ lz) {
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int e = q + P.m;
        double s = 0.0;
        for (int k = 0; k < P.K; ++k) s += (double)p[(size_t)k * P.fs + e];
        lz[e] = s * P.invK;
        lz[(size_t)P.fs + e] = 0.0;
    }
}
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_var_lazy_finish(MfParams P, S *__restrict__ p, const double *__restrict__ lz) {
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int e = q + P.m;
        const double beta = lz[(size_t)P.fs + e];
        for (int k = 0; k < P.K; ++k) p[(size_t)k * P.fs + e] = (S)((double)p[(size_t)k * P.fs + e] - beta);
    }
}

struct VarEFusedArgs {
    int nblk;
    int n;               // completes primal row n and dual row n, starts dual row n+1; start = 1: only the last (after dual step 1)
    int start, check_stop;
    int reg;             // 1: variance-curvature: F2, F2* and the stop rule belong to k_var_curv_fin, which runs after k_curv_y2
    double tol, mu1, mu2, tau, Ntot;
};

__global__ __launch_bounds__(256) void k_var_efused(VarEFusedArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                    double *hist_p, double *hist_d) {
    __shared__ double sc[CP_NSCAL], stage[256];
    if (ctrl->stop) return;
    reduce_scalars(part, A.nblk, sc, stage);
    if (threadIdx.x != 0) return;
    const int n = A.n;
    double *hd1 = hist_d + (size_t)(n + 1) * CP_HCOLS_VAR;
    if (!A.start) {
        double *hp = hist_p + (size_t)n * CP_HCOLS_VAR, *hd = hist_d + (size_t)n * CP_HCOLS_VAR;
        const double F1 = A.mu1 * 0.5 * sc[DS_F1], F2 = A.mu2 * sc[DS_F2];           // NNE/SSD.m:48, norm21.m:56
        hp[1] = F1; hp[3] = 0.0;
        if (A.reg == 0) { hp[2] = F2; hp[0] = hp[3] + (F1 + F2); }                   // chambolle_pock.m:170-171
        hp[4] = 0.0;
        hp[5] = fmax(fabs(sc[DS_UX] / A.Ntot), fabs(sc[DS_UY] / A.Ntot));            // mean_zero_indicator.m:65
        const double mw = sc[PS_WX] / A.Ntot;                                        // x-mean used for both (:78,:81)
        const double dx = fmax(sc[PM_WXMAX] - mw, mw - (-sc[PM_WXMIN]));
        const double dy = fmax(sc[PM_WYMAX] - mw, mw - (-sc[PM_WYMIN]));
        hd[3] = 0.0;
        hd[5] = fmax(dx, dy);
        hd[0] = -(hd[3] + (hd[1] + hd[2]));                                          // chambolle_pock.m:175-176
    }
    hd1[1] = A.mu1 * (0.5 * sc[VS_Y1SQ] + sc[VS_Y1G]);                               // NNE/SSD.m:66-67
    if (A.reg == 0) {
        hd1[2] = 0.0;                                                                // norm21.m:80
        hd1[4] = (sc[DM_S1] > A.mu2) ? sc[DM_S1] - A.mu2 : 0.0;                      // norm21.m:81-88
        ctrl->cn[0] = sc[DS_UX] / A.Ntot - A.tau * (sc[VS_SGX] / A.Ntot);
        ctrl->cn[1] = sc[DS_UY] / A.Ntot - A.tau * (sc[VS_SGY] / A.Ntot);
    } else {
        hd1[4] = 0.0;                                                                // NNE/SSD.m: no constraint
        if (A.start) {   // after k_var_dual<0, 1>: F2*(y2_1) and the boundary term of mean(Lap y2_1) come with the same partials
            hd1[2] = A.mu2 * (0.5 * sc[DS_F2S]);                                     // NNE/SSD.m:66-67 (g = 0)
            ctrl->cn[0] = sc[DS_UX] / A.Ntot - A.tau * (sc[VSC_SGX] / A.Ntot) - A.tau * (sc[DS_BSX] / A.Ntot);
            ctrl->cn[1] = sc[DS_UY] / A.Ntot - A.tau * (sc[VS_SGY] / A.Ntot) - A.tau * (sc[DS_BSY] / A.Ntot);
        } else {         // after k_var_lazy_curv: k_var_curv_fin adds the boundary term of y2_{n+1}
            ctrl->cn[0] = sc[DS_UX] / A.Ntot - A.tau * (sc[VS_SGX] / A.Ntot);
            ctrl->cn[1] = sc[DS_UY] / A.Ntot - A.tau * (sc[VS_SGY] / A.Ntot);
        }
    }
    if (!A.start && A.check_stop && n >= 4 && A.reg == 0) {                          // chambolle_pock.m:142-153
        bool ok = true;
        for (int r = n - 4; r <= n; ++r) {
            const double *p_ = hist_p + (size_t)r * CP_HCOLS_VAR, *d_ = hist_d + (size_t)r * CP_HCOLS_VAR;
            const double gap = fabs((p_[0] - d_[0]) / d_[0]);
            if (!(gap <= A.tol) || !(p_[4] <= A.tol) || !(p_[5] <= A.tol) || !(d_[4] <= A.tol) || !(d_[5] <= A.tol))
                ok = false;
        }
        if (ok) { ctrl->stop = 1; ctrl->stop_iter = n; }
    }
}

// after k_curv_y2(n) of the variance-curvature model: F2(A x_n) completes primal row n, F2*(y2_{n+1}) goes into dual row n+1,
// the boundary term of the mean of the next un-projected u, the stop rule for i = n (cf. k_curv_fin)
__global__ __launch_bounds__(256) void k_var_curv_fin(CurvFinArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                      double *hist_p, double *hist_d) {
    __shared__ double sc[CP_NSCAL], stage[256];
    if (ctrl->stop) return;
    reduce_scalars(part, A.nblk, sc, stage);
    if (threadIdx.x != 0) return;
    const int n = A.n;
    double *hp = hist_p + (size_t)n * CP_HCOLS_VAR, *hd1 = hist_d + (size_t)(n + 1) * CP_HCOLS_VAR;
    const double F2 = A.mu2 * 0.5 * sc[DS_F2];                                       // NNE/SSD.m:48 (g = 0)
    hp[2] = F2;
    hp[0] = hp[3] + (hp[1] + F2);                                                    // chambolle_pock.m:170-171
    hd1[2] = A.mu2 * (0.5 * sc[DS_F2S]);                                             // NNE/SSD.m:66-67 (g = 0)
    ctrl->cn[0] -= A.tau * (sc[DS_BSX] / A.Ntot);                                    // - tau * mean(Lap y2_{n+1}), zero for neumann rows
    ctrl->cn[1] -= A.tau * (sc[DS_BSY] / A.Ntot);
    if (A.check_stop && n >= 4) {                                                    // chambolle_pock.m:142-153
        bool ok = true;
        for (int r = n - 4; r <= n; ++r) {
            const double *p_ = hist_p + (size_t)r * CP_HCOLS_VAR, *d_ = hist_d + (size_t)r * CP_HCOLS_VAR;
            const double gap = fabs((p_[0] - d_[0]) / d_[0]);
            if (!(gap <= A.tol) || !(p_[4] <= A.tol) || !(p_[5] <= A.tol) || !(d_[4] <= A.tol) || !(d_[5] <= A.tol))
                ok = false;
        }
        if (ok) { ctrl->stop = 1; ctrl->stop_iter = n; }
    }
}
