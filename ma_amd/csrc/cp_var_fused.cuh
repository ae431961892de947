// Variance-TV (NNE/var_tv_registration_no_ref_ml.m): one field kernel per Chambolle-Pock iteration.
//
// k_var_fused_tv(n) = primal step n followed by dual step n+1 of PDO/chambolle_pock.m:162-165, per pixel, with the state
// read and written once:   reads  u_{n-1} (2), y1_n (1), y2_n (4), grad T (2), B*b (1)     writes u_n (2), y1 (1), y2 (4)
// = 17 elements per pixel-image instead of the 28 of the split passes (k_var_dual 17 + k_var_primal 11).
// Same organisation as k_mf_fused_tv (cp_mfnn_fused.cuh): a wavefront marches along one image column in aligned segments
// of 64 rows, (i+1, j) from lane+1 / the LDS hand-over of the previous step, (i, j+1) recomputed from y_n (read-only
// here), four adjacent columns per workgroup, no workgroup barrier in the loop. What is different:
//   * A = [B*diag(grad T); D]: B (mean over the K images of a pixel) enters twice. (A'y)_u,k = grad T_k * (B*y1)_k + D'y2_k
//     needs the mean of y1 over the images at every position where the primal step is evaluated -- the own pixel and
//     (i, j+1): one pass over the images that loads y1 at both and keeps the own values in registers. (A xbar)_1,k =
//     (B*(grad T . ubar))_k needs the own-pixel primal step of ALL images before the first y1 update: the per-image loop
//     runs twice, the first leaves u_n (and ubar_n) of every image in wave-private LDS tiles, the second updates y.
//   * the mean of the next un-projected u (mean_zero_indicator, known before the pass starts as in k_mf_fused_tv):
//     mean(u - tau*A'y) = mean(u) - tau*mean(grad T * (B*y1)) since D*1 = 0; the dual step accumulates
//     sum_k grad T_k*(B*y1_new)_k = sum_k g_k y1_k - mean_k(y1) * sum_k g_k per pixel (VS_SGX, VS_SGY).
// k_var_efused: reduction of the scalars, history rows, mean for the next pass, stop rule (no Gram, no eigensolve).
#pragma once
#include "cp_mfnn_fused.cuh"
#include "cp_var.cuh"

#ifndef CP_VAR_UNROLL_B
#define CP_VAR_UNROLL_B 4
#endif
#ifndef CP_VAR_UNROLL_C
#define CP_VAR_UNROLL_C 1
#endif
enum { VS_SGX = DS_F2S, VS_SGY = DS_SGY };   // TV only: sum of grad T * (B*y1_new) per component (DS_F2S belongs to curvature)

// u_n, u_{n-1} and w = -(A'y_n)_u at (i, jg) of one image; y1c = (B*y1_n)_k at that position
template <bool LINEAR>
__device__ __forceinline__ void var_primal_u_at(const MfParams &P, int i, int jg, PixOff o, double y1c,
                                                const double *__restrict__ y2k, const double *__restrict__ gxk,
                                                const double *__restrict__ uxk, double cnx, double cny, double &uxn,
                                                double &uyn, double &uxo, double &uyo, double &wx, double &wy) {
    const int m = P.m;
    const double gx = ldo(gxk, o.c), gy = ldo(gxk + P.fs, o.c);
    double adj[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double *f = y2k + (size_t)c * P.fs;
        const double wc = ldo(f, o.c);
        if ((c & 1) == 0) {
            const double wraw = ldo(f, o.c, -(int)sizeof(double));
            const double wm = (i >= 1) ? wraw : 0.0;
            const double wl = LINEAR ? ldo(f, o.c + 8u * (unsigned)((m - 1) - i)) : 0.0;
            adj[c] = fwd_diff_adj(wc, wm, wl, i, m, LINEAR, P.inv_h1);
        } else {
            const double wraw = ldo(f, o.cm);
            const double wm = (jg >= 1) ? wraw : 0.0;
            double wl = 0.0;
            if (LINEAR) {
                const int dj = (jg >= P.n_global - 2) ? (P.n_global - 1) - jg : 0;
                wl = ldo(f, o.c + 8u * (unsigned)(dj * m));
            }
            adj[c] = fwd_diff_adj(wc, wm, wl, jg, P.n_global, LINEAR, P.inv_h2);
        }
    }
    const double vux = gx * y1c + (adj[0] + adj[1]);          // k_var_primal
    const double vuy = gy * y1c + (adj[2] + adj[3]);
    uxo = ldo(uxk, o.c);
    uyo = ldo(uxk + P.fs, o.c);
    uxn = (uxo - P.tau * vux) - cnx;                           // NNE/mean_zero_indicator.m:50
    uyn = (uyo - P.tau * vuy) - cny;                           // :53
    wx = -vux;
    wy = -vuy;
}

// NT = per-image values of the own-pixel primal step kept in wave-private LDS tiles between the two per-image loops:
//   4: u_n and ubar_n (32 KB per wave -> one workgroup per CU, 1 wave/SIMD)
//   2: u_n only; ubar_n is rebuilt from a second load of u_{n-1} (16 KB per wave -> two workgroups per CU)
// y1_n of the own pixel is loaded three times (mean, primal step, dual step) either way. The per-image loops are runtime
// loops: unrolled with register arrays the kernel needed > 512 VGPRs (measured: 630 spilled at 256, 35 ms per iteration).
template <bool LINEAR, int NT>
__global__ __launch_bounds__(CP_BLOCK, NT == 4 ? 1 : 2) void k_var_fused_tv(MfParams P, const double *__restrict__ xo,
                                                               double *__restrict__ xn, const double *__restrict__ pin,
                                                               double *__restrict__ pout, const double *__restrict__ g,
                                                               const double *__restrict__ bbar,
                                                               const CpCtrl *__restrict__ ctrl, double *__restrict__ part) {
    constexpr int NWV = CP_WAVES;
    __shared__ double carry_all[NWV * CP_KMAX * 4];   // [wave][image][ux, uy, uxbar, uybar] of lane 0 (see k_mf_fused_tv)
    __shared__ double stage_all[NWV * CP_NSCAL];
    __shared__ double tiles[NWV * NT * CP_KMAX * 64];  // [wave][ux | uy | uxbar | uybar][image][lane], wave-private
    const int stopped = ctrl->stop;
    const double cnx = ctrl->cn[0], cny = ctrl->cn[1];
    if (stopped) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *carry = carry_all + wave * CP_KMAX * 4;
    double *tX = tiles + wave * NT * CP_KMAX * 64, *tY = tX + CP_KMAX * 64;
    double *tXB = tX + (NT == 4 ? 2 : 0) * CP_KMAX * 64, *tYB = tX + (NT == 4 ? 3 : 1) * CP_KMAX * 64;   // NT == 2: unused aliases
    const int K = P.K, m = P.m;
    const size_t fs = (size_t)P.fs;
    const bool linear = LINEAR;
    const double sigma = P.sigma, theta = P.theta, inv_sigma = 1.0 / P.sigma;
    constexpr unsigned ES = 8u;
    const unsigned mb = ES * (unsigned)m;

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
        const bool last_j = (jg == P.n_global - 1);
        const int oj = (last_j && LINEAR) ? -1 : 1;
        for (int sg = seg_hi; sg >= seg_lo; --sg) {
            const int ii = sg * 64 + lane;
            const bool valid = ii < m;
            const int i = min(ii, m - 1);
            const unsigned off = ES * (unsigned)((j + 1) * m + i);
            const bool last_i = (i == m - 1);
            const int oi = (last_i && LINEAR) ? -1 : 1;
            const int nb_lane = lane + oi;
            const bool want = valid && (LINEAR || !last_i);
            const bool from_carry = want && nb_lane == 64 && sg < seg_hi;
            const bool recompute = want && (nb_lane < 0 || nb_lane > 63) && !from_carry;
            const bool any_rc = __any(recompute);
            const PixOff p0 = {off, off - mb};
            const PixOff pj = {oj > 0 ? off + mb : off - mb, oj > 0 ? off : off - 2u * mb};
            const PixOff pi = {oi > 0 ? off + ES : off - ES, (oi > 0 ? off + ES : off - ES) - mb};
            const double vmask = valid ? 1.0 : 0.0;

            // ---- pass A: the means of y1 over the images that B needs: own pixel, (i, j +- 1), and (i +- 1, j) if recomputed --
            double s1 = 0.0, s1j = 0.0, s1i = 0.0;
#pragma unroll 4
            for (int k = 0; k < K; ++k) {
                s1 += ldo(pin + (size_t)k * fs, off);
                s1j += ldo(pin + (size_t)k * fs, pj.c);
            }
            if (any_rc) {
                for (int k = 0; k < K; ++k) s1i += ldo(pin + (size_t)k * fs, pi.c);
            }
            const double mean1 = s1 * P.invK, mean1j = s1j * P.invK, mean1i = s1i * P.invK;

            // ---- pass B: primal step at the own pixel for every image; u_n (and ubar_n) go to the wave's LDS tiles ---------
            double src = 0.0, srb = 0.0;
#pragma unroll CP_VAR_UNROLL_B
            for (int k = 0; k < K; ++k) {
                const double *y2k = pin + ((size_t)K + 4 * (size_t)k) * fs;
                const double *gxk = g + 2 * (size_t)k * fs, *uxk = xo + 2 * (size_t)k * fs;
                double ux, uy, uxo_, uyo_, wx, wy;
                var_primal_u_at<LINEAR>(P, i, jg, p0, ldo(pin + (size_t)k * fs, off) - mean1, y2k, gxk, uxk, cnx, cny, ux, uy, uxo_, uyo_, wx, wy);
                const double gx = ldo(gxk, off), gy = ldo(gxk + fs, off);
                const double uxb = ux + theta * (ux - uxo_), uyb = uy + theta * (uy - uyo_);
                tX[k * 64 + lane] = ux; tY[k * 64 + lane] = uy;
                if (NT == 4) { tXB[k * 64 + lane] = uxb; tYB[k * 64 + lane] = uyb; }
                src += gx * ux + gy * uy;
                srb += gx * uxb + gy * uyb;
                if (valid) {
                    double *uo = xn + 2 * (size_t)k * fs;
                    sto(uo, off, ux);
                    sto(uo + fs, off, uy);
                    maxs[PM_WXMAX - CP_NSUM] = fmax(maxs[PM_WXMAX - CP_NSUM], wx);
                    maxs[PM_WXMIN - CP_NSUM] = fmax(maxs[PM_WXMIN - CP_NSUM], -wx);
                    maxs[PM_WYMAX - CP_NSUM] = fmax(maxs[PM_WYMAX - CP_NSUM], wy);
                    maxs[PM_WYMIN - CP_NSUM] = fmax(maxs[PM_WYMIN - CP_NSUM], -wy);
                    sums[PS_WX] += wx;
                    sums[DS_UX] += ux;
                    sums[DS_UY] += uy;
                }
            }
            const double mrc = src * P.invK, mrb = srb * P.invK;
            wave_lds_sync();

            // ---- pass C: dual step per image ------------------------------------------------------------------------------
            double sy = 0.0, sgx = 0.0, sgy = 0.0, sgxy = 0.0, sgyy = 0.0;
#pragma unroll CP_VAR_UNROLL_C
            for (int k = 0; k < K; ++k) {
                const double *y1k = pin + (size_t)k * fs, *y2k = pin + ((size_t)K + 4 * (size_t)k) * fs;
                const double *gxk = g + 2 * (size_t)k * fs, *uxk = xo + 2 * (size_t)k * fs;
                const int nl_ = nb_lane & 63;
                const double ux = tX[k * 64 + lane], uy = tY[k * 64 + lane];
                double uxi = tX[k * 64 + nl_], uyi = tY[k * 64 + nl_];
                double uxb, uyb, uxbi, uybi;
                if (NT == 4) {
                    uxb = tXB[k * 64 + lane]; uyb = tYB[k * 64 + lane];
                    uxbi = tXB[k * 64 + nl_]; uybi = tYB[k * 64 + nl_];
                } else {   // ubar_n = u_n + theta*(u_n - u_{n-1}) with u_{n-1} loaded again (own pixel and i +- 1)
                    const double o0 = ldo(uxk, off), o1 = ldo(uxk + fs, off);
                    const double q0 = ldo(uxk, pi.c), q1 = ldo(uxk + fs, pi.c);
                    uxb = ux + theta * (ux - o0); uyb = uy + theta * (uy - o1);
                    uxbi = uxi + theta * (uxi - q0); uybi = uyi + theta * (uyi - q1);
                }
                // (i, j +- 1): recomputed from y_n
                double uxj, uyj, oj0, oj1, wd0, wd1;
                var_primal_u_at<LINEAR>(P, i, jg + oj, pj, ldo(y1k, pj.c) - mean1j, y2k, gxk, uxk, cnx, cny, uxj, uyj, oj0, oj1, wd0, wd1);
                const double uxbj = uxj + theta * (uxj - oj0), uybj = uyj + theta * (uyj - oj1);
                // (i +- 1, j): lane +- 1's tile entry; lane 63: the hand-over of the previous step; recomputed where neither exists
                if (from_carry) { uxi = carry[4 * k]; uyi = carry[4 * k + 1]; uxbi = carry[4 * k + 2]; uybi = carry[4 * k + 3]; }
                if (any_rc) {
                    if (recompute) {
                        double o0, o1;
                        var_primal_u_at<LINEAR>(P, i + oi, jg, pi, ldo(y1k, pi.c) - mean1i, y2k, gxk, uxk, cnx, cny, uxi, uyi, o0, o1, wd0, wd1);
                        uxbi = uxi + theta * (uxi - o0);
                        uybi = uyi + theta * (uyi - o1);
                    }
                }
                if (lane == 0) { carry[4 * k] = ux; carry[4 * k + 1] = uy; carry[4 * k + 2] = uxb; carry[4 * k + 3] = uyb; }

                const double gx = ldo(gxk, off), gy = ldo(gxk + fs, off);
                const double gg = -ldo_stream(bbar + (size_t)k * fs, off);               // F(y, (-b), ...)   NNE/var_tv...:172
                const double y1 = ldo(y1k, off);
                const double v0 = ldo(y2k, off), v1 = ldo(y2k + fs, off), v2 = ldo(y2k + 2 * fs, off), v3 = ldo(y2k + 3 * fs, off);
                // ---- y1: SSD* ------------------------------------------------------------------------------------------------
                const double a_cur = (gx * ux + gy * uy) - mrc;                          // (A x_n)_1 = B(grad T . u_n)
                const double a_bar = (gx * uxb + gy * uyb) - mrb;
                const double dc = a_cur - gg;
                sums[DS_F1] += vmask * (dc * dc);                                        // NNE/SSD.m:48
                const double yt = y1 + sigma * a_bar;
                double y1n;
                if (P.ssd_swap) {                                                        // NNE/var_tv_registration_no_ref.m:173
                    y1n = yt - (yt + sigma * gg) / (1.0 + sigma);
                } else {
                    const double lt = P.mu1 * inv_sigma;
                    y1n = yt - sigma * ((yt * inv_sigma + lt * gg) / (1.0 + lt));        // NNE/SSD.m:60-61 with :43
                }
                const double z = y1n / P.mu1;
                sums[VS_Y1SQ] += vmask * (z * z);                                        // NNE/SSD.m:66-67
                sums[VS_Y1G] += vmask * (z * gg);
                sy += y1n; sgx += gx; sgy += gy; sgxy += gx * y1n; sgyy += gy * y1n;
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
                const double t0 = v0 + sigma * e0, t1 = v1 + sigma * e1, t2 = v2 + sigma * e2, t3 = v3 + sigma * e3;
                const double nn = fmax(sqrt(t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3), P.mu2);   // :70
                const double n0 = (P.mu2 * t0) / nn, n1 = (P.mu2 * t1) / nn, n2 = (P.mu2 * t2) / nn, n3 = (P.mu2 * t3) / nn;
                if (valid) {
                    double *o1 = pout + (size_t)k * fs, *o2 = pout + ((size_t)K + 4 * (size_t)k) * fs;
                    sto(o1, off, y1n);
                    sto(o2, off, n0); sto(o2 + fs, off, n1); sto(o2 + 2 * fs, off, n2); sto(o2 + 3 * fs, off, n3);
                    maxs[DM_S1 - CP_NSUM] = fmax(maxs[DM_S1 - CP_NSUM], sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3));   // :81
                }
            }
            // sum_k grad T_k * (B*y1_new)_k at this pixel: mean of the next un-projected u
            const double my = sy * P.invK;
            sums[VS_SGX] += vmask * (sgxy - my * sgx);
            sums[VS_SGY] += vmask * (sgyy - my * sgy);
            wave_lds_sync();   // tiles and hand-over of this step are complete before the next step touches them
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

struct VarEFusedArgs {
    int nblk;
    int n;               // completes primal row n and dual row n, starts dual row n+1; start = 1: only the last (after dual step 1)
    int start, check_stop;
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
        hp[1] = F1; hp[2] = F2; hp[3] = 0.0;
        hp[0] = hp[3] + (F1 + F2);                                                   // chambolle_pock.m:170-171
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
    hd1[2] = 0.0;                                                                    // norm21.m:80
    hd1[4] = (sc[DM_S1] > A.mu2) ? sc[DM_S1] - A.mu2 : 0.0;                          // norm21.m:81-88
    ctrl->cn[0] = sc[DS_UX] / A.Ntot - A.tau * (sc[VS_SGX] / A.Ntot);
    ctrl->cn[1] = sc[DS_UY] / A.Ntot - A.tau * (sc[VS_SGY] / A.Ntot);
    if (!A.start && A.check_stop && n >= 4) {                                        // chambolle_pock.m:142-153
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
