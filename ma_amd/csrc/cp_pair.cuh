// Fused Chambolle-Pock passes for the pairwise models of PDO/test_TV_L1_registration.m,
// PDO/test_TV_L2_registration.m and PDO/test_TV_L1_denoising.m.
//
// One iteration of PDO/chambolle_pock.m:156-179 = k_pw_dual -> k_pw_primal -> k_pw_fin.
// Layout: fields are m x n column-major with one halo column on each side (fs = m*(n+2));
// x: NC fields (u_x, u_y | u), v: 2*NC fields (Dx u_c, Dy u_c per component c).
// blockIdx.y selects the problem of a batch (independent problems of identical shape).
#pragma once
#include "cp_device.cuh"

#define CP_HCOLS_PW 5    // [p, F, G, F_con, G_con] / [q, F*, G*, F*_con, G*_con]
enum { GT_SAD_REG = 0, GT_SSD_REG = 1, GT_SAD_DEN = 2 };

struct PwCtrl {
    int stop, stop_iter, pad0, pad1;
};

struct PwParams {
    int m, n, npix, fs;
    int nxf, nvf;             // fields per problem in x / v buffers
    double inv_h1, inv_h2;
    double sigma, tau, theta, lambda;
};

#define PW_NS 2
#define PW_NM 2
#define PW_PSTRIDE (PW_NS + PW_NM)

// S = storage type of the state and linearisation fields (double, or float in CP_FLAG_FP32 mode); arithmetic, partial
// sums and histories are fp64 either way, so the fp32 mode halves the HBM traffic and nothing else.
template <typename S, int MODE, int NC>
__global__ __launch_bounds__(CP_BLOCK) void k_pw_dual(PwParams P, const S *__restrict__ xc_,
                                                      const S *__restrict__ xo_,
                                                      const S *__restrict__ vin_, S *__restrict__ vout_,
                                                      const PwCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double red[CP_WAVES * PW_PSTRIDE];
    const int prob = blockIdx.y;
    if (ctrl[prob].stop) return;
    const int m = P.m, fs = P.fs;
    const S *xc = xc_ + (size_t)prob * P.nxf * fs, *xo = xo_ + (size_t)prob * P.nxf * fs;
    const S *vin = vin_ + (size_t)prob * P.nvf * fs;
    S *vout = vout_ + (size_t)prob * P.nvf * fs;
    const double sigma = P.sigma, theta = P.theta;
    double sums[PW_NS] = {0.0, 0.0}, maxs[PW_NM] = {0.0, 0.0};
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int j = q / m, i = q - j * m, e = q + m;
        const bool last_i = (i == m - 1), last_j = (j == P.n - 1);
        double dcur[2 * NC], dbar[2 * NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const S *uc = xc + (size_t)c * fs + e, *uo = xo + (size_t)c * fs + e;
            const double u0 = uc[0], ub0 = u0 + theta * (u0 - uo[0]);              // chambolle_pock.m:179
            double ui = 0.0, ubi = 0.0, uj = 0.0, ubj = 0.0;
            if (!last_i) { ui = uc[1]; ubi = ui + theta * (ui - uo[1]); }
            if (!last_j) { uj = uc[m]; ubj = uj + theta * (uj - uo[m]); }
            // neumann rows: PDO/test_TV_L1_registration.m:67-70
            dcur[2 * c] = last_i ? 0.0 : (ui - u0) * P.inv_h1;
            dcur[2 * c + 1] = last_j ? 0.0 : (uj - u0) * P.inv_h2;
            dbar[2 * c] = last_i ? 0.0 : (ubi - ub0) * P.inv_h1;
            dbar[2 * c + 1] = last_j ? 0.0 : (ubj - ub0) * P.inv_h2;
        }
        double n2 = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) n2 += dcur[c] * dcur[c];
        sums[0] += sqrt(n2);                                                       // TV_registration.m:38,44
        double vn[2 * NC];
        double nn2 = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) {
            const double v = vin[(size_t)c * fs + e];
            vn[c] = (MODE == 0) ? v + sigma * dbar[c] : v;
            nn2 += vn[c] * vn[c];
        }
        double nrm = sqrt(nn2);
        if (MODE == 0) {
            const double den = fmax(nrm, 1.0);                                     // TV_registration.m:68-69
            nn2 = 0.0;
#pragma unroll
            for (int c = 0; c < 2 * NC; ++c) {
                vn[c] = (double)(S)(vn[c] / den);          // the stored (rounded) value is the iterate
                vout[(size_t)c * fs + e] = (S)vn[c];
                nn2 += vn[c] * vn[c];
            }
            nrm = sqrt(nn2);
        }
        maxs[0] = fmax(maxs[0], nrm);                                              // TV_registration.m:60-62
    }
    block_reduce_store<PW_NS, PW_NM>(sums, maxs, red,
                                     part + ((size_t)prob * gridDim.x + blockIdx.x) * PW_PSTRIDE);
}

template <typename S, int MODE, int GT>
__global__ __launch_bounds__(CP_BLOCK) void k_pw_primal(PwParams P, const S *__restrict__ xc_,
                                                        S *__restrict__ xn_, const S *__restrict__ v_,
                                                        const S *__restrict__ g_, const S *__restrict__ b_,
                                                        const PwCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double red[CP_WAVES * PW_PSTRIDE];
    constexpr int NC = (GT == GT_SAD_DEN) ? 1 : 2;
    const int prob = blockIdx.y;
    if (ctrl[prob].stop) return;
    const int m = P.m, fs = P.fs;
    const S *xc = xc_ + (size_t)prob * P.nxf * fs;
    S *xn = xn_ + (size_t)prob * P.nxf * fs;
    const S *v = v_ + (size_t)prob * P.nvf * fs;
    const S *g = g_ + (size_t)prob * 2 * fs, *b = b_ + (size_t)prob * fs;
    const double tau = P.tau, lam = P.lambda, lt = P.lambda * P.tau;
    double sums[PW_NS] = {0.0, 0.0}, maxs[PW_NM] = {0.0, 0.0};
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int j = q / m, i = q - j * m, e = q + m;
        double w[NC], ut[NC], un[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const S *fx = v + (size_t)(2 * c) * fs + e, *fy = v + (size_t)(2 * c + 1) * fs + e;
            const double ax = fwd_diff_adj(fx[0], (i >= 1) ? fx[-1] : 0.0, 0.0, i, m, false, P.inv_h1);
            const double ay = fwd_diff_adj(fy[0], (j >= 1) ? fy[-m] : 0.0, 0.0, j, P.n, false, P.inv_h2);
            w[c] = ax + ay;                                                         // (K'v)_c
            const double u0 = xc[(size_t)c * fs + e];
            ut[c] = u0 - tau * w[c];                                                // chambolle_pock.m:165
            un[c] = u0;
        }
        if (GT == GT_SAD_REG) {
            const double gx = g[e], gy = g[(size_t)fs + e], bb = b[e];
            const double ngs = gx * gx + gy * gy;
            if (MODE == 0) {
                const double phi = bb + (gx * ut[0] + gy * ut[NC - 1]);             // SAD_registration.m:46
                if (phi < (-tau) * lam * ngs) { un[0] = ut[0] + tau * lam * gx; un[NC - 1] = ut[NC - 1] + tau * lam * gy; }
                else if (phi > tau * lam * ngs) { un[0] = ut[0] - tau * lam * gx; un[NC - 1] = ut[NC - 1] - tau * lam * gy; }
                else if (ngs > 1e-14) { const double r = phi / ngs; un[0] = ut[0] - gx * r; un[NC - 1] = ut[NC - 1] - gy * r; }
                else { un[0] = ut[0]; un[NC - 1] = ut[NC - 1]; }                    // :58-76
            }
            if (MODE == 0 && sizeof(S) == 4) { un[0] = (double)(S)un[0]; un[NC - 1] = (double)(S)un[NC - 1]; }
            sums[0] += fabs(bb + (gx * un[0] + gy * un[NC - 1]));                   // :47
            // conjugate at -K'v, scaled by 1/lambda  (:86-138)
            const double zx = -w[0] / lam, zy = -w[NC - 1] / lam;
            double val, con;
            if (ngs < 1e-14) {
                val = -fabs(bb);
                con = sqrt(zx * zx + zy * zy);
            } else {
                const double f = -bb / ngs;
                val = zx * (f * gx) + zy * (f * gy);
                const double ng = sqrt(ngs);
                const double r0 = (gx * zx + gy * zy) / ng, r1 = (-gy * zx + gx * zy) / ng;
                if (r0 > ng) con = (zx - gx) * (zx - gx) + (zy - gy) * (zy - gy);   // squared (:126)
                else if (r0 < -ng) con = (zx + gx) * (zx + gx) + (zy + gy) * (zy + gy);   // squared (:130)
                else con = fabs(r1);                                                // plain (:134)
            }
            sums[1] += val;
            maxs[0] = fmax(maxs[0], con);
        } else if (GT == GT_SSD_REG) {
            const double gx = g[e], gy = g[(size_t)fs + e], bb = b[e];
            const double ngs = gx * gx + gy * gy;
            if (MODE == 0) {
                // (I + lt*g g') r = ut - lt*g*b     SSD_registration.m:56-63 (2x2 block solve)
                const double cx = ut[0] - lt * gx * bb, cy = ut[NC - 1] - lt * gy * bb;
                const double a11 = 1.0 + lt * gx * gx, a12 = lt * gx * gy, a22 = 1.0 + lt * gy * gy;
                const double det = a11 * a22 - a12 * a12;
                un[0] = (a22 * cx - a12 * cy) / det;
                un[NC - 1] = (a11 * cy - a12 * cx) / det;
            }
            if (MODE == 0 && sizeof(S) == 4) { un[0] = (double)(S)un[0]; un[NC - 1] = (double)(S)un[NC - 1]; }
            const double res = (gx * un[0] + gy * un[NC - 1]) + bb;
            sums[0] += res * res;                                                   // :50-51
            const double zx = -w[0] / lam, zy = -w[NC - 1] / lam;                   // :70
            const double dd = (zx - bb * gx) * gx + (zy - bb * gy) * gy;
            double r1 = 0.5 * (dd * dd);                                            // :84
            double con;
            if (ngs > 1e-7) {                                                       // :81
                r1 = r1 / (ngs * ngs);
                const double ng = sqrt(ngs);
                con = fabs((-gy / ng) * zx + (gx / ng) * zy);                       // :100-102
            } else {
                con = sqrt(zx * zx + zy * zy);                                      // :96
            }
            sums[1] += r1 - 0.5 * (bb * bb);                                        // :87
            maxs[0] = fmax(maxs[0], con);
        } else {
            const double gi = b[e];
            if (MODE == 0) {
                const double d = ut[0] - gi;                                        // SAD_denoise.m:46-52
                un[0] = (d > lam * tau) ? ut[0] - lam * tau : ((d < (-1.0) * lam * tau) ? ut[0] + lam * tau : gi);
            }
            if (MODE == 0 && sizeof(S) == 4) un[0] = (double)(S)un[0];
            sums[0] += fabs(un[0] - gi);                                            // :40
            const double z = -w[0] / lam;                                           // :60
            sums[1] += z * gi;                                                      // :66
            maxs[0] = fmax(maxs[0], fabs(z));                                       // :63-65
        }
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) xn[(size_t)c * fs + e] = (S)un[c];
        }
    }
    block_reduce_store<PW_NS, PW_NM>(sums, maxs, red,
                                     part + ((size_t)prob * gridDim.x + blockIdx.x) * PW_PSTRIDE);
}

struct PwFinArgs {
    int nblk, gt;
    int rpF, rpG, rd;     // primal row receiving F (TV of K u), primal row receiving G, dual row
    int check_stop;
    int hist_rows;        // rows allocated per problem
    double tol, lambda;
};

// one workgroup (256 threads) per problem: 32 sub-groups stride the per-block partials of the 8 slots (4 dual, 4
// primal), then a fixed-order combine -- deterministic, and identical for a problem alone or inside a batch
#define PW_FIN_THREADS 256
__global__ __launch_bounds__(PW_FIN_THREADS) void k_pw_fin(PwFinArgs A, const double *__restrict__ partD,
                                                           const double *__restrict__ partP, PwCtrl *ctrl,
                                                           double *hist_p_, double *hist_d_) {
    __shared__ double stage[PW_FIN_THREADS], tot[8];
    const int prob = blockIdx.x;
    if (ctrl[prob].stop) return;
    const int t = threadIdx.x, slot8 = t & 7, sub = t >> 3;
    const double *pd = partD + (size_t)prob * A.nblk * PW_PSTRIDE, *pp = partP + (size_t)prob * A.nblk * PW_PSTRIDE;
    {
        const double *src = (slot8 < 4) ? pd : pp;
        const int slot = slot8 & 3;
        const bool is_sum = slot < PW_NS;
        double v = 0.0;                                   // all maxima are of non-negative quantities
        for (int bI = sub; bI < A.nblk; bI += PW_FIN_THREADS / 8) {
            const double o = src[(size_t)bI * PW_PSTRIDE + slot];
            v = is_sum ? v + o : fmax(v, o);
        }
        stage[t] = v;
    }
    __syncthreads();
    if (t < 8) {
        const bool is_sum = (t & 3) < PW_NS;
        double v = stage[t];
        for (int u = 1; u < PW_FIN_THREADS / 8; ++u) {
            const double o = stage[u * 8 + t];
            v = is_sum ? v + o : fmax(v, o);
        }
        tot[t] = v;
    }
    __syncthreads();
    if (t != 0) return;
    const double tvsum = tot[0], vmax = tot[2], gsum = tot[4], gssum = tot[5], gscon = tot[6];
    double *hist_p = hist_p_ + (size_t)prob * A.hist_rows * CP_HCOLS_PW;
    double *hist_d = hist_d_ + (size_t)prob * A.hist_rows * CP_HCOLS_PW;
    if (A.rpG >= 0) {
        double *hp = hist_p + (size_t)A.rpG * CP_HCOLS_PW;
        hp[2] = (A.gt == GT_SSD_REG) ? 0.5 * A.lambda * gsum : A.lambda * gsum;
        hp[3] = 0.0; hp[4] = 0.0;
    }
    if (A.rpF >= 0) {
        double *hp = hist_p + (size_t)A.rpF * CP_HCOLS_PW;
        hp[1] = tvsum;
        hp[0] = hp[2] + hp[1];                                                     // chambolle_pock.m:170-171
    }
    if (A.rd >= 0) {
        double *hd = hist_d + (size_t)A.rd * CP_HCOLS_PW;
        hd[1] = 0.0;
        hd[3] = (vmax > 1.0) ? vmax - 1.0 : 0.0;                                   // TV_registration.m:60-62
        double gs, gc;
        if (A.gt == GT_SAD_DEN) { gs = A.lambda * gssum; gc = (gscon > 1.0) ? gscon - 1.0 : 0.0; }
        else { gs = A.lambda * gssum; gc = gscon; }
        hd[2] = gs; hd[4] = gc;
        hd[0] = -(hd[2] + hd[1]);                                                  // chambolle_pock.m:175-176
    }
    if (A.check_stop && A.rpF >= 4) {                                              // chambolle_pock.m:142-153
        bool ok = true;
        for (int r = A.rpF - 4; r <= A.rpF; ++r) {
            const double *p_ = hist_p + (size_t)r * CP_HCOLS_PW, *d_ = hist_d + (size_t)r * CP_HCOLS_PW;
            const double gap = fabs((p_[0] - d_[0]) / d_[0]);
            if (!(gap <= A.tol) || !(p_[3] <= A.tol) || !(p_[4] <= A.tol) || !(d_[3] <= A.tol) || !(d_[4] <= A.tol))
                ok = false;
        }
        if (ok) { ctrl[prob].stop = 1; ctrl[prob].stop_iter = A.rpF; }
    }
}

// T(u0), grad T(u0), b = T(u0) - grad T . u0 - R   (PDO/SAD_registration.m:39-40); see k_mf_warp
struct PwWarpParams {
    int m, n, npix, fs, nxf;
    double hx, hy, inv_h1, inv_h2;
};
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_pw_warp(PwWarpParams P, const double *__restrict__ T_,
                                                      const double *__restrict__ R_, const S *__restrict__ x_,
                                                      S *__restrict__ Tu_, S *__restrict__ g_,
                                                      S *__restrict__ b_) {
    const int prob = blockIdx.y;
    const int q = blockIdx.x * CP_BLOCK + threadIdx.x;
    if (q >= P.npix) return;
    const int j = q / P.m, i = q - j * P.m, e = q + P.m;
    const S *x = x_ + (size_t)prob * P.nxf * P.fs;
    const double ux = x[e], uy = x[(size_t)P.fs + e];
    const double px = __dadd_rn(__dadd_rn(0.0, __dmul_rn(P.hx, (double)i + 0.5)), ux);
    const double py = __dadd_rn(__dadd_rn(0.0, __dmul_rn(P.hy, (double)j + 0.5)), uy);
    double val, dx, dy;
    bilinear_tap(T_ + (size_t)prob * P.npix, P.m, P.n, px, py, P.inv_h1, P.inv_h2, val, dx, dy);
    // the stored (rounded) gradient is the one the iteration uses, so b is formed from it
    const double dxs = (double)(S)dx, dys = (double)(S)dy;
    Tu_[(size_t)prob * P.fs + e] = (S)val;
    g_[(size_t)prob * 2 * P.fs + e] = (S)dx;
    g_[(size_t)prob * 2 * P.fs + P.fs + e] = (S)dy;
    b_[(size_t)prob * P.fs + e] = (S)(val - (dxs * ux + dys * uy) - R_[(size_t)prob * P.npix + q]);
}

// host double <-> device float field conversion (fp32 mode transfers)
__global__ void k_cvt_d2f(int n, const double *__restrict__ src, float *__restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
__global__ void k_cvt_f2d(int n, const float *__restrict__ src, double *__restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = (double)src[i];
}
