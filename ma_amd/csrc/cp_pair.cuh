// Fused Chambolle-Pock passes for the pairwise models of PDO/test_TV_L1_registration.m,
// PDO/test_TV_L2_registration.m and PDO/test_TV_L1_denoising.m.
//
// One iteration of PDO/chambolle_pock.m:156-179 = k_pw_dual -> k_pw_primal -> k_pw_fin (split passes: row 0, the first
// dual step) or, in steady state, k_pw_fused (primal step n + dual step n+1) -> k_pw_fin_fused.
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
            const double iden = 1.0 / fmax(nrm, 1.0);                              // TV_registration.m:68-69
            nn2 = 0.0;
#pragma unroll
            for (int c = 0; c < 2 * NC; ++c) {
                vn[c] = (double)(S)(vn[c] * iden);         // the stored (rounded) value is the iterate
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

// Primal step and G-side monitors at ONE pixel (e = field index, (i, j) its row / column):
//   MODE 0: un = prox_{tau G}(u - tau K'v) rounded to the storage type (chambolle_pock.m:165), MODE 1: un = u.
//   MON: gs[0] += G(un) summand, gs[1] += G*(-K'v) summand, gm[0] = max(G* violation).
// uo receives u (the iterate before the step). Shared by k_pw_primal and by k_pw_fused, which also evaluates it at the
// two forward neighbours of a pixel -- same code, same inputs, so the recomputed values equal the stored ones.
// the values one evaluation reads; loading them is split from the arithmetic so that k_pw_march can issue the loads of
// the next column before it works on the current one
template <int NC>
struct PwIn {
    double fx0[NC], fxm[NC], fy0[NC], fym[NC], u[NC], gx, gy, bb;
};
// all loads unconditional: (i-1, j) at i = 0 is the previous column's last element, (i, j-1) at j = 0 the halo column
template <typename S, int GT>
__device__ __forceinline__ void pw_load_point(const PwParams &P, const S *__restrict__ xc, const S *__restrict__ v,
                                              const S *__restrict__ g, const S *__restrict__ b, int e,
                                              PwIn<(GT == GT_SAD_DEN) ? 1 : 2> &in) {
    constexpr int NC = (GT == GT_SAD_DEN) ? 1 : 2;
    const int m = P.m, fs = P.fs;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const S *fx = v + (size_t)(2 * c) * fs + e, *fy = v + (size_t)(2 * c + 1) * fs + e;
        in.fx0[c] = fx[0]; in.fxm[c] = fx[-1];
        in.fy0[c] = fy[0]; in.fym[c] = fy[-m];
        in.u[c] = xc[(size_t)c * fs + e];
    }
    in.gx = in.gy = 0.0;
    if (GT != GT_SAD_DEN) { in.gx = g[e]; in.gy = g[(size_t)fs + e]; }
    in.bb = b[e];
}
template <typename S, int MODE, int GT, bool MON>
__device__ __forceinline__ void pw_eval_point(const PwParams &P, const PwIn<(GT == GT_SAD_DEN) ? 1 : 2> &in, int i, int j,
                                              double *un, double *uo, double *gs, double *gm, double *vc = nullptr) {
    constexpr int NC = (GT == GT_SAD_DEN) ? 1 : 2;
    const int m = P.m;
    // one reciprocal per quantity instead of a division per use (the passes are bound by fp64 issue slots once fused;
    // a product with the reciprocal differs from the quotient by at most one more ulp)
    const double tau = P.tau, lam = P.lambda, lt = P.lambda * P.tau, ilam = 1.0 / P.lambda;
    double w[NC], ut[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double fx0 = in.fx0[c], fy0 = in.fy0[c];
        if (vc) { vc[2 * c] = fx0; vc[2 * c + 1] = fy0; }                       // v at the pixel, for the caller's dual step
        const double ax = fwd_diff_adj(fx0, (i >= 1) ? in.fxm[c] : 0.0, 0.0, i, m, false, P.inv_h1);
        const double ay = fwd_diff_adj(fy0, (j >= 1) ? in.fym[c] : 0.0, 0.0, j, P.n, false, P.inv_h2);
        w[c] = ax + ay;                                                         // (K'v)_c
        const double u0 = in.u[c];
        ut[c] = u0 - tau * w[c];                                                // chambolle_pock.m:165
        un[c] = u0;
        uo[c] = u0;
    }
    if (GT == GT_SAD_REG) {
        const double gx = in.gx, gy = in.gy, bb = in.bb;
        const double ngs = gx * gx + gy * gy;
        const double ings = 1.0 / ngs;                                          // only used where ngs > 1e-14
        if (MODE == 0) {
            const double phi = bb + (gx * ut[0] + gy * ut[NC - 1]);             // SAD_registration.m:46
            if (phi < (-tau) * lam * ngs) { un[0] = ut[0] + tau * lam * gx; un[NC - 1] = ut[NC - 1] + tau * lam * gy; }
            else if (phi > tau * lam * ngs) { un[0] = ut[0] - tau * lam * gx; un[NC - 1] = ut[NC - 1] - tau * lam * gy; }
            else if (ngs > 1e-14) { const double r = phi * ings; un[0] = ut[0] - gx * r; un[NC - 1] = ut[NC - 1] - gy * r; }
            else { un[0] = ut[0]; un[NC - 1] = ut[NC - 1]; }                    // :58-76
        }
        if (MODE == 0 && sizeof(S) == 4) { un[0] = (double)(S)un[0]; un[NC - 1] = (double)(S)un[NC - 1]; }
        if (MON) {
            gs[0] += fabs(bb + (gx * un[0] + gy * un[NC - 1]));                 // :47
            // conjugate at -K'v, scaled by 1/lambda  (:86-138)
            const double zx = -w[0] * ilam, zy = -w[NC - 1] * ilam;
            double val, con;
            if (ngs < 1e-14) {
                val = -fabs(bb);
                con = sqrt(zx * zx + zy * zy);
            } else {
                const double f = -bb * ings;
                val = zx * (f * gx) + zy * (f * gy);
                const double ng = sqrt(ngs), ing = ng * ings;
                const double r0 = (gx * zx + gy * zy) * ing, r1 = (-gy * zx + gx * zy) * ing;
                if (r0 > ng) con = (zx - gx) * (zx - gx) + (zy - gy) * (zy - gy);   // squared (:126)
                else if (r0 < -ng) con = (zx + gx) * (zx + gx) + (zy + gy) * (zy + gy);   // squared (:130)
                else con = fabs(r1);                                            // plain (:134)
            }
            gs[1] += val;
            gm[0] = fmax(gm[0], con);
        }
    } else if (GT == GT_SSD_REG) {
        const double gx = in.gx, gy = in.gy, bb = in.bb;
        const double ngs = gx * gx + gy * gy;
        if (MODE == 0) {
            // (I + lt*g g') r = ut - lt*g*b     SSD_registration.m:56-63 (2x2 block solve)
            const double cx = ut[0] - lt * gx * bb, cy = ut[NC - 1] - lt * gy * bb;
            const double a11 = 1.0 + lt * gx * gx, a12 = lt * gx * gy, a22 = 1.0 + lt * gy * gy;
            const double idet = 1.0 / (a11 * a22 - a12 * a12);
            un[0] = (a22 * cx - a12 * cy) * idet;
            un[NC - 1] = (a11 * cy - a12 * cx) * idet;
        }
        if (MODE == 0 && sizeof(S) == 4) { un[0] = (double)(S)un[0]; un[NC - 1] = (double)(S)un[NC - 1]; }
        if (MON) {
            const double res = (gx * un[0] + gy * un[NC - 1]) + bb;
            gs[0] += res * res;                                                 // :50-51
            const double zx = -w[0] * ilam, zy = -w[NC - 1] * ilam;             // :70
            const double dd = (zx - bb * gx) * gx + (zy - bb * gy) * gy;
            double r1 = 0.5 * (dd * dd);                                        // :84
            double con;
            if (ngs > 1e-7) {                                                   // :81
                const double ings = 1.0 / ngs, ing = sqrt(ngs) * ings;
                r1 = r1 * (ings * ings);
                con = fabs((-gy * ing) * zx + (gx * ing) * zy);                 // :100-102
            } else {
                con = sqrt(zx * zx + zy * zy);                                  // :96
            }
            gs[1] += r1 - 0.5 * (bb * bb);                                      // :87
            gm[0] = fmax(gm[0], con);
        }
    } else {
        const double gi = in.bb;
        if (MODE == 0) {
            const double d = ut[0] - gi;                                        // SAD_denoise.m:46-52
            un[0] = (d > lam * tau) ? ut[0] - lam * tau : ((d < (-1.0) * lam * tau) ? ut[0] + lam * tau : gi);
        }
        if (MODE == 0 && sizeof(S) == 4) un[0] = (double)(S)un[0];
        if (MON) {
            gs[0] += fabs(un[0] - gi);                                          // :40
            const double z = -w[0] * ilam;                                      // :60
            gs[1] += z * gi;                                                    // :66
            gm[0] = fmax(gm[0], fabs(z));                                       // :63-65
        }
    }
}

template <typename S, int MODE, int GT, bool MON>
__device__ __forceinline__ void pw_primal_point(const PwParams &P, const S *__restrict__ xc, const S *__restrict__ v,
                                                const S *__restrict__ g, const S *__restrict__ b, int e, int i, int j,
                                                double *un, double *uo, double *gs, double *gm, double *vc = nullptr) {
    PwIn<(GT == GT_SAD_DEN) ? 1 : 2> in;
    pw_load_point<S, GT>(P, xc, v, g, b, e, in);
    pw_eval_point<S, MODE, GT, MON>(P, in, i, j, un, uo, gs, gm, vc);
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
    double sums[PW_NS] = {0.0, 0.0}, maxs[PW_NM] = {0.0, 0.0};
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int j = q / m, i = q - j * m, e = q + m;
        double un[NC], uo[NC];
        pw_primal_point<S, MODE, GT, true>(P, xc, v, g, b, e, i, j, un, uo, sums, maxs);
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) xn[(size_t)c * fs + e] = (S)un[c];
        }
    }
    block_reduce_store<PW_NS, PW_NM>(sums, maxs, red,
                                     part + ((size_t)prob * gridDim.x + blockIdx.x) * PW_PSTRIDE);
}

// ---- fused iteration: primal step n and dual step n+1 in one pass ------------------------------------------------
// x_{n-1} (xo), v_n (vin) -> x_n (xn), v_{n+1} (vout). The dual step needs xbar_n = x_n + theta (x_n - x_{n-1}) at the
// pixel and at its two forward neighbours, so the primal step is evaluated at three pixels (the neighbours' inputs
// come from cache lines this or the adjacent workgroup streams anyway). Moves 15 instead of 23 elements per pixel:
// reads u_{n-1}(2) v_n(4) grad T(2) b(1), writes u_n(2) v_{n+1}(4). Partials per block (PWF_PSTRIDE):
//   [0] sum ||D u_n||  (F, primal row n)   [1] G(u_n) summand  [2] G*(-K'v_n) summand
//   [3] max ||v_{n+1}|| (F* violation of dual row n+1)  [4] max G* violation (dual row n)
#define PWF_NS 3
#define PWF_NM 2
#define PWF_PSTRIDE 6
template <typename S, int GT>
__global__ __launch_bounds__(CP_BLOCK) void k_pw_fused(PwParams P, const S *__restrict__ xo_, S *__restrict__ xn_,
                                                       const S *__restrict__ vin_, S *__restrict__ vout_,
                                                       const S *__restrict__ g_, const S *__restrict__ b_,
                                                       const PwCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double red[CP_WAVES * PWF_PSTRIDE];
    constexpr int NC = (GT == GT_SAD_DEN) ? 1 : 2;
    const int prob = blockIdx.y;
    if (ctrl[prob].stop) return;
    const int m = P.m, fs = P.fs;
    const S *xo = xo_ + (size_t)prob * P.nxf * fs;
    S *xn = xn_ + (size_t)prob * P.nxf * fs;
    const S *vin = vin_ + (size_t)prob * P.nvf * fs;
    S *vout = vout_ + (size_t)prob * P.nvf * fs;
    const S *g = g_ + (size_t)prob * 2 * fs, *b = b_ + (size_t)prob * fs;
    const double sigma = P.sigma, theta = P.theta;
    double sums[PWF_NS] = {0.0, 0.0, 0.0}, maxs[PWF_NM] = {0.0, 0.0};
    // workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give every XCD a contiguous range of
    // 256-pixel chunks, so that the chunk holding a pixel's (i, j+1) neighbour -- m/256 chunks further -- is streamed
    // by the same XCD at about the same time and the neighbour loads hit its L2
    int bx = blockIdx.x;
    if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    for (int q = bx * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int j = q / m, i = q - j * m, e = q + m;
        const bool last_i = (i == m - 1), last_j = (j == P.n - 1);
        double unA[NC], uoA[NC], unB[NC], uoB[NC], unC[NC], uoC[NC];
        pw_primal_point<S, 0, GT, true>(P, xo, vin, g, b, e, i, j, unA, uoA, sums + 1, maxs + 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) xn[(size_t)c * fs + e] = (S)unA[c];
        // forward neighbours: evaluated unconditionally (the addresses are inside the allocation: next column or the
        // halo column), unused at the last row / column
        pw_primal_point<S, 0, GT, false>(P, xo, vin, g, b, e + 1, i + 1, j, unB, uoB, nullptr, nullptr);
        pw_primal_point<S, 0, GT, false>(P, xo, vin, g, b, e + m, i, j + 1, unC, uoC, nullptr, nullptr);
        double dcur[2 * NC], dbar[2 * NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double u0 = unA[c], ub0 = u0 + theta * (u0 - uoA[c]);           // chambolle_pock.m:179
            const double ui = unB[c], ubi = ui + theta * (ui - uoB[c]);
            const double uj = unC[c], ubj = uj + theta * (uj - uoC[c]);
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
            vn[c] = (double)vin[(size_t)c * fs + e] + sigma * dbar[c];
            nn2 += vn[c] * vn[c];
        }
        const double iden = 1.0 / fmax(sqrt(nn2), 1.0);                            // TV_registration.m:68-69
        nn2 = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) {
            vn[c] = (double)(S)(vn[c] * iden);         // the stored (rounded) value is the iterate
            vout[(size_t)c * fs + e] = (S)vn[c];
            nn2 += vn[c] * vn[c];
        }
        maxs[0] = fmax(maxs[0], sqrt(nn2));                                        // TV_registration.m:60-62
    }
    block_reduce_store<PWF_NS, PWF_NM>(sums, maxs, red,
                                       part + ((size_t)prob * gridDim.x + blockIdx.x) * PWF_PSTRIDE);
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

// ---- fused iteration, marching form ---------------------------------------------------------------------------------
// Same step as k_pw_fused with one primal evaluation per pixel instead of three. A wave owns 63 rows x PW_STRIP-ish
// columns and marches along j: at step j it evaluates the primal step at column j+1 (which becomes the pixel's own
// value at the next step), takes the (i+1, j) neighbour from lane+1 (cross-lane shift, no LDS, no barrier) and then
// does the dual step at (i, j). Lane 63 is a halo lane (its row belongs to the next wave: evaluated, never stored);
// the evaluation of the column after the strip is the only other redundancy: (64/63) * (C+1)/C evaluations per pixel.
// Partials as k_pw_fused, one row per workgroup of 4 independent waves.
template <typename S, int GT>
__global__ __launch_bounds__(CP_BLOCK) void k_pw_march(PwParams P, int strip, int ngroups, int nwaves,
                                                       const S *__restrict__ xo_, S *__restrict__ xn_,
                                                       const S *__restrict__ vin_, S *__restrict__ vout_,
                                                       const S *__restrict__ g_, const S *__restrict__ b_,
                                                       const PwCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double red[CP_WAVES * PWF_PSTRIDE];
    constexpr int NC = (GT == GT_SAD_DEN) ? 1 : 2;
    const int prob = blockIdx.y;
    if (ctrl[prob].stop) return;
    const int m = P.m, fs = P.fs;
    const S *xo = xo_ + (size_t)prob * P.nxf * fs;
    S *xn = xn_ + (size_t)prob * P.nxf * fs;
    const S *vin = vin_ + (size_t)prob * P.nvf * fs;
    S *vout = vout_ + (size_t)prob * P.nvf * fs;
    const S *g = g_ + (size_t)prob * 2 * fs, *b = b_ + (size_t)prob * fs;
    const double sigma = P.sigma, theta = P.theta;
    double sums[PWF_NS] = {0.0, 0.0, 0.0}, maxs[PWF_NM] = {0.0, 0.0};
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * CP_WAVES + (threadIdx.x >> 6);       // wave index: row group fastest, then strip
    if (w < nwaves) {
        const int grp = w % ngroups, st = w / ngroups;
        const int i = grp * 63 + lane;
        const bool own = (lane < 63) && (i < m);                        // this lane stores / counts row i
        const int ic = min(i, m - 1);                                   // rows past the end: clamped address, masked out
        const bool last_i = (i == m - 1);
        const int j0 = st * strip, j1 = min(j0 + strip, P.n);
        double unA[NC], uoA[NC], vA[2 * NC];
        {
            double gs[2] = {0.0, 0.0}, gm[1] = {0.0};
            const int e = (j0 + 1) * m + ic;
            pw_primal_point<S, 0, GT, true>(P, xo, vin, g, b, e, ic, j0, unA, uoA, gs, gm, vA);
            if (own) {
#pragma unroll
                for (int c = 0; c < NC; ++c) xn[(size_t)c * fs + e] = (S)unA[c];
                sums[1] += gs[0]; sums[2] += gs[1]; maxs[1] = fmax(maxs[1], gm[0]);
            }
        }
        PwIn<NC> nxt;
        pw_load_point<S, GT>(P, xo, vin, g, b, (j0 + 2) * m + ic, nxt);
        for (int j = j0; j < j1; ++j) {
            const int e = (j + 1) * m + ic;
            const bool last_j = (j == P.n - 1);
            // primal step at (i, j+1): this wave's own pixel if the column is inside the strip, else the halo column.
            // Its inputs were loaded one step ago; the loads of column j+2 go out now (clamped to the halo column)
            const PwIn<NC> cur = nxt;
            pw_load_point<S, GT>(P, xo, vin, g, b, (min(j + 2, P.n) + 1) * m + ic, nxt);
            double unC[NC], uoC[NC], vC[2 * NC];
            {
                double gs[2] = {0.0, 0.0}, gm[1] = {0.0};
                pw_eval_point<S, 0, GT, true>(P, cur, ic, j + 1, unC, uoC, gs, gm, vC);
                if (own && j + 1 < j1) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) xn[(size_t)c * fs + e + m] = (S)unC[c];
                    sums[1] += gs[0]; sums[2] += gs[1]; maxs[1] = fmax(maxs[1], gm[0]);
                }
            }
            double dcur[2 * NC], dbar[2 * NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double u0 = unA[c], uold = uoA[c];
                const double ui = __shfl_down(u0, 1), uiold = __shfl_down(uold, 1);    // (i+1, j) from lane+1
                const double ub0 = u0 + theta * (u0 - uold);                           // chambolle_pock.m:179
                const double ubi = ui + theta * (ui - uiold);
                const double uj = unC[c], ubj = uj + theta * (uj - uoC[c]);
                // neumann rows: PDO/test_TV_L1_registration.m:67-70
                dcur[2 * c] = last_i ? 0.0 : (ui - u0) * P.inv_h1;
                dcur[2 * c + 1] = last_j ? 0.0 : (uj - u0) * P.inv_h2;
                dbar[2 * c] = last_i ? 0.0 : (ubi - ub0) * P.inv_h1;
                dbar[2 * c + 1] = last_j ? 0.0 : (ubj - ub0) * P.inv_h2;
            }
            double n2 = 0.0;
#pragma unroll
            for (int c = 0; c < 2 * NC; ++c) n2 += dcur[c] * dcur[c];
            double vn[2 * NC];
            double nn2 = 0.0;
#pragma unroll
            for (int c = 0; c < 2 * NC; ++c) {
                vn[c] = vA[c] + sigma * dbar[c];
                nn2 += vn[c] * vn[c];
            }
            const double iden = 1.0 / fmax(sqrt(nn2), 1.0);                            // TV_registration.m:68-69
            nn2 = 0.0;
#pragma unroll
            for (int c = 0; c < 2 * NC; ++c) {
                vn[c] = (double)(S)(vn[c] * iden);         // the stored (rounded) value is the iterate
                nn2 += vn[c] * vn[c];
            }
            if (own) {
#pragma unroll
                for (int c = 0; c < 2 * NC; ++c) vout[(size_t)c * fs + e] = (S)vn[c];
                sums[0] += sqrt(n2);                                                   // TV_registration.m:38,44
                maxs[0] = fmax(maxs[0], sqrt(nn2));                                    // TV_registration.m:60-62
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) { unA[c] = unC[c]; uoA[c] = uoC[c]; }
#pragma unroll
            for (int c = 0; c < 2 * NC; ++c) vA[c] = vC[c];
        }
    }
    block_reduce_store<PWF_NS, PWF_NM>(sums, maxs, red,
                                       part + ((size_t)prob * gridDim.x + blockIdx.x) * PWF_PSTRIDE);
}

// ---- k_pw_march2: the marching form with TWO rows per lane (m even) ----------------------------------------------------------
// k_pw_march's segments of 63 rows start at multiples of 504 bytes: every wave-wide access straddles five 128-byte lines instead
// of four, stores write partial lines, and the lines at a segment's ends are fetched by two waves -- config 5 moves 10.6 GB per
// launch for 8.05 GB of operands. Here a lane owns rows (i, i+1) and a wave 128 ALIGNED rows: every access is one 1 KB run of
// 16 bytes per lane (8 in fp32 storage). The row above the segment -- its primal value is the (i+1, j) neighbour of the wave's last
// row and belongs to the wave above -- is evaluated once more from loads at ONE address for the whole wave (a single 64-byte
// request per field; the evaluation runs in all lanes, the VALU is at 8 % in these kernels): no halo lane, (129/128) (C+1)/C
// evaluations per pixel. (i-1, j) of a lane's first row is the second row of lane - 1 (cross-lane shift; lane 0 loads it), of
// its second row its own first.
template <typename S>
struct __attribute__((aligned(2 * sizeof(S)))) pw_two { S a, b; };
template <typename S>
__device__ __forceinline__ pw_two<S> pw_ld2(const S *p) { return *reinterpret_cast<const pw_two<S> *>(p); }
template <typename S>
__device__ __forceinline__ void pw_st2(S *p, double a, double b) { *reinterpret_cast<pw_two<S> *>(p) = pw_two<S>{(S)a, (S)b}; }
template <int NC>
struct PwIn2 {
    PwIn<NC> A, B;          // first / second row of the lane; A.fxm is completed at evaluation time (needs lane - 1's B.fx0)
    double xm0[NC];         // lane 0: v_x at (i - 1, j) of its first row
};
template <typename S, int GT>
__device__ __forceinline__ void pw_load_two(const PwParams &P, const S *__restrict__ xc, const S *__restrict__ v,
                                            const S *__restrict__ g, const S *__restrict__ b, int e, int lane,
                                            PwIn2<(GT == GT_SAD_DEN) ? 1 : 2> &in) {
    constexpr int NC = (GT == GT_SAD_DEN) ? 1 : 2;
    const int m = P.m, fs = P.fs;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const S *fx = v + (size_t)(2 * c) * fs + e, *fy = v + (size_t)(2 * c + 1) * fs + e;
        const pw_two<S> x2 = pw_ld2(fx), y2 = pw_ld2(fy), ym = pw_ld2(fy - m), u2 = pw_ld2(xc + (size_t)c * fs + e);
        in.xm0[c] = fx[(lane == 0) ? -1 : 0];                  // lanes 1..63: a line the pair load fetches anyway, value unused
        in.A.fx0[c] = x2.a; in.B.fx0[c] = x2.b; in.B.fxm[c] = x2.a; in.A.fxm[c] = 0.0;
        in.A.fy0[c] = y2.a; in.B.fy0[c] = y2.b; in.A.fym[c] = ym.a; in.B.fym[c] = ym.b;
        in.A.u[c] = u2.a; in.B.u[c] = u2.b;
    }
    in.A.gx = in.A.gy = in.B.gx = in.B.gy = 0.0;
    if (GT != GT_SAD_DEN) {
        const pw_two<S> gx = pw_ld2(g + e), gy = pw_ld2(g + (size_t)fs + e);
        in.A.gx = gx.a; in.B.gx = gx.b; in.A.gy = gy.a; in.B.gy = gy.b;
    }
    const pw_two<S> bb = pw_ld2(b + e);
    in.A.bb = bb.a; in.B.bb = bb.b;
}
template <int NC>
__device__ __forceinline__ void pw_two_finish(PwIn2<NC> &in, int lane) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double up = __shfl_up(in.B.fx0[c], 1);           // every lane takes part
        in.A.fxm[c] = (lane == 0) ? in.xm0[c] : up;
    }
}

template <typename S, int GT>
__global__ __launch_bounds__(CP_BLOCK) void k_pw_march2(PwParams P, int strip, int ngroups, int nwaves,
                                                        const S *__restrict__ xo_, S *__restrict__ xn_,
                                                        const S *__restrict__ vin_, S *__restrict__ vout_,
                                                        const S *__restrict__ g_, const S *__restrict__ b_,
                                                        const PwCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double red[CP_WAVES * PWF_PSTRIDE];
    constexpr int NC = (GT == GT_SAD_DEN) ? 1 : 2;
    const int prob = blockIdx.y;
    if (ctrl[prob].stop) return;
    const int m = P.m, fs = P.fs;
    const S *xo = xo_ + (size_t)prob * P.nxf * fs;
    S *xn = xn_ + (size_t)prob * P.nxf * fs;
    const S *vin = vin_ + (size_t)prob * P.nvf * fs;
    S *vout = vout_ + (size_t)prob * P.nvf * fs;
    const S *g = g_ + (size_t)prob * 2 * fs, *b = b_ + (size_t)prob * fs;
    const double sigma = P.sigma, theta = P.theta;
    double sums[PWF_NS] = {0.0, 0.0, 0.0}, maxs[PWF_NM] = {0.0, 0.0};
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * CP_WAVES + (threadIdx.x >> 6);       // wave index: row group fastest, then strip
    if (w < nwaves) {
        const int grp = w % ngroups, st = w / ngroups;
        const int i0 = grp * 128 + 2 * lane;                            // rows i0, i0 + 1 (m even: both exist or neither)
        const bool own = i0 < m;
        const int ic = min(i0, m - 2);                                  // rows past the end: clamped address, masked out
        const bool last_i1 = (ic + 1 == m - 1);                         // of the lane's second row
        const int ih = min(grp * 128 + 128, m - 1);                     // the row above the segment (clamped: then it is never used)
        const int j0 = st * strip, j1 = min(j0 + strip, P.n);
        double unA[2][NC], uoA[2][NC], vA[2][2 * NC], unH[NC], uoH[NC];
        {
            PwIn2<NC> t;
            PwIn<NC> th;
            const int e = (j0 + 1) * m + ic;
            pw_load_two<S, GT>(P, xo, vin, g, b, e, lane, t);
            pw_load_point<S, GT>(P, xo, vin, g, b, (j0 + 1) * m + ih, th);
            pw_two_finish<NC>(t, lane);
            double gs[2] = {0.0, 0.0}, gm[1] = {0.0};
            pw_eval_point<S, 0, GT, true>(P, t.A, ic, j0, unA[0], uoA[0], gs, gm, vA[0]);
            pw_eval_point<S, 0, GT, true>(P, t.B, ic + 1, j0, unA[1], uoA[1], gs, gm, vA[1]);
            pw_eval_point<S, 0, GT, false>(P, th, ih, j0, unH, uoH, nullptr, nullptr);
            if (own) {
#pragma unroll
                for (int c = 0; c < NC; ++c) pw_st2<S>(xn + (size_t)c * fs + e, unA[0][c], unA[1][c]);
                sums[1] += gs[0]; sums[2] += gs[1]; maxs[1] = fmax(maxs[1], gm[0]);
            }
        }
        PwIn2<NC> nxt;
        PwIn<NC> nxth;
        pw_load_two<S, GT>(P, xo, vin, g, b, (j0 + 2) * m + ic, lane, nxt);
        pw_load_point<S, GT>(P, xo, vin, g, b, (j0 + 2) * m + ih, nxth);
        for (int j = j0; j < j1; ++j) {
            const int e = (j + 1) * m + ic;
            const bool last_j = (j == P.n - 1);
            // primal step at column j+1 (the wave's own pixels if the column is inside the strip, else the halo column): its
            // inputs were loaded one step ago; the loads of column j+2 go out now (clamped to the halo column)
            PwIn2<NC> cur = nxt;
            const PwIn<NC> curh = nxth;
            const int jn = (min(j + 2, P.n) + 1) * m;
            pw_load_two<S, GT>(P, xo, vin, g, b, jn + ic, lane, nxt);
            pw_load_point<S, GT>(P, xo, vin, g, b, jn + ih, nxth);
            pw_two_finish<NC>(cur, lane);
            double unC[2][NC], uoC[2][NC], vC[2][2 * NC], unHC[NC], uoHC[NC];
            {
                double gs[2] = {0.0, 0.0}, gm[1] = {0.0};
                pw_eval_point<S, 0, GT, true>(P, cur.A, ic, j + 1, unC[0], uoC[0], gs, gm, vC[0]);
                pw_eval_point<S, 0, GT, true>(P, cur.B, ic + 1, j + 1, unC[1], uoC[1], gs, gm, vC[1]);
                pw_eval_point<S, 0, GT, false>(P, curh, ih, j + 1, unHC, uoHC, nullptr, nullptr);
                if (own && j + 1 < j1) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) pw_st2<S>(xn + (size_t)c * fs + e + m, unC[0][c], unC[1][c]);
                    sums[1] += gs[0]; sums[2] += gs[1]; maxs[1] = fmax(maxs[1], gm[0]);
                }
            }
            double vn[2][2 * NC];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool last_i = (r == 1) && last_i1;
                double dcur[2 * NC], dbar[2 * NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double u0 = unA[r][c], uold = uoA[r][c];
                    double ui, uiold;                                                   // (i+1, j)
                    if (r == 0) { ui = unA[1][c]; uiold = uoA[1][c]; }
                    else {
                        const double su = __shfl_down(unA[0][c], 1), so = __shfl_down(uoA[0][c], 1);
                        ui = (lane == 63) ? unH[c] : su;
                        uiold = (lane == 63) ? uoH[c] : so;
                    }
                    const double ub0 = u0 + theta * (u0 - uold);                           // chambolle_pock.m:179
                    const double ubi = ui + theta * (ui - uiold);
                    const double uj = unC[r][c], ubj = uj + theta * (uj - uoC[r][c]);
                    // neumann rows: PDO/test_TV_L1_registration.m:67-70
                    dcur[2 * c] = last_i ? 0.0 : (ui - u0) * P.inv_h1;
                    dcur[2 * c + 1] = last_j ? 0.0 : (uj - u0) * P.inv_h2;
                    dbar[2 * c] = last_i ? 0.0 : (ubi - ub0) * P.inv_h1;
                    dbar[2 * c + 1] = last_j ? 0.0 : (ubj - ub0) * P.inv_h2;
                }
                double n2 = 0.0;
#pragma unroll
                for (int c = 0; c < 2 * NC; ++c) n2 += dcur[c] * dcur[c];
                double nn2 = 0.0;
#pragma unroll
                for (int c = 0; c < 2 * NC; ++c) {
                    vn[r][c] = vA[r][c] + sigma * dbar[c];
                    nn2 += vn[r][c] * vn[r][c];
                }
                const double iden = 1.0 / fmax(sqrt(nn2), 1.0);                            // TV_registration.m:68-69
                nn2 = 0.0;
#pragma unroll
                for (int c = 0; c < 2 * NC; ++c) {
                    vn[r][c] = (double)(S)(vn[r][c] * iden);         // the stored (rounded) value is the iterate
                    nn2 += vn[r][c] * vn[r][c];
                }
                if (own) {
                    sums[0] += sqrt(n2);                                                   // TV_registration.m:38,44
                    maxs[0] = fmax(maxs[0], sqrt(nn2));                                    // TV_registration.m:60-62
                }
            }
            if (own) {
#pragma unroll
                for (int c = 0; c < 2 * NC; ++c) pw_st2<S>(vout + (size_t)c * fs + e, vn[0][c], vn[1][c]);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int c = 0; c < NC; ++c) { unA[r][c] = unC[r][c]; uoA[r][c] = uoC[r][c]; }
#pragma unroll
                for (int c = 0; c < 2 * NC; ++c) vA[r][c] = vC[r][c];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) { unH[c] = unHC[c]; uoH[c] = uoHC[c]; }
        }
    }
    block_reduce_store<PWF_NS, PWF_NM>(sums, maxs, red,
                                       part + ((size_t)prob * gridDim.x + blockIdx.x) * PWF_PSTRIDE);
}

// history rows of the fused iteration n: primal row n complete, G* part of dual row n (its F* violation was written by
// the previous launch), F* violation of dual row n+1; then the stop rule on rows n-4..n (chambolle_pock.m:142-153)
struct PwFusedFinArgs {
    int nblk, gt, n, check_stop, hist_rows;
    double tol, lambda;
};
// 1024 threads = 128 sub-groups x 8 slots, 4 unconditional loads (row clamped) in flight per thread: the kernel is a
// chain of global round trips on one workgroup per problem (2112 partial rows at 2048^2: 25 us with 256 threads and one
// load at a time).
#define PW_FINF_THREADS 1024
__global__ __launch_bounds__(PW_FINF_THREADS) void k_pw_fin_fused(PwFusedFinArgs A, const double *__restrict__ part_,
                                                                  PwCtrl *ctrl, double *hist_p_, double *hist_d_) {
    __shared__ double stage[PW_FINF_THREADS], stage2[64], tot[8];
    const int prob = blockIdx.x;
    const int stopped = ctrl[prob].stop;
    const int t = threadIdx.x, slot = t & 7, sub = t >> 3;
    constexpr int NSUB = PW_FINF_THREADS / 8;
    const double *part = part_ + (size_t)prob * A.nblk * PWF_PSTRIDE;
    {
        const bool live = slot < PWF_NS + PWF_NM, is_sum = slot < PWF_NS;
        const int sl = live ? slot : 0;
        double v[4] = {0.0, 0.0, 0.0, 0.0};               // all maxima are of non-negative quantities
        for (int bI = sub; bI < A.nblk; bI += 4 * NSUB) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = bI + u * NSUB;
                const double o = part[(size_t)min(r, A.nblk - 1) * PWF_PSTRIDE + sl];
                const double w = is_sum ? v[u] + o : fmax(v[u], o);
                v[u] = (r < A.nblk) ? w : v[u];
            }
        }
        stage[t] = live ? (is_sum ? (v[0] + v[1]) + (v[2] + v[3]) : fmax(fmax(v[0], v[1]), fmax(v[2], v[3]))) : 0.0;
    }
    if (stopped) return;
    __syncthreads();
    if (t < 64) {   // 8 lanes per slot combine 16 sub-groups each, then lane 0 of the slot the 8 partial results
        const int s8 = t & 7, grp = t >> 3;
        const bool is_sum = s8 < PWF_NS;
        double v = stage[(grp * 16) * 8 + s8];
        for (int u = 1; u < 16; ++u) {
            const double o = stage[(grp * 16 + u) * 8 + s8];
            v = is_sum ? v + o : fmax(v, o);
        }
        stage2[t] = v;
    }
    __syncthreads();
    if (t < 8) {
        const bool is_sum = t < PWF_NS;
        double v = stage2[t];
        for (int u = 1; u < 8; ++u) {
            const double o = stage2[u * 8 + t];
            v = is_sum ? v + o : fmax(v, o);
        }
        tot[t] = v;
    }
    __syncthreads();
    if (t != 0) return;
    const double tvsum = tot[0], gsum = tot[1], gssum = tot[2], vmax = tot[3], gscon = tot[4];
    double *hp = hist_p_ + ((size_t)prob * A.hist_rows + A.n) * CP_HCOLS_PW;
    double *hd = hist_d_ + ((size_t)prob * A.hist_rows + A.n) * CP_HCOLS_PW;
    hp[2] = (A.gt == GT_SSD_REG) ? 0.5 * A.lambda * gsum : A.lambda * gsum;
    hp[3] = 0.0; hp[4] = 0.0;
    hp[1] = tvsum;
    hp[0] = hp[2] + hp[1];                                                         // chambolle_pock.m:170-171
    hd[1] = 0.0;
    hd[2] = A.lambda * gssum;
    hd[4] = (A.gt == GT_SAD_DEN) ? ((gscon > 1.0) ? gscon - 1.0 : 0.0) : gscon;
    hd[0] = -(hd[2] + hd[1]);                                                      // chambolle_pock.m:175-176
    hd[CP_HCOLS_PW + 3] = (vmax > 1.0) ? vmax - 1.0 : 0.0;                         // row n+1, TV_registration.m:60-62
    if (A.check_stop && A.n >= 4) {
        bool ok = true;
        for (int r = A.n - 4; r <= A.n; ++r) {
            const double *p_ = hist_p_ + ((size_t)prob * A.hist_rows + r) * CP_HCOLS_PW;
            const double *d_ = hist_d_ + ((size_t)prob * A.hist_rows + r) * CP_HCOLS_PW;
            const double gap = fabs((p_[0] - d_[0]) / d_[0]);
            if (!(gap <= A.tol) || !(p_[3] <= A.tol) || !(p_[4] <= A.tol) || !(d_[3] <= A.tol) || !(d_[4] <= A.tol))
                ok = false;
        }
        if (ok) { ctrl[prob].stop = 1; ctrl[prob].stop_iter = A.n; }
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
