// Fused Chambolle-Pock passes for the mf_nn (multi-image, nuclear-norm) models.
//
// One CP iteration of NNE/mf_nn_{tv,curvature}_registration_no_ref_ml.m with PDO/chambolle_pock.m:156-179
// is executed as   k_mf_dual -> k_mf_eig -> k_mf_primal -> k_mf_fin :
//
//   k_mf_dual    y  <- prox_{sigma F*}(y + sigma*A*xbar)   (:162) for y1 (SAD*), y2 (pointwise 2x2 nn* | SSD*)
//                and the un-projected y3~ = y3 + sigma*B*Lbar together with its K x K Gram matrix (MFMA);
//                also the primal monitors F1(Ax), F2(Ax), mean(u) of the CURRENT x (they need neighbours /
//                the mean offset of x and therefore trail the primal pass by one kernel).
//   k_mf_eig     eig(Gram) -> singular values -> l1-ball projection -> W with y3 = y3~ * W (applied lazily by
//                the readers of y3~), F3*; completes primal row n-1 / dual row n; evaluates the stop rule.
//   k_mf_primal  x  <- prox_{tau G}(x - tau*A'*y)           (:165); the mean-zero projection is kept as a
//                pending scalar offset c (true u = stored - c); Gram(B*L_new) and the G* monitors.
//   k_mf_fin     offsets c, nuclear-norm feasibility of B*L, G* measures, completes dual row n.
//
// Data layout: every field is an m x nl column-major slab with one halo column on each side
// (field stride fs = m*(nl+2), element (i,j) at fs*field + m*(j+1) + i).
// x buffers: fields [2k+0]=u_x,k  [2k+1]=u_y,k  [2K+k]=L_k.
// p buffers: fields [k]=y1_k ; TV: [K+4k+c]=y2_k,c (c: Dx u_x, Dy u_x, Dx u_y, Dy u_y), [5K+k]=y3_k
//                              curv: [K+2k+c]=y2_k,c (c: Lap u_x, Lap u_y),            [3K+k]=y3_k
#pragma once
#include "cp_device.cuh"

struct CpCtrl {
    int stop;          // set by the stop rule; every later kernel of the solve returns immediately
    int stop_iter;     // i at which the rule fired
    int hs_fallbacks;  // hand-over time-outs / fall-backs of the 8-wave kernels since cp_create (hs_report; must stay 0)
    int pad1;
    double c[2][2];    // pending mean offsets of the two x buffers: true u = stored - c[buf][comp] (split passes)
    double cn[2];      // mean of the next un-projected u, known before the fused pass runs (see k_mf_fused_tv)
    double nu;         // radius of the nuclear-norm ball
    double sv[CP_KBIG];      // last singular values computed by k_mf_eig (descending not guaranteed)
    double svL[CP_KBIG];     // last singular values of B*L computed by k_mf_fin
};

struct MfParams {
    int m, nl, K, npix;
    int n_global, j_off;
    int bc;
    int fs;
    int ssd_swap;             // CP_FLAG_VAR_SSD_SWAP (variance models only)
    int parts8;               // > 0: parts per column of the 8-wave kernels (CP_TUNE_PARTS8; 0: the rule of fused2_cap)
    int norm21;               // CP_FLAG_MFNN_NORM21: F2 = mu*||.||_{2,1} per image instead of the pointwise 2x2 nuclear norm
    double inv_h1, inv_h2;
    double mu1, mu2;          // h1*h2, mu*h1*h2
    double sigma, tau, theta;
    double invK, Ntot;
};

// slots of the per-block partial buffer
enum { DS_F1 = 0, DS_F2, DS_F1S, DS_UX, DS_UY, DS_F2S, DS_SGX, DS_SGY,   // dual-side sums
       PS_WX, PS_UX, PS_UY };                                             // primal-side sums
enum { DS_BSX = PS_UX, DS_BSY = PS_UY };   // dual-side kernels of the curvature model: sum of y2_new * rowsum(Lap) per component
enum { DM_Y1 = CP_NSUM, DM_S1,                                            // dual-side maxima
       PM_WXMAX, PM_WXMIN, PM_WYMAX, PM_WYMIN, PM_VL };                   // primal-side maxima

// y3 = y3~ * W for the wave's pixel: loads the K stored values, multiplies by the K x K matrix W (staged in LDS,
// broadcast reads) and leaves the result in tile[k][lane]; returns sum_k y3[k].
// KM: the extent the kernel is built for (16: one MFMA tile; 32 = CP_KBIG: the general path for K > 16, W is KM x KM)
template <typename S, int KM = CP_KMAX>
__device__ __forceinline__ double load_y3_apply_W(const S *__restrict__ y3base, int fs, int e, bool valid,
                                                  int K, const double *Wl, double *tile, int lane) {
    double a[KM];
#pragma unroll
    for (int kp = 0; kp < KM; ++kp) a[kp] = 0.0;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        if (k < K) {
            const double t = valid ? y3base[(size_t)k * fs + e] : 0.0;
#pragma unroll
            for (int kp = 0; kp < KM; ++kp) a[kp] = fma(t, Wl[k * KM + kp], a[kp]);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int kp = 0; kp < KM; ++kp) {
        tile[kp * CP_TS + lane] = a[kp];
        s += a[kp];
    }
    return s;
}

// stage the KM x KM matrix W (global) into LDS; the caller synchronises
template <int KM = CP_KMAX>
__device__ __forceinline__ void stage_W(const double *__restrict__ W, double *Wl) {
    for (int e = threadIdx.x; e < KM * KM; e += blockDim.x) Wl[e] = W[e];
}

// MODE 0: update (writes pout); MODE 1: monitor only (y_new := y_in, nothing written but the partials).
// REG 0: TV (forward differences + pointwise 2x2 nuclear norm); REG 1: curvature (Laplacian + SSD).
template <typename S, int MODE, int REG, int KM = CP_KMAX>
__global__ __launch_bounds__(CP_BLOCK) void k_mf_dual(MfParams P, const S *__restrict__ xc,
                                                      const S *__restrict__ xo,
                                                      const S *__restrict__ pin, S *__restrict__ pout,
                                                      const S *__restrict__ g, const S *__restrict__ b,
                                                      const double *__restrict__ W, const CpCtrl *__restrict__ ctrl,
                                                      int cbuf_c, int cbuf_o, double *__restrict__ part) {
    constexpr int NT = KM / 16;
    __shared__ double tiles[CP_WAVES * KM * CP_TS];
    __shared__ double red[CP_WAVES * CP_NSCAL];
    __shared__ double Wl[KM * KM];
    if (ctrl->stop) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *tile = tiles + wave * KM * CP_TS;
    const int K = P.K, m = P.m, fs = P.fs;
    const bool linear = (P.bc == 1), neumann = (P.bc == 0);
    stage_W<KM>(W, Wl);
    __syncthreads();
    const double ccx = ctrl->c[cbuf_c][0], ccy = ctrl->c[cbuf_c][1];
    const double cox = ctrl->c[cbuf_o][0], coy = ctrl->c[cbuf_o][1];
    const double sigma = P.sigma, theta = P.theta, inv_sigma = 1.0 / P.sigma;
    const int NY2 = (REG == 0) ? 4 : 2;
    const S *y3in = pin + (size_t)(K + NY2 * K) * fs;
    S *y3out = pout + (size_t)(K + NY2 * K) * fs;

    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = 0.0;
    GramAcc<NT> acc;
    gram_acc_zero(acc);

    for (int base = blockIdx.x * CP_BLOCK; base < P.npix; base += gridDim.x * CP_BLOCK) {
        const int q = base + tid;
        const bool valid = q < P.npix;
        int i = 0, j = 0;
        if (valid) { j = q / m; i = q - j * m; }
        const int e = q + m;
        const int jg = P.j_off + j;
        const bool last_i = (i == m - 1), last_j = (jg == P.n_global - 1);
        const bool first_i = (i == 0), first_j = (jg == 0);

        load_y3_apply_W<S, KM>(y3in, fs, e, valid, K, Wl, tile, lane);

        double sumLbar = 0.0;
        for (int k = 0; k < K; ++k) {
            if (valid) {
                const S *uxc = xc + (size_t)(2 * k) * fs + e, *uyc = xc + (size_t)(2 * k + 1) * fs + e;
                const S *uxo = xo + (size_t)(2 * k) * fs + e, *uyo = xo + (size_t)(2 * k + 1) * fs + e;
                const double Lc = xc[(size_t)(2 * K + k) * fs + e], Lo = xo[(size_t)(2 * K + k) * fs + e];
                const double Lbar = Lc + theta * (Lc - Lo);
                // true displacement values (stored - pending mean offset) and the over-relaxed point
                const double ux = uxc[0] - ccx, uy = uyc[0] - ccy;
                const double uxb = ux + theta * (ux - (uxo[0] - cox)), uyb = uy + theta * (uy - (uyo[0] - coy));
                const double gx = g[(size_t)(2 * k) * fs + e], gy = g[(size_t)(2 * k + 1) * fs + e];
                const double bb = b[(size_t)k * fs + e];

                // ---- y1: SAD* -------------------------------------------------------------------
                const double r_cur = -(gx * ux + gy * uy) + Lc;       // (A x)_1
                const double r_bar = -(gx * uxb + gy * uyb) + Lbar;   // (A xbar)_1
                sums[DS_F1] += fabs(r_cur - bb);                      // NNE/SAD.m:54
                const double y1 = pin[(size_t)k * fs + e];
                double y1n;
                if (MODE == 0) {
                    const double yt = y1 + sigma * r_bar;
                    // NNE/SAD.m:71-72 with the inner call :42-49 (sigma -> 1/sigma)
                    const double Ls = yt * inv_sigma, thr = inv_sigma * P.mu1, dd = Ls - bb;
                    const double prox = (dd > thr) ? (Ls - thr) : ((dd < -thr) ? (Ls + thr) : bb);
                    y1n = rnd<S>(yt - sigma * prox);
                    pout[(size_t)k * fs + e] = (S)y1n;
                } else {
                    y1n = y1;
                }
                sums[DS_F1S] += y1n * bb;                             // NNE/SAD.m:83
                sums[DS_SGX] += gx * y1n;                             // sum of grad T' y1: mean of the next u (fused pass)
                sums[DS_SGY] += gy * y1n;
                maxs[DM_Y1 - CP_NSUM] = fmax(maxs[DM_Y1 - CP_NSUM], fabs(y1n));

                // ---- y2 -------------------------------------------------------------------------
                if (REG == 0) {
                    // neighbours (offset-invariant differences are still taken on true values)
                    double uxi, uyi, uxj, uyj, uxbi, uybi, uxbj, uybj;   // "next" along i / j (or previous at a linear boundary)
                    {
                        // neighbour loads are unconditional (halo columns make them legal), selection afterwards
                        const int oi = last_i ? -1 : 1;
                        const bool need_i = !last_i || linear;
                        const double ra0 = uxc[oi], ra1 = uyc[oi], ro0 = uxo[oi], ro1 = uyo[oi];
                        const int oj = last_j ? -m : m;
                        const bool need_j = !last_j || linear;
                        const double rc0 = uxc[oj], rc1 = uyc[oj], rp0 = uxo[oj], rp1 = uyo[oj];
                        const double a0 = need_i ? ra0 - ccx : 0.0, a1 = need_i ? ra1 - ccy : 0.0;
                        const double o0 = need_i ? ro0 - cox : 0.0, o1 = need_i ? ro1 - coy : 0.0;
                        uxi = a0; uyi = a1;
                        uxbi = a0 + theta * (a0 - o0); uybi = a1 + theta * (a1 - o1);
                        const double c0 = need_j ? rc0 - ccx : 0.0, c1 = need_j ? rc1 - ccy : 0.0;
                        const double p0 = need_j ? rp0 - cox : 0.0, p1 = need_j ? rp1 - coy : 0.0;
                        uxj = c0; uyj = c1;
                        uxbj = c0 + theta * (c0 - p0); uybj = c1 + theta * (c1 - p1);
                    }
                    // D x (monitor) and D xbar; at a boundary row "next" holds the previous sample
                    const double d0 = fwd_diff(ux, uxi, uxi, last_i, linear, P.inv_h1);
                    const double d1 = fwd_diff(ux, uxj, uxj, last_j, linear, P.inv_h2);
                    const double d2 = fwd_diff(uy, uyi, uyi, last_i, linear, P.inv_h1);
                    const double d3 = fwd_diff(uy, uyj, uyj, last_j, linear, P.inv_h2);
                    const double e0 = fwd_diff(uxb, uxbi, uxbi, last_i, linear, P.inv_h1);
                    const double e1 = fwd_diff(uxb, uxbj, uxbj, last_j, linear, P.inv_h2);
                    const double e2 = fwd_diff(uyb, uybi, uybi, last_i, linear, P.inv_h1);
                    const double e3 = fwd_diff(uyb, uybj, uybj, last_j, linear, P.inv_h2);
                    double s1v, s2v;
                    if (P.norm21) sums[DS_F2] += sqrt(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);   // NNE/norm21.m:52-56
                    else {
                        sv_2x2(d0, d1, d2, d3, s1v, s2v);
                        sums[DS_F2] += s1v + s2v;                      // NNE/pointwise_2x2_nn.m:32
                    }
                    const S *y2 = pin + (size_t)(K + 4 * k) * fs + e;
                    const double v0 = y2[0], v1 = y2[(size_t)fs], v2 = y2[2 * (size_t)fs], v3 = y2[3 * (size_t)fs];
                    double n0, n1, n2, n3;
                    if (MODE == 0) {
                        const double t0 = v0 + sigma * e0, t1 = v1 + sigma * e1, t2 = v2 + sigma * e2,
                                     t3 = v3 + sigma * e3;
                        if (P.norm21) {
                            const double nn = fmax(sqrt(t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3), P.mu2);   // NNE/norm21.m:70
                            n0 = rnd<S>((P.mu2 * t0) / nn); n1 = rnd<S>((P.mu2 * t1) / nn);               // :71
                            n2 = rnd<S>((P.mu2 * t2) / nn); n3 = rnd<S>((P.mu2 * t3) / nn);
                        } else {
                            // NNE/pointwise_2x2_nn.m:46-48: M - tau * SVT_{mu/tau}(M / tau)
                            double r0, r1, r2, r3;
                            svt_2x2(t0 * inv_sigma, t1 * inv_sigma, t2 * inv_sigma, t3 * inv_sigma, P.mu2 * inv_sigma, r0,
                                    r1, r2, r3);
                            n0 = rnd<S>(t0 - sigma * r0); n1 = rnd<S>(t1 - sigma * r1);
                            n2 = rnd<S>(t2 - sigma * r2); n3 = rnd<S>(t3 - sigma * r3);
                        }
                        S *o2 = pout + (size_t)(K + 4 * k) * fs + e;
                        o2[0] = (S)n0; o2[(size_t)fs] = (S)n1; o2[2 * (size_t)fs] = (S)n2; o2[3 * (size_t)fs] = (S)n3;
                    } else {
                        n0 = v0; n1 = v1; n2 = v2; n3 = v3;
                    }
                    if (P.norm21) s1v = sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);   // NNE/norm21.m:65,81
                    else sv_2x2(n0, n1, n2, n3, s1v, s2v);
                    maxs[DM_S1 - CP_NSUM] = fmax(maxs[DM_S1 - CP_NSUM], s1v);   // NNE/pointwise_2x2_nn.m:56
                } else {
                    // 5-point Laplacian, NNE/discrete_laplacian.m:23-46: a missing neighbour is the centre value
                    // (neumann, diagonal -1/h^2) or zero (dirichlet, diagonal -2/h^2)
                    const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
                    double lapc[2], lapb[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const S *pc = c ? uyc : uxc, *po = c ? uyo : uxo;
                        const double oc = c ? ccy : ccx, oo = c ? coy : cox;
                        const double uc0 = c ? uy : ux, ub0 = c ? uyb : uxb;
                        double nb[4], nbb[4];   // i-1, i+1, j-1, j+1
                        const bool have[4] = {!first_i, !last_i, !first_j, !last_j};
                        const int off[4] = {-1, 1, -m, m};
#pragma unroll
                        for (int d = 0; d < 4; ++d) {   // unconditional loads (halo), selection afterwards
                            const double ra = pc[off[d]], ro = po[off[d]];
                            const double a = ra - oc, o = ro - oo;
                            nb[d] = have[d] ? a : (neumann ? uc0 : 0.0);
                            nbb[d] = have[d] ? a + theta * (a - o) : (neumann ? ub0 : 0.0);
                        }
                        lapc[c] = (nb[0] + nb[1] - 2.0 * uc0) * ih1 + (nb[2] + nb[3] - 2.0 * uc0) * ih2;
                        lapb[c] = (nbb[0] + nbb[1] - 2.0 * ub0) * ih1 + (nbb[2] + nbb[3] - 2.0 * ub0) * ih2;
                    }
                    sums[DS_F2] += lapc[0] * lapc[0] + lapc[1] * lapc[1];        // NNE/SSD.m:48 (g = 0)
                    const S *y2 = pin + (size_t)(K + 2 * k) * fs + e;
                    const double v0 = y2[0], v1 = y2[(size_t)fs];
                    double n0, n1;
                    if (MODE == 0) {
                        const double t0 = v0 + sigma * lapb[0], t1 = v1 + sigma * lapb[1];
                        // NNE/SSD.m:60-61 with :43 (g = 0): x - tau * ((x/tau) / (1 + lambda/tau))
                        const double den = 1.0 + P.mu2 * inv_sigma;
                        n0 = rnd<S>(t0 - sigma * ((t0 * inv_sigma) / den));
                        n1 = rnd<S>(t1 - sigma * ((t1 * inv_sigma) / den));
                        S *o2 = pout + (size_t)(K + 2 * k) * fs + e;
                        o2[0] = (S)n0; o2[(size_t)fs] = (S)n1;
                    } else {
                        n0 = v0; n1 = v1;
                    }
                    const double z0 = n0 / P.mu2, z1 = n1 / P.mu2;
                    sums[DS_F2S] += z0 * z0 + z1 * z1;                           // NNE/SSD.m:66-67
                    if (!neumann) {   // sum of Lap(y2) = sum of y2 * rowsum(Lap): the mean of the next u (fused curvature passes)
                        const double rs = -((first_i ? 1.0 : 0.0) + (last_i ? 1.0 : 0.0)) * ih1 - ((first_j ? 1.0 : 0.0) + (last_j ? 1.0 : 0.0)) * ih2;
                        sums[DS_BSX] += n0 * rs;
                        sums[DS_BSY] += n1 * rs;
                    }
                }
                sums[DS_UX] += ux;
                sums[DS_UY] += uy;

                // ---- y3 (un-projected): y3 + sigma*(Lbar - mean_k Lbar) ---------------------------
                if (MODE == 0) tile[k * CP_TS + lane] += sigma * Lbar;
                sumLbar += Lbar;
            }
        }
        if (MODE == 0) {
            const double sm = sigma * (sumLbar * P.invK);
            for (int k = 0; k < K; ++k) {
                const double v = rnd<S>(tile[k * CP_TS + lane] - sm);
                if (valid) {
                    tile[k * CP_TS + lane] = v;
                    y3out[(size_t)k * fs + e] = (S)v;
                }
            }
        }
        __syncthreads();
        gram_tile_accum_n<NT>(tile, acc);
        __syncthreads();
    }
    double *out = part + (size_t)blockIdx.x * cp_pstride<KM>();
    block_reduce_store<CP_NSUM, CP_NMAX>(sums, maxs, red, out);
    gram_block_store_n<NT>(acc, tiles, out + CP_NSCAL);
}

template <typename S, int MODE, int REG, int KM = CP_KMAX>
__global__ __launch_bounds__(CP_BLOCK) void k_mf_primal(MfParams P, const S *__restrict__ xc,
                                                        S *__restrict__ xn, const S *__restrict__ p,
                                                        const S *__restrict__ g, const double *__restrict__ W,
                                                        const CpCtrl *__restrict__ ctrl, int cbuf_c,
                                                        double *__restrict__ part) {
    constexpr int NT = KM / 16;
    __shared__ double tiles[CP_WAVES * KM * CP_TS];
    __shared__ double red[CP_WAVES * CP_NSCAL];
    __shared__ double Wl[KM * KM];
    if (ctrl->stop) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *tile = tiles + wave * KM * CP_TS;
    const int K = P.K, m = P.m, fs = P.fs;
    const bool linear = (P.bc == 1), neumann = (P.bc == 0);
    stage_W<KM>(W, Wl);
    __syncthreads();
    const double ccx = ctrl->c[cbuf_c][0], ccy = ctrl->c[cbuf_c][1];
    const double tau = P.tau;
    const int NY2 = (REG == 0) ? 4 : 2;
    const S *y3 = p + (size_t)(K + NY2 * K) * fs;

    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = -1.0e300;
    GramAcc<NT> acc;
    gram_acc_zero(acc);

    for (int base = blockIdx.x * CP_BLOCK; base < P.npix; base += gridDim.x * CP_BLOCK) {
        const int q = base + tid;
        const bool valid = q < P.npix;
        int i = 0, j = 0;
        if (valid) { j = q / m; i = q - j * m; }
        const int e = q + m;
        const int jg = P.j_off + j;

        const double s3 = load_y3_apply_W<S, KM>(y3, fs, e, valid, K, Wl, tile, lane);
        const double mean3 = s3 * P.invK;
        double sumL = 0.0;
        for (int k = 0; k < K; ++k) {
            if (valid) {
                const double y1 = p[(size_t)k * fs + e];
                // (A'y)_L = y1 + B*y3     NNE/mf_nn_tv...:186-188 transposed
                const double vL = y1 + (tile[k * CP_TS + lane] - mean3);
                const double Lc = xc[(size_t)(2 * K + k) * fs + e];
                double Ln;
                if (MODE == 0) {
                    Ln = rnd<S>(Lc - tau * vL);                        // zero_function prox = identity
                    xn[(size_t)(2 * K + k) * fs + e] = (S)Ln;
                } else {
                    Ln = Lc;
                }
                tile[k * CP_TS + lane] = Ln;
                sumL += Ln;
                maxs[PM_VL - CP_NSUM] = fmax(maxs[PM_VL - CP_NSUM], fabs(vL));   // NNE/zero_function.m:32

                const double gx = g[(size_t)(2 * k) * fs + e], gy = g[(size_t)(2 * k + 1) * fs + e];
                double vux, vuy;
                if (REG == 0) {
                    const S *y2 = p + (size_t)(K + 4 * k) * fs + e;
                    double adj[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const S *f = y2 + (size_t)c * fs;
                        const double wc = f[0];
                        if ((c & 1) == 0) {   // Dx component: adjoint along i
                            const double wm = (i >= 1) ? f[-1] : 0.0;
                            const double wl = linear ? f[(m - 1) - i] : 0.0;
                            adj[c] = fwd_diff_adj(wc, wm, wl, i, m, linear, P.inv_h1);
                        } else {              // Dy component: adjoint along j
                            const double wm = (jg >= 1) ? f[-m] : 0.0;
                            const double wl = (linear && jg >= P.n_global - 2) ? f[(size_t)((P.n_global - 1) - jg) * m] : 0.0;
                            adj[c] = fwd_diff_adj(wc, wm, wl, jg, P.n_global, linear, P.inv_h2);
                        }
                    }
                    vux = -gx * y1 + (adj[0] + adj[1]);
                    vuy = -gy * y1 + (adj[2] + adj[3]);
                } else {
                    const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
                    const S *y2 = p + (size_t)(K + 2 * k) * fs + e;
                    double lap[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const S *f = y2 + (size_t)c * fs;
                        const double w0 = f[0];
                        const double a0 = (i > 0) ? f[-1] : (neumann ? w0 : 0.0);
                        const double a1 = (i < m - 1) ? f[1] : (neumann ? w0 : 0.0);
                        const double a2 = (jg > 0) ? f[-m] : (neumann ? w0 : 0.0);
                        const double a3 = (jg < P.n_global - 1) ? f[m] : (neumann ? w0 : 0.0);
                        lap[c] = (a0 + a1 - 2.0 * w0) * ih1 + (a2 + a3 - 2.0 * w0) * ih2;
                    }
                    vux = -gx * y1 + lap[0];
                    vuy = -gy * y1 + lap[1];
                }
                const double ux = xc[(size_t)(2 * k) * fs + e] - ccx, uy = xc[(size_t)(2 * k + 1) * fs + e] - ccy;
                if (MODE == 0) {
                    const double ntx = rnd<S>(ux - tau * vux), nty = rnd<S>(uy - tau * vuy);    // mean subtracted lazily (c of xn)
                    xn[(size_t)(2 * k) * fs + e] = (S)ntx;
                    xn[(size_t)(2 * k + 1) * fs + e] = (S)nty;
                    sums[PS_UX] += ntx;
                    sums[PS_UY] += nty;
                }
                // G*(-A'y): NNE/mean_zero_indicator.m:78-84 on w = -(A'y)_u
                const double wx = -vux, wy = -vuy;
                sums[PS_WX] += wx;
                maxs[PM_WXMAX - CP_NSUM] = fmax(maxs[PM_WXMAX - CP_NSUM], wx);
                maxs[PM_WXMIN - CP_NSUM] = fmax(maxs[PM_WXMIN - CP_NSUM], -wx);
                maxs[PM_WYMAX - CP_NSUM] = fmax(maxs[PM_WYMAX - CP_NSUM], wy);
                maxs[PM_WYMIN - CP_NSUM] = fmax(maxs[PM_WYMIN - CP_NSUM], -wy);
            }
        }
        // B * L (new L in update mode, current L in monitor mode) -> Gram for the nuclear-norm feasibility
        {
            const double mL = sumL * P.invK;
            for (int k = 0; k < K; ++k) {
                const double v = valid ? tile[k * CP_TS + lane] - mL : 0.0;
                tile[k * CP_TS + lane] = v;
            }
            for (int k = K; k < KM; ++k) tile[k * CP_TS + lane] = 0.0;
        }
        __syncthreads();
        gram_tile_accum_n<NT>(tile, acc);
        __syncthreads();
    }
    double *out = part + (size_t)blockIdx.x * cp_pstride<KM>();
    block_reduce_store<CP_NSUM, CP_NMAX>(sums, maxs, red, out);
    gram_block_store_n<NT>(acc, tiles, out + CP_NSCAL);
}
