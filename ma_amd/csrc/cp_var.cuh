// Chambolle-Pock passes for the variance models (NNE/var_tv_registration_no_ref_ml.m,
// NNE/var_curvature_registration_no_ref_ml.m):
//     x = u (2K fields),  A = [ B * diag(grad T) ; D | Lap ],  p = [ y1 (K) ; y2 (4K TV | 2K curvature) ]
//     F1 = SSD(y1, -B*b)   F2 = mu * norm21 per image (TV) | SSD(y2, 0) (curvature)   G = mean_zero_indicator
// One CP iteration of PDO/chambolle_pock.m:156-179 = k_var_dual -> k_var_fin_dual -> k_var_primal -> k_var_fin_primal.
// B (mean over the K images of a pixel) couples the images of one pixel only, so a thread owns a pixel and loops
// over the images twice: once for the mean, once for the update. No nuclear norm, no Gram, no eigensolve.
// History rows have 6 columns: [p, F1, F2, G, F_con, G_con] / [q, F1*, F2*, G*, F*_con, G*_con].
#pragma once
#include "cp_mfnn.cuh"

#define CP_HCOLS_VAR 6

// additional names for the slots of the shared partial layout (cp_mfnn.cuh)
enum { VS_Y1SQ = DS_F1S, VS_Y1G = DS_SGX };

struct VarFinArgs {
    int nblk, reg;
    int rp, rd;          // primal / dual history row this kernel contributes to (-1: none)
    int check_stop, update, newbuf;
    double tol, mu1, mu2, Ntot;
};

// fixed-order reduction of the scalar part of the partial rows (256 threads: 8 sub-groups x 32 slots)
__device__ __forceinline__ void reduce_scalars(const double *__restrict__ part, int nblk, double *sc, double *stage) {
    const int t = threadIdx.x, slot = t & 31, sub = t >> 5;
    const bool is_sum = slot < CP_NSUM;
    double v = is_sum ? 0.0 : -1.0e300;
    if (slot < CP_NSCAL) {
        // 8 unconditional loads (row clamped) in flight per thread: the kernel is a chain of global round trips
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = v;
        for (int bI = sub; bI < nblk; bI += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = bI + 8 * u;
                const double o = part[(size_t)min(r, nblk - 1) * CP_PSTRIDE + slot];
                const double w = is_sum ? a[u] + o : fmax(a[u], o);
                a[u] = (r < nblk) ? w : a[u];
            }
        }
        v = is_sum ? ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))
                   : fmax(fmax(fmax(a[0], a[1]), fmax(a[2], a[3])), fmax(fmax(a[4], a[5]), fmax(a[6], a[7])));
    }
    stage[t] = v;
    __syncthreads();
    if (t < CP_NSCAL) {
        const bool s = t < CP_NSUM;
        double a = stage[t];
        for (int u = 1; u < 8; ++u) {
            const double o = stage[u * 32 + t];
            a = s ? a + o : fmax(a, o);
        }
        sc[t] = a;
    }
    __syncthreads();
}

// b <- B*b (centering over the images, NNE/var_tv...:153-154)
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_var_center_b(MfParams P, S *__restrict__ b) {
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int e = q + P.m;
        double s = 0.0;
        for (int k = 0; k < P.K; ++k) s += (double)b[(size_t)k * P.fs + e];
        const double mean = s * P.invK;
        for (int k = 0; k < P.K; ++k) b[(size_t)k * P.fs + e] = (S)((double)b[(size_t)k * P.fs + e] - mean);
    }
}

// MODE 0: update (writes pout); MODE 1: monitor only (y_new := y_in). REG 0: TV, REG 1: curvature.
// S = storage type of the fields (double, or float with CP_FLAG_FP32): values are computed in double and rounded when stored
template <typename S, int MODE, int REG>
__global__ __launch_bounds__(CP_BLOCK) void k_var_dual(MfParams P, const S *__restrict__ xc,
                                                       const S *__restrict__ xo,
                                                       const S *__restrict__ pin, S *__restrict__ pout,
                                                       const S *__restrict__ g, const S *__restrict__ bbar,
                                                       const CpCtrl *__restrict__ ctrl, int cbuf_c, int cbuf_o,
                                                       double *__restrict__ part) {
    __shared__ double red[CP_WAVES * CP_NSCAL];
    if (ctrl->stop) return;
    const int tid = threadIdx.x;
    const int K = P.K, m = P.m, fs = P.fs;
    const bool linear = (P.bc == 1), neumann = (P.bc == 0);
    const double ccx = ctrl->c[cbuf_c][0], ccy = ctrl->c[cbuf_c][1];
    const double cox = ctrl->c[cbuf_o][0], coy = ctrl->c[cbuf_o][1];
    const double sigma = P.sigma, theta = P.theta, inv_sigma = 1.0 / P.sigma;

    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = 0.0;

    for (int base = blockIdx.x * CP_BLOCK; base < P.npix; base += gridDim.x * CP_BLOCK) {
        const int q = base + tid;
        if (q >= P.npix) continue;
        const int j = q / m, i = q - j * m, e = q + m, jg = P.j_off + j;
        const bool last_i = (i == m - 1), last_j = (jg == P.n_global - 1);
        const bool first_i = (i == 0), first_j = (jg == 0);

        // mean over the images of grad T . u (current and over-relaxed point)
        double src = 0.0, srb = 0.0;
        for (int k = 0; k < K; ++k) {
            const double ux = xc[(size_t)(2 * k) * fs + e] - ccx, uy = xc[(size_t)(2 * k + 1) * fs + e] - ccy;
            const double uxb = ux + theta * (ux - (xo[(size_t)(2 * k) * fs + e] - cox));
            const double uyb = uy + theta * (uy - (xo[(size_t)(2 * k + 1) * fs + e] - coy));
            const double gx = g[(size_t)(2 * k) * fs + e], gy = g[(size_t)(2 * k + 1) * fs + e];
            src += gx * ux + gy * uy;
            srb += gx * uxb + gy * uyb;
        }
        const double mrc = src * P.invK, mrb = srb * P.invK;
        double sy = 0.0, sgx = 0.0, sgy = 0.0, sgxy = 0.0, sgyy = 0.0;   // for the mean of the next un-projected u (cp_var_fused.cuh: k_var_lazy_tv continues from this dual step)

        for (int k = 0; k < K; ++k) {
            const S *uxc = xc + (size_t)(2 * k) * fs + e, *uyc = xc + (size_t)(2 * k + 1) * fs + e;
            const S *uxo = xo + (size_t)(2 * k) * fs + e, *uyo = xo + (size_t)(2 * k + 1) * fs + e;
            const double ux = uxc[0] - ccx, uy = uyc[0] - ccy;
            const double uxb = ux + theta * (ux - (uxo[0] - cox)), uyb = uy + theta * (uy - (uyo[0] - coy));
            const double gx = g[(size_t)(2 * k) * fs + e], gy = g[(size_t)(2 * k + 1) * fs + e];
            const double gg = -bbar[(size_t)k * fs + e];              // F(y, (-b), ...)   NNE/var_tv...:172

            // ---- y1: SSD* ------------------------------------------------------------------------
            const double a_cur = (gx * ux + gy * uy) - mrc;           // (A x)_1 = B(grad T . u)
            const double a_bar = (gx * uxb + gy * uyb) - mrb;
            const double dc = a_cur - gg;
            sums[DS_F1] += dc * dc;                                   // NNE/SSD.m:48
            const double y1 = pin[(size_t)k * fs + e];
            double y1n;
            if (MODE == 0) {
                const double yt = y1 + sigma * a_bar;
                // NNE/SSD.m:60-61 with :43 (tau -> 1/tau): x - tau*((x/tau + (lambda/tau) g)/(1 + lambda/tau))
                if (P.ssd_swap) {   // NNE/var_tv_registration_no_ref.m:173: SSD(y1, b, sigma, 1, flag) -> lambda = sigma, tau = 1
                    y1n = yt - (yt + sigma * gg) / (1.0 + sigma);
                } else {
                    const double lt = P.mu1 * inv_sigma;
                    y1n = yt - sigma * ((yt * inv_sigma + lt * gg) / (1.0 + lt));
                }
                y1n = rnd<S>(y1n);                                    // what the store keeps
                pout[(size_t)k * fs + e] = (S)y1n;
            } else {
                y1n = y1;
            }
            const double z = y1n / P.mu1;
            sums[VS_Y1SQ] += z * z;                                   // NNE/SSD.m:66-67
            sums[VS_Y1G] += z * gg;
            sy += y1n; sgx += gx; sgy += gy; sgxy += gx * y1n; sgyy += gy * y1n;

            // ---- y2 ------------------------------------------------------------------------------
            if (REG == 0) {
                double uxi, uyi, uxj, uyj, uxbi, uybi, uxbj, uybj;
                {
                    // neighbour loads are unconditional (halo columns make them legal), selection happens afterwards
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
                const double d0 = fwd_diff(ux, uxi, uxi, last_i, linear, P.inv_h1);
                const double d1 = fwd_diff(ux, uxj, uxj, last_j, linear, P.inv_h2);
                const double d2 = fwd_diff(uy, uyi, uyi, last_i, linear, P.inv_h1);
                const double d3 = fwd_diff(uy, uyj, uyj, last_j, linear, P.inv_h2);
                const double e0 = fwd_diff(uxb, uxbi, uxbi, last_i, linear, P.inv_h1);
                const double e1 = fwd_diff(uxb, uxbj, uxbj, last_j, linear, P.inv_h2);
                const double e2 = fwd_diff(uyb, uybi, uybi, last_i, linear, P.inv_h1);
                const double e3 = fwd_diff(uyb, uybj, uybj, last_j, linear, P.inv_h2);
                sums[DS_F2] += sqrt(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);      // NNE/norm21.m:56
                const S *y2 = pin + (size_t)(K + 4 * k) * fs + e;
                const double v0 = y2[0], v1 = y2[(size_t)fs], v2 = y2[2 * (size_t)fs], v3 = y2[3 * (size_t)fs];
                double n0, n1, n2, n3;
                if (MODE == 0) {
                    const double t0 = v0 + sigma * e0, t1 = v1 + sigma * e1, t2 = v2 + sigma * e2, t3 = v3 + sigma * e3;
                    const double nn = fmax(sqrt(t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3), P.mu2);   // NNE/norm21.m:70
                    n0 = rnd<S>((P.mu2 * t0) / nn); n1 = rnd<S>((P.mu2 * t1) / nn); n2 = rnd<S>((P.mu2 * t2) / nn); n3 = rnd<S>((P.mu2 * t3) / nn);  // :71-72
                    S *o2 = pout + (size_t)(K + 4 * k) * fs + e;
                    o2[0] = (S)n0; o2[(size_t)fs] = (S)n1; o2[2 * (size_t)fs] = (S)n2; o2[3 * (size_t)fs] = (S)n3;
                } else {
                    n0 = v0; n1 = v1; n2 = v2; n3 = v3;
                }
                maxs[DM_S1 - CP_NSUM] = fmax(maxs[DM_S1 - CP_NSUM], sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3));   // :81
            } else {
                const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
                double lapc[2], lapb[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const S *pc = c ? uyc : uxc, *po = c ? uyo : uxo;
                    const double oc = c ? ccy : ccx, oo = c ? coy : cox;
                    const double uc0 = c ? uy : ux, ub0 = c ? uyb : uxb;
                    double nb[4], nbb[4];
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
                sums[DS_F2] += lapc[0] * lapc[0] + lapc[1] * lapc[1];            // NNE/SSD.m:48 (g = 0)
                const S *y2 = pin + (size_t)(K + 2 * k) * fs + e;
                const double v0 = y2[0], v1 = y2[(size_t)fs];
                double n0, n1;
                if (MODE == 0) {
                    const double t0 = v0 + sigma * lapb[0], t1 = v1 + sigma * lapb[1];
                    const double den = 1.0 + P.mu2 * inv_sigma;
                    n0 = rnd<S>(t0 - sigma * ((t0 * inv_sigma) / den));
                    n1 = rnd<S>(t1 - sigma * ((t1 * inv_sigma) / den));
                    S *o2 = pout + (size_t)(K + 2 * k) * fs + e;
                    o2[0] = (S)n0; o2[(size_t)fs] = (S)n1;
                } else {
                    n0 = v0; n1 = v1;
                }
                const double z0 = n0 / P.mu2, z1 = n1 / P.mu2;
                sums[DS_F2S] += z0 * z0 + z1 * z1;
                if (MODE == 0) {   // sum of y2_new * rowsum(Lap): mean(Lap y2_new) for the fused path that continues from here
                    const double rs = lap_rowsum(neumann, first_i, last_i, first_j, last_j, ih1, ih2);
                    sums[DS_BSX] += n0 * rs;
                    sums[DS_BSY] += n1 * rs;
                }
            }
            sums[DS_UX] += ux;
            sums[DS_UY] += uy;
        }
        if (REG == 0) {   // sum_k grad T_k * (B*y1_new)_k (slots VS_SGX = DS_F2S, VS_SGY = DS_SGY; DS_F2S belongs to curvature)
            const double my = sy * P.invK;
            sums[DS_F2S] += sgxy - my * sgx;
            sums[DS_SGY] += sgyy - my * sgy;
        } else {          // curvature: DS_F2S holds F2*(y2); the x component goes to slot 11 (VSC_SGX, cp_var_fused.cuh)
            const double my = sy * P.invK;
            sums[11] += sgxy - my * sgx;
            sums[DS_SGY] += sgyy - my * sgy;
        }
    }
    block_reduce_store<CP_NSUM, CP_NMAX>(sums, maxs, red, part + (size_t)blockIdx.x * CP_PSTRIDE);
}

template <typename S, int MODE, int REG>
__global__ __launch_bounds__(CP_BLOCK) void k_var_primal(MfParams P, const S *__restrict__ xc,
                                                         S *__restrict__ xn, const S *__restrict__ p,
                                                         const S *__restrict__ g, const CpCtrl *__restrict__ ctrl,
                                                         int cbuf_c, double *__restrict__ part) {
    __shared__ double red[CP_WAVES * CP_NSCAL];
    if (ctrl->stop) return;
    const int tid = threadIdx.x;
    const int K = P.K, m = P.m, fs = P.fs;
    const bool linear = (P.bc == 1), neumann = (P.bc == 0);
    const double ccx = ctrl->c[cbuf_c][0], ccy = ctrl->c[cbuf_c][1];
    const double tau = P.tau;

    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = -1.0e300;

    for (int base = blockIdx.x * CP_BLOCK; base < P.npix; base += gridDim.x * CP_BLOCK) {
        const int q = base + tid;
        if (q >= P.npix) continue;
        const int j = q / m, i = q - j * m, e = q + m, jg = P.j_off + j;
        double s1 = 0.0;
        for (int k = 0; k < K; ++k) s1 += p[(size_t)k * fs + e];
        const double mean1 = s1 * P.invK;
        for (int k = 0; k < K; ++k) {
            const double yc = p[(size_t)k * fs + e] - mean1;          // B*y1   (A1' = diag(grad T)' * B)
            const double gx = g[(size_t)(2 * k) * fs + e], gy = g[(size_t)(2 * k + 1) * fs + e];
            double vux, vuy;
            if (REG == 0) {
                const S *y2 = p + (size_t)(K + 4 * k) * fs + e;
                double adj[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const S *f = y2 + (size_t)c * fs;
                    const double wc = f[0];
                    if ((c & 1) == 0) {
                        const double wm = (i >= 1) ? f[-1] : 0.0;
                        const double wl = linear ? f[(m - 1) - i] : 0.0;
                        adj[c] = fwd_diff_adj(wc, wm, wl, i, m, linear, P.inv_h1);
                    } else {
                        const double wm = (jg >= 1) ? f[-m] : 0.0;
                        const double wl = (linear && jg >= P.n_global - 2) ? f[(size_t)((P.n_global - 1) - jg) * m] : 0.0;
                        adj[c] = fwd_diff_adj(wc, wm, wl, jg, P.n_global, linear, P.inv_h2);
                    }
                }
                vux = gx * yc + (adj[0] + adj[1]);
                vuy = gy * yc + (adj[2] + adj[3]);
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
                vux = gx * yc + lap[0];
                vuy = gy * yc + lap[1];
            }
            const double ux = xc[(size_t)(2 * k) * fs + e] - ccx, uy = xc[(size_t)(2 * k + 1) * fs + e] - ccy;
            if (MODE == 0) {
                const double ntx = rnd<S>(ux - tau * vux), nty = rnd<S>(uy - tau * vuy);    // mean subtracted lazily (c of xn)
                xn[(size_t)(2 * k) * fs + e] = (S)ntx;
                xn[(size_t)(2 * k + 1) * fs + e] = (S)nty;
                sums[PS_UX] += ntx;
                sums[PS_UY] += nty;
            }
            const double wx = -vux, wy = -vuy;                               // NNE/mean_zero_indicator.m:78-84
            sums[PS_WX] += wx;
            maxs[PM_WXMAX - CP_NSUM] = fmax(maxs[PM_WXMAX - CP_NSUM], wx);
            maxs[PM_WXMIN - CP_NSUM] = fmax(maxs[PM_WXMIN - CP_NSUM], -wx);
            maxs[PM_WYMAX - CP_NSUM] = fmax(maxs[PM_WYMAX - CP_NSUM], wy);
            maxs[PM_WYMIN - CP_NSUM] = fmax(maxs[PM_WYMIN - CP_NSUM], -wy);
        }
    }
    block_reduce_store<CP_NSUM, CP_NMAX>(sums, maxs, red, part + (size_t)blockIdx.x * CP_PSTRIDE);
}

// after the dual pass: F* part of dual row rd, primal row rp (F(Ax), mean-zero violation), stop rule for i = rp
__global__ __launch_bounds__(256) void k_var_fin_dual(VarFinArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                      double *hist_p, double *hist_d) {
    __shared__ double sc[CP_NSCAL], stage[256];
    if (ctrl->stop) return;
    reduce_scalars(part, A.nblk, sc, stage);
    if (threadIdx.x != 0) return;
    if (A.rd >= 0) {
        double *hd = hist_d + (size_t)A.rd * CP_HCOLS_VAR;
        hd[1] = A.mu1 * (0.5 * sc[VS_Y1SQ] + sc[VS_Y1G]);                          // NNE/SSD.m:66-67
        hd[2] = (A.reg == 0) ? 0.0 : A.mu2 * (0.5 * sc[DS_F2S]);                   // norm21.m:80 | SSD.m:66-67 (g = 0)
        hd[4] = (A.reg == 0 && sc[DM_S1] > A.mu2) ? sc[DM_S1] - A.mu2 : 0.0;       // norm21.m:81-88
    }
    if (A.rp >= 0) {
        double *hp = hist_p + (size_t)A.rp * CP_HCOLS_VAR;
        const double F1 = A.mu1 * 0.5 * sc[DS_F1];                                 // NNE/SSD.m:48
        const double F2 = (A.reg == 0) ? A.mu2 * sc[DS_F2] : A.mu2 * 0.5 * sc[DS_F2];
        hp[1] = F1; hp[2] = F2; hp[3] = 0.0;
        hp[0] = hp[3] + (F1 + F2);                                                 // chambolle_pock.m:170-171
        hp[4] = 0.0;
        hp[5] = fmax(fabs(sc[DS_UX] / A.Ntot), fabs(sc[DS_UY] / A.Ntot));          // mean_zero_indicator.m:65
        if (A.check_stop && A.rp >= 4) {                                            // chambolle_pock.m:142-153
            bool ok = true;
            for (int r = A.rp - 4; r <= A.rp; ++r) {
                const double *p_ = hist_p + (size_t)r * CP_HCOLS_VAR, *d_ = hist_d + (size_t)r * CP_HCOLS_VAR;
                const double gap = fabs((p_[0] - d_[0]) / d_[0]);
                if (!(gap <= A.tol) || !(p_[4] <= A.tol) || !(p_[5] <= A.tol) || !(d_[4] <= A.tol) || !(d_[5] <= A.tol))
                    ok = false;
            }
            if (ok) { ctrl->stop = 1; ctrl->stop_iter = A.rp; }
        }
    }
}

// after the primal pass: mean offsets of the new x, G* part of dual row rd
__global__ __launch_bounds__(256) void k_var_fin_primal(VarFinArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                        double *hist_p, double *hist_d) {
    __shared__ double sc[CP_NSCAL], stage[256];
    if (ctrl->stop) return;
    reduce_scalars(part, A.nblk, sc, stage);
    if (threadIdx.x != 0) return;
    if (A.update) {
        ctrl->c[A.newbuf][0] = sc[PS_UX] / A.Ntot;                                 // mean_zero_indicator.m:50,53
        ctrl->c[A.newbuf][1] = sc[PS_UY] / A.Ntot;
    }
    if (A.rd >= 0) {
        double *hd = hist_d + (size_t)A.rd * CP_HCOLS_VAR;
        const double mw = sc[PS_WX] / A.Ntot;                                      // x-mean used for both (:78,:81)
        const double dx = fmax(sc[PM_WXMAX] - mw, mw - (-sc[PM_WXMIN]));
        const double dy = fmax(sc[PM_WYMAX] - mw, mw - (-sc[PM_WYMIN]));
        hd[3] = 0.0;
        hd[5] = fmax(dx, dy);
        hd[0] = -(hd[3] + (hd[1] + hd[2]));                                        // chambolle_pock.m:175-176
    }
}

// canonical state: x - pending mean offsets into xd, dual copied if the buffers differ
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_var_materialize(MfParams P, int npf, const S *__restrict__ xs,
                                                              S *__restrict__ xd, const S *__restrict__ ps,
                                                              S *__restrict__ pd, const CpCtrl *__restrict__ ctrl,
                                                              int cbuf) {
    const double cx = ctrl->c[cbuf][0], cy = ctrl->c[cbuf][1];
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int e = q + P.m;
        for (int k = 0; k < P.K; ++k) {
            xd[(size_t)(2 * k) * P.fs + e] = (S)((double)xs[(size_t)(2 * k) * P.fs + e] - cx);
            xd[(size_t)(2 * k + 1) * P.fs + e] = (S)((double)xs[(size_t)(2 * k + 1) * P.fs + e] - cy);
        }
        if (ps != pd)
            for (int f = 0; f < npf; ++f) pd[(size_t)f * P.fs + e] = ps[(size_t)f * P.fs + e];
    }
}

// ---- power iteration of NNE/matrix_norm.m on A = [B*diag(grad T); D|Lap] -------------------------------------
template <typename S, int REG>
__global__ __launch_bounds__(CP_BLOCK) void k_var_applyA(MfParams P, const double *__restrict__ x, double scale,
                                                         const S *__restrict__ g, double *__restrict__ y) {
    const int K = P.K, m = P.m, fs = P.fs;
    const bool linear = (P.bc == 1), neumann = (P.bc == 0);
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int j = q / m, i = q - j * m, e = q + m, jg = P.j_off + j;
        const bool last_i = (i == m - 1), last_j = (jg == P.n_global - 1), first_i = (i == 0), first_j = (jg == 0);
        double s = 0.0;
        for (int k = 0; k < K; ++k)
            s += g[(size_t)(2 * k) * fs + e] * (scale * x[(size_t)(2 * k) * fs + e]) +
                 g[(size_t)(2 * k + 1) * fs + e] * (scale * x[(size_t)(2 * k + 1) * fs + e]);
        const double mean = s * P.invK;
        for (int k = 0; k < K; ++k) {
            const double *ux = x + (size_t)(2 * k) * fs + e, *uy = x + (size_t)(2 * k + 1) * fs + e;
            const double u0 = scale * ux[0], v0 = scale * uy[0];
            y[(size_t)k * fs + e] = (g[(size_t)(2 * k) * fs + e] * u0 + g[(size_t)(2 * k + 1) * fs + e] * v0) - mean;
            if (REG == 0) {
                const int oi = last_i ? -1 : 1, oj = last_j ? -m : m;
                const bool ni_ = !last_i || linear, nj_ = !last_j || linear;
                const double uxi = ni_ ? scale * ux[oi] : 0.0, uyi = ni_ ? scale * uy[oi] : 0.0;
                const double uxj = nj_ ? scale * ux[oj] : 0.0, uyj = nj_ ? scale * uy[oj] : 0.0;
                double *o = y + (size_t)(K + 4 * k) * fs + e;
                o[0] = fwd_diff(u0, uxi, uxi, last_i, linear, P.inv_h1);
                o[(size_t)fs] = fwd_diff(u0, uxj, uxj, last_j, linear, P.inv_h2);
                o[2 * (size_t)fs] = fwd_diff(v0, uyi, uyi, last_i, linear, P.inv_h1);
                o[3 * (size_t)fs] = fwd_diff(v0, uyj, uyj, last_j, linear, P.inv_h2);
            } else {
                const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double *f = c ? uy : ux;
                    const double w0 = scale * f[0];
                    const double a0 = !first_i ? scale * f[-1] : (neumann ? w0 : 0.0);
                    const double a1 = !last_i ? scale * f[1] : (neumann ? w0 : 0.0);
                    const double a2 = !first_j ? scale * f[-m] : (neumann ? w0 : 0.0);
                    const double a3 = !last_j ? scale * f[m] : (neumann ? w0 : 0.0);
                    y[(size_t)(K + 2 * k + c) * fs + e] = (a0 + a1 - 2.0 * w0) * ih1 + (a2 + a3 - 2.0 * w0) * ih2;
                }
            }
        }
    }
}

template <typename S, int REG>
__global__ __launch_bounds__(CP_BLOCK) void k_var_applyAT(MfParams P, const double *__restrict__ y,
                                                          const S *__restrict__ g, double *__restrict__ xo,
                                                          double *__restrict__ part) {
    __shared__ double red[CP_WAVES * 2];
    const int K = P.K, m = P.m, fs = P.fs;
    const bool linear = (P.bc == 1), neumann = (P.bc == 0);
    double ss[1] = {0.0}, mm[1] = {0.0};
    for (int base = blockIdx.x * CP_BLOCK; base < P.npix; base += gridDim.x * CP_BLOCK) {
        const int q = base + threadIdx.x;
        if (q < P.npix) {
            const int j = q / m, i = q - j * m, e = q + m, jg = P.j_off + j;
            double s1 = 0.0;
            for (int k = 0; k < K; ++k) s1 += y[(size_t)k * fs + e];
            const double mean1 = s1 * P.invK;
            for (int k = 0; k < K; ++k) {
                const double yc = y[(size_t)k * fs + e] - mean1;
                const double gx = g[(size_t)(2 * k) * fs + e], gy = g[(size_t)(2 * k + 1) * fs + e];
                double vux, vuy;
                if (REG == 0) {
                    const double *y2 = y + (size_t)(K + 4 * k) * fs + e;
                    double adj[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double *f = y2 + (size_t)c * fs;
                        const double wc = f[0];
                        if ((c & 1) == 0) {
                            const double wm = (i >= 1) ? f[-1] : 0.0;
                            const double wl = linear ? f[(m - 1) - i] : 0.0;
                            adj[c] = fwd_diff_adj(wc, wm, wl, i, m, linear, P.inv_h1);
                        } else {
                            const double wm = (jg >= 1) ? f[-m] : 0.0;
                            const double wl = (linear && jg >= P.n_global - 2) ? f[(size_t)((P.n_global - 1) - jg) * m] : 0.0;
                            adj[c] = fwd_diff_adj(wc, wm, wl, jg, P.n_global, linear, P.inv_h2);
                        }
                    }
                    vux = gx * yc + (adj[0] + adj[1]);
                    vuy = gy * yc + (adj[2] + adj[3]);
                } else {
                    const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
                    const double *y2 = y + (size_t)(K + 2 * k) * fs + e;
                    double lap[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double *f = y2 + (size_t)c * fs;
                        const double w0 = f[0];
                        const double a0 = (i > 0) ? f[-1] : (neumann ? w0 : 0.0);
                        const double a1 = (i < m - 1) ? f[1] : (neumann ? w0 : 0.0);
                        const double a2 = (jg > 0) ? f[-m] : (neumann ? w0 : 0.0);
                        const double a3 = (jg < P.n_global - 1) ? f[m] : (neumann ? w0 : 0.0);
                        lap[c] = (a0 + a1 - 2.0 * w0) * ih1 + (a2 + a3 - 2.0 * w0) * ih2;
                    }
                    vux = gx * yc + lap[0];
                    vuy = gy * yc + lap[1];
                }
                xo[(size_t)(2 * k) * fs + e] = vux;
                xo[(size_t)(2 * k + 1) * fs + e] = vuy;
                ss[0] += vux * vux + vuy * vuy;
            }
        }
    }
    block_reduce_store<1, 1>(ss, mm, red, part + (size_t)blockIdx.x * CP_PSTRIDE);
}
