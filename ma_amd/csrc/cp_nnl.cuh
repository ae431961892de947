// The soft nuclear norm (NNE/nuclear_norm.m) and the L-only Chambolle-Pock problems of the landscape experiment
// NNE/test_nuclear_norm_measure.m:122-243:
//   version 1   min_L mu ||L||_* + ||L - I||_1                 K = id, F = SAD(., I, 1, sigma),  G = nuclear_norm(., k, tau, mu)
//   version 2   min_L delta{||L||_* <= nu} + ||L - I||_1       K = id, F = SAD,                   G = nuclear_norm_constraint(., k, tau, nu)
//   version 3   min_L delta{||B L||_* <= nu} + ||L - I||_1     K = B,  F = nuclear_norm_constraint, G = SAD
// No stencils: a matrix is npix x K, column major (image k at k*npix), all work is per pixel row plus K x K spectral
// work. A spectral function of a tall matrix M is applied as M * W with W = V diag(f(S)/S) V' from the eigen-
// decomposition of the Gram matrix M'M (MFMA contraction, one-wave Jacobi: the machinery of the mf_nn models).
// Singular values below sqrt(eps) * s_max are noise with this route; the problems here threshold or project them away.
#pragma once
#include "cp_mfnn_small.cuh"

struct NnlCtrl {
    int stop, stop_iter, pad0, pad1;
    double sv[CP_KMAX];     // singular values of the matrix the last k_nnl_spec decomposed
    double fsv[CP_KMAX];    // ... after the spectral function: the singular values of M * W
    double smax_out;        // largest singular value of M * W
};

enum { NNL_SOFT = 0,        // f = max(S - thr, 0)                    NNE/nuclear_norm.m:52
       NNL_BALL = 1,        // f = nu * P_l1(S / nu)                  NNE/nuclear_norm_constraint.m:45
       NNL_CONJ_BALL = 2,   // f = S - mu_ * P_l1(S / mu_)            NNE/nuclear_norm_constraint.m:72-76
       NNL_VALUES = 3 };    // singular values only (W = identity)

// out = ca * a + cb * (useB ? B b : b), B = mean-free operator over the images of a pixel (NNE/mean_free_operator.m:20,
// symmetric); a may be null (0). With part != null also the per-workgroup partial of Gram(out) (slot 0).
__global__ __launch_bounds__(CP_BLOCK) void k_nnl_lin(const NnlCtrl *__restrict__ ctrl, size_t npix, int K,
                                                      const double *a, double ca, const double *b, double cb, int useB,
                                                      double *out, double *__restrict__ part) {
    __shared__ double tiles[CP_WAVES * 16 * CP_TS];
    if (ctrl->stop) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *tile = tiles + wave * 16 * CP_TS;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const double invK = 1.0 / (double)K;
    for (size_t base = (size_t)blockIdx.x * CP_BLOCK; base < npix; base += (size_t)gridDim.x * CP_BLOCK) {
        const size_t q = base + threadIdx.x;
        const bool valid = q < npix;
        double vb[CP_KMAX], va[CP_KMAX];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < CP_KMAX; ++k) {
            const bool on = valid && k < K;
            vb[k] = on ? b[(size_t)k * npix + q] : 0.0;
            va[k] = (on && a) ? a[(size_t)k * npix + q] : 0.0;
            s += vb[k];
        }
        const double mean = useB ? s * invK : 0.0;
#pragma unroll
        for (int k = 0; k < CP_KMAX; ++k) {
            const bool on = valid && k < K;
            const double o = on ? ca * va[k] + cb * (vb[k] - mean) : 0.0;
            if (on) out[(size_t)k * npix + q] = o;
            tile[k * CP_TS + lane] = o;
        }
        if (part) {
            __syncthreads();
            gram_tile_accum(tile, acc);
            __syncthreads();
        }
    }
    if (part) gram_block_store(acc, tiles, part + (size_t)blockIdx.x * CP_PSTRIDE + CP_NSCAL);
}

// in place: conj = 1: v <- v - s * prox_{SAD, 1/s}(v / s)  (NNE/SAD.m:71-72 with the inner call :42-49, mu = 1)
//           conj = 0: v <- prox_{SAD, s}(v)                 (NNE/SAD.m:42-49)
__global__ __launch_bounds__(CP_BLOCK) void k_nnl_sad(const NnlCtrl *__restrict__ ctrl, size_t n, double *v,
                                                      const double *__restrict__ I, double s, int conj) {
    if (ctrl->stop) return;
    const double inv_s = 1.0 / s;
    for (size_t q = (size_t)blockIdx.x * CP_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * CP_BLOCK) {
        const double x = v[q], Iq = I[q];
        const double L = conj ? x * inv_s : x, thr = conj ? inv_s : s;
        const double d = L - Iq;
        const double prox = (d > thr) ? (L - thr) : ((d < -thr) ? (L + thr) : Iq);
        v[q] = conj ? x - s * prox : prox;
    }
}

// out = in * W (in place allowed)
__global__ __launch_bounds__(CP_BLOCK) void k_nnl_applyW(const NnlCtrl *__restrict__ ctrl, size_t npix, int K,
                                                         const double *in, const double *__restrict__ W, double *out) {
    __shared__ double Wl[CP_KMAX * CP_KMAX];
    if (ctrl->stop) return;
    stage_W(W, Wl);
    __syncthreads();
    for (size_t q = (size_t)blockIdx.x * CP_BLOCK + threadIdx.x; q < npix; q += (size_t)gridDim.x * CP_BLOCK) {
        double r[CP_KMAX];
#pragma unroll
        for (int kp = 0; kp < CP_KMAX; ++kp) r[kp] = 0.0;
#pragma unroll
        for (int k = 0; k < CP_KMAX; ++k)
            if (k < K) {
                const double t = in[(size_t)k * npix + q];
#pragma unroll
                for (int kp = 0; kp < CP_KMAX; ++kp) r[kp] = fma(t, Wl[k * CP_KMAX + kp], r[kp]);
            }
#pragma unroll
        for (int kp = 0; kp < CP_KMAX; ++kp)
            if (kp < K) out[(size_t)kp * npix + q] = r[kp];
    }
}

// xbar = xn + theta * (xn - x); x = xn     (PDO/chambolle_pock.m:179)
__global__ __launch_bounds__(CP_BLOCK) void k_nnl_extrap(const NnlCtrl *__restrict__ ctrl, size_t n,
                                                         const double *__restrict__ xn, double *x, double *xbar, double theta) {
    if (ctrl->stop) return;
    for (size_t q = (size_t)blockIdx.x * CP_BLOCK + threadIdx.x; q < n; q += (size_t)gridDim.x * CP_BLOCK) {
        const double a = xn[q], o = x[q];
        xbar[q] = a + theta * (a - o);
        x[q] = a;
    }
}

// monitors of (x, y): sums [0] = sum |x - I|, [1] = <Ky, I>; max [0] = max |Ky| (K = id or B); Gram(Kx) and Gram(y)
__global__ __launch_bounds__(CP_BLOCK) void k_nnl_stats(const NnlCtrl *__restrict__ ctrl, size_t npix, int K, int useB,
                                                        const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ I, double *__restrict__ part) {
    __shared__ double tiles[CP_WAVES * 16 * CP_TS];
    __shared__ double red[CP_WAVES * CP_NSCAL];
    if (ctrl->stop) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *tile = tiles + wave * 16 * CP_TS;
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int i = 0; i < CP_NSUM; ++i) sums[i] = 0.0;
#pragma unroll
    for (int i = 0; i < CP_NMAX; ++i) maxs[i] = 0.0;
    const double invK = 1.0 / (double)K;
    for (size_t base = (size_t)blockIdx.x * CP_BLOCK; base < npix; base += (size_t)gridDim.x * CP_BLOCK) {
        const size_t q = base + threadIdx.x;
        const bool valid = q < npix;
        double vx[CP_KMAX], vy[CP_KMAX];
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int k = 0; k < CP_KMAX; ++k) {
            const bool on = valid && k < K;
            vx[k] = on ? x[(size_t)k * npix + q] : 0.0;
            vy[k] = on ? y[(size_t)k * npix + q] : 0.0;
            sx += vx[k];
            sy += vy[k];
        }
        const double mx = useB ? sx * invK : 0.0, my = useB ? sy * invK : 0.0;
#pragma unroll
        for (int k = 0; k < CP_KMAX; ++k) {
            const bool on = valid && k < K;
            const double Ik = on ? I[(size_t)k * npix + q] : 0.0;
            const double ky = on ? vy[k] - my : 0.0;
            sums[0] += on ? fabs(vx[k] - Ik) : 0.0;
            sums[1] += ky * Ik;
            maxs[0] = fmax(maxs[0], fabs(ky));
            tile[k * CP_TS + lane] = on ? vx[k] - mx : 0.0;
        }
        __syncthreads();
        gram_tile_accum(tile, acc0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CP_KMAX; ++k) tile[k * CP_TS + lane] = vy[k];
        __syncthreads();
        gram_tile_accum(tile, acc1);
        __syncthreads();
    }
    double *row = part + (size_t)blockIdx.x * CP_PSTRIDE;
    block_reduce_store<CP_NSUM, CP_NMAX>(sums, maxs, red, row);
    gram_block_store(acc0, tiles, row + CP_NSCAL);
    gram_block_store(acc1, tiles, row + CP_NSCAL + 256);
}

// Gram partials (slot 0) -> singular values S, spectral function, W = V diag(f(S)/S) V' (CONJ_BALL: I - V diag(mu_ P/S) V').
// One workgroup of CP_RED_THREADS threads.
__global__ __launch_bounds__(CP_RED_THREADS) void k_nnl_spec(NnlCtrl *ctrl, int nblk, int K, int mode, double param,
                                                             const double *__restrict__ part, double *__restrict__ Wout) {
    __shared__ double GG[1][16][17], VV[1][16][17], V0[1][16][17], Tm[1][16][17], sc[CP_NSCAL], dvec[16],
        stage[CP_RED_THREADS], z[16], zs[16], res[2], fs_[16];
    double (*G)[17] = GG[0], (*V)[17] = VV[0];
    const int t = threadIdx.x;
    const int stopped = ctrl->stop;
    reduce_partials(part, nblk, G, sc, stage);
    if (stopped || t >= 256) return;
    eig16_warm<1>(GG, VV, V0, Tm, nullptr, 0);
    if (mode == NNL_SOFT || mode == NNL_VALUES) {
        if (t < 16) {
            const double S = (t < K) ? sqrt(fmax(G[t][t], 0.0)) : 0.0;
            const double f = (mode == NNL_SOFT) ? fmax(S - param, 0.0) : S;
            dvec[t] = (S > 0.0) ? f / S : 0.0;
            fs_[t] = f;
            zs[t] = S;
        }
    } else if (t < 64) {
        l1ball_scale16(G, K, param, z, zs, dvec, res, ctrl->sv);   // dvec = mu_ P / S, zs = S, z = S - mu_ P
        if (t < 16) fs_[t] = (mode == NNL_BALL) ? dvec[t] * zs[t] : z[t];
    }
    __syncthreads();
    if (t < 16) { ctrl->sv[t] = zs[t]; ctrl->fsv[t] = fs_[t]; }
    if (t == 0) {
        double mx = 0.0;
        for (int i = 0; i < 16; ++i) mx = fmax(mx, fs_[i]);
        ctrl->smax_out = mx;
    }
    {
        const int k = t >> 4, kp = t & 15;
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += V[k][i] * dvec[i] * V[kp][i];
        if (mode == NNL_CONJ_BALL) a = ((k == kp) ? 1.0 : 0.0) - a;
        if (mode == NNL_VALUES) a = (k == kp) ? 1.0 : 0.0;
        Wout[k * CP_KMAX + kp] = a;
    }
}

struct NnlHistArgs {
    int nblk, K, version, row, use_fsv, check_stop;
    double param, tol;
};
#define CP_HCOLS_NNL 5   // [p, F, G, F_con, G_con] / [q, F*, G*, F*_con, G*_con]   PDO/chambolle_pock.m:225-228

// history row `row` from the monitors of k_nnl_stats (+ the singular values k_nnl_spec left in ctrl when use_fsv), stop rule
__global__ __launch_bounds__(CP_RED_THREADS) void k_nnl_hist(NnlHistArgs A, NnlCtrl *ctrl, const double *__restrict__ part,
                                                             double *hist_p, double *hist_d) {
    __shared__ double GG[2][16][17], VV[2][16][17], V0[2][16][17], Tm[2][16][17], sc[CP_NSCAL], stage[CP_RED_THREADS],
        stage2[CP_RED_THREADS];
    const int t = threadIdx.x;
    const int stopped = ctrl->stop;
    reduce_partials2(part, A.nblk, GG[0], GG[1], sc, stage, stage2);
    if (stopped || t >= 512) return;
    int Kx = A.K;
    if (A.version == 3) {   // Gram(B x): exactly mean-free over the images, restrict to the complement of (1,..,1)
        if (t < 256) helmert_rows(GG[0], Tm[0], A.K, t >> 4, t & 15);
        __syncthreads();
        if (t < 256) helmert_cols(Tm[0], GG[0], A.K, t >> 4, t & 15);
        __syncthreads();
        Kx = A.K - 1;
    }
    eig16_warm<2>(GG, VV, V0, Tm, nullptr, 0);
    if (t == 0) {
        double sumSx = 0.0, maxSy = 0.0;
        // versions 1, 2: x = L~ * W, so its singular values are the f(S) k_nnl_spec left in ctrl (a fresh Gram would turn
        // the thresholded ones into sqrt(eps) * s_max); version 3: y = y~ * W likewise
        const bool fx = A.use_fsv && A.version != 3, fy = A.use_fsv && A.version == 3;
        for (int i = 0; i < Kx; ++i) sumSx += fx ? ctrl->fsv[i] : sqrt(fmax(GG[0][i][i], 0.0));
        if (fy) maxSy = ctrl->smax_out;
        else for (int i = 0; i < A.K; ++i) maxSy = fmax(maxSy, sqrt(fmax(GG[1][i][i], 0.0)));
        const double sad = sc[0], yI = sc[1], maxKy = sc[CP_NSUM];
        double F, G, Fc = 0.0, Gc = 0.0, Fs, Gs, Fsc = 0.0, Gsc = 0.0;
        if (A.version == 1) {
            F = sad; G = A.param * sumSx;                                         // NNE/SAD.m:54, nuclear_norm.m:47
            Fs = yI; Fsc = (maxKy > 1.0) ? maxKy - 1.0 : 0.0;                     // NNE/SAD.m:77-83 (mu = 1)
            Gs = 0.0;                                                             // nuclear_norm.m:71
            const double ms = maxSy / A.param;                                    // svd(-y / mu) (:63-67)
            Gsc = (ms > 1.0) ? ms - 1.0 : 0.0;                                    // (:72-74)
        } else if (A.version == 2) {
            F = sad; G = 0.0; Gc = (sumSx > A.param) ? (sumSx - A.param) / A.param : 0.0;   // nuclear_norm_constraint.m:55-62
            Fs = yI; Fsc = (maxKy > 1.0) ? maxKy - 1.0 : 0.0;
            Gs = A.param * maxSy;                                                 // (:89)
        } else {
            F = 0.0; Fc = (sumSx > A.param) ? (sumSx - A.param) / A.param : 0.0;
            G = sad;
            Fs = A.param * maxSy;
            Gs = -yI; Gsc = (maxKy > 1.0) ? maxKy - 1.0 : 0.0;                    // SAD*(-B'y): <-By, I>, max|By|
        }
        double *hp = hist_p + (size_t)A.row * CP_HCOLS_NNL, *hd = hist_d + (size_t)A.row * CP_HCOLS_NNL;
        hp[0] = G + F; hp[1] = F; hp[2] = G; hp[3] = Fc; hp[4] = Gc;              // chambolle_pock.m:91,170-171
        hd[0] = -(Gs + Fs); hd[1] = Fs; hd[2] = Gs; hd[3] = Fsc; hd[4] = Gsc;     // (:101-102,175-176)
        if (A.check_stop && A.row >= 4) {                                         // (:142-153)
            bool ok = true;
            for (int r = A.row - 4; r <= A.row; ++r) {
                const double *p_ = hist_p + (size_t)r * CP_HCOLS_NNL, *d_ = hist_d + (size_t)r * CP_HCOLS_NNL;
                const double gap = fabs((p_[0] - d_[0]) / d_[0]);
                if (!(gap <= A.tol) || !(p_[3] <= A.tol) || !(p_[4] <= A.tol) || !(d_[3] <= A.tol) || !(d_[4] <= A.tol))
                    ok = false;
            }
            if (ok) { ctrl->stop = 1; ctrl->stop_iter = A.row; }
        }
    }
}
