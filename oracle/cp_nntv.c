/* TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C + OpenMP) of the NN-TV and NN-curvature hot paths of roland1993/MA.
 * PARITY UNPINNED: the reference is MATLAB source, cannot be run in this image and stores no golden vectors; this file
 * is checked against the numpy restatement (the oracle .py files, themselves pinned only by the reference's invariants) in
 * tests/test_oracle_c.py. Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load it.
 *
 * What it follows (matrix-free, same arithmetic per element as the .m files):
 *   PDO/chambolle_pock.m:134-193            iteration, histories [p,F1,F2,F3,G,F_con,G_con], stop rule
 *   NNE/mf_nn_tv_registration_no_ref_ml.m:183-188  A = [-grad T, I; D, 0; 0, B];  :235-346 local F, G
 *   NNE/finite_difference_operator.m:29-50  forward differences (neumann | linear last row, unscaled)
 *   NNE/SAD.m:42-83, NNE/pointwise_2x2_nn.m:12-57 + NNE/svd_2x2.m:10-47 (atan2/cos/sin form),
 *   NNE/nuclear_norm_constraint.m:34-94 + NNE/l1ball_projection.m:23-68, NNE/mean_zero_indicator.m:33-84,
 *   NNE/zero_function.m:26-34, NNE/mean_free_operator.m:20
 * svd(mn x K, 'econ') (MATLAB builtin) is replaced by eig of the K x K Gram matrix (cyclic Jacobi); B*L is exactly
 * mean-free over the images, its Gram matrix is deflated by (1,..,1) first (Helmert basis).
 *
 * Vector layout = the reference's: x = [u_1x; u_1y; ...; u_Kx; u_Ky; L_1; ...; L_K], each block mn column-major;
 * y = [y1 (K mn); per image (Dx ux; Dy ux; Dx uy; Dy uy) (4 K mn); y3 (K mn)]; g = per image (gx; gy); b (K mn).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KMAX 16

typedef struct {
    int m, n, K, linear;
    double ih1, ih2, mu1, mu2, nu;
    const double *g, *b;
    int curv, neumann, nc;   /* curv: NNE/mf_nn_curvature_registration_no_ref_ml.m (A2 = discrete_laplacian, F2 = SSD(y2, 0));
                                neumann: Laplacian boundary rows (else dirichlet); nc = blocks of y2 per image (4 | 2) */
} prob_t;

int nntv_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- K x K symmetric eigenproblem: cyclic Jacobi, eigenvalues on the diagonal, V columns = eigenvectors ---- */
static void jacobi_eig(double G[KMAX][KMAX], double V[KMAX][KMAX], int K) {
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) V[i][j] = (i == j);
    for (int i = 0; i < K; ++i)
        for (int j = i + 1; j < K; ++j) G[i][j] = G[j][i] = 0.5 * (G[i][j] + G[j][i]);
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rot = 0;
        for (int p = 0; p < K - 1; ++p)
            for (int q = p + 1; q < K; ++q) {
                const double apq = G[p][q], app = G[p][p], aqq = G[q][q];
                if (!(fabs(apq) > 1e-17 * sqrt(fabs(app * aqq)) && fabs(apq) > 1e-300)) continue;
                rot = 1;
                const double th = (aqq - app) / (2.0 * apq);
                const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int l = 0; l < K; ++l) {
                    const double gp = G[l][p], gq = G[l][q];
                    G[l][p] = c * gp - s * gq; G[l][q] = s * gp + c * gq;
                    const double vp = V[l][p], vq = V[l][q];
                    V[l][p] = c * vp - s * vq; V[l][q] = s * vp + c * vq;
                }
                for (int l = 0; l < K; ++l) {
                    const double gp = G[p][l], gq = G[q][l];
                    G[p][l] = c * gp - s * gq; G[q][l] = s * gp + c * gq;
                }
            }
        if (!rot) break;
    }
}

/* Gram matrix of the mn x K matrix whose columns are the K blocks of Y (optionally of B*Y: mean over k removed) */
static void gram(const double *Y, long mn, int K, int centre, double G[KMAX][KMAX]) {
    double acc[KMAX * KMAX];
    memset(acc, 0, sizeof acc);
#pragma omp parallel
    {
        double loc[KMAX * KMAX];
        memset(loc, 0, sizeof loc);
#pragma omp for schedule(static)
        for (long q = 0; q < mn; ++q) {
            double v[KMAX], s = 0.0;
            for (int k = 0; k < K; ++k) { v[k] = Y[(long)k * mn + q]; s += v[k]; }
            if (centre) { const double mean = s / K; for (int k = 0; k < K; ++k) v[k] -= mean; }
            for (int a = 0; a < K; ++a)
                for (int c = a; c < K; ++c) loc[a * KMAX + c] += v[a] * v[c];
        }
#pragma omp critical
        for (int e = 0; e < KMAX * KMAX; ++e) acc[e] += loc[e];
    }
    for (int a = 0; a < K; ++a)
        for (int c = a; c < K; ++c) G[a][c] = G[c][a] = acc[a * KMAX + c];
}

/* singular values (descending not guaranteed) and right singular vectors of the mn x K matrix Y */
static void sv_of(const double *Y, long mn, int K, double S[KMAX], double V[KMAX][KMAX]) {
    double G[KMAX][KMAX];
    gram(Y, mn, K, 0, G);
    jacobi_eig(G, V, K);
    for (int i = 0; i < K; ++i) S[i] = sqrt(fmax(G[i][i], 0.0));
}

/* sum of the singular values of B*L (exactly mean-free over k: deflate (1,..,1) before the eigensolve) */
static double nuc_of_BL(const double *L, long mn, int K) {
    double G[KMAX][KMAX], H[KMAX][KMAX], V[KMAX][KMAX];
    gram(L, mn, K, 1, G);
    memset(H, 0, sizeof H);
    for (int a = 0; a < K - 1; ++a)
        for (int c = 0; c < K - 1; ++c) {
            const double na = 1.0 / sqrt((double)(a + 1) * (a + 2)), nc = 1.0 / sqrt((double)(c + 1) * (c + 2));
            double acc = 0.0;
            for (int i = 0; i <= a + 1; ++i) {
                const double qa = (i <= a) ? na : -(double)(a + 1) * na;
                double row = 0.0;
                for (int j = 0; j <= c + 1; ++j) row += G[i][j] * ((j <= c) ? nc : -(double)(c + 1) * nc);
                acc += qa * row;
            }
            H[a][c] = acc;
        }
    jacobi_eig(H, V, K - 1);
    double s = 0.0;
    for (int i = 0; i < K - 1; ++i) s += sqrt(fmax(H[i][i], 0.0));
    return s;
}

/* NNE/l1ball_projection.m:23-68 for non-negative input (singular values); sort form of the same threshold */
static void l1ball(const double *z, int K, double *out) {
    double s = 0.0, zs[KMAX];
    for (int i = 0; i < K; ++i) { s += z[i]; zs[i] = z[i]; }
    if (s <= 1.0) { for (int i = 0; i < K; ++i) out[i] = z[i]; return; }
    for (int i = 1; i < K; ++i) {
        const double v = zs[i];
        int j = i - 1;
        while (j >= 0 && zs[j] < v) { zs[j + 1] = zs[j]; --j; }
        zs[j + 1] = v;
    }
    double cum = 0.0, theta = 0.0;
    for (int i = 0; i < K; ++i) {
        cum += zs[i];
        const double th = (cum - 1.0) / (double)(i + 1);
        if (zs[i] - th > 0.0) theta = th;
    }
    for (int i = 0; i < K; ++i) out[i] = fmax(z[i] - theta, 0.0);
}

static inline double sgn(double v) { return (v > 0.0) - (v < 0.0); }

/* NNE/svd_2x2.m:10-47 on (a, c, b, d) = (m11, m21, m12, m22) */
static inline void svd2(double a, double c, double b, double d, double *sig, double *U, double *V) {
    const double s1 = a * a + b * b + c * c + d * d;
    const double xt = a * a + b * b - c * c - d * d;
    const double s2 = sqrt(xt * xt + 4.0 * (a * c + b * d) * (a * c + b * d));
    sig[0] = sqrt(fmax((s1 + s2) / 2.0, 0.0));
    sig[1] = sqrt(fmax((s1 - s2) / 2.0, 0.0));
    if (!U) return;
    const double th = atan2(2.0 * a * c + 2.0 * b * d, xt) / 2.0;
    const double ct = cos(th), st = sin(th);
    const double ph = atan2(2.0 * a * b + 2.0 * c * d, a * a - b * b + c * c - d * d) / 2.0;
    const double cp = cos(ph), sp = sin(ph);
    const double s11 = (a * ct + c * st) * cp + (b * ct + d * st) * sp;
    const double s22 = (a * st - c * ct) * sp + (-b * st + d * ct) * cp;
    const double g1 = sgn(s11), g2 = sgn(s22);
    U[0] = ct; U[1] = st; U[2] = -st; U[3] = ct;
    V[0] = g1 * cp; V[1] = g1 * sp; V[2] = -g2 * sp; V[3] = g2 * cp;
}

/* forward difference and its adjoint along one axis (NNE/finite_difference_operator.m:29-36) */
static inline double fd(const double *u, long idx, long stride, int pos, int len, int linear, double ih) {
    if (pos < len - 1) return (u[idx + stride] - u[idx]) * ih;
    return linear ? (u[idx] - u[idx - stride]) : 0.0;
}
static inline double fd_adj(const double *w, long idx, long stride, int pos, int len, int linear, double ih) {
    double r = 0.0;
    if (pos >= 1) r += w[idx - stride] * ih;
    if (pos <= len - 2) r -= w[idx] * ih;
    if (linear) {
        const long last = idx + (long)(len - 1 - pos) * stride;
        if (pos == len - 2) r -= w[last];
        if (pos == len - 1) r += w[last];
    }
    return r;
}

/* 5-point Laplacian of one field at pixel q = (i, j): a missing neighbour is the centre value (neumann, diagonal -1/h^2)
 * or zero (dirichlet, diagonal -2/h^2)                                        NNE/discrete_laplacian.m:23-46 */
static inline double lap5(const double *u, long q, int i, int j, int m, int n, int neumann, double ih1, double ih2) {
    const double w0 = u[q], mis = neumann ? w0 : 0.0;
    const double a0 = (i > 0) ? u[q - 1] : mis, a1 = (i < m - 1) ? u[q + 1] : mis;
    const double a2 = (j > 0) ? u[q - m] : mis, a3 = (j < n - 1) ? u[q + m] : mis;
    return (a0 + a1 - 2.0 * w0) * (ih1 * ih1) + (a2 + a3 - 2.0 * w0) * (ih2 * ih2);
}

/* y = A*x */
static void apply_A(const prob_t *P, const double *x, double *y) {
    const int m = P->m, n = P->n, K = P->K;
    const long mn = (long)m * n, kmn = mn * K;
    const double *L = x + 2 * kmn;
#pragma omp parallel for schedule(static)
    for (long q = 0; q < mn; ++q) {
        const int j = (int)(q / m), i = (int)(q - (long)j * m);
        double sL = 0.0;
        for (int k = 0; k < K; ++k) sL += L[(long)k * mn + q];
        const double mean = sL / K;
        for (int k = 0; k < K; ++k) {
            const double *ux = x + (long)(2 * k) * mn, *uy = ux + mn;
            const double *gx = P->g + (long)(2 * k) * mn, *gy = gx + mn;
            y[(long)k * mn + q] = -(gx[q] * ux[q] + gy[q] * uy[q]) + L[(long)k * mn + q];
            double *y2 = y + kmn + (long)(P->nc * k) * mn;
            if (P->curv) {
                y2[q] = lap5(ux, q, i, j, m, n, P->neumann, P->ih1, P->ih2);
                y2[mn + q] = lap5(uy, q, i, j, m, n, P->neumann, P->ih1, P->ih2);
            } else {
                y2[q] = fd(ux, q, 1, i, m, P->linear, P->ih1);
                y2[mn + q] = fd(ux, q, m, j, n, P->linear, P->ih2);
                y2[2 * mn + q] = fd(uy, q, 1, i, m, P->linear, P->ih1);
                y2[3 * mn + q] = fd(uy, q, m, j, n, P->linear, P->ih2);
            }
            y[(1 + P->nc) * kmn + (long)k * mn + q] = L[(long)k * mn + q] - mean;
        }
    }
}

/* x = A'*y */
static void apply_At(const prob_t *P, const double *y, double *x) {
    const int m = P->m, n = P->n, K = P->K;
    const long mn = (long)m * n, kmn = mn * K;
#pragma omp parallel for schedule(static)
    for (long q = 0; q < mn; ++q) {
        const int j = (int)(q / m), i = (int)(q - (long)j * m);
        double s3 = 0.0;
        const long o3 = (1 + P->nc) * kmn;
        for (int k = 0; k < K; ++k) s3 += y[o3 + (long)k * mn + q];
        const double mean3 = s3 / K;
        for (int k = 0; k < K; ++k) {
            const double y1 = y[(long)k * mn + q];
            const double *y2 = y + kmn + (long)(P->nc * k) * mn;
            const double *gx = P->g + (long)(2 * k) * mn, *gy = gx + mn;
            if (P->curv) {   /* the Laplacian is symmetric */
                x[(long)(2 * k) * mn + q] = -gx[q] * y1 + lap5(y2, q, i, j, m, n, P->neumann, P->ih1, P->ih2);
                x[(long)(2 * k + 1) * mn + q] = -gy[q] * y1 + lap5(y2 + mn, q, i, j, m, n, P->neumann, P->ih1, P->ih2);
            } else {
                x[(long)(2 * k) * mn + q] = -gx[q] * y1 + (fd_adj(y2, q, 1, i, m, P->linear, P->ih1) +
                                                           fd_adj(y2 + mn, q, m, j, n, P->linear, P->ih2));
                x[(long)(2 * k + 1) * mn + q] = -gy[q] * y1 + (fd_adj(y2 + 2 * mn, q, 1, i, m, P->linear, P->ih1) +
                                                               fd_adj(y2 + 3 * mn, q, m, j, n, P->linear, P->ih2));
            }
            x[2 * kmn + (long)k * mn + q] = y1 + (y[o3 + (long)k * mn + q] - mean3);
        }
    }
}

/* primal row: F(Ax) values [F1,F2,F3], G value, violations; y = A*x given            chambolle_pock.m:233-237 */
static void primal_row(const prob_t *P, const double *x, const double *y, double *row) {
    const long mn = (long)P->m * P->n, kmn = mn * P->K;
    double f1 = 0.0, f2 = 0.0, sx = 0.0, sy = 0.0;
#pragma omp parallel for reduction(+ : f1, f2, sx, sy) schedule(static)
    for (long e = 0; e < kmn; ++e) {
        f1 += fabs(y[e] - P->b[e]);                                  /* NNE/SAD.m:54 */
        const long k = e / mn, q = e - k * mn;
        const double *y2 = y + kmn + P->nc * k * mn;
        if (P->curv) {
            f2 += y2[q] * y2[q] + y2[mn + q] * y2[mn + q];           /* NNE/SSD.m:48 (g = 0) */
        } else {
            double sig[2];
            svd2(y2[q], y2[mn + q], y2[2 * mn + q], y2[3 * mn + q], sig, NULL, NULL);
            f2 += sig[0] + sig[1];                                   /* NNE/pointwise_2x2_nn.m:32 */
        }
        sx += x[2 * k * mn + q];
        sy += x[(2 * k + 1) * mn + q];
    }
    const double F1 = P->mu1 * f1, F2 = P->curv ? P->mu2 * 0.5 * f2 : P->mu2 * f2;
    const double sS = nuc_of_BL(y + (1 + P->nc) * kmn, mn, P->K);
    row[1] = F1; row[2] = F2; row[3] = 0.0; row[4] = 0.0;
    row[0] = row[4] + (F1 + F2 + 0.0);                               /* chambolle_pock.m:170-171 */
    row[5] = (sS > P->nu) ? (sS - P->nu) / P->nu : 0.0;              /* nuclear_norm_constraint.m:55-59 */
    row[6] = fmax(fabs(sx / (double)kmn), fabs(sy / (double)kmn));   /* mean_zero_indicator.m:65 */
}

/* dual row: F*(y) values, G*(-A'y), violations; w = A'*y given                         chambolle_pock.m:239-243 */
static void dual_row(const prob_t *P, const double *y, const double *w, double *row) {
    const long mn = (long)P->m * P->n, kmn = mn * P->K;
    double f1s = 0.0, f2s = 0.0, y1max = 0.0, s1max = 0.0, swx = 0.0;
#pragma omp parallel for reduction(+ : f1s, f2s, swx) reduction(max : y1max, s1max) schedule(static)
    for (long e = 0; e < kmn; ++e) {
        f1s += y[e] * P->b[e];                                       /* NNE/SAD.m:83 */
        y1max = fmax(y1max, fabs(y[e]));
        const long k = e / mn, q = e - k * mn;
        const double *y2 = y + kmn + P->nc * k * mn;
        if (P->curv) {
            const double z0 = y2[q] / P->mu2, z1 = y2[mn + q] / P->mu2;
            f2s += z0 * z0 + z1 * z1;                                /* NNE/SSD.m:66-67 (g = 0) */
        } else {
            double sig[2];
            svd2(y2[q], y2[mn + q], y2[2 * mn + q], y2[3 * mn + q], sig, NULL, NULL);
            s1max = fmax(s1max, sig[0]);
        }
        swx += -w[2 * k * mn + q];
    }
    double S[KMAX], V[KMAX][KMAX], smax = 0.0;
    sv_of(y + (1 + P->nc) * kmn, mn, P->K, S, V);
    for (int i = 0; i < P->K; ++i) smax = fmax(smax, S[i]);
    const double mwx = swx / (double)kmn;                            /* x-mean for both, mean_zero_indicator.m:78,81 */
    double dmax = 0.0, lmax = 0.0;
#pragma omp parallel for reduction(max : dmax, lmax) schedule(static)
    for (long e = 0; e < kmn; ++e) {
        const long k = e / mn, q = e - k * mn;
        dmax = fmax(dmax, fmax(fabs(-w[2 * k * mn + q] - mwx), fabs(-w[(2 * k + 1) * mn + q] - mwx)));
        lmax = fmax(lmax, fabs(w[2 * kmn + e]));                     /* zero_function.m:32 */
    }
    row[1] = f1s; row[2] = P->curv ? P->mu2 * (0.5 * f2s) : 0.0; row[3] = P->nu * smax;   /* nuclear_norm_constraint.m:89 */
    row[4] = 0.0;
    const double con1 = (y1max > P->mu1) ? y1max - P->mu1 : 0.0;     /* SAD.m:77-81 */
    const double con2 = P->curv ? 0.0 : fmax((s1max - P->mu2) / P->mu2, 0.0);   /* pointwise_2x2_nn.m:56 | SSD: none */
    row[5] = fmax(con1, con2);
    row[6] = fmax(dmax, lmax);
    row[0] = -(row[4] + (row[1] + row[2] + row[3]));                 /* chambolle_pock.m:175-176 */
}

/* y <- prox_{sigma F*}(y)  (in place; y holds y_old + sigma*A*xbar)                  mf_nn_tv...:251-276 */
static void prox_Fstar(const prob_t *P, double *y, double sigma) {
    const long mn = (long)P->m * P->n, kmn = mn * P->K;
    const int K = P->K;
#pragma omp parallel for schedule(static)
    for (long e = 0; e < kmn; ++e) {
        /* SAD*: L - sigma * prox_{SAD,1/sigma}(L/sigma)   NNE/SAD.m:71-72 with :42-49 */
        const double Ls = y[e] / sigma, thr = (1.0 / sigma) * P->mu1, dd = Ls - P->b[e];
        const double prox = (dd > thr) ? (Ls - thr) : ((dd < -thr) ? (Ls + thr) : P->b[e]);
        y[e] = y[e] - sigma * prox;
        const long k = e / mn, q = e - k * mn;
        double *y2 = y + kmn + P->nc * k * mn;
        if (P->curv) {   /* SSD*, g = 0: x - tau*((x/tau)/(1 + lambda/tau))              NNE/SSD.m:60-61 with :43 */
            const double den = 1.0 + P->mu2 * (1.0 / sigma);
            y2[q] = y2[q] - sigma * ((y2[q] * (1.0 / sigma)) / den);
            y2[mn + q] = y2[mn + q] - sigma * ((y2[mn + q] * (1.0 / sigma)) / den);
            continue;
        }
        /* pointwise 2x2 nuclear norm*: M - sigma*SVT_{mu/sigma}(M/sigma)   pointwise_2x2_nn.m:46-48, :16-24 */
        const double a = y2[q] / sigma, c = y2[mn + q] / sigma, b = y2[2 * mn + q] / sigma, d = y2[3 * mn + q] / sigma;
        double sig[2], U[4], V[4];
        svd2(a, c, b, d, sig, U, V);
        const double t1 = fmax(sig[0] - P->mu2 * (1.0 / sigma), 0.0), t2 = fmax(sig[1] - P->mu2 * (1.0 / sigma), 0.0);
        const double r11 = t1 * U[0] * V[0] + t2 * U[2] * V[2], r21 = t1 * U[1] * V[0] + t2 * U[3] * V[2];
        const double r12 = t1 * U[0] * V[1] + t2 * U[2] * V[3], r22 = t1 * U[1] * V[1] + t2 * U[3] * V[3];
        y2[q] -= sigma * r11; y2[mn + q] -= sigma * r21; y2[2 * mn + q] -= sigma * r12; y2[3 * mn + q] -= sigma * r22;
    }
    /* nuclear-norm ball*: U diag(S - mu*P_l1(S/mu)) V' = Y (I - V diag(d) V'), mu = nu*sigma   :72-79 */
    double *y3 = y + (1 + P->nc) * kmn, S[KMAX], V[KMAX][KMAX], z[KMAX], Pz[KMAX], W[KMAX][KMAX];
    sv_of(y3, mn, K, S, V);
    const double mu_ = P->nu * sigma;
    for (int i = 0; i < K; ++i) z[i] = S[i] / mu_;
    l1ball(z, K, Pz);
    for (int a = 0; a < K; ++a)
        for (int c = 0; c < K; ++c) {
            double acc = (a == c) ? 1.0 : 0.0;
            for (int i = 0; i < K; ++i) acc -= V[a][i] * ((S[i] > 0.0) ? (mu_ * Pz[i]) / S[i] : 0.0) * V[c][i];
            W[a][c] = acc;
        }
#pragma omp parallel for schedule(static)
    for (long q = 0; q < mn; ++q) {
        double v[KMAX], o[KMAX];
        for (int k = 0; k < K; ++k) v[k] = y3[(long)k * mn + q];
        for (int c = 0; c < K; ++c) { double acc = 0.0; for (int k = 0; k < K; ++k) acc += v[k] * W[k][c]; o[c] = acc; }
        for (int k = 0; k < K; ++k) y3[(long)k * mn + q] = o[k];
    }
}

/* Runs at most max_iter iterations from (x, y); histories are (max_iter+1) x 7 row-major; returns the number of rows
 * filled (iterations + 1). tol < 0 disables the stop rule. */
static int nn_chambolle_pock(int curv, int bc, int m, int n, int K, double h1, double h2, double mu, double nu, double theta,
                            double tau, double sigma, int max_iter, double tol, const double *g, const double *b,
                            double *x, double *y, double *hist_p, double *hist_d) {
    if (K < 2 || K > KMAX) return -1;
    prob_t P = {m, n, K, bc == 1, 1.0 / h1, 1.0 / h2, h1 * h2, mu * (h1 * h2), nu, g, b, curv, bc == 0, curv ? 2 : 4};
    const long mn = (long)m * n, kmn = mn * K, nx = 3 * kmn, ny = (2 + P.nc) * kmn;
    double *xbar = malloc(sizeof(double) * nx), *xold = malloc(sizeof(double) * nx), *w = malloc(sizeof(double) * nx);
    double *t = malloc(sizeof(double) * ny);
    if (!xbar || !xold || !w || !t) return -2;
    memcpy(xbar, x, sizeof(double) * nx);
    apply_A(&P, x, t);
    primal_row(&P, x, t, hist_p);
    apply_At(&P, y, w);
    dual_row(&P, y, w, hist_d);
    int i = 0;
    for (;;) {
        if (i == max_iter) break;                                                   /* chambolle_pock.m:137 */
        if (tol >= 0.0 && i >= 4) {                                                 /* :142-153 */
            int ok = 1;
            for (int r = i - 4; r <= i; ++r) {
                const double *p_ = hist_p + 7 * r, *d_ = hist_d + 7 * r;
                const double gap = fabs((p_[0] - d_[0]) / d_[0]);
                if (!(gap <= tol) || !(p_[5] <= tol) || !(p_[6] <= tol) || !(d_[5] <= tol) || !(d_[6] <= tol)) ok = 0;
            }
            if (ok) break;
        }
        ++i;
        memcpy(xold, x, sizeof(double) * nx);
        apply_A(&P, xbar, t);
#pragma omp parallel for schedule(static)
        for (long e = 0; e < ny; ++e) y[e] = y[e] + sigma * t[e];
        prox_Fstar(&P, y, sigma);                                                   /* :162 */
        apply_At(&P, y, w);
        double sx = 0.0, sy = 0.0;
#pragma omp parallel for reduction(+ : sx, sy) schedule(static)
        for (long e = 0; e < kmn; ++e) {
            const long k = e / mn, q = e - k * mn;
            x[2 * k * mn + q] -= tau * w[2 * k * mn + q];
            x[(2 * k + 1) * mn + q] -= tau * w[(2 * k + 1) * mn + q];
            x[2 * kmn + e] -= tau * w[2 * kmn + e];                                 /* zero_function prox = identity */
            sx += x[2 * k * mn + q];
            sy += x[(2 * k + 1) * mn + q];
        }
        const double mx = sx / (double)kmn, my = sy / (double)kmn;                  /* mean_zero_indicator.m:50,53 */
#pragma omp parallel for schedule(static)
        for (long e = 0; e < kmn; ++e) {
            const long k = e / mn, q = e - k * mn;
            x[2 * k * mn + q] -= mx;
            x[(2 * k + 1) * mn + q] -= my;
        }
        apply_A(&P, x, t);
        primal_row(&P, x, t, hist_p + 7 * i);
        dual_row(&P, y, w, hist_d + 7 * i);
#pragma omp parallel for schedule(static)
        for (long e = 0; e < nx; ++e) xbar[e] = x[e] + theta * (x[e] - xold[e]);    /* :179 */
    }
    free(xbar); free(xold); free(w); free(t);
    return i + 1;
}

/* NN-TV: bc 0 = neumann, 1 = linear (kept signature: linear flag) */
int nntv_chambolle_pock(int m, int n, int K, int linear, double h1, double h2, double mu, double nu, double theta,
                        double tau, double sigma, int max_iter, double tol, const double *g, const double *b,
                        double *x, double *y, double *hist_p, double *hist_d) {
    return nn_chambolle_pock(0, linear ? 1 : 0, m, n, K, h1, h2, mu, nu, theta, tau, sigma, max_iter, tol, g, b, x, y, hist_p, hist_d);
}
/* NN-curvature (NNE/mf_nn_curvature_registration_no_ref_ml.m:182-202,230-332): y = [y1 (K mn); per image (Lap ux; Lap uy)
 * (2 K mn); y3 (K mn)]; bc 0 = neumann, 2 = dirichlet (NNE/discrete_laplacian.m:25-28) */
int nncurv_chambolle_pock(int m, int n, int K, int bc, double h1, double h2, double mu, double nu, double theta,
                          double tau, double sigma, int max_iter, double tol, const double *g, const double *b,
                          double *x, double *y, double *hist_p, double *hist_d) {
    if (bc != 0 && bc != 2) return -3;
    return nn_chambolle_pock(1, bc, m, n, K, h1, h2, mu, nu, theta, tau, sigma, max_iter, tol, g, b, x, y, hist_p, hist_d);
}
