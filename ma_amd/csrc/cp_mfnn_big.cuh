// mf_nn models with 17 .. CP_KBIG (32) images: the single-workgroup kernels of the general path.
//
// The reference has no limit on the number of images (NNE/nuclear_norm_constraint.m:34-37 reshapes to any numImg; the
// kidney demo takes k = size(data, 3), NNE/demo_mf_nn_curvature_registration_no_ref_ml.m:43-79). Up to 16 images everything
// K x K lives on one 16 x 16 MFMA tile and one wavefront (cp_mfnn_small.cuh, cp_mfnn_fused.cuh); beyond that the iteration
// runs on the split passes k_mf_dual / k_mf_primal instantiated for KM = 32 (2 x 2 MFMA tiles per Gram matrix, W is
// 32 x 32) and on the kernels of this file, which do for 32 x 32 matrices what k_mf_eig / k_mf_fin / k_mf_sv_eig do for
// 16 x 16 ones -- written for clarity, one workgroup of 1024 threads = one thread per matrix entry, workgroup barriers
// between the phases of a Jacobi round. Same formulas, same history columns, same stop rule (the tails are those of
// k_mf_eig / k_mf_fin line by line).
#pragma once
#include "cp_mfnn_small.cuh"

#define CP_BIG_THREADS 1024
typedef double (*BigMat)[CP_KBIG + 1];

// fixed-order reduction of the partial rows written by the KM = 32 field kernels: Gram entry e by thread e, the scalars by
// the first CP_NSCAL threads. which: 0 -> the Gram block at +CP_NSCAL.
__device__ __forceinline__ void big_reduce(const double *__restrict__ part, int nblk, BigMat G, double *sc) {
    constexpr int PS = cp_pstride<CP_KBIG>();
    const int t = threadIdx.x;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const double *src = part + CP_NSCAL + t;
    int bI = 0;
    for (; bI + 3 < nblk; bI += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += src[(size_t)(bI + u) * PS];
    }
    for (int u = 0; bI < nblk; ++bI, ++u) a[u] += src[(size_t)bI * PS];
    G[t >> 5][t & 31] = (a[0] + a[1]) + (a[2] + a[3]);
    if (t < CP_NSCAL) {
        const bool is_sum = t < CP_NSUM;
        double v = is_sum ? 0.0 : -1.0e300;
        for (int bJ = 0; bJ < nblk; ++bJ) {
            const double o = part[(size_t)bJ * PS + t];
            v = is_sum ? v + o : fmax(v, o);
        }
        sc[t] = v;
    }
    __syncthreads();
}

// Cyclic Jacobi on a symmetric 32 x 32 matrix in LDS, 1024 threads, round-robin ordering: 31 rounds of 16 disjoint
// rotations per sweep (jacobi16 of cp_mfnn_small.cuh with N = 32). On exit diag(G) holds the eigenvalues and the columns
// of V the eigenvectors: G_in = V diag V'. V must hold the identity (or the basis to be rotated) on entry.
// Kp: the tournament runs over the even number Kp >= K of players -- rows and columns K..31 of a zero-padded Gram matrix are zero,
// rotations against them are identities (as the 16 x 16 kernels do): K = 20 plays 19 rounds per sweep instead of 31.
__device__ void big_jacobi(BigMat G, BigMat V, double (*cs)[2], int (*pq)[2], int *rotflag, int Kp = CP_KBIG) {
    const int N = Kp;
    const int t = threadIdx.x, r_ = t >> 5, c_ = t & 31;
    {
        const double s = 0.5 * (G[r_][c_] + G[c_][r_]);
        __syncthreads();
        G[r_][c_] = s;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (t == 0) *rotflag = 0;
        __syncthreads();
        for (int r = 0; r < N - 1; ++r) {
            if (t < N / 2) {
                int p, q;
                if (t == 0) { p = N - 1; q = r; }
                else { p = (r + t) % (N - 1); q = (r - t + (N - 1)) % (N - 1); }
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                double c, s;
                jacobi_rot(G[p][p], G[q][q], G[p][q], c, s);
                if (s != 0.0) *rotflag = 1;
                cs[t][0] = c; cs[t][1] = s; pq[t][0] = p; pq[t][1] = q;
            }
            __syncthreads();
            if (((t >> 5) & 15) < N / 2) {   // columns: G <- G J (threads 0..511), V <- V J (threads 512..1023)
                const int pi = (t >> 5) & 15, l = t & 31;
                const int p = pq[pi][0], q = pq[pi][1];
                const double c = cs[pi][0], s = cs[pi][1];
                BigMat M = (t < 512) ? G : V;
                const double mp = M[l][p], mq = M[l][q];
                M[l][p] = c * mp - s * mq;
                M[l][q] = s * mp + c * mq;
            }
            __syncthreads();
            if (t < 512 && (t >> 5) < N / 2) {   // rows: G <- J' G
                const int pi = t >> 5, l = t & 31;
                const int p = pq[pi][0], q = pq[pi][1];
                const double c = cs[pi][0], s = cs[pi][1];
                const double gp = G[p][l], gq = G[q][l];
                G[p][l] = c * gp - s * gq;
                G[q][l] = s * gp + c * gq;
            }
            __syncthreads();
        }
        const int done = (*rotflag == 0);
        __syncthreads();
        if (done) break;
    }
}

// eigen-decomposition with the warm start of eig16_warm: G' = V0' G V0 with the previous eigenvectors is almost diagonal;
// eigenvectors in the original basis V0 * V, stored back to vwarm (1024 doubles) for the next call. Ends with a barrier.
__device__ void big_eig_warm(BigMat G, BigMat V, BigMat V0, BigMat Tm, double (*cs)[2], int (*pq)[2], int *rotflag,
                             double *__restrict__ vwarm, int warm, int Kp = CP_KBIG) {
    constexpr int N = CP_KBIG;
    const int t = threadIdx.x, r = t >> 5, c = t & 31;
    const bool w = warm && vwarm;
    V[r][c] = (r == c) ? 1.0 : 0.0;
    if (w) {
        V0[r][c] = vwarm[t];
        __syncthreads();
        double acc = 0.0;
        for (int i = 0; i < N; ++i) acc += G[r][i] * V0[i][c];
        Tm[r][c] = acc;
        __syncthreads();
        acc = 0.0;
        for (int i = 0; i < N; ++i) acc += V0[i][r] * Tm[i][c];
        __syncthreads();
        G[r][c] = acc;
    }
    __syncthreads();
    big_jacobi(G, V, cs, pq, rotflag, Kp);
    if (w) {
        double acc = 0.0;
        for (int i = 0; i < N; ++i) acc += V0[r][i] * V[i][c];
        __syncthreads();
        V[r][c] = acc;
    }
    if (vwarm) vwarm[t] = V[r][c];
    __syncthreads();
}

// NNE/nuclear_norm_constraint.m:72-76 with NNE/l1ball_projection.m:23-68 on up to 32 singular values, by ONE thread in the
// serial order of the .m file (sort descending, running sums, the last index with a positive gap): S_i = sqrt(max(G_ii, 0)),
// z = S / mu_, P = projection of z onto the unit l1-ball, d_i = mu_ P_i / S_i -> dvec. res[0] = max_i (S_i - mu_ P_i),
// res[1] = max_i S_i. (l1ball_scale16 of cp_mfnn_small.cuh computes the same with 16 lanes.)
__device__ void big_l1ball_scale(BigMat G, int K, double mu_, double *dvec, double *res, double *__restrict__ sv_out) {
    double S[CP_KBIG], z[CP_KBIG], zs[CP_KBIG];
    double sumz = 0.0;
    for (int i = 0; i < K; ++i) {
        S[i] = sqrt(fmax(G[i][i], 0.0));
        z[i] = S[i] / mu_;
        sumz += z[i];
        sv_out[i] = S[i];
    }
    double theta = 0.0;
    const bool inside = (sumz <= 1.0);                         // l1ball_projection.m:28
    if (!inside) {
        for (int i = 0; i < K; ++i) {                          // descending by rank counting (ties keep their order)
            int rank = 0;
            for (int j = 0; j < K; ++j) rank += (z[j] > z[i] || (z[j] == z[i] && j < i)) ? 1 : 0;
            zs[rank] = z[i];
        }
        double cum = 0.0;
        for (int i = 0; i < K; ++i) {
            cum += zs[i];
            const double th = (cum - 1.0) / (double)(i + 1);
            if (zs[i] - th > 0.0) theta = th;                  // the largest such index wins
        }
    }
    double spmax = 0.0, smax = 0.0;
    for (int i = 0; i < CP_KBIG; ++i) {
        double d = 0.0;
        if (i < K) {
            const double Pi = inside ? z[i] : fmax(z[i] - theta, 0.0);   // :65-66 (sign(z) = +)
            const double sp = S[i] - mu_ * Pi;                             // nuclear_norm_constraint.m:76
            d = (S[i] > 0.0) ? (mu_ * Pi) / S[i] : 0.0;
            spmax = fmax(spmax, sp);
            smax = fmax(smax, S[i]);
        }
        dvec[i] = d;
    }
    res[0] = spmax; res[1] = smax;
}

#define CP_BIG_SHARED                                                                                     \
    __shared__ double Gs[CP_KBIG][CP_KBIG + 1], Vs[CP_KBIG][CP_KBIG + 1], V0s[CP_KBIG][CP_KBIG + 1],     \
        Tms[CP_KBIG][CP_KBIG + 1], sc[CP_NSCAL], dvec[CP_KBIG], res[2], cs[CP_KBIG / 2][2];             \
    __shared__ int pq[CP_KBIG / 2][2], rotflag

// ---- dual-side finalisation (k_mf_eig for K > 16): W for y3, F* values, primal row rp, stop rule ----------------------
__global__ __launch_bounds__(CP_BIG_THREADS) void k_big_eig(EigArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                            double *__restrict__ Wout, double *hist_p, double *hist_d) {
    CP_BIG_SHARED;
    const int t = threadIdx.x;
    const int stopped = ctrl->stop;
    const double nu = ctrl->nu;
    big_reduce(part, A.nblk, Gs, sc);
    if (stopped) return;
    big_eig_warm(Gs, Vs, V0s, Tms, cs, pq, &rotflag, A.vwarm, A.warm, (A.K + 1) & ~1);
    if (t == 0) big_l1ball_scale(Gs, A.K, nu * A.sigma, dvec, res, ctrl->sv);   // NNE/nuclear_norm_constraint.m:72
    __syncthreads();
    if (t == 0) {
        const double spmax = res[0], smax = res[1];
        if (A.rd >= 0) {
            double *hd = hist_d + (size_t)A.rd * CP_HCOLS_MF;
            hd[1] = sc[DS_F1S];                                                    // NNE/SAD.m:83
            hd[2] = (A.reg == 0) ? 0.0 : A.mu2 * (0.5 * sc[DS_F2S]);               // NNE/SSD.m:66-67 (g = 0)
            hd[3] = nu * (A.monitor ? smax : spmax);                               // nuclear_norm_constraint.m:89
            const double con1 = (sc[DM_Y1] > A.mu1) ? sc[DM_Y1] - A.mu1 : 0.0;     // NNE/SAD.m:77-81
            const double con2 = (A.reg != 0) ? 0.0
                                : (A.norm21 ? ((sc[DM_S1] > A.mu2) ? sc[DM_S1] - A.mu2 : 0.0)
                                            : fmax((sc[DM_S1] - A.mu2) / A.mu2, 0.0));
            hd[5] = fmax(fmax(con1, con2), 0.0);
        }
        if (A.compute_cn) {
            ctrl->cn[0] = sc[DS_UX] / A.Ntot + A.tau * (sc[DS_SGX] / A.Ntot);
            ctrl->cn[1] = sc[DS_UY] / A.Ntot + A.tau * (sc[DS_SGY] / A.Ntot);
            if (A.reg != 0) {
                ctrl->cn[0] -= A.tau * (sc[DS_BSX] / A.Ntot);
                ctrl->cn[1] -= A.tau * (sc[DS_BSY] / A.Ntot);
            }
        }
        if (A.rp >= 0) {
            double *hp = hist_p + (size_t)A.rp * CP_HCOLS_MF;
            const double F1 = A.mu1 * sc[DS_F1];
            const double F2 = (A.reg == 0) ? A.mu2 * sc[DS_F2] : A.mu2 * 0.5 * sc[DS_F2];
            hp[1] = F1; hp[2] = F2; hp[3] = 0.0; hp[4] = 0.0;
            hp[0] = hp[4] + (F1 + F2 + 0.0);                                       // chambolle_pock.m:170-171
            hp[6] = fmax(fabs(sc[DS_UX] / A.Ntot), fabs(sc[DS_UY] / A.Ntot));      // mean_zero_indicator.m:65
            if (A.check_stop && A.rp >= 4) {                                        // chambolle_pock.m:142-153
                bool ok = true;
                for (int r = A.rp - 4; r <= A.rp; ++r) {
                    const double *p_ = hist_p + (size_t)r * CP_HCOLS_MF, *d_ = hist_d + (size_t)r * CP_HCOLS_MF;
                    const double gap = fabs((p_[0] - d_[0]) / d_[0]);
                    if (!(gap <= A.tol) || !(p_[5] <= A.tol) || !(p_[6] <= A.tol) || !(d_[5] <= A.tol) ||
                        !(d_[6] <= A.tol))
                        ok = false;
                }
                if (ok) { ctrl->stop = 1; ctrl->stop_iter = A.rp; }
            }
        }
    }
    __syncthreads();
    {   // W = I - V diag(d) V'
        const int k = t >> 5, kp = t & 31;
        double a = (k == kp) ? 1.0 : 0.0;
        for (int i = 0; i < CP_KBIG; ++i) a -= Vs[k][i] * dvec[i] * Vs[kp][i];
        Wout[k * CP_KBIG + kp] = a;
    }
}

// ---- primal-side finalisation (k_mf_fin for K > 16): offsets c, feasibility of B*L, G* measures, dual row rd ----------
__global__ __launch_bounds__(CP_BIG_THREADS) void k_big_fin(EigArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                            double *hist_p, double *hist_d) {
    CP_BIG_SHARED;
    (void)dvec; (void)res;
    const int t = threadIdx.x, a_ = t >> 5, b_ = t & 31;
    const int stopped = ctrl->stop;
    const double nu = ctrl->nu;
    big_reduce(part, A.nblk, Gs, sc);
    if (stopped) return;
    {   // Gram(B*L) restricted to the complement of (1,..,1): out = Q' G Q with the Helmert basis (helmert_rows / helmert_cols)
        double row = 0.0;
        if (b_ < A.K - 1) {
            const double nb = 1.0 / sqrt((double)(b_ + 1) * (b_ + 2));
            for (int j = 0; j <= b_ + 1; ++j) row += Gs[a_][j] * ((j <= b_) ? nb : -(double)(b_ + 1) * nb);
        }
        Tms[a_][b_] = row;
        __syncthreads();
        double acc = 0.0;
        if (a_ < A.K - 1 && b_ < A.K - 1) {
            const double na = 1.0 / sqrt((double)(a_ + 1) * (a_ + 2));
            for (int i = 0; i <= a_ + 1; ++i) acc += ((i <= a_) ? na : -(double)(a_ + 1) * na) * Tms[i][b_];
        }
        __syncthreads();
        Gs[a_][b_] = acc;
        __syncthreads();
    }
    big_eig_warm(Gs, Vs, V0s, Tms, cs, pq, &rotflag, A.vwarm, A.warm, (A.K + 1) & ~1);
    if (t == 0) {
        double sumS = 0.0;
        for (int i = 0; i < A.K; ++i) ctrl->svL[i] = 0.0;
        for (int i = 0; i < A.K - 1; ++i) {
            const double s = sqrt(fmax(Gs[i][i], 0.0));
            ctrl->svL[i] = s;
            sumS += s;
        }
        if (A.rp >= 0)                                                             // nuclear_norm_constraint.m:55-59
            hist_p[(size_t)A.rp * CP_HCOLS_MF + 5] = (sumS > nu) ? (sumS - nu) / nu : 0.0;
        if (A.update) {
            ctrl->c[A.newbuf][0] = sc[PS_UX] / A.Ntot;                             // mean_zero_indicator.m:50,53
            ctrl->c[A.newbuf][1] = sc[PS_UY] / A.Ntot;
        }
        if (A.rd >= 0) {
            double *hd = hist_d + (size_t)A.rd * CP_HCOLS_MF;
            const double mw = sc[PS_WX] / A.Ntot;                                  // mean of u_x, used for BOTH (:78,:81)
            const double dx = fmax(sc[PM_WXMAX] - mw, mw - (-sc[PM_WXMIN]));
            const double dy = fmax(sc[PM_WYMAX] - mw, mw - (-sc[PM_WYMIN]));
            hd[4] = 0.0;
            hd[6] = fmax(fmax(dx, dy), sc[PM_VL]);                                 // mf_nn_tv...:342
            hd[0] = -(hd[4] + (hd[1] + hd[2] + hd[3]));                            // chambolle_pock.m:175-176
        }
    }
}

// reduce + Jacobi for cp_singular_values (k_mf_sv_eig for K > 16): out[0..31] eigenvalues, out[32..] = Vin * V
__global__ __launch_bounds__(CP_BIG_THREADS) void k_big_sv_eig(int nblk, const double *__restrict__ part,
                                                               const double *__restrict__ Vin, double *__restrict__ out) {
    CP_BIG_SHARED;
    (void)V0s; (void)Tms; (void)dvec; (void)res;
    constexpr int PS = cp_pstride<CP_KBIG>();
    const int t = threadIdx.x, k = t >> 5, kp = t & 31;
    double a0 = 0.0;
    for (int bI = 0; bI < nblk; ++bI) a0 += part[(size_t)bI * PS + CP_NSCAL + t];   // only the Gram part is valid here
    Gs[k][kp] = a0;
    Vs[k][kp] = (k == kp) ? 1.0 : 0.0;
    if (t < CP_NSCAL) sc[t] = 0.0;
    __syncthreads();
    big_jacobi(Gs, Vs, cs, pq, &rotflag);
    if (t < CP_KBIG) out[t] = Gs[t][t];
    double a = 0.0;
    for (int i = 0; i < CP_KBIG; ++i) a += Vin[k * CP_KBIG + i] * Vs[i][kp];
    out[CP_KBIG + t] = a;
}
