// Single-workgroup kernels of the mf_nn models: partial reduction, K x K Jacobi eigensolve,
// l1-ball projection of the singular values, history rows and the stop rule; plus the auxiliary
// full-grid kernels (linearisation/warp, materialise, Gram of B*field, power iteration).
#pragma once
#include "cp_mfnn.cuh"

#ifdef CP_EPROF
__device__ unsigned long long g_eprof[32];
#define EPROF(i) do { if (threadIdx.x == 0) g_eprof[i] = wall_clock64(); } while (0)
#define EPROF_ADD(i) do { if (threadIdx.x == 0) g_eprof[i] += 1; } while (0)
#else
#define EPROF(i)
#define EPROF_ADD(i)
#endif
#define CP_HCOLS_MF 7   // [p, F1, F2, F3, G, F_con, G_con] / [q, F1*, F2*, F3*, G*, F*_con, G*_con]

struct EigArgs {
    int nblk, K, reg;
    int rp, rd;          // primal / dual history row this kernel contributes to (-1: none)
    int check_stop;
    int update;          // fin: store the new mean offsets into ctrl->c[newbuf]
    int newbuf;
    int monitor;         // eig: the dual row describes y_in itself (row 0), not the projected y
    int compute_cn;      // eig: store the mean of the next un-projected u into ctrl->cn (start of the fused passes)
    int norm21;          // F2 = mu*||.||_{2,1} (CP_FLAG_MFNN_NORM21): absolute violation measure, NNE/norm21.m:81-85
    double tau;
    double tol, mu1, mu2, sigma, Ntot;
    double *vwarm;       // 256 doubles: eigenvector basis of this kernel's matrix from the previous iteration (or null)
    int warm;            // start the eigensolve from vwarm (see eig16_warm)
};

// ---- reduction of the per-block partial buffer (fixed order => deterministic) -------------------
// Runs with CP_RED_THREADS (1024) threads: 4 groups x 256 Gram entries, each group striding the blocks with 8
// independent accumulators (loads in flight), then a fixed-order combine through LDS. The 16 scalars use
// 64 sub-groups of 16 threads. stage: LDS scratch of at least 1024 doubles.
#define CP_RED_THREADS 1024
__device__ __forceinline__ void reduce_partials(const double *__restrict__ part, int nblk, double (*G)[17],
                                                double *sc, double *stage) {
    __shared__ double stage_sc[CP_RED_THREADS];
    const int t = threadIdx.x;
    {   // one phase: the loads of the Gram entries and of the scalars are all in flight before the first barrier
        const int grp = t >> 8, e = t & 255;
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.0;
        const double *src = part + CP_NSCAL + e;
        int bI = grp;
        for (; bI + 28 < nblk; bI += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += src[(size_t)(bI + 4 * u) * CP_PSTRIDE];
        }
        for (int u = 0; bI < nblk; bI += 4, ++u) a[u & 7] += src[(size_t)bI * CP_PSTRIDE];
        const int slot = t & 31, sub = t >> 5;   // 32 sub-groups of 32 threads (CP_NSCAL <= 32 slots used)
        const bool is_sum = slot < CP_NSUM;
        double v = is_sum ? 0.0 : -1.0e300;
        if (slot < CP_NSCAL)
            for (int bJ = sub; bJ < nblk; bJ += 32) {
                const double o = part[(size_t)bJ * CP_PSTRIDE + slot];
                v = is_sum ? v + o : fmax(v, o);
            }
        stage[t] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        stage_sc[t] = v;
    }
    __syncthreads();
    if (t < 256) {
        G[t >> 4][t & 15] = (stage[t] + stage[256 + t]) + (stage[512 + t] + stage[768 + t]);
    } else if (t < 256 + CP_NSCAL) {
        const int q = t - 256;
        const bool is_sum = q < CP_NSUM;
        double v = stage_sc[q];
        for (int sub = 1; sub < 32; ++sub) {
            const double o = stage_sc[sub * 32 + q];
            v = is_sum ? v + o : fmax(v, o);
        }
        sc[q] = v;
    }
    __syncthreads();
}

// Both Gram blocks of the fused pass (Gram(y3~) at +CP_NSCAL, Gram(B*L) 256 further) and the scalars in ONE phase:
// every load of the kernel's reduction is in flight before the first barrier (k_mf_efused is latency-bound: a single
// workgroup, a few dependent global round trips). 2 groups x 512 Gram entries stride the rows with 8 accumulators,
// the scalars use 32 sub-groups of 32 threads; fixed order, deterministic. stage, stage2: 1024 doubles each.
__device__ __forceinline__ void reduce_partials2(const double *__restrict__ part, int nblk, double (*G)[17],
                                                 double (*G2)[17], double *sc, double *stage, double *stage2) {
    const int t = threadIdx.x;
    {
        const int grp = t >> 9, e = t & 511;
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.0;
        const double *src = part + CP_NSCAL + e;
        int bI = grp;
        for (; bI + 14 < nblk; bI += 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += src[(size_t)(bI + 2 * u) * CP_PSTRIDE];
        }
        for (int u = 0; bI < nblk; bI += 2, ++u) a[u & 7] += src[(size_t)bI * CP_PSTRIDE];
        const int slot = t & 31, sub = t >> 5;   // 32 sub-groups of 32 threads (CP_NSCAL <= 32 slots used)
        const bool is_sum = slot < CP_NSUM;
        double v = is_sum ? 0.0 : -1.0e300;
        if (slot < CP_NSCAL)
            for (int bJ = sub; bJ < nblk; bJ += 32) {
                const double o = part[(size_t)bJ * CP_PSTRIDE + slot];
                v = is_sum ? v + o : fmax(v, o);
            }
        stage[t] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        stage2[t] = v;
    }
    __syncthreads();
    if (t < 512) {
        const double v = stage[t] + stage[512 + t];
        const int e = t & 255;
        if (t < 256) G[e >> 4][e & 15] = v;
        else G2[e >> 4][e & 15] = v;
    } else if (t < 512 + CP_NSCAL) {
        const int q = t - 512;
        const bool is_sum = q < CP_NSUM;
        double v = stage2[q];
        for (int sub = 1; sub < 32; ++sub) {
            const double o = stage2[sub * 32 + q];
            v = is_sum ? v + o : fmax(v, o);
        }
        sc[q] = v;
    }
    __syncthreads();
}

// Jacobi rotation (c, s) that annihilates a_pq: with d = a_qq - a_pp, h = |(d, 2 a_pq)| and w = |d| + h the half-angle
// vector is (c, s) = (w, sign(d) 2 a_pq) / |(w, 2 a_pq)|. Branch-free (two calls interleave as independent instruction
// streams) and built on v_rsq_f64: h needs no more than one Newton step (an error in h only perturbs the angle, the
// normalisation below restores c^2 + s^2 = 1), the normalising reciprocal square root gets two. Identity below the
// rotation threshold |a_pq| <= eps sqrt(a_pp a_qq) (the relative criterion of one-sided / positive definite Jacobi;
// a tighter one only adds a sweep that rotates rounding noise).
#define CP_JACOBI_TOL2 4.9e-32   // (2^-52)^2
__device__ __forceinline__ void jacobi_rot(double app, double aqq, double apq, double &c, double &s) {
    const bool rot = (apq * apq > CP_JACOBI_TOL2 * fabs(app * aqq)) && (fabs(apq) > 1.0e-300);
    const double d = aqq - app, a2 = apq + apq;
    const double x = rot ? fma(d, d, a2 * a2) : 1.0;
    const double y0 = __builtin_amdgcn_rsq(x);
    const double g0 = x * y0;                                   // ~ sqrt(x)
    const double h = fma(fma(-g0, g0, x), 0.5 * y0, g0);        // one Newton step
    const double w = fabs(d) + h;
    const double x2 = rot ? fma(w, w, a2 * a2) : 1.0;
    double y = __builtin_amdgcn_rsq(x2);
    y = fma(0.5 * y, fma(-x2 * y, y, 1.0), y);
    y = fma(0.5 * y, fma(-x2 * y, y, 1.0), y);
    c = rot ? w * y : 1.0;
    s = rot ? copysign(a2, d * apq) * y : 0.0;
}
// ---- parallel cyclic Jacobi on a symmetric 16 x 16 matrix in LDS (256 threads) -------------------
// Round-robin ordering: 15 rounds of 8 disjoint rotations per sweep. On exit diag(G) holds the
// eigenvalues and (WITH_V) the columns of V the eigenvectors: G_in = V diag V'.
template <bool WITH_V>
__device__ void jacobi16(double (*G)[17], double (*V)[17], double (*cs)[2], int (*pq)[2], int *rotflag) {
    const int t = threadIdx.x;   // the first 256 threads work, the others only join the barriers
    if (WITH_V && t < 256) V[t >> 4][t & 15] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;
    {   // symmetrise
        const int r = (t >> 4) & 15, c = t & 15;
        const double s = 0.5 * (G[r][c] + G[c][r]);
        __syncthreads();
        if (t < 256) G[r][c] = s;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 30; ++sweep) {
        if (t == 0) *rotflag = 0;
        __syncthreads();
        for (int r = 0; r < 15; ++r) {
            if (t < 8) {
                int p, q;
                if (t == 0) { p = 15; q = r; }
                else { p = (r + t) % 15; q = (r - t + 15) % 15; }
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                const double app = G[p][p], aqq = G[q][q], apq = G[p][q];
                double c, s;
                jacobi_rot(app, aqq, apq, c, s);
                if (s != 0.0) *rotflag = 1;
                cs[t][0] = c; cs[t][1] = s; pq[t][0] = p; pq[t][1] = q;
            }
            __syncthreads();
            {   // columns: G <- G J (threads 0..127), V <- V J (threads 128..255)
                const int pi = (t >> 4) & 7, l = t & 15;
                const int p = pq[pi][0], q = pq[pi][1];
                const double c = cs[pi][0], s = cs[pi][1];
                if (t < 128) {
                    const double gp = G[l][p], gq = G[l][q];
                    G[l][p] = c * gp - s * gq;
                    G[l][q] = s * gp + c * gq;
                } else if (WITH_V && t < 256) {
                    const double vp = V[l][p], vq = V[l][q];
                    V[l][p] = c * vp - s * vq;
                    V[l][q] = s * vp + c * vq;
                }
            }
            __syncthreads();
            if (t < 128) {   // rows: G <- J' G
                const int pi = t >> 4, l = t & 15;
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

// Ordering point for LDS traffic between the lanes of ONE wavefront (its DS instructions execute in order; this only
// keeps the compiler from moving accesses across it). No s_barrier.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cyclic Jacobi on a symmetric 16 x 16 matrix in LDS by ONE wavefront, in place, without workgroup barriers (the
// barrier-per-round versions spent 0.7 us per round, 30 us per warm-started call: k_mf_efused is pure latency).
// Round-robin ordering of jacobi16: 15 rounds of 8 disjoint index pairs per sweep. The lanes form an 8 x 8 grid; in a
// round lane (I, J) owns the 2 x 2 block (rows of pair I) x (columns of pair J) of G and rows {2I, 2I+1} x (columns of
// pair J) of V, reads them together with the two diagonal blocks that define the rotations of pair I and pair J
// (recomputed per lane from the same LDS data, so bitwise the same in every lane -- nothing is broadcast), applies
// G' = J' G J, V' = V J and writes its block back: all reads of a round precede all writes in program order, and the
// blocks of a round are disjoint. The two rotation chains of a lane are independent instruction streams.
// On exit diag(G) holds the eigenvalues, the columns of V the eigenvectors (G_in = V diag V'). V must hold the
// identity (or the basis to be rotated) on entry. Call with the 64 lanes of one wavefront; returns the sweep count.
// nact: rows/columns from nact on are zero (K < 16: zero padding). The tournament then runs over N = the even number >= nact
// players -- N - 1 rounds of N/2 pairs instead of 15 of 8 (a rotation against a zero row is the identity, so the rounds
// left out are no-ops: K = 10: 9 rounds per sweep, 18 -> 11 us per warm-started call); lanes whose pair index is >= N/2
// repeat the work of lane (I mod N/2, J mod N/2) -- same inputs, same values written to the same places.
__device__ __forceinline__ int jacobi16_wave(double (*G)[17], double (*V)[17], int nact = 16) {
    const int lane = threadIdx.x & 63, I = lane >> 3, J = lane & 7;
    const int r0 = 2 * I, c0 = 2 * J;
    const int N = min(16, max(2, (nact + 1) & ~1)), Nm1 = N - 1, hN = N >> 1;
    const int Ie = I % hN, Je = J % hN;
    {   // symmetrise (static 2 x 2 tile per lane)
        const double s00 = 0.5 * (G[r0][c0] + G[c0][r0]), s01 = 0.5 * (G[r0][c0 + 1] + G[c0 + 1][r0]);
        const double s10 = 0.5 * (G[r0 + 1][c0] + G[c0][r0 + 1]), s11 = 0.5 * (G[r0 + 1][c0 + 1] + G[c0 + 1][r0 + 1]);
        wave_lds_sync();
        G[r0][c0] = s00; G[r0][c0 + 1] = s01; G[r0 + 1][c0] = s10; G[r0 + 1][c0 + 1] = s11;
        wave_lds_sync();
    }
    int sweep = 0;
    for (; sweep < 30; ++sweep) {
        int need = 0;   // an off-diagonal entry above the rotation threshold somewhere in the upper triangle?
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int a = r0 + (e >> 1), b = c0 + (e & 1);
            const double apq = G[a][b];
            if (a < b && apq * apq > CP_JACOBI_TOL2 * fabs(G[a][a] * G[b][b]) && fabs(apq) > 1.0e-300) need = 1;
        }
        if (!__any(need)) break;
        int xI = Ie % Nm1, yI = (Nm1 - Ie) % Nm1, xJ = Je % Nm1, yJ = (Nm1 - Je) % Nm1;   // (rd + k) % (N-1), (rd - k + N-1) % (N-1) at rd = 0
        for (int rd = 0; rd < Nm1; ++rd) {
            // pair I = (p, q), pair J = (r, u): (rd, N-1) for k = 0 and the two residues above for k = 1..N/2-1, lower first
            const int p = (Ie == 0) ? rd : min(xI, yI), q = (Ie == 0) ? Nm1 : max(xI, yI);
            const int r = (Je == 0) ? rd : min(xJ, yJ), u = (Je == 0) ? Nm1 : max(xJ, yJ);
            xI = (xI == Nm1 - 1) ? 0 : xI + 1; yI = (yI == Nm1 - 1) ? 0 : yI + 1;
            xJ = (xJ == Nm1 - 1) ? 0 : xJ + 1; yJ = (yJ == Nm1 - 1) ? 0 : yJ + 1;
            const double gpp = G[p][p], gqq = G[q][q], gpq = G[p][q];
            const double grr = G[r][r], guu = G[u][u], gru = G[r][u];
            const double b00 = G[p][r], b01 = G[p][u], b10 = G[q][r], b11 = G[q][u];
            const double v00 = V[r0][r], v01 = V[r0][u], v10 = V[r0 + 1][r], v11 = V[r0 + 1][u];
            double cI, sI, cJ, sJ;
            jacobi_rot(gpp, gqq, gpq, cI, sI);
            jacobi_rot(grr, guu, gru, cJ, sJ);
            // columns (pair J): x_r' = c x_r - s x_u, x_u' = s x_r + c x_u; then rows (pair I) likewise
            const double t00 = cJ * b00 - sJ * b01, t01 = sJ * b00 + cJ * b01;
            const double t10 = cJ * b10 - sJ * b11, t11 = sJ * b10 + cJ * b11;
            double n00 = cI * t00 - sI * t10, n01 = cI * t01 - sI * t11;
            double n10 = sI * t00 + cI * t10, n11 = sI * t01 + cI * t11;
            if (Ie == Je && sI != 0.0) { n01 = 0.0; n10 = 0.0; }   // the annihilated entry, exactly
            wave_lds_sync();
            G[p][r] = n00; G[p][u] = n01; G[q][r] = n10; G[q][u] = n11;
            V[r0][r] = cJ * v00 - sJ * v01; V[r0][u] = sJ * v00 + cJ * v01;
            V[r0 + 1][r] = cJ * v10 - sJ * v11; V[r0 + 1][u] = sJ * v10 + cJ * v11;
            wave_lds_sync();
        }
    }
    return sweep;
}

// B*L is exactly mean-free over the images, so (1,..,1) is a null vector of its Gram matrix. Rounding would turn that
// zero into sqrt(eps)*s_max; the matrix is restricted to the orthogonal complement instead (Helmert basis
// q_j = (1,..,1,-j,0,..)/sqrt(j(j+1)), j = 1..K-1) before the eigensolve: out = Q' Gin Q, padded with zeros.
// Two phases of <= 17 terms per thread: helmert_rows (T = Gin Q), a workgroup barrier, helmert_cols (out = Q' T).
// Called by the 256 threads (a, b) = (t >> 4, t & 15) of a matrix.
__device__ __forceinline__ void helmert_rows(const double (*Gin)[17], double (*T)[17], int K, int a, int b) {
    double row = 0.0;
    if (b < K - 1) {
        const double nb = 1.0 / sqrt((double)(b + 1) * (b + 2));
        for (int j = 0; j <= b + 1; ++j) {
            const double qb = (j <= b) ? nb : -(double)(b + 1) * nb;
            row += Gin[a][j] * qb;
        }
    }
    T[a][b] = row;
}
__device__ __forceinline__ void helmert_cols(const double (*T)[17], double (*out)[17], int K, int a, int b) {
    double acc = 0.0;
    if (a < K - 1 && b < K - 1) {
        const double na = 1.0 / sqrt((double)(a + 1) * (a + 2));
        for (int i = 0; i <= a + 1; ++i) {
            const double qa = (i <= a) ? na : -(double)(a + 1) * na;
            acc += qa * T[i][b];
        }
    }
    out[a][b] = acc;
}

// Conjugate prox of the nuclear-norm constraint on the singular values (NNE/nuclear_norm_constraint.m:72-76 with
// NNE/l1ball_projection.m:23-68), by the first 16 lanes of ONE wavefront (lane i = singular value i; call with the
// whole wavefront, no workgroup barrier inside): S_i = sqrt(max(G_ii, 0)), z = S / mu_, P = projection of z onto the
// unit l1-ball (sort by rank counting, running sums in the serial order of the .m file), d_i = mu_ P_i / S_i -> dvec.
// res[0] = max_i (S_i - mu_ P_i) (largest singular value of the prox), res[1] = max_i S_i. zv, zs: LDS scratch of 16.
__device__ __forceinline__ void l1ball_scale16(const double (*G)[17], int K, double mu_, double *zv, double *zs,
                                               double *dvec, double *res, double *__restrict__ sv_out) {
    const int i = threadIdx.x & 63;
    const bool on = i < K;
    const double Si = on ? sqrt(fmax(G[i & 15][i & 15], 0.0)) : 0.0;
    const double zi = Si / mu_;
    if (i < 16) { zv[i] = on ? zi : 0.0; zs[i] = 0.0; }
    if (on) sv_out[i] = Si;
    wave_lds_sync();
    double sumz = 0.0;
    int rank = 0;
    for (int j = 0; j < K; ++j) {
        const double zj = zv[j];
        sumz += zj;
        rank += (zj > zi || (zj == zi && j < i)) ? 1 : 0;
    }
    const bool inside = (sumz <= 1.0);                    // NNE/l1ball_projection.m:28
    if (on) zs[rank] = zi;                                // descending
    wave_lds_sync();
    double cum = 0.0;
    for (int j = 0; j < 16; ++j) cum += (j <= i) ? zs[j] : 0.0;
    const double th = (cum - 1.0) / (double)(i + 1);
    const bool okth = on && (zs[i & 15] - th > 0.0);
    const unsigned long long mk = __ballot(okth);
    const int last = mk ? 63 - __clzll((long long)mk) : 0;   // the largest such index wins (monotone)
    const double theta = (inside || !mk) ? 0.0 : __shfl(th, last);
    const double Pi = inside ? zi : fmax(zi - theta, 0.0);   // :65-66 (sign(z) = +)
    const double sp = Si - mu_ * Pi;                         // nuclear_norm_constraint.m:76
    if (i < 16) {
        dvec[i] = (on && Si > 0.0) ? (mu_ * Pi) / Si : 0.0;
        zv[i] = on ? sp : 0.0;
        zs[i] = Si;
    }
    wave_lds_sync();
    if (i == 0) {
        double spmax = 0.0, smax = 0.0;
        for (int j = 0; j < 16; ++j) { spmax = fmax(spmax, zv[j]); smax = fmax(smax, zs[j]); }
        res[0] = spmax; res[1] = smax;
    }
}

// Eigen-decomposition of NM (1 or 2) symmetric 16 x 16 matrices GG[mid] on NM*256 live threads, optionally warm-started:
// the matrices of consecutive CP iterations differ little, so G' = V0' G V0 with the previous eigenvectors V0 is almost
// diagonal and needs 2-3 sweeps instead of ~8; the eigenvectors in the original basis are V0 * V. On exit diag(GG[mid])
// holds the eigenvalues, VV[mid] the eigenvectors; they are also stored to vwarm (if given) for the next call.
// V0, Tm: LDS scratch matrices. The Jacobi iteration itself runs on one wavefront per matrix (jacobi16_wave); the
// others wait at the barrier that follows. Ends with a barrier.
template <int NM>
__device__ __forceinline__ void eig16_warm(double (*GG)[16][17], double (*VV)[16][17], double (*V0)[16][17],
                                           double (*Tm)[16][17], double *__restrict__ vwarm, int warm, int nact = 16) {
    const int t = threadIdx.x, mid = t >> 8, tl = t & 255, r = tl >> 4, c = tl & 15;
    const bool w = warm && vwarm;
    VV[mid][r][c] = (r == c) ? 1.0 : 0.0;
    if (w) {
        V0[mid][r][c] = vwarm[mid * 256 + tl];
        __syncthreads();
        double acc = 0.0;
        for (int i = 0; i < 16; ++i) acc += GG[mid][r][i] * V0[mid][i][c];
        Tm[mid][r][c] = acc;
        __syncthreads();
        acc = 0.0;
        for (int i = 0; i < 16; ++i) acc += V0[mid][i][r] * Tm[mid][i][c];
        __syncthreads();
        GG[mid][r][c] = acc;
    }
    __syncthreads();
    EPROF(3);
    if ((t >> 6) < NM) {   // one wavefront per matrix (wavefronts 0 and 1: different SIMDs -- the iteration is issue-bound), no workgroup barrier inside
        const int sw = jacobi16_wave(GG[t >> 6], VV[t >> 6], nact);
        (void)sw;
#ifdef CP_EPROF
        if ((t & 63) == 0) g_eprof[20 + (t >> 6)] = sw;
#endif
    }
    __syncthreads();
    EPROF(4);
    if (w) {   // eigenvectors in the original basis: V0 * V
        double acc = 0.0;
        for (int i = 0; i < 16; ++i) acc += V0[mid][r][i] * VV[mid][i][c];
        __syncthreads();
        VV[mid][r][c] = acc;
    }
    if (vwarm) vwarm[mid * 256 + tl] = VV[mid][r][c];
    __syncthreads();
}

// ---- dual-side finalisation: W for y3, F* values, primal row rp, stop rule ------------------------
__global__ __launch_bounds__(CP_RED_THREADS) void k_mf_eig(EigArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                           double *__restrict__ Wout, double *hist_p, double *hist_d) {
    __shared__ double GG[1][16][17], VV[1][16][17], V0[1][16][17], Tm[1][16][17], sc[CP_NSCAL],
        dvec[16], stage[CP_RED_THREADS];
    double (*G)[17] = GG[0], (*V)[17] = VV[0];
    const int t = threadIdx.x;
    __shared__ double z[16], zs[16], res[2];
    const int stopped = ctrl->stop;   // requested together with the reduction's loads; see k_mf_efused
    const double nu = ctrl->nu;
    reduce_partials(part, A.nblk, G, sc, stage);
    if (stopped) return;
    if (t >= 256) return;        // the eigensolve needs 256 threads; ended waves do not take part in barriers
    eig16_warm<1>(GG, VV, V0, Tm, A.vwarm, A.warm, A.K);
    if (t < 64) l1ball_scale16(G, A.K, nu * A.sigma, z, zs, dvec, res, ctrl->sv);   // NNE/nuclear_norm_constraint.m:72
    __syncthreads();
    if (t == 0) {
        const double spmax = res[0], smax = res[1];
        if (A.rd >= 0) {
            double *hd = hist_d + (size_t)A.rd * CP_HCOLS_MF;
            hd[1] = sc[DS_F1S];                                                    // NNE/SAD.m:83
            hd[2] = (A.reg == 0) ? 0.0 : A.mu2 * (0.5 * sc[DS_F2S]);               // NNE/SSD.m:66-67 (g = 0)
            // nuclear_norm_constraint.m:89: nu * max(svd(y3)); the singular values of the projected y3 are S_prox
            hd[3] = nu * (A.monitor ? smax : spmax);
            const double con1 = (sc[DM_Y1] > A.mu1) ? sc[DM_Y1] - A.mu1 : 0.0;     // NNE/SAD.m:77-81
            const double con2 = (A.reg != 0) ? 0.0
                                : (A.norm21 ? ((sc[DM_S1] > A.mu2) ? sc[DM_S1] - A.mu2 : 0.0)
                                            : fmax((sc[DM_S1] - A.mu2) / A.mu2, 0.0));
            hd[5] = fmax(fmax(con1, con2), 0.0);
        }
        if (A.compute_cn) {
            // mean(u - tau*(A'y)_u) = mean(u) + tau*mean(grad T' y1): D'y2 has zero mean because D*1 = 0
            ctrl->cn[0] = sc[DS_UX] / A.Ntot + A.tau * (sc[DS_SGX] / A.Ntot);
            ctrl->cn[1] = sc[DS_UY] / A.Ntot + A.tau * (sc[DS_SGY] / A.Ntot);
            if (A.reg != 0) {   // curvature: (A'y)_u = -grad T' y1 + Lap y2, and mean(Lap y2) = sum(y2 * rowsum(Lap)) / N (dirichlet rows)
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
    if (t < 256) {   // W = I - V diag(d) V'
        const int k = t >> 4, kp = t & 15;
        double a = (k == kp) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) a -= V[k][i] * dvec[i] * V[kp][i];
        Wout[k * CP_KMAX + kp] = a;
    }
}

// ---- primal-side finalisation: offsets c, feasibility of B*L, G* measures, dual row rd ------------
__global__ __launch_bounds__(CP_RED_THREADS) void k_mf_fin(EigArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                           double *hist_p, double *hist_d) {
    __shared__ double GG[1][16][17], VV[1][16][17], V0[1][16][17], Tm[1][16][17], sc[CP_NSCAL],
        stage[CP_RED_THREADS];
    double (*G)[17] = GG[0];
    const int stopped = ctrl->stop;
    const double nu = ctrl->nu;
    reduce_partials(part, A.nblk, G, sc, stage);
    if (stopped) return;
    {   // Gram(B*L) restricted to the complement of (1,..,1) (see helmert_rows)
        const int t = threadIdx.x;
        if (t >= 256) return;    // the eigensolve needs 256 threads; ended waves do not take part in barriers
        helmert_rows(G, Tm[0], A.K, t >> 4, t & 15);
        __syncthreads();
        helmert_cols(Tm[0], G, A.K, t >> 4, t & 15);
        __syncthreads();
    }
    eig16_warm<1>(GG, VV, V0, Tm, A.vwarm, A.warm, A.K);
    if (threadIdx.x == 0) {
        double sumS = 0.0;
        for (int i = 0; i < A.K; ++i) ctrl->svL[i] = 0.0;
        for (int i = 0; i < A.K - 1; ++i) {
            const double s = sqrt(fmax(G[i][i], 0.0));
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

// ---- state canonicalisation: apply pending offsets / W so that the stored vectors are the true ones ---
template <typename S, int REG, int KM = CP_KMAX>
__global__ __launch_bounds__(CP_BLOCK) void k_mf_materialize(MfParams P, const S *__restrict__ xs,
                                                             S *__restrict__ xd, const S *__restrict__ ps,
                                                             S *__restrict__ pd, const double *__restrict__ W,
                                                             const CpCtrl *__restrict__ ctrl, int cbuf) {
    __shared__ double tiles[CP_WAVES * KM * CP_TS];
    __shared__ double Wl[KM * KM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *tile = tiles + wave * KM * CP_TS;
    const int K = P.K, fs = P.fs;
    stage_W<KM>(W, Wl);
    __syncthreads();
    const double cx = ctrl->c[cbuf][0], cy = ctrl->c[cbuf][1];
    const int NY2 = (REG == 0) ? 4 : 2;
    for (int base = blockIdx.x * CP_BLOCK; base < P.npix; base += gridDim.x * CP_BLOCK) {
        const int q = base + tid;
        const bool valid = q < P.npix;
        const int e = q + P.m;
        load_y3_apply_W<S, KM>(ps + (size_t)(K + NY2 * K) * fs, fs, e, valid, K, Wl, tile, lane);
        if (valid) {
            for (int k = 0; k < K; ++k) {
                xd[(size_t)(2 * k) * fs + e] = (S)((double)xs[(size_t)(2 * k) * fs + e] - cx);
                xd[(size_t)(2 * k + 1) * fs + e] = (S)((double)xs[(size_t)(2 * k + 1) * fs + e] - cy);
                xd[(size_t)(2 * K + k) * fs + e] = xs[(size_t)(2 * K + k) * fs + e];
                pd[(size_t)(K + NY2 * K + k) * fs + e] = (S)tile[k * CP_TS + lane];
            }
            if (ps != pd)
                for (int f = 0; f < K + NY2 * K; ++f) pd[(size_t)f * fs + e] = ps[(size_t)f * fs + e];
        }
    }
}

// km: extent of the W matrices (16, or CP_KBIG on the general path)
__global__ void k_mf_reset_ctrl(CpCtrl *ctrl, double *W0, double *W1, int km = CP_KMAX) {
    const int t = threadIdx.x;
    if (t == 0) {
        ctrl->stop = 0; ctrl->stop_iter = -1;
        ctrl->c[0][0] = ctrl->c[0][1] = ctrl->c[1][0] = ctrl->c[1][1] = 0.0;
    }
    for (int e = t; e < km * km; e += blockDim.x) {
        const double v = ((e / km) == (e % km)) ? 1.0 : 0.0;
        W0[e] = v;
        W1[e] = v;
    }
}

// ---- linearisation: T(u0), grad T(u0), b = T(u0) - grad T . u0   --------------------------------------
// IRE/evaluate_displacement.m:38-49 -> IRE/cell_centered_grid.m:20-29 -> IRE/bilinear_interpolation.m:29-105.
// Coordinate arithmetic uses explicitly rounded (non-fused) operations so that floor() -- the index
// logic -- is bit-identical to the CPU restatement.
struct WarpParams {
    int m, nl, K, npix, n_global, j_off, fs;
    int jlo;                 // first local column processed (-1: the left halo column), npix = m * number of columns
    int mi, ni;              // image size (rows, cols) -- equals (m, n_global) in the drivers
    int ifs;                 // image field stride (mi*ni)
    double om0, om2, hx, hy; // grid origin and cell size
    double inv_h1, inv_h2;   // 1/h of the image (h passed to evaluate_displacement)
};

// one thread per (pixel, image); u is read from x fields [2k], [2k+1] (true values: canonical state)
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_mf_warp(WarpParams P, const double *__restrict__ imgs,
                                                      const S *__restrict__ x, S *__restrict__ Tu,
                                                      S *__restrict__ g, S *__restrict__ b) {
    const long long gid = (long long)blockIdx.x * CP_BLOCK + threadIdx.x;
    if (gid >= (long long)P.npix * P.K) return;
    const int k = (int)(gid / P.npix), q = (int)(gid - (long long)k * P.npix);
    const int jj = q / P.m, i = q - jj * P.m, j = jj + P.jlo, e = (j + 1) * P.m + i;
    const double ux = x[(size_t)(2 * k) * P.fs + e], uy = x[(size_t)(2 * k + 1) * P.fs + e];
    const double px = __dadd_rn(__dadd_rn(P.om0, __dmul_rn(P.hx, (double)i + 0.5)), ux);
    const double py = __dadd_rn(__dadd_rn(P.om2, __dmul_rn(P.hy, (double)(P.j_off + j) + 0.5)), uy);
    double val, dx, dy;
    bilinear_tap(imgs + (size_t)k * P.ifs, P.mi, P.ni, px, py, P.inv_h1, P.inv_h2, val, dx, dy);
    Tu[(size_t)k * P.fs + e] = (S)val;
    g[(size_t)(2 * k) * P.fs + e] = (S)dx;
    g[(size_t)(2 * k + 1) * P.fs + e] = (S)dy;
    // fp32 storage: b is formed from the rounded gradient, the one the iteration multiplies with
    const double dxs = rnd<S>(dx), dys = rnd<S>(dy);
    b[(size_t)k * P.fs + e] = (S)(val - (dxs * ux + dys * uy));        // NNE/mf_nn_tv...:167-168
}

// ---- Gram of (B*field)*Vm for accurate singular values (nu_0, SV_history) -----------------------------
template <typename S, int KM = CP_KMAX>
__global__ __launch_bounds__(CP_BLOCK) void k_mf_gram_field(MfParams P, const S *__restrict__ field,
                                                            const double *__restrict__ Vm,
                                                            double *__restrict__ part) {
    constexpr int NT = KM / 16;
    __shared__ double tiles[CP_WAVES * KM * CP_TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *tile = tiles + wave * KM * CP_TS;
    const int K = P.K, fs = P.fs;
    GramAcc<NT> acc;
    gram_acc_zero(acc);
    for (int base = blockIdx.x * CP_BLOCK; base < P.npix; base += gridDim.x * CP_BLOCK) {
        const int q = base + tid;
        const bool valid = q < P.npix;
        const int e = q + P.m;
        double t3[KM];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            t3[k] = (valid && k < K) ? field[(size_t)k * fs + e] : 0.0;
            s += t3[k];
        }
        const double mean = s * P.invK;
#pragma unroll
        for (int k = 0; k < KM; ++k) t3[k] = (valid && k < K) ? t3[k] - mean : 0.0;   // mean_free_operator.m:20
#pragma unroll
        for (int kp = 0; kp < KM; ++kp) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < KM; ++k) a = fma(t3[k], Vm[k * KM + kp], a);
            tile[kp * CP_TS + lane] = a;
        }
        __syncthreads();
        gram_tile_accum_n<NT>(tile, acc);
        __syncthreads();
    }
    gram_block_store_n<NT>(acc, tiles, part + (size_t)blockIdx.x * cp_pstride<KM>() + CP_NSCAL);
}

// reduce + Jacobi; out[0..15] eigenvalues (diag), out[16..271] eigenvector matrix composed with Vin
__global__ __launch_bounds__(CP_BLOCK) void k_mf_sv_eig(int nblk, const double *__restrict__ part,
                                                        const double *__restrict__ Vin, double *__restrict__ out) {
    __shared__ double G[16][17], V[16][17], sc[CP_NSCAL], cs[8][2];
    __shared__ int pq[8][2], rotflag;
    const int t = threadIdx.x;
    {   // only the Gram part of the partials is valid here
        double a0 = 0.0;
        for (int bI = 0; bI < nblk; ++bI) a0 += part[(size_t)bI * CP_PSTRIDE + CP_NSCAL + t];
        G[t >> 4][t & 15] = a0;
        if (t < CP_NSCAL) sc[t] = 0.0;
        __syncthreads();
    }
    jacobi16<true>(G, V, cs, pq, &rotflag);
    if (t < 16) out[t] = G[t][t];
    {   // Vout = Vin * V
        const int k = t >> 4, kp = t & 15;
        double a = 0.0;
        for (int i = 0; i < 16; ++i) a += Vin[k * CP_KMAX + i] * V[i][kp];
        out[16 + t] = a;
    }
}

// ---- power iteration of NNE/matrix_norm.m on A = [-grad T, I; D|Lap, 0; 0, B] -------------------------------
// y = A*(scale*x)
// SG: storage type of the gradient field (the vectors of the power iteration are doubles in every mode)
template <typename SG, int REG>
__global__ __launch_bounds__(CP_BLOCK) void k_mf_applyA(MfParams P, const double *__restrict__ x, double scale,
                                                        const SG *__restrict__ g, double *__restrict__ y) {
    const int K = P.K, m = P.m, fs = P.fs;
    const bool linear = (P.bc == 1), neumann = (P.bc == 0);
    const int NY2 = (REG == 0) ? 4 : 2;
    for (int q = blockIdx.x * CP_BLOCK + threadIdx.x; q < P.npix; q += gridDim.x * CP_BLOCK) {
        const int j = q / m, i = q - j * m, e = q + m, jg = P.j_off + j;
        const bool last_i = (i == m - 1), last_j = (jg == P.n_global - 1), first_i = (i == 0), first_j = (jg == 0);
        double sumL = 0.0;
        for (int k = 0; k < K; ++k) sumL += scale * x[(size_t)(2 * K + k) * fs + e];
        const double mean = sumL * P.invK;
        for (int k = 0; k < K; ++k) {
            const double *ux = x + (size_t)(2 * k) * fs + e, *uy = x + (size_t)(2 * k + 1) * fs + e;
            const double L = scale * x[(size_t)(2 * K + k) * fs + e];
            const double u0 = scale * ux[0], v0 = scale * uy[0];
            y[(size_t)k * fs + e] = -((double)g[(size_t)(2 * k) * fs + e] * u0 + (double)g[(size_t)(2 * k + 1) * fs + e] * v0) + L;
            y[(size_t)(K + NY2 * K + k) * fs + e] = L - mean;
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

// xo = A'*y and per-block partial sums of ||xo||^2 (part[blk*CP_PSTRIDE + 0])
template <typename SG, int REG>
__global__ __launch_bounds__(CP_BLOCK) void k_mf_applyAT(MfParams P, const double *__restrict__ y,
                                                         const SG *__restrict__ g, double *__restrict__ xo,
                                                         double *__restrict__ part) {
    __shared__ double red[CP_WAVES * 2];
    const int K = P.K, m = P.m, fs = P.fs;
    const bool linear = (P.bc == 1), neumann = (P.bc == 0);
    const int NY2 = (REG == 0) ? 4 : 2;
    double ss[1] = {0.0}, mm[1] = {0.0};
    for (int base = blockIdx.x * CP_BLOCK; base < P.npix; base += gridDim.x * CP_BLOCK) {
        const int q = base + threadIdx.x;
        if (q < P.npix) {
            const int j = q / m, i = q - j * m, e = q + m, jg = P.j_off + j;
            double s3 = 0.0;
            for (int k = 0; k < K; ++k) s3 += y[(size_t)(K + NY2 * K + k) * fs + e];
            const double mean3 = s3 * P.invK;
            for (int k = 0; k < K; ++k) {
                const double y1 = y[(size_t)k * fs + e];
                const double vL = y1 + (y[(size_t)(K + NY2 * K + k) * fs + e] - mean3);
                xo[(size_t)(2 * K + k) * fs + e] = vL;
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
                    vux = -gx * y1 + (adj[0] + adj[1]);
                    vuy = -gy * y1 + (adj[2] + adj[3]);
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
                    vux = -gx * y1 + lap[0];
                    vuy = -gy * y1 + lap[1];
                }
                xo[(size_t)(2 * k) * fs + e] = vux;
                xo[(size_t)(2 * k + 1) * fs + e] = vuy;
                ss[0] += vux * vux + vuy * vuy + vL * vL;
            }
        }
    }
    block_reduce_store<1, 1>(ss, mm, red, part + (size_t)blockIdx.x * CP_PSTRIDE);
}

// deterministic start vector of the power iteration (global MATLAB index -> hash)
__global__ __launch_bounds__(CP_BLOCK) void k_mf_hash_start(MfParams P, int nfields, double *__restrict__ x) {
    const long long tot = (long long)P.npix * nfields;
    for (long long gid = (long long)blockIdx.x * CP_BLOCK + threadIdx.x; gid < tot;
         gid += (long long)gridDim.x * CP_BLOCK) {
        const int f = (int)(gid / P.npix), q = (int)(gid - (long long)f * P.npix);
        const int j = q / P.m, i = q - j * P.m;
        const unsigned long long idx =
            (unsigned long long)f * ((unsigned long long)P.m * P.n_global) + (unsigned long long)(P.j_off + j) * P.m + i;
        const unsigned long long hsh = (idx * 2654435761ULL) & 0xFFFFFFFFULL;
        x[(size_t)f * P.fs + q + P.m] = (double)hsh / 2147483648.0 - 1.0;
    }
}
