// NN-curvature (NNE/mf_nn_curvature_registration_no_ref_ml.m) in two field passes per Chambolle-Pock iteration.
//
// The 5-point Laplacian sits on both sides of the iteration (A2 = discrete_laplacian, symmetric), so the one-kernel
// recomputation of k_mf_fused_tv would need the primal step at 5 positions and y2 at 13. The iteration is split where
// the stencils are instead:
//   k_mf_fused_curv(n)  primal step n -- u_n from the Laplacian of y2_n (read-only here), L_n -- and everything of dual
//                       step n+1 that is pointwise in xbar_n: y1_{n+1} (SAD*), y3~_{n+1} with both Gram matrices.
//                       reads u(2) L y1 y2(2) y3 gradT(2) b, writes u(2) L y1 y3 = 15 elements per pixel-image
//   k_mf_efused(n)      reduction, eigensolves, W_{n+1}, history rows (cp_mfnn_fused.cuh, reg = 1)
//   k_curv_y2(n)        y2_{n+1} = prox_{sigma F2*}(y2_n + sigma*Lap(ubar_n)), F2(A x_n), F2*(y2_{n+1}).
//                       reads u_n(2) u_{n-1}(2) y2(2), writes y2(2) = 8 elements
//   k_curv_fin(n)       F2 / F2* into the history rows, mean of the next un-projected u, stop rule
// 23 elements instead of the 29 of the split passes (k_mf_dual + k_mf_primal), and no position is recomputed.
// The mean of the next u follows as in k_mf_fused_tv with one more term: mean(Lap y2) is not zero under Dirichlet rows,
// but equals sum_q y2(q) * rowsum_q(Lap) / N (the Laplacian is symmetric), accumulated over the boundary pixels while
// y2_{n+1} is written.
#pragma once
#include "cp_mfnn_fused.cuh"
#include "cp_var.cuh"   // reduce_scalars



// Same organisation as k_mf_fused_tv: a wave marches down one image column in segments of 64 rows, four adjacent
// columns per workgroup, no workgroup barrier in the loop. y2_n at (i+1, j) is lane+1's own value (lane 63: the value
// lane 0 loaded one step earlier, through LDS; at the top of a part a load in lane 63 alone), at (i-1, j) one
// element-shifted load, at (i, j-1) and (i, j+1) the lines the neighbouring waves stream.
// Round 3: the per-image body is one basic block of buffer loads and range-dropped stores (below); NWV = 8 adjacent columns per
// workgroup (one workgroup per CU, CP_TUNE_CURV_WAVES=8) would halve the (i, j +- 1) lines that belong to another workgroup but
// measured slower (6.25 vs 6.0 ms); (i-1, j) by cross-lane shift is CP_CURV_SHF (off).
// CP_CURV_SHF=1: (i-1, j) as lane - 1's value (lane_shr1) + one load in the first lane; measured after the per-image body became one
// basic block: 6.03 / 6.58 ms against 5.97 / 6.49 with the element-shifted wave-wide load (the two placement modes of one box,
// 4096^2 x 16) -- the shifted load is an L1 hit on lines the wave fetches anyway and costs less than the cross-lane moves. Default 0.
#ifndef CP_CURV_SHF
#define CP_CURV_SHF 0
#endif
template <typename S, bool NEUMANN, bool FULLK, int KA = CP_KMAX, int NWV = CP_WAVES>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void k_mf_fused_curv(MfParams P, const S *__restrict__ xo, S *__restrict__ xn,
                                                                const S *__restrict__ pin, S *__restrict__ pout,
                                                                const S *__restrict__ g, const S *__restrict__ b,
                                                                const double *__restrict__ W,
                                                                const CpCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double tiles[NWV * 2 * 16 * CP_TS];
    __shared__ double Wl[CP_KMAX * CP_KMAX];
    __shared__ double carry_all[NWV * 2 * CP_KMAX * 2];   // [wave][step parity][image][y2 component] of lane 0
    const int stopped = ctrl->stop;
    const double cnx = ctrl->cn[0], cny = ctrl->cn[1];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *tileA = tiles + wave * 2 * 16 * CP_TS, *tileB = tileA + 16 * CP_TS;
    double *carry = carry_all + wave * 2 * CP_KMAX * 2;
    const int K = P.K, m = P.m;
    const size_t fs = (size_t)P.fs;
    stage_W(W, Wl);
    if (stopped) return;
    if (KA < CP_KMAX) {
        for (int k = KA; k < CP_KMAX; ++k) tileA[k * CP_TS + lane] = 0.0;
    }
    __syncthreads();
    const double sigma = P.sigma, tau = P.tau, theta = P.theta, inv_sigma = 1.0 / P.sigma;
    const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
    const S *y3in = pin + 3 * (size_t)K * fs;
    S *y3out = pout + 3 * (size_t)K * fs;
    const S *Lin = xo + 2 * (size_t)K * fs;
    S *Lout = xn + 2 * (size_t)K * fs;
    constexpr unsigned ES = (unsigned)sizeof(S);
    const unsigned mb = ES * (unsigned)m;

    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = -1.0e300;
    maxs[DM_Y1 - CP_NSUM] = 0.0; maxs[DM_S1 - CP_NSUM] = 0.0; maxs[PM_VL - CP_NSUM] = 0.0;
    double4_t acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};

    const Fused2Geom GM = fused2_geom(m);
    const int ntasks = GM.parts * ((P.nl + NWV - 1) / NWV);
    const unsigned off_l0 = (lane == 0) ? 0u : 0xFFFFFFFFu;       // only the segment's first lane loads (i-1, j)
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
            const bool first_i = (i == 0), last_i = (i == m - 1);

            fused_phase1<S, FULLK, KA>(P, K, valid, off, lane, fs, y3in, pin, Lin, y3out, Lout, Wl, tileA, tileB, sigma, tau,
                                       theta, maxs[PM_VL - CP_NSUM], acc1, acc2);

            // ---- per image: primal step of u at the own pixel (Laplacian of y2_n), y1_{n+1} ---------------------------------
            const bool want = valid && !last_i;                    // y2_n at (i+1, j) is used
            const bool from_carry = want && lane == 63 && sg < seg_hi;
            const bool reload = want && lane == 63 && sg == seg_hi;
            const double *cin = carry + (par ^ 1) * CP_KMAX * 2;
            double *cout_ = carry + par * CP_KMAX * 2;
            const double vmask = valid ? 1.0 : 0.0;
            // Round 3: the per-image body is ONE basic block. Every load of the image is issued first, through buffer descriptors
            // (uniform field base in SGPRs + one 32-bit lane offset; the second field of a pair through the uniform offset), and
            // the stores are unconditional: positions that do not exist -- lanes past the end of the column, (i-1, j) in all
            // lanes but the first, the one-lane (i+1, j) load at the top of a part -- carry the offset 2^32-1 and are dropped by
            // the range check. Before, the second y2 component was loaded after the first had been consumed and every image
            // ended in a full wait that included its own stores (branches around memory operations hide the counts from the
            // wait-count pass): three dependent round trips per image.
            constexpr unsigned kNone = 0xFFFFFFFFu;
            const unsigned offv = valid ? off : kNone;
            const unsigned off_re = reload ? off + ES : kNone;
            const unsigned fsb = (unsigned)(fs * ES);               // fields of the multi-image models stay below 2 GiB (cp_create)
            for (int k = 0; k < K; ++k) {
                const cp_rsrc r_y1 = mk_rsrc(pin + (size_t)k * fs), r_y2 = mk_rsrc(pin + ((size_t)K + 2 * (size_t)k) * fs);
                const cp_rsrc r_g = mk_rsrc(g + 2 * (size_t)k * fs), r_b = mk_rsrc(b + (size_t)k * fs);
                const cp_rsrc r_xo = mk_rsrc(xo + 2 * (size_t)k * fs), r_xn = mk_rsrc(xn + 2 * (size_t)k * fs);
                const cp_rsrc r_y1o = mk_rsrc(pout + (size_t)k * fs);
                const S q_y1 = bld_raw<S>(r_y1, off), q_gx = bld_raw<S>(r_g, off), q_gy = bld_raw<S>(r_g, off, fsb);
                const S q_b = bld_raw_stream<S>(r_b, off);
                const S q_uxo = bld_raw_stream<S>(r_xo, off), q_uyo = bld_raw_stream<S>(r_xo, off, fsb);
                S q_w0[2], q_wim[2], q_wjm[2], q_wjp[2], q_wre[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned so = c ? fsb : 0u;
                    q_w0[c] = bld_raw<S>(r_y2, off, so);
#if CP_CURV_SHF
                    q_wim[c] = bld_raw<S>(r_y2, (off - ES) | off_l0, so);   // lanes 1..63: out of range, dropped
#else
                    q_wim[c] = bld_raw<S>(r_y2, off - ES, so);
#endif
                    q_wjm[c] = bld_raw<S>(r_y2, off - mb, so);
                    q_wjp[c] = bld_raw<S>(r_y2, off + mb, so);
                    q_wre[c] = bld_raw<S>(r_y2, off_re, so);                // (i+1, j) for lane 63 at the top of a part
                }
                if (sizeof(S) == 4) __builtin_amdgcn_sched_barrier(0);
                const double Lc = tileA[k * CP_TS + lane], Lbar = tileB[k * CP_TS + lane];
                const double y1 = (double)q_y1, gx = (double)q_gx, gy = (double)q_gy, bb = (double)q_b;
                const double uxo_ = (double)q_uxo, uyo_ = (double)q_uyo;
                double lap[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double w0 = (double)q_w0[c];
#if CP_CURV_SHF
                    const double wsh = lane_shr1<double>(w0);         // evaluated by every lane, before the select
                    const double wim = (lane == 0) ? (double)q_wim[c] : wsh;
#else
                    const double wim = (double)q_wim[c];
#endif
                    const double wjm = (double)q_wjm[c], wjp = (double)q_wjp[c];
                    const double cv = cin[2 * k + c];                 // uniform address: one broadcast read
                    double wip = __shfl(w0, lane + 1);
                    wip = from_carry ? cv : wip;
                    wip = reload ? (double)q_wre[c] : wip;
                    if (lane == 0) cout_[2 * k + c] = w0;
                    // NNE/discrete_laplacian.m:23-46: a missing neighbour is the centre value (neumann) or zero (dirichlet)
                    const double a0 = !first_i ? wim : (NEUMANN ? w0 : 0.0);
                    const double a1 = !last_i ? wip : (NEUMANN ? w0 : 0.0);
                    const double a2 = !first_j ? wjm : (NEUMANN ? w0 : 0.0);
                    const double a3 = !last_j ? wjp : (NEUMANN ? w0 : 0.0);
                    lap[c] = (a0 + a1 - 2.0 * w0) * ih1 + (a2 + a3 - 2.0 * w0) * ih2;
                }
                const double vux = -gx * y1 + lap[0];              // (A'y)_u = -grad T' y1 + Lap y2
                const double vuy = -gy * y1 + lap[1];
                const double ux = rnd<S>((uxo_ - tau * vux) - cnx);    // NNE/mean_zero_indicator.m:50,53
                const double uy = rnd<S>((uyo_ - tau * vuy) - cny);
                const double wx = -vux, wy = -vuy;
                bst<S>(r_xn, offv, 0u, ux);
                bst<S>(r_xn, offv, fsb, uy);
                const double big = -1.0e300;
                maxs[PM_WXMAX - CP_NSUM] = fmax(maxs[PM_WXMAX - CP_NSUM], valid ? wx : big);     // NNE/mean_zero_indicator.m:78-84
                maxs[PM_WXMIN - CP_NSUM] = fmax(maxs[PM_WXMIN - CP_NSUM], valid ? -wx : big);
                maxs[PM_WYMAX - CP_NSUM] = fmax(maxs[PM_WYMAX - CP_NSUM], valid ? wy : big);
                maxs[PM_WYMIN - CP_NSUM] = fmax(maxs[PM_WYMIN - CP_NSUM], valid ? -wy : big);
                sums[PS_WX] += vmask * wx;
                const double uxb = ux + theta * (ux - uxo_), uyb = uy + theta * (uy - uyo_);
                // ---- y1: SAD* (identical to k_mf_dual) ----------------------------------------------------------------
                const double r_cur = -(gx * ux + gy * uy) + Lc;
                const double r_bar = -(gx * uxb + gy * uyb) + Lbar;
                sums[DS_F1] += vmask * fabs(r_cur - bb);                              // NNE/SAD.m:54
                const double yt = y1 + sigma * r_bar;
                const double Ls = yt * inv_sigma, thr = inv_sigma * P.mu1, dd = Ls - bb;
                const double prox = (dd > thr) ? (Ls - thr) : ((dd < -thr) ? (Ls + thr) : bb);   // :42-49
                const double y1n = rnd<S>(yt - sigma * prox);                         // :71-72
                sums[DS_F1S] += vmask * (y1n * bb);                                   // :83
                sums[DS_SGX] += vmask * (gx * y1n);
                sums[DS_SGY] += vmask * (gy * y1n);
                bst<S>(r_y1o, offv, 0u, y1n);
                maxs[DM_Y1 - CP_NSUM] = fmax(maxs[DM_Y1 - CP_NSUM], valid ? fabs(y1n) : 0.0);
                sums[DS_UX] += vmask * ux;
                sums[DS_UY] += vmask * uy;
            }
            wave_lds_sync();
        }
    }
    double *out = part + (size_t)blockIdx.x * CP_PSTRIDE;
    {
        constexpr int WS = 2 * 16 * CP_TS;
        double *st = tiles + wave * WS;
        wave_lds_sync();
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = ((lane >> 4) + 4 * r) * 16 + (lane & 15);
            st[CP_NSCAL + e] = acc1[r];
            st[CP_NSCAL + 256 + e] = acc2[r];
        }
        __syncthreads();
        for (int e = tid; e < CP_PSTRIDE; e += NWV * 64) {
            const bool is_max = (e >= CP_NSUM && e < CP_NSCAL);
            double v = tiles[e];
#pragma unroll
            for (int w = 1; w < NWV; ++w) {
                const double o = tiles[w * WS + e];
                v = is_max ? fmax(v, o) : v + o;
            }
            out[e] = v;
        }
    }
}

// ---- k_curv_y2: y2_{n+1} = prox_{sigma F2*}(y2_n + sigma*Lap(ubar_n)), ubar_n = u_n + theta*(u_n - u_{n-1}) -----------------
// The 2K fields do not couple here. A wave owns 64 rows and marches along j through a strip of columns, one field at a
// time, with a three-column window of u_n and u_{n-1} in registers: every element is loaded once by its wave; (i +- 1, j)
// are lane +- 1's window values (cross-lane shift), lanes 0 and 63 load the two rows outside the segment. The four waves of
// a workgroup are four vertically adjacent segments (256 rows), so three of every four of those rows are lines the
// neighbouring wave streams at the same moment. Tasks: (row block, column strip, field group).
// The march issues the loads of CP_Y2_CB columns as one group through buffer descriptors (3 full loads and 4 one-lane
// loads per column; positions that do not exist are out of the descriptor's range) and stores unconditionally the same way:
// the first version loaded one column per step behind a branch and a full wait (one round trip per column, 7.8 MB in
// flight on the chip: 4.06 ms at 4096^2 x 16); groups of 4 and tasks of 4 fields (16384 tasks, no tail of half-empty
// CUs): 3.55 ms = 5.2 TB/s of the 18.4 GB it moves (groups of 2: 3.71, of 8: 3.62).
#ifndef CP_Y2_CB
#define CP_Y2_CB 4          // columns whose loads are issued together
#endif
#ifndef CP_Y2_MINTASKS
#define CP_Y2_MINTASKS 16384
#endif
struct CurvY2Geom { int rblocks, strips, strips8, strip_w, fgroups, fields_per_group; };
__host__ __device__ inline CurvY2Geom curv_y2_geom(int m, int nl, int K, int rows_per_block = 256) {
    CurvY2Geom G;
    G.rblocks = (m + rows_per_block - 1) / rows_per_block;
#ifndef CP_Y2_STRIP
#define CP_Y2_STRIP 32
#endif
    G.strip_w = nl < CP_Y2_STRIP ? nl : CP_Y2_STRIP;
    G.strips = (nl + G.strip_w - 1) / G.strip_w;
    G.fgroups = 1;
    while ((long long)G.rblocks * G.strips * G.fgroups < CP_Y2_MINTASKS && G.fgroups < 2 * K) G.fgroups *= 2;
    G.fields_per_group = (2 * K + G.fgroups - 1) / G.fgroups;
    G.fgroups = (2 * K + G.fields_per_group - 1) / G.fields_per_group;
    G.strips8 = (G.strips + 7) / 8;
    return G;
}
// task ids: strip index mod 8 fastest (= the XCD, workgroups are dealt round robin), then the row block: vertically
// adjacent row blocks of one strip are consecutive workgroups of one XCD and share the rows between them in its L2
__host__ __device__ inline long long curv_y2_tasks(int m, int nl, int K, int rows_per_block = 256) {
    const CurvY2Geom G = curv_y2_geom(m, nl, K, rows_per_block);
    return 8LL * G.rblocks * G.strips8 * G.fgroups;
}

#ifndef CP_Y2_WPE
#define CP_Y2_WPE 1         // waves per SIMD the register allocation has to leave room for (launch bounds)
#endif
template <typename S, bool NEUMANN, int CB = CP_Y2_CB>
__global__ __launch_bounds__(CP_BLOCK, CP_Y2_WPE) void k_curv_y2(MfParams P, const S *__restrict__ xc, const S *__restrict__ xo,
                                                      const S *__restrict__ pin, S *__restrict__ pout,
                                                      const CpCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double red[CP_WAVES * CP_NSCAL];
    if (ctrl->stop) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = P.K, m = P.m;
    const size_t fs = (size_t)P.fs;
    const double sigma = P.sigma, theta = P.theta, inv_sigma = 1.0 / P.sigma;
    const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
    const double den = 1.0 + P.mu2 * inv_sigma;
    constexpr unsigned ES = (unsigned)sizeof(S), kNone = 0xFFFFFFFFu;
    const unsigned mb = ES * (unsigned)m;
    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = 0.0;

    const CurvY2Geom GM = curv_y2_geom(m, P.nl, K);
    const int ntasks = 8 * GM.rblocks * GM.strips8 * GM.fgroups;
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        int q = task >> 3;
        const int rb = q % GM.rblocks;
        q /= GM.rblocks;
        const int cs = (q % GM.strips8) * 8 + (task & 7), fg = q / GM.strips8;
        const int ii = rb * 256 + wave * 64 + lane;
        if (cs >= GM.strips || rb * 256 + wave * 64 >= m) continue;   // wave-uniform
        const bool valid = ii < m;
        const int i = min(ii, m - 1);
        const bool first_i = (i == 0), last_i = (i == m - 1);
        const int j0 = cs * GM.strip_w, j1 = min(j0 + GM.strip_w, P.nl);
        const int f0 = fg * GM.fields_per_group, f1 = min(f0 + GM.fields_per_group, 2 * K);
        // lanes 0 and 63 fetch the rows just outside the segment (if they exist); a lane past the end of the image
        // never needs one (its value is not used)
        const bool edge_dn = (lane == 0) && !first_i, edge_up = (lane == 63) && valid && !last_i;
        for (int f = f0; f < f1; ++f) {
            // one descriptor per field (uniform: SALU), one 32-bit offset per position; positions that do not exist (columns
            // past the strip, the outside rows of lanes 1..62, lanes past the end of a column for the store) carry the
            // offset 2^32-1 and the range check drops them: no branch around a memory operation, exact wait counts
            const cp_rsrc r_un = mk_rsrc(xc + (size_t)f * fs), r_uo = mk_rsrc(xo + (size_t)f * fs);
            const cp_rsrc r_yi = mk_rsrc(pin + ((size_t)K + f) * fs), r_yo = mk_rsrc(pout + ((size_t)K + f) * fs);
            unsigned off = ES * (unsigned)((j0 + 1) * m + i);     // (i, j0)
            double an_m = bld<S>(r_un, off - mb), ao_m = bld<S>(r_uo, off - mb);   // column j0 - 1 (halo column at the slab's edge)
            double an_c = bld_stream<S>(r_un, off), ao_c = bld_stream<S>(r_uo, off);
            for (int jb = j0; jb < j1; jb += CB, off += CB * mb) {
                // all loads of CB columns first: column j+1 of u_n, u_{n-1} (the last one of the block with a plain load:
                // at the end of a strip it is the first column of the next strip), y2 at j, and the four outside rows
                S rp_n[CB], rp_o[CB], rv[CB], ru_n[CB], ru_o[CB], rd_n[CB], rd_o[CB];
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const bool in = jb + c < j1;                  // wave-uniform
                    const unsigned o = off + (unsigned)c * mb;
                    const unsigned o_p = in ? o + mb : kNone, o_c = in ? o : kNone;
                    if (c + 1 < CB) { rp_n[c] = bld_raw_stream<S>(r_un, o_p); rp_o[c] = bld_raw_stream<S>(r_uo, o_p); }
                    else { rp_n[c] = bld_raw<S>(r_un, o_p); rp_o[c] = bld_raw<S>(r_uo, o_p); }
                    rv[c] = bld_raw_stream<S>(r_yi, o_c);
                    const unsigned o_u = (in && edge_up) ? o + ES : kNone, o_d = (in && edge_dn) ? o - ES : kNone;
                    ru_n[c] = bld_raw<S>(r_un, o_u); ru_o[c] = bld_raw<S>(r_uo, o_u);
                    rd_n[c] = bld_raw<S>(r_un, o_d); rd_o[c] = bld_raw<S>(r_uo, o_d);
                }
                if (sizeof(S) == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const int j = jb + c;
                    const bool in = j < j1;
                    const int jg = P.j_off + j;
                    const bool first_j = (jg == 0), last_j = (jg == P.n_global - 1);
                    const double an_p = (double)rp_n[c], ao_p = (double)rp_o[c], v = (double)rv[c];
                    double an_u = __shfl(an_c, lane + 1), ao_u = __shfl(ao_c, lane + 1);   // (i+1, j)
                    double an_d = __shfl(an_c, lane - 1), ao_d = __shfl(ao_c, lane - 1);   // (i-1, j)
                    if (edge_up) { an_u = (double)ru_n[c]; ao_u = (double)ru_o[c]; }
                    if (edge_dn) { an_d = (double)rd_n[c]; ao_d = (double)rd_o[c]; }
                    // as k_mf_dual: xbar at the five points first, then the Laplacian (NNE/discrete_laplacian.m:23-46)
                    const double uc0 = an_c, ub0 = an_c + theta * (an_c - ao_c);
                    const bool have[4] = {!first_i, !last_i, !first_j, !last_j};
                    const double na[4] = {an_d, an_u, an_m, an_p}, no[4] = {ao_d, ao_u, ao_m, ao_p};
                    double nb[4], nbb[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        nb[d] = have[d] ? na[d] : (NEUMANN ? uc0 : 0.0);
                        nbb[d] = have[d] ? na[d] + theta * (na[d] - no[d]) : (NEUMANN ? ub0 : 0.0);
                    }
                    const double lapc = (nb[0] + nb[1] - 2.0 * uc0) * ih1 + (nb[2] + nb[3] - 2.0 * uc0) * ih2;
                    const double lapb = (nbb[0] + nbb[1] - 2.0 * ub0) * ih1 + (nbb[2] + nbb[3] - 2.0 * ub0) * ih2;
                    const double t0 = v + sigma * lapb;
                    const double n0 = rnd<S>(t0 - sigma * ((t0 * inv_sigma) / den));      // NNE/SSD.m:60-61 with :43 (g = 0)
                    const bool live = valid && in;
                    bst<S>(r_yo, live ? off + (unsigned)c * mb : kNone, 0u, n0);
                    if (live) {
                        sums[DS_F2] += lapc * lapc;                                       // NNE/SSD.m:48 (g = 0)
                        const double z0 = n0 / P.mu2;
                        sums[DS_F2S] += z0 * z0;                                          // NNE/SSD.m:66-67
                        const double rs = lap_rowsum(NEUMANN, first_i, last_i, first_j, last_j, ih1, ih2);
                        if (f & 1) sums[DS_BSY] += n0 * rs; else sums[DS_BSX] += n0 * rs;
                    }
                    an_m = an_c; ao_m = ao_c; an_c = an_p; ao_c = ao_p;
                }
            }
        }
    }
    block_reduce_store<CP_NSUM, CP_NMAX>(sums, maxs, red, part + (size_t)blockIdx.x * CP_PSTRIDE);
}

// ---- k_curv_y2_r2: the same pass with TWO rows per lane (m even) --------------------------------------------------------------
// A lane owns rows (i, i+1), a wave 128 rows, a workgroup 512: every wave-wide access is one 1 KB run of 16 bytes per lane in
// fp64 storage (8 in fp32), the access shape the streaming copy needs for its 6.5 TB/s (8 bytes per lane: 5.7), and the two rows
// outside a segment are fetched once per 128 rows instead of once per 64. (i+1) of the lane's first row and (i-1) of its second
// are its own values; the other two come from lane +- 1 (lanes 0 and 63: one-lane loads). Arithmetic per pixel is that of
// k_curv_y2, expression by expression.
#ifndef CP_Y2R2_CB
#define CP_Y2R2_CB 4
#endif
template <typename S, bool NEUMANN, int CB = CP_Y2R2_CB>
__global__ __launch_bounds__(CP_BLOCK) void k_curv_y2_r2(MfParams P, const S *__restrict__ xc, const S *__restrict__ xo,
                                                         const S *__restrict__ pin, S *__restrict__ pout,
                                                         const CpCtrl *__restrict__ ctrl, double *__restrict__ part) {
    __shared__ double red[CP_WAVES * CP_NSCAL];
    if (ctrl->stop) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = P.K, m = P.m;
    const size_t fs = (size_t)P.fs;
    const double sigma = P.sigma, theta = P.theta, inv_sigma = 1.0 / P.sigma;
    const double ih1 = P.inv_h1 * P.inv_h1, ih2 = P.inv_h2 * P.inv_h2;
    const double den = 1.0 + P.mu2 * inv_sigma;
    constexpr unsigned ES = (unsigned)sizeof(S), kNone = 0xFFFFFFFFu;
    const unsigned mb = ES * (unsigned)m;
    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = 0.0;

    const CurvY2Geom GM = curv_y2_geom(m, P.nl, K, 512);
    const int ntasks = 8 * GM.rblocks * GM.strips8 * GM.fgroups;
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        int q = task >> 3;
        const int rb = q % GM.rblocks;
        q /= GM.rblocks;
        const int cs = (q % GM.strips8) * 8 + (task & 7), fg = q / GM.strips8;
        const int ii = rb * 512 + wave * 128 + 2 * lane;             // rows ii, ii + 1 (m is even: both exist or neither)
        if (cs >= GM.strips || rb * 512 + wave * 128 >= m) continue;   // wave-uniform
        const bool valid = ii < m;
        const int i = min(ii, m - 2);
        const bool first_i = (i == 0), last_i = (i + 1 == m - 1);    // of the lane's first / second row
        const int j0 = cs * GM.strip_w, j1 = min(j0 + GM.strip_w, P.nl);
        const int f0 = fg * GM.fields_per_group, f1 = min(f0 + GM.fields_per_group, 2 * K);
        const bool edge_dn = (lane == 0) && !first_i, edge_up = (lane == 63) && valid && !last_i;
        for (int f = f0; f < f1; ++f) {
            const cp_rsrc r_un = mk_rsrc(xc + (size_t)f * fs), r_uo = mk_rsrc(xo + (size_t)f * fs);
            const cp_rsrc r_yi = mk_rsrc(pin + ((size_t)K + f) * fs), r_yo = mk_rsrc(pout + ((size_t)K + f) * fs);
            unsigned off = ES * (unsigned)((j0 + 1) * m + i);     // (i, j0)
            cp_pair<S> t;
            double an_m[2], ao_m[2], an_c[2], ao_c[2];
            t = bld2_raw<S, 0>(r_un, off - mb); an_m[0] = (double)t.a; an_m[1] = (double)t.b;   // column j0 - 1
            t = bld2_raw<S, 0>(r_uo, off - mb); ao_m[0] = (double)t.a; ao_m[1] = (double)t.b;
            t = bld2_raw<S, 2>(r_un, off); an_c[0] = (double)t.a; an_c[1] = (double)t.b;
            t = bld2_raw<S, 2>(r_uo, off); ao_c[0] = (double)t.a; ao_c[1] = (double)t.b;
            for (int jb = j0; jb < j1; jb += CB, off += CB * mb) {
                cp_pair<S> rp_n[CB], rp_o[CB], rv[CB];
                S ru_n[CB], ru_o[CB], rd_n[CB], rd_o[CB];
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const bool in = jb + c < j1;                  // wave-uniform
                    const unsigned o = off + (unsigned)c * mb;
                    const unsigned o_p = in ? o + mb : kNone, o_c = in ? o : kNone;
                    if (c + 1 < CB) { rp_n[c] = bld2_raw<S, 2>(r_un, o_p); rp_o[c] = bld2_raw<S, 2>(r_uo, o_p); }
                    else { rp_n[c] = bld2_raw<S, 0>(r_un, o_p); rp_o[c] = bld2_raw<S, 0>(r_uo, o_p); }
                    rv[c] = bld2_raw<S, 2>(r_yi, o_c);
                    const unsigned o_u = (in && edge_up) ? o + 2 * ES : kNone, o_d = (in && edge_dn) ? o - ES : kNone;
                    ru_n[c] = bld_raw<S>(r_un, o_u); ru_o[c] = bld_raw<S>(r_uo, o_u);
                    rd_n[c] = bld_raw<S>(r_un, o_d); rd_o[c] = bld_raw<S>(r_uo, o_d);
                }
                if (sizeof(S) == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const int j = jb + c;
                    const bool in = j < j1;
                    const int jg = P.j_off + j;
                    const bool first_j = (jg == 0), last_j = (jg == P.n_global - 1);
                    const double an_p[2] = {(double)rp_n[c].a, (double)rp_n[c].b}, ao_p[2] = {(double)rp_o[c].a, (double)rp_o[c].b};
                    const double v[2] = {(double)rv[c].a, (double)rv[c].b};
                    double an_u = __shfl(an_c[0], lane + 1), ao_u = __shfl(ao_c[0], lane + 1);   // (i + 2, j): above the second row
                    double an_d = __shfl(an_c[1], lane - 1), ao_d = __shfl(ao_c[1], lane - 1);   // (i - 1, j): below the first row
                    if (edge_up) { an_u = (double)ru_n[c]; ao_u = (double)ru_o[c]; }
                    if (edge_dn) { an_d = (double)rd_n[c]; ao_d = (double)rd_o[c]; }
                    double n0[2];
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const bool fi = (r == 0) && first_i, li = (r == 1) && last_i;
                        const double uc0 = an_c[r], ub0 = an_c[r] + theta * (an_c[r] - ao_c[r]);
                        const bool have[4] = {!fi, !li, !first_j, !last_j};
                        const double na[4] = {r == 0 ? an_d : an_c[0], r == 0 ? an_c[1] : an_u, an_m[r], an_p[r]};
                        const double no[4] = {r == 0 ? ao_d : ao_c[0], r == 0 ? ao_c[1] : ao_u, ao_m[r], ao_p[r]};
                        double nb[4], nbb[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            nb[d] = have[d] ? na[d] : (NEUMANN ? uc0 : 0.0);
                            nbb[d] = have[d] ? na[d] + theta * (na[d] - no[d]) : (NEUMANN ? ub0 : 0.0);
                        }
                        const double lapc = (nb[0] + nb[1] - 2.0 * uc0) * ih1 + (nb[2] + nb[3] - 2.0 * uc0) * ih2;
                        const double lapb = (nbb[0] + nbb[1] - 2.0 * ub0) * ih1 + (nbb[2] + nbb[3] - 2.0 * ub0) * ih2;
                        const double t0 = v[r] + sigma * lapb;
                        n0[r] = rnd<S>(t0 - sigma * ((t0 * inv_sigma) / den));               // NNE/SSD.m:60-61 with :43 (g = 0)
                        if (valid && in) {
                            sums[DS_F2] += lapc * lapc;                                       // NNE/SSD.m:48 (g = 0)
                            const double z0 = n0[r] / P.mu2;
                            sums[DS_F2S] += z0 * z0;                                          // NNE/SSD.m:66-67
                            const double rs = lap_rowsum(NEUMANN, fi, li, first_j, last_j, ih1, ih2);
                            if (f & 1) sums[DS_BSY] += n0[r] * rs; else sums[DS_BSX] += n0[r] * rs;
                        }
                    }
                    bst2<S>(r_yo, (valid && in) ? off + (unsigned)c * mb : kNone, n0[0], n0[1]);
#pragma unroll
                    for (int r = 0; r < 2; ++r) { an_m[r] = an_c[r]; ao_m[r] = ao_c[r]; an_c[r] = an_p[r]; ao_c[r] = ao_p[r]; }
                }
            }
        }
    }
    block_reduce_store<CP_NSUM, CP_NMAX>(sums, maxs, red, part + (size_t)blockIdx.x * CP_PSTRIDE);
}

struct CurvFinArgs {
    int nblk, n, check_stop;
    double tol, mu2, tau, Ntot;
};

// after k_curv_y2(n): F2(A x_n) completes primal row n, F2*(y2_{n+1}) goes into dual row n+1, the boundary term of the
// mean of the next un-projected u, and the stop rule for i = n (PDO/chambolle_pock.m:142-153; every entry of rows n-4..n
// is final now: k_mf_efused(n) ran before on the same stream)
__global__ __launch_bounds__(256) void k_curv_fin(CurvFinArgs A, const double *__restrict__ part, CpCtrl *ctrl,
                                                  double *hist_p, double *hist_d) {
    __shared__ double sc[CP_NSCAL], stage[256];
    if (ctrl->stop) return;
    reduce_scalars(part, A.nblk, sc, stage);
    if (threadIdx.x != 0) return;
    const int n = A.n;
    double *hp = hist_p + (size_t)n * CP_HCOLS_MF, *hd1 = hist_d + (size_t)(n + 1) * CP_HCOLS_MF;
    const double F2 = A.mu2 * 0.5 * sc[DS_F2];
    hp[2] = F2;
    hp[0] = hp[4] + (hp[1] + F2 + 0.0);                        // chambolle_pock.m:170-171
    hd1[2] = A.mu2 * (0.5 * sc[DS_F2S]);                       // NNE/SSD.m:66-67 (g = 0)
    ctrl->cn[0] -= A.tau * (sc[DS_BSX] / A.Ntot);              // - tau * mean(Lap y2_{n+1}), zero for neumann rows
    ctrl->cn[1] -= A.tau * (sc[DS_BSY] / A.Ntot);
    if (A.check_stop && n >= 4) {
        bool ok = true;
        for (int r = n - 4; r <= n; ++r) {
            const double *p_ = hist_p + (size_t)r * CP_HCOLS_MF, *d_ = hist_d + (size_t)r * CP_HCOLS_MF;
            const double gap = fabs((p_[0] - d_[0]) / d_[0]);
            if (!(gap <= A.tol) || !(p_[5] <= A.tol) || !(p_[6] <= A.tol) || !(d_[5] <= A.tol) || !(d_[6] <= A.tol)) ok = false;
        }
        if (ok) { ctrl->stop = 1; ctrl->stop_iter = n; }
    }
}
