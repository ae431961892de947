// NN-TV: one kernel per Chambolle-Pock iteration.
//
// k_mf_fused_tv(n) = primal step n followed by dual step n+1 of PDO/chambolle_pock.m:162-165 for the model of
// NNE/mf_nn_tv_registration_no_ref_ml.m, per pixel, with the state read and written exactly once:
//     reads  x_{n-1} (3), y_n (6), grad T (2), b (1)          writes x_n (3), y_{n+1} (6)     [per pixel-image]
// i.e. 21 elements instead of the 35 the separate passes move (the dual pass no longer re-reads x and x_old).
//
// What makes the fusion legal:
//   * the dual step at (i,j) needs x_n at (i,j), (i+1,j), (i,j+1) (forward differences). The displacement part of
//     the primal step is cheap (adjoint stencil + one fma), so each thread recomputes it at its two neighbours
//     from y_n, which is never written by this kernel -- no inter-thread exchange, no halo tiles;
//   * the mean-zero projection (NNE/mean_zero_indicator.m:50-53) subtracts mean(u - tau*(A'y)_u). Since D*1 = 0
//     for both boundary conditions, mean((A'y)_u) = -mean(grad T' y1), which the PREVIOUS dual step accumulates
//     while it writes y1; together with the measured mean of the previous u this gives the mean before the pass
//     starts (ctrl->cn), identical to the reference up to rounding and without drift;
//   * the nuclear-norm prox of y3 stays lazy (y3 = y3~ * W, W from k_mf_efused of the previous iteration).
// k_mf_efused: reduction of the partials, eig of Gram(y3~_{n+1}) -> W_{n+1}, eig of Gram(B*L_n) -> feasibility,
// history rows n (primal and dual complete), F* part of row n+1, mean for the next pass, stop rule for i = n.
#pragma once
#include "cp_mfnn_small.cuh"

// u_n = (u_{n-1} - tau*(A'y_n)_u) - mean  and  u_{n-1} at a position (i, jg) of one image; w = -(A'y_n)_u for the G* monitors.
// All loads are unconditional (every field has a halo column on both sides, so the +-1 element and +-1 column neighbours
// are always inside the allocation); boundary handling happens in the arithmetic. This keeps the whole per-image body one
// basic block, so every load of the positions is issued before the first wait.
// Addressing goes through buffer descriptors (cp_device.cuh: bld): the resources of one image's field groups; the
// evaluation returns the values the own-pixel position loads anyway, so the dual step does not load them a second time.
struct ImgRsrc { cp_rsrc y1, y2a, y2b, g, u; };     // y1_k | y2_k components 0,1 | components 2,3 | grad T_k (x, y) | u_k (x, y)
struct PrimalAt { double ux, uy, uxo, uyo, wx, wy, y1, gx, gy, v0, v1, v2, v3; };
// what one evaluation loads: y1, grad T, the four y2 components at the position and at their adjoint neighbours, u_{n-1}
template <typename S>
struct ImgLoads { S y1, gx, gy, v0, v1, v2, v3, w0m, w1m, w2m, w3m, uxo, uyo, l0, l1, l2, l3; };   // as stored (see bld_raw)
#ifndef CP_FUSED_SHF
#define CP_FUSED_SHF 1
#endif
#define CP_BND_BLOCK 32     // values per (segment, column) block of the boundary array: [Dx u_x | Dx u_y][CP_KMAX images]
// element index of the block of segment sg (row 64 sg - 1), storage column jc = j + 1 (0 and nl + 1: halo columns)
__host__ __device__ inline size_t bnd_block(int sg, int jc, int nl) { return ((size_t)sg * (size_t)(nl + 2) + (size_t)jc) * CP_BND_BLOCK; }
__host__ __device__ inline size_t bnd_elems(int m, int nl) { return bnd_block((m + 63) / 64 + 1, 0, nl); }
// value of lane - 1 (lane 0 receives 0): one v_mov_b32 dpp wave_shr:1 per dword, no LDS traffic. Must be called where every
// lane of the wave is active (a DPP read from a disabled lane delivers nothing): never inside one arm of a ?: on a
// lane-dependent condition -- assign the result to a variable first. The empty volatile asm keeps the moves where the call
// is, so that a later select on a lane-dependent condition cannot pull them into its arm either.
template <typename S>
__device__ __forceinline__ S lane_shr1(S v) {
    if constexpr (sizeof(S) == 8) {
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
        asm volatile("" : "+v"(lo), "+v"(hi));
        return __hiloint2double(hi, lo);
    } else {
        int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false);
        asm volatile("" : "+v"(r));
        return __builtin_bit_cast(float, r);
    }
}
// value of lane `src` (wave-uniform index) in every lane: v_readlane_b32 per dword
template <typename S>
__device__ __forceinline__ S lane_bcast(S v, int src) {
    if constexpr (sizeof(S) == 8) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    } else {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
    }
}
// SHF (round 3; lanes = consecutive rows, evaluated by every lane of the wave): the Dx components of y2 at (i-1, j) are the
// values lane - 1 loaded for its own row, so they come from a cross-lane shift; what is left is the row under the
// segment, which only the segment's first lane needs. The two element-shifted wave-wide loads this replaces straddled five
// lines instead of four: 1.2 GB of reads and 0.5 ms per launch at 4096^2 x 16 (profiles/r03_exp_reads.txt) -- and a load in
// the first lane alone still fetches that fifth line (profiles/r03_exp_handover.txt: no gain). So that row comes from a
// BOUNDARY ARRAY instead: for every segment sg and column j one block of 32 values [Dx u_x | Dx u_y][image] of row
// 64 sg - 1, written by the launch that produces y (its lane 63 holds exactly these numbers) and read with ONE 256-byte
// load per wave and step (bnd_block_*): 16 times fewer bytes than the 32 lines of the one-lane loads.
// img_load<SHF>: w0m / w2m are not loaded, the caller passes the boundary values (uniform over the wave).
template <typename S, bool LINEAR, bool SHF = false>
__device__ __forceinline__ void img_load(const MfParams &P, int i, int jg, unsigned oc, unsigned fsb, const ImgRsrc &R,
                                         ImgLoads<S> &q, S bnd0 = (S)0, S bnd2 = (S)0) {
    constexpr unsigned ES = (unsigned)sizeof(S);
    const int m = P.m;
    const unsigned oim = oc - ES, ojm = oc - ES * (unsigned)m;        // (i-1, j) and (i, j-1): inside the allocation (halo column)
    q.y1 = bld_raw<S>(R.y1, oc);
    q.gx = bld_raw<S>(R.g, oc); q.gy = bld_raw<S>(R.g, oc, fsb);
    q.v0 = bld_raw<S>(R.y2a, oc); q.v1 = bld_raw<S>(R.y2a, oc, fsb); q.v2 = bld_raw<S>(R.y2b, oc); q.v3 = bld_raw<S>(R.y2b, oc, fsb);
    if (SHF) { q.w0m = bnd0; q.w2m = bnd2; }                                             // first lane of the segment; others: lane - 1 (img_eval)
    else { q.w0m = bld_raw<S>(R.y2a, oim); q.w2m = bld_raw<S>(R.y2b, oim); }             // Dx components: adjoint along i
    q.w1m = bld_raw<S>(R.y2a, ojm, fsb); q.w3m = bld_raw<S>(R.y2b, ojm, fsb);            // Dy components: adjoint along j
    q.uxo = bld_raw<S>(R.u, oc);
    q.uyo = bld_raw<S>(R.u, oc, fsb);
#ifdef CP_EXP   // timing / traffic experiments only (results wrong): 1 no (i, j-1) loads, 2 no (i-1, j) loads
    if (CP_EXP & 1) { q.w1m = q.v1; q.w3m = q.v3; }
    if (CP_EXP & 2) { q.w0m = q.v0; q.w2m = q.v2; }
#endif
    q.l0 = q.l1 = q.l2 = q.l3 = (S)0;
    if (LINEAR) {
        const unsigned oli = oc + ES * (unsigned)((m - 1) - i);
        const int dj = (jg >= P.n_global - 2) ? (P.n_global - 1) - jg : 0;
        const unsigned olj = oc + ES * (unsigned)(dj * m);
        q.l0 = bld_raw<S>(R.y2a, oli); q.l2 = bld_raw<S>(R.y2b, oli);
        q.l1 = bld_raw<S>(R.y2a, olj, fsb); q.l3 = bld_raw<S>(R.y2b, olj, fsb);
    }
}
template <typename S, bool LINEAR, bool SHF = false>
__device__ __forceinline__ void img_eval(const MfParams &P, int i, int jg, const ImgLoads<S> &q, double cnx, double cny,
                                         PrimalAt &r, int li = 0) {
    const int m = P.m;
    S w0m = q.w0m, w2m = q.w2m;
    if (SHF) {
        const S s0 = lane_shr1<S>(q.v0), s2 = lane_shr1<S>(q.v2);
        w0m = (li == 0) ? q.w0m : s0;
        w2m = (li == 0) ? q.w2m : s2;
    }
    r.y1 = (double)q.y1; r.gx = (double)q.gx; r.gy = (double)q.gy; r.v0 = (double)q.v0; r.v1 = (double)q.v1; r.v2 = (double)q.v2;
    r.v3 = (double)q.v3; r.uxo = (double)q.uxo; r.uyo = (double)q.uyo;
    const double a0 = fwd_diff_adj(r.v0, (i >= 1) ? (double)w0m : 0.0, (double)q.l0, i, m, LINEAR, P.inv_h1);
    const double a1 = fwd_diff_adj(r.v1, (jg >= 1) ? (double)q.w1m : 0.0, (double)q.l1, jg, P.n_global, LINEAR, P.inv_h2);
    const double a2 = fwd_diff_adj(r.v2, (i >= 1) ? (double)w2m : 0.0, (double)q.l2, i, m, LINEAR, P.inv_h1);
    const double a3 = fwd_diff_adj(r.v3, (jg >= 1) ? (double)q.w3m : 0.0, (double)q.l3, jg, P.n_global, LINEAR, P.inv_h2);
    const double vux = -r.gx * r.y1 + (a0 + a1);
    const double vuy = -r.gy * r.y1 + (a2 + a3);
    r.ux = rnd<S>((r.uxo - P.tau * vux) - cnx);  // NNE/mean_zero_indicator.m:50 (the stored value is the iterate)
    r.uy = rnd<S>((r.uyo - P.tau * vuy) - cny);  // :53
    r.wx = -vux;
    r.wy = -vuy;
}
template <typename S, bool LINEAR>
__device__ __forceinline__ void primal_u_at_b(const MfParams &P, int i, int jg, unsigned oc, unsigned fsb, const ImgRsrc &R,
                                              double cnx, double cny, PrimalAt &r) {
    ImgLoads<S> q;
    img_load<S, LINEAR>(P, i, jg, oc, fsb, R, q);
    img_eval<S, LINEAR>(P, i, jg, q, cnx, cny, r);
}
template <typename S>
__device__ __forceinline__ ImgRsrc img_rsrc(const S *pin, const S *g, const S *xo, int K, int k, size_t fs) {
    const S *y2k = pin + ((size_t)K + 4 * (size_t)k) * fs;
    return ImgRsrc{mk_rsrc(pin + (size_t)k * fs), mk_rsrc(y2k), mk_rsrc(y2k + 2 * fs), mk_rsrc(g + 2 * (size_t)k * fs),
                   mk_rsrc(xo + 2 * (size_t)k * fs)};
}

// Phase 1 of the fused passes (NN-TV and NN-curvature): everything that couples the K images of one pixel.
//   y3_n = y3~_n * W_n (lazy nuclear-norm prox), (A'y)_L = y1_n + B*y3_n, L_n = L_{n-1} - tau*(A'y)_L (zero_function prox),
//   Lbar = L_n + theta*(L_n - L_{n-1}), y3~_{n+1} = y3_n + sigma*B*Lbar; stores L_n and y3~_{n+1}; accumulates the Gram
//   matrices of y3~_{n+1} (acc1) and of B*L_n (acc2) on the matrix cores; leaves L_n in tileA and Lbar in tileB (row =
//   image, column = lane) for the per-image phase. The tiles are wave-private: only the order of the wave's own LDS
//   instructions matters (wave_lds_sync), no workgroup barrier.
template <typename S, bool FULLK, int KA>
__device__ __forceinline__ void fused_phase1(const MfParams &P, int K, bool valid, unsigned off, int lane, size_t fs,
                                             const S *__restrict__ y3in, const S *__restrict__ pin,
                                             const S *__restrict__ Lin, S *__restrict__ y3out, S *__restrict__ Lout,
                                             const double *Wl, double *tileA, double *tileB, double sigma, double tau,
                                             double theta, double &vl_max, double4_t &acc1, double4_t &acc2) {
    double a[KA], Lr[KA], y1v[KA];
    {
        double t3[KA];
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            const int kk = (FULLK || k < K) ? k : K - 1;
            t3[k] = ldo_stream(y3in + (size_t)kk * fs, off);
            y1v[k] = ldo(pin + (size_t)kk * fs, off);
            Lr[k] = ldo_stream(Lin + (size_t)kk * fs, off);
        }
#pragma unroll
        for (int kp = 0; kp < KA; ++kp) a[kp] = 0.0;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            const double t = (FULLK || k < K) ? t3[k] : 0.0;
#pragma unroll
            for (int kp = 0; kp < KA; ++kp) a[kp] = fma(t, Wl[k * CP_KMAX + kp], a[kp]);
        }
    }
    double s3 = 0.0;
#pragma unroll
    for (int kp = 0; kp < KA; ++kp) s3 += a[kp];
    const double mean3 = s3 * P.invK;
    double sumL = 0.0, sumLb = 0.0, vLmax = 0.0;
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        const bool act = FULLK || (k < K);
        const double Lo = Lr[k];
        const double vL = y1v[k] + (a[k] - mean3);
        const double Ln = rnd<S>(Lo - tau * vL);
        const double Lb = Ln + theta * (Ln - Lo);
        vLmax = fmax(vLmax, act ? fabs(vL) : 0.0);
        Lr[k] = act ? Ln : 0.0;
        a[k] = a[k] + sigma * Lb;
        sumL += act ? Ln : 0.0;
        sumLb += act ? Lb : 0.0;
        tileB[k * CP_TS + lane] = Lb;
    }
    if (valid) vl_max = fmax(vl_max, vLmax);
    const double smb = sigma * (sumLb * P.invK), mL = sumL * P.invK;
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        const bool act = FULLK || (k < K);
        const double v = (valid && act) ? rnd<S>(a[k] - smb) : 0.0;
        tileA[k * CP_TS + lane] = v;
        // unconditional (round 3): a lane past the end of the column / an image past K carries the offset 2^32-1, which the
        // range check of the buffer store drops -- 2 x KA branches around stores less per step
        const unsigned so = (valid && act) ? off : 0xFFFFFFFFu;
        bst<S>(mk_rsrc(y3out + (size_t)k * fs), so, 0u, v);
        bst<S>(mk_rsrc(Lout + (size_t)k * fs), so, 0u, Lr[k]);
    }
    wave_lds_sync();
    gram_tile_accum(tileA, acc1);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < KA; ++k)
        tileA[k * CP_TS + lane] = (valid && (FULLK || k < K)) ? Lr[k] - mL : 0.0;
    wave_lds_sync();
    gram_tile_accum(tileA, acc2);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < KA; ++k) tileA[k * CP_TS + lane] = Lr[k];
}

// ---- k_mf_fused_tv: organised by wavefront -----------------------------------------------------------------------------
// A wavefront owns one image column and marches DOWN it in aligned segments of 64 rows (a "part" of `steps` segments per
// task), so that the (i+1, j) neighbour of every row is an own-pixel result that already exists: lane+1's in the same
// step (cross-lane shift), and for lane 63 the value lane 0 produced one step earlier, handed over through 32 doubles of
// LDS per image. The 13 loads per image of the recomputation at (i+1, j) -- a third of the kernel's load instructions and
// L2 requests, and the spill-over lines that belong to another XCD -- are gone; only the first step of a part that does
// not start at the image's last row recomputes that one neighbour, in lane 63 alone. The four waves of a workgroup take
// four adjacent columns and run free (the tiles are wave-private: the Gram steps need only the ordering of the wave's own
// LDS instructions, wave_lds_sync, not a workgroup barrier; they meet once, for the partial row), so the (i, j+1) loads of
// waves 0..2 are the lines the next wave streams at the same moment. Tasks (part, column group) are dealt so that a part
// stays on one XCD (8 parts per column in the 4-wave kernels when there are at least 8 segments, see fused2_cap;
// gridDim.x % 8 == 0) and the next column
// group's task -- whose own loads are wave 3's (i, j+1) loads -- is the next workgroup of that XCD.
#ifndef CP_FUSED_PARTS
#define CP_FUSED_PARTS 8
#endif
struct Fused2Geom { int nseg, parts, steps; };
__host__ __device__ inline Fused2Geom fused2_geom(int m, int R = 64, int cap = CP_FUSED_PARTS) {
    Fused2Geom G;
    G.nseg = (m + R - 1) / R;
    G.parts = G.nseg < cap ? G.nseg : cap;
    G.steps = (G.nseg + G.parts - 1) / G.parts;
    G.parts = (G.nseg + G.steps - 1) / G.steps;   // drop parts that would be empty
    return G;
}
// Parts per column. The 8-wave hand-over kernels (k_mf_fused_tv, k_var_lazy_tv) pay a staircase start per task (wave w
// waits for wave w+1's first image) and a recomputed (i+1, j) row at every part boundary, so they want long tasks: 2 parts
// per column as long as that still leaves a task for every CU (4096^2 x 16: 8 parts 9.35 ms, 4: 9.19, 3: 9.30, 2: 8.83-9.12,
// 1: 9.03; var-TV 7.66 / 7.53 / - / 7.01 / 7.15), more parts for fewer column groups: slabs of 4096 rows x 512 / 1024 / 2048
// columns 1.38 / 2.57 / 4.67 ms with >= 1024 tasks, 1.31 / 2.49 / 4.58 with >= 256 (4, 2, 2 parts).
// The 4-wave kernels keep 8 (k_mf_fused_curv: no difference; k_var_lazy_curv: 7.81 ms with 8, 8.31 with 2, 8.87 with 1).
__host__ __device__ inline int fused2_cap(int colgroups, int waves, int forced = 0) {
    if (waves != 8) return CP_FUSED_PARTS;
    if (forced > 0) return forced;        // CP_TUNE_PARTS8 (MfParams::parts8)
    int p = 2;
#ifndef CP_FUSED_MINTASKS8
#define CP_FUSED_MINTASKS8 256     // one workgroup per CU
#endif
    while (p < CP_FUSED_PARTS && (long long)p * colgroups < CP_FUSED_MINTASKS8) p *= 2;
    return p;
}
__host__ __device__ inline long long fused2_tasks(int m, int nl, int cw = 1, int waves = CP_WAVES, int forced = 0) {
    const int colgroups = (nl + waves * cw - 1) / (waves * cw);
    return (long long)fused2_geom(m, 64 / cw, fused2_cap(colgroups, waves, forced)).parts * colgroups;
}

// ---- hand-over of own-pixel results between the waves of a workgroup (no workgroup barrier) --------------------------------
// Wave w+1 computes u_n, u_{n-1} at (i, j+1) as its own-pixel result of image k; wave w needs exactly these four numbers for
// its forward differences along j. One 2 KB slot per pair of neighbouring waves ([value][lane], row i = lane in both waves)
// and two counters: pub = number of results the producer has written, ack = number the consumer has read. The waves of a
// workgroup walk the same (task, step, image) sequence, so the n-th result given is the n-th taken.
//   hs_give: wait until the consumer has read result n-1 (ack >= n-1), write the four values, then pub = n;
//   hs_take: wait until pub >= n, read the four values, then ack = n.
// DS instructions of one wave execute in program order and the LDS serves the waves of its CU one instruction at a time,
// so "values, then counter" on one side and "counter, then values" on the other is all the ordering there is to get; the
// accesses are volatile so that the compiler keeps that order. No cycle: a wave gives before it takes, the last wave only
// gives. The wait of hs_take is bounded (2^22 polls, ~0.1 s): on a time-out it acknowledges anyway and reports false, the
// caller then loads and evaluates the position itself -- always correct, so the grid drains whatever happens; hs_give can
// then only wait for a consumer that is guaranteed to move on.
#ifndef CP_FUSED_HS
#define CP_FUSED_HS 1
#endif
#ifndef CP_HS_MAXPOLL
#define CP_HS_MAXPOLL (1 << 22)    // ~0.1 s of s_sleep(1) polls
#endif
// The order "values, then counter" / "counter, then values" is what the protocol needs, from two sides:
//   * the hardware: DS instructions of one wave execute in program order (see above);
//   * the compiler: the accesses are volatile AND separated by __atomic_signal_fence(seq_cst), which forbids the compiler
//     to move any memory access across it and emits no instruction.
// Round 3 measured the alternative the memory model offers, a workgroup-scope release / acquire fence restricted to the LDS
// address space (__builtin_amdgcn_fence(..., "workgroup", "local"), CP_HS_FENCE=1): it waits for every DS instruction of
// the wave in flight (lgkmcnt(0)), the tile reads of the Gram step included -- 8.59 ms instead of 8.24 per launch at
// 4096^2 x 16 on the same box (profiles/r03_exp_handover.txt) -- so the compiler-only fence stays the default.
#ifndef CP_HS_FENCE
#define CP_HS_FENCE 0
#endif
__device__ __forceinline__ void hs_release() {
#if CP_HS_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
#else
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#endif
}
__device__ __forceinline__ void hs_acquire() {
#if CP_HS_FENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#endif
}
// Both waits are bounded (CP_HS_MAXPOLL). A giver that times out marks the pair broken and never writes the slot again (the
// consumer may still be reading result n-1); a consumer that finds the pair broken, or times out, acknowledges and reports
// false -- its caller loads and evaluates the position itself, always correct. Every such event is counted (return value of
// hs_give, !hs_take) and added to CpCtrl::hs_fallbacks at the end of the kernel: cp_handover_fallbacks() must read 0.
// The slot and flag pointers carry the LDS address space in their type (no generic pointers whose address space the compiler
// would have to infer: one instantiation failed in the backend with a flat-pointer aperture test otherwise).
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ int hs_give(lds_double *slot, lds_int *ack, lds_int *pub, lds_int *broken, int n, int lane, double a,
                                       double b, double c, double d) {
    volatile lds_int *vack = ack, *vbr = broken;
    if (*vbr) return 0;
    for (int polls = 0; *vack < n - 1; ++polls) {
        if (polls > CP_HS_MAXPOLL) {
            if (lane == 0) *vbr = 1;
            return 1;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    hs_acquire();                       // the consumer's reads of result n-1 precede its ack
    volatile lds_double *vs = slot;
    vs[lane] = a; vs[64 + lane] = b; vs[128 + lane] = c; vs[192 + lane] = d;
    hs_release();
    if (lane == 0) *(volatile lds_int *)pub = n;
    return 0;
}
__device__ __forceinline__ bool hs_take(const lds_double *slot, lds_int *pub, lds_int *ack, lds_int *broken, int n, int lane,
                                        double &a, double &b, double &c, double &d) {
    volatile lds_int *vpub = pub, *vbr = broken;
    bool ok = true;
    for (int polls = 0; *vpub < n; ++polls) {
        if (polls > CP_HS_MAXPOLL || *vbr) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (ok) {
        hs_acquire();
        const volatile lds_double *vs = slot;
        a = vs[lane]; b = vs[64 + lane]; c = vs[128 + lane]; d = vs[192 + lane];
        hs_release();
    }
    if (lane == 0) *(volatile lds_int *)ack = n;
    return ok;
}
// end of a kernel that used the hand-over: add this wave's fall-backs to the handle's counter
__device__ __forceinline__ void hs_report(const CpCtrl *ctrl, int fallbacks, int lane) {
    if (fallbacks != 0 && lane == 0) atomicAdd(const_cast<int *>(&ctrl->hs_fallbacks), fallbacks);
}

// KA = extent of the per-pixel image arrays (4, 8, 12 or 16 >= K): phase 1 loads, multiplies by W and stores KA images, so
// K = 10 does not pay for 16 (the Gram tiles stay 16 x 16, zero padded). N21: F2 = mu*||.||_{2,1} (NNE/norm21.m, the
// single-level driver) instead of the pointwise 2x2 nuclear norm.
// CW = image columns per wavefront (1, 2 or 4): the 64 lanes form a tile of R = 64/CW rows x CW columns (lane = lc*R + li).
// With CW > 1 the (i, j+1) neighbour of every column of the tile but the last is the own-pixel result of lane + R (a
// cross-lane read, bitwise what the recomputation gives); only the last column's lanes recompute it (loads masked to
// those lanes), and only the last wave's last column reads lines of another workgroup.
// NWV = wavefronts (= adjacent image columns) per workgroup: 4 (two workgroups per CU) or 8 (one workgroup per CU, the same
// two waves per SIMD): only the last wave's (i, j+1) lines and the first wave's (i, j-1) lines belong to another workgroup,
// so 8 columns halve the cross-workgroup re-reads.
// KS (the coarsest pyramid levels: a handful of workgroups, one wave per SIMD, so the per-image phase is the issue rate of
// ONE wave -- 1.5 us per image): the four waves of a workgroup share one column; each runs phase 1 (same values, same
// stores; only wave 0's Gram tiles count) and then every fourth image of phase 2.
template <typename S, bool LINEAR, bool FULLK, int KA = CP_KMAX, bool N21 = false, int CW = 1, int NWV = CP_WAVES, bool KS = false>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void k_mf_fused_tv(MfParams P, const S *__restrict__ xo, S *__restrict__ xn,
                                                              const S *__restrict__ pin, S *__restrict__ pout,
                                                              const S *__restrict__ g, const S *__restrict__ b,
                                                              const double *__restrict__ W,
                                                              const CpCtrl *__restrict__ ctrl, double *__restrict__ part,
                                                              const S *__restrict__ bin, S *__restrict__ bout) {
    __shared__ double tiles[NWV * 2 * 16 * CP_TS];
    __shared__ double Wl[CP_KMAX * CP_KMAX];
    constexpr int R = 64 / CW;
    // [wave][image][tile column][ux, uy, ux_old, uy_old] of the tile's first row: read (by the last row's lanes) before it is
    // rewritten within a step, in program order of the wave's own LDS instructions
    __shared__ double carry_all[NWV * CP_KMAX * CW * 4];
    // measured at 4096^2 x 16: 8 waves with the hand-over 9.09-9.34 ms and 50.4 GB moved (31.0 GB read) per launch; 4 waves
    // without it 9.20-9.26 ms and 57.3 GB (37.9 GB read: neighbouring waves miss the same line at the same moment and both
    // fetch it); 4 waves with it 9.31-9.63 ms, 51.8 GB; 8 waves without it 10.2-10.6 ms
    constexpr bool HS = (CP_FUSED_HS != 0) && !LINEAR && CW == 1 && NWV == 8;
    constexpr bool SHF = (CP_FUSED_SHF != 0) && CW == 1;   // (i-1, j) from lane - 1 and the boundary array (img_load)
    __shared__ double xch_all[HS ? (NWV - 1) * 256 : 1];   // hand-over slots, one per pair of neighbouring waves
    __shared__ int hs_flags[3 * NWV];                       // pub[wave], ack[wave], broken[wave]
    const int stopped = ctrl->stop;
    const double cnx = ctrl->cn[0], cny = ctrl->cn[1];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *tileA = tiles + wave * 2 * 16 * CP_TS, *tileB = tileA + 16 * CP_TS;
    double *carry = carry_all + wave * CP_KMAX * CW * 4;
    const cp_rsrc Rbin = mk_rsrc(bin), Rbout = mk_rsrc(bout);
    const bool bnd_on = lane < CP_BND_BLOCK;                                // the lanes that hold one value of a block
    const unsigned bnd_lane = (unsigned)sizeof(S) * (unsigned)(lane & (CP_BND_BLOCK - 1));
    // KS: a wave stores the boundary values of its own images only
    const bool bnd_mine = !KS || (((lane & (CP_KMAX - 1)) & (NWV - 1)) == wave);
    const int li = lane & (R - 1), lc = lane / R;
    const int K = P.K, m = P.m;
    const size_t fs = (size_t)P.fs;
    const bool linear = LINEAR;
    stage_W(W, Wl);
    if (stopped) return;
    if (KA < CP_KMAX) {
        for (int k = KA; k < CP_KMAX; ++k) tileA[k * CP_TS + lane] = 0.0;
    }
    if (tid < 3 * NWV) hs_flags[tid] = 0;
    __syncthreads();
    lds_int *fl = (lds_int *)hs_flags;
    int hs_fb = 0;                                                         // hand-over fall-backs of this wave (hs_report)
    lds_double *xl = (lds_double *)xch_all + (HS && wave > 0 ? (wave - 1) * 256 : 0);       // slot this wave gives into (to wave - 1)
    const lds_double *xr = (lds_double *)xch_all + (HS && wave < NWV - 1 ? wave * 256 : 0);  // slot this wave takes from (wave + 1)
    int pseq = 0, cseq = 0;
    const double sigma = P.sigma, tau = P.tau, theta = P.theta, inv_sigma = 1.0 / P.sigma;
    const S *y3in = pin + 5 * (size_t)K * fs;
    S *y3out = pout + 5 * (size_t)K * fs;
    const S *Lin = xo + 2 * (size_t)K * fs;
    S *Lout = xn + 2 * (size_t)K * fs;
    constexpr unsigned ES = (unsigned)sizeof(S);
    const unsigned mb = ES * (unsigned)m, fsb = ES * (unsigned)P.fs;

    double sums[CP_NSUM], maxs[CP_NMAX];
#pragma unroll
    for (int a = 0; a < CP_NSUM; ++a) sums[a] = 0.0;
#pragma unroll
    for (int a = 0; a < CP_NMAX; ++a) maxs[a] = -1.0e300;
    maxs[DM_Y1 - CP_NSUM] = 0.0; maxs[DM_S1 - CP_NSUM] = 0.0; maxs[PM_VL - CP_NSUM] = 0.0;
    double4_t acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};

    const Fused2Geom GM = fused2_geom(m, R, fused2_cap((P.nl + NWV * CW - 1) / (NWV * CW), NWV, P.parts8));
    static_assert(!KS || (CW == 1 && NWV == CP_WAVES && !LINEAR), "KS: one column per workgroup of 4 waves");
    const int ntasks = KS ? GM.parts * P.nl : GM.parts * ((P.nl + NWV * CW - 1) / (NWV * CW));
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int cg = task / GM.parts, pt = task - cg * GM.parts;
        const int jw = KS ? cg : (cg * NWV + wave) * CW;           // first column of the wave's tile
        if (jw >= P.nl) continue;                                  // wave-uniform
        const bool colv = jw + lc < P.nl;
        const int j = min(jw + lc, P.nl - 1);                      // lanes past the last column redo it, nothing is stored
        const int seg_lo = pt * GM.steps, seg_hi = min(seg_lo + GM.steps, GM.nseg) - 1;
        const int jg = P.j_off + j;
        const bool last_j = (jg == P.n_global - 1);
        const int oj = (last_j && LINEAR) ? -1 : 1;
        // where the (i, j +- 1) result comes from: the lane R further in the tile, or a recomputation
        const int nbj_lane = lane + oj * R;
        const bool wantj = LINEAR || !last_j;
        const bool rcj = wantj && !(lc + oj >= 0 && lc + oj < CW && jw + lc + oj < P.nl);
        // hand-over roles (wave-uniform): this wave gives to wave - 1 (whose column jw - 1 is in this column group, hence active)
        // and takes from wave + 1 if that wave has a column (jw + 1 < nl: then it is active and gives)
        const bool hs_p = HS && wave > 0, hs_c = HS && wave < NWV - 1 && jw + 1 < P.nl;
        for (int sg = seg_hi; sg >= seg_lo; --sg) {
            const int ii = sg * R + li;
            const bool valid = ii < m && colv;
            const int i = min(ii, m - 1);                          // lanes past the end redo the last row, nothing is stored
            const unsigned off = ES * (unsigned)((j + 1) * m + i);
            const bool last_i = (i == m - 1);

            // boundary blocks of this segment: row 64 sg - 1 of the own column and (waves that evaluate (i, j +- 1) from
            // memory) of the neighbour column, requested before phase 1 so that they have arrived when the per-image loop starts; bacc collects lane 63's new Dx values, image by image, for the segment above
            S bq0 = (S)0, bqj = (S)0, bacc = (S)0;
#ifndef CP_BND_LATE
#define CP_BND_LATE 0
#endif
            if (SHF && !CP_BND_LATE) {
                bq0 = bld_raw<S>(Rbin, bnd_on ? (unsigned)(bnd_block(sg, j + 1, P.nl) * sizeof(S)) + bnd_lane : 0xFFFFFFFFu);
                if (!hs_c) bqj = bld_raw<S>(Rbin, bnd_on ? (unsigned)(bnd_block(sg, j + 1 + oj, P.nl) * sizeof(S)) + bnd_lane : 0xFFFFFFFFu);
            }
            // ---- phase 1: everything that couples the K images of this pixel ------------------------------------------
            fused_phase1<S, FULLK, KA>(P, K, valid, off, lane, fs, y3in, pin, Lin, y3out, Lout, Wl, tileA, tileB, sigma, tau,
                                       theta, maxs[PM_VL - CP_NSUM], acc1, acc2);
            if (SHF && CP_BND_LATE) {
                bq0 = bld_raw<S>(Rbin, bnd_on ? (unsigned)(bnd_block(sg, j + 1, P.nl) * sizeof(S)) + bnd_lane : 0xFFFFFFFFu);
                if (!hs_c) bqj = bld_raw<S>(Rbin, bnd_on ? (unsigned)(bnd_block(sg, j + 1 + oj, P.nl) * sizeof(S)) + bnd_lane : 0xFFFFFFFFu);
            }

            // ---- phase 2: per image: primal step at the own pixel and at (i, j+1); (i +- 1, j) from a neighbour lane ------
            // where the i-neighbour's own-pixel result comes from: lane +- 1 of this step; for lane 63 the hand-over of the
            // previous step (the segment above); recomputed (lane-masked) only where neither exists
            const int oi = (last_i && LINEAR) ? -1 : 1;
            const int nb_lane = lane + oi, nb_li = li + oi;
            const bool want = valid && (LINEAR || !last_i);          // the neighbour's value is used at all
            const bool from_carry = want && nb_li == R && sg < seg_hi;
            const bool recompute = want && (nb_li < 0 || nb_li >= R) && !from_carry;
            const bool any_rc = __any(recompute);                    // wave-uniform: first step of a part / m % R == 1 with linear rows
            const unsigned offj = oj > 0 ? off + mb : off - mb;            // (i, j +- 1)
            const unsigned offi = oi > 0 ? off + ES : off - ES;            // (i +- 1, j), only where it is recomputed
            double *cslot = carry + lc * 4;
            const double vmask = valid ? 1.0 : 0.0;
            const unsigned off_st = valid ? off : 0xFFFFFFFFu;
            for (int k = KS ? wave : 0; k < K; k += KS ? NWV : 1) {
                // descriptors of this image's field groups (uniform: SALU); every position is one 32-bit VGPR offset
                S *o2k = pout + ((size_t)K + 4 * (size_t)k) * fs;
                const ImgRsrc R = img_rsrc<S>(pin, g, xo, K, k, fs);
                const cp_rsrc Run = mk_rsrc(xn + 2 * (size_t)k * fs),
                              Ro1 = mk_rsrc(pout + (size_t)k * fs), Ro2a = mk_rsrc(o2k), Ro2b = mk_rsrc(o2k + 2 * fs);
                const double Lc = tileA[k * CP_TS + lane], Lbar = tileB[k * CP_TS + lane];
                PrimalAt r0, rj;
                ImgLoads<S> q0, qj;
                img_load<S, LINEAR, SHF>(P, i, jg, off, fsb, R, q0, lane_bcast<S>(bq0, k), lane_bcast<S>(bq0, CP_KMAX + k));
                // (i, j+1): the wave of the next column computes exactly this as its own-pixel result and hands it over through
                // LDS (hs_take below); only a wave without such a neighbour -- the last of the workgroup, the last column of a
                // slab -- loads and evaluates the position itself. Evaluated by every lane (no lane-dependent branch).
#ifdef CP_EXP   // 4: no (i, j+1) evaluation from memory at all
                if (CP_EXP & 4) qj = q0; else
#endif
                if (!hs_c) img_load<S, LINEAR, SHF>(P, i, jg + oj, offj, fsb, R, qj, lane_bcast<S>(bqj, k), lane_bcast<S>(bqj, CP_KMAX + k));
                const S bb_raw = bld_raw_stream<S>(mk_rsrc(b + (size_t)k * fs), off);
                // fp32 storage: keep the float -> double conversions behind the whole group of loads (left alone, the compiler
                // converts each value as it arrives into the register the next load is waiting for: one load in flight)
                if (sizeof(S) == 4) __builtin_amdgcn_sched_barrier(0);
                const double bb = (double)bb_raw;
                img_eval<S, LINEAR, SHF>(P, i, jg, q0, cnx, cny, r0, li);
                if (HS) {
                    if (hs_p) hs_fb += hs_give(xl, fl + NWV + (wave - 1), fl + wave, fl + 2 * NWV + wave, ++pseq, lane, r0.ux, r0.uy, r0.uxo, r0.uyo);
                    if (hs_c) {
                        if (!hs_take(xr, fl + (wave + 1), fl + NWV + wave, fl + 2 * NWV + (wave + 1), ++cseq, lane, rj.ux, rj.uy, rj.uxo, rj.uyo)) {
                            ++hs_fb;
                            img_load<S, LINEAR, false>(P, i, jg + oj, offj, fsb, R, qj);         // never in practice: see hs_take
                            img_eval<S, LINEAR, false>(P, i, jg + oj, qj, cnx, cny, rj);
                        }
                    } else {
                        img_eval<S, LINEAR, SHF>(P, i, jg + oj, qj, cnx, cny, rj, li);
                    }
                } else {
                    img_eval<S, LINEAR, SHF>(P, i, jg + oj, qj, cnx, cny, rj, li);
                }
                const double ux = r0.ux, uy = r0.uy, uxo_ = r0.uxo, uyo_ = r0.uyo, wx = r0.wx, wy = r0.wy;
                const double gx = r0.gx, gy = r0.gy, y1 = r0.y1, v0 = r0.v0, v1 = r0.v1, v2 = r0.v2, v3 = r0.v3;
                const double uxj = rj.ux, uyj = rj.uy, oj0 = rj.uxo, oj1 = rj.uyo;
                // bitwise what a recomputation at (i +- 1, j) would give
                double uxi = __shfl(ux, nb_lane), uyi = __shfl(uy, nb_lane);
                double oi0 = __shfl(uxo_, nb_lane), oi1 = __shfl(uyo_, nb_lane);
                if (from_carry) {
                    const double *cs = cslot + k * CW * 4;
                    uxi = cs[0]; uyi = cs[1]; oi0 = cs[2]; oi1 = cs[3];
                }
                if (any_rc) {
                    if (recompute) {
                        PrimalAt ri;
                        primal_u_at_b<S, LINEAR>(P, i + oi, jg, offi, fsb, R, cnx, cny, ri);
                        uxi = ri.ux; uyi = ri.uy; oi0 = ri.uxo; oi1 = ri.uyo;
                    }
                }
                if (li == 0) {
                    double *cs = cslot + k * CW * 4;
                    cs[0] = ux; cs[1] = uy; cs[2] = uxo_; cs[3] = uyo_;
                }

                // stores are unconditional instructions: lanes past the end of the column / the slab carry an offset beyond
                // the descriptor's range (off_st = 2^32 - 1) and the buffer range check drops them. A branch around the
                // stores would make the number of memory operations in flight unknown to the compiler, which then waits for
                // ALL of them (vmcnt(0)) -- the stores of image k included -- before it touches the prefetched image k+1.
                bst<S>(Run, off_st, 0u, ux);
                bst<S>(Run, off_st, fsb, uy);
                maxs[PM_WXMAX - CP_NSUM] = fmax(maxs[PM_WXMAX - CP_NSUM], valid ? wx : -1.0e300);
                maxs[PM_WXMIN - CP_NSUM] = fmax(maxs[PM_WXMIN - CP_NSUM], valid ? -wx : -1.0e300);
                maxs[PM_WYMAX - CP_NSUM] = fmax(maxs[PM_WYMAX - CP_NSUM], valid ? wy : -1.0e300);
                maxs[PM_WYMIN - CP_NSUM] = fmax(maxs[PM_WYMIN - CP_NSUM], valid ? -wy : -1.0e300);
                sums[PS_WX] += vmask * wx;
                const double uxb = ux + theta * (ux - uxo_), uyb = uy + theta * (uy - uyo_);
                const double uxbi = uxi + theta * (uxi - oi0), uybi = uyi + theta * (uyi - oi1);
                const double uxbj = uxj + theta * (uxj - oj0), uybj = uyj + theta * (uyj - oj1);

                const double r_cur = -(gx * ux + gy * uy) + Lc;
                const double r_bar = -(gx * uxb + gy * uyb) + Lbar;
                sums[DS_F1] += vmask * fabs(r_cur - bb);
                const double yt = y1 + sigma * r_bar;
                const double Ls = yt * inv_sigma, thr = inv_sigma * P.mu1, dd = Ls - bb;
                const double prox = (dd > thr) ? (Ls - thr) : ((dd < -thr) ? (Ls + thr) : bb);
                const double y1n = rnd<S>(yt - sigma * prox);
                sums[DS_F1S] += vmask * (y1n * bb);
                sums[DS_SGX] += vmask * (gx * y1n);
                sums[DS_SGY] += vmask * (gy * y1n);

                const double d0 = fwd_diff(ux, uxi, uxi, last_i, linear, P.inv_h1);
                const double d1 = fwd_diff(ux, uxj, uxj, last_j, linear, P.inv_h2);
                const double d2 = fwd_diff(uy, uyi, uyi, last_i, linear, P.inv_h1);
                const double d3 = fwd_diff(uy, uyj, uyj, last_j, linear, P.inv_h2);
                const double e0 = fwd_diff(uxb, uxbi, uxbi, last_i, linear, P.inv_h1);
                const double e1 = fwd_diff(uxb, uxbj, uxbj, last_j, linear, P.inv_h2);
                const double e2 = fwd_diff(uyb, uybi, uybi, last_i, linear, P.inv_h1);
                const double e3 = fwd_diff(uyb, uybj, uybj, last_j, linear, P.inv_h2);
                const double t0 = v0 + sigma * e0, t1 = v1 + sigma * e1, t2 = v2 + sigma * e2, t3 = v3 + sigma * e3;
                double n0, n1, n2, n3, s1v;
                if (N21) {
                    sums[DS_F2] += vmask * sqrt(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
                    const double nn = fmax(sqrt(t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3), P.mu2);
                    n0 = (P.mu2 * t0) / nn; n1 = (P.mu2 * t1) / nn; n2 = (P.mu2 * t2) / nn; n3 = (P.mu2 * t3) / nn;
                    n0 = rnd<S>(n0); n1 = rnd<S>(n1); n2 = rnd<S>(n2); n3 = rnd<S>(n3);
                    s1v = sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
                } else {
                    sums[DS_F2] += vmask * nuc_2x2(d0, d1, d2, d3);
                    proj_spectral_2x2(t0, t1, t2, t3, P.mu2, n0, n1, n2, n3, s1v);
                }
                if (sizeof(S) == 4) { n0 = rnd<S>(n0); n1 = rnd<S>(n1); n2 = rnd<S>(n2); n3 = rnd<S>(n3); }
                bst<S>(Ro1, off_st, 0u, y1n);
                bst<S>(Ro2a, off_st, 0u, n0); bst<S>(Ro2a, off_st, fsb, n1); bst<S>(Ro2b, off_st, 0u, n2); bst<S>(Ro2b, off_st, fsb, n3);
                if (SHF) {   // row 64 sg + 63 = the row under segment sg + 1: lane k keeps Dx u_x of image k, lane 16 + k Dx u_y
                    const S t0 = lane_bcast<S>((S)n0, 63), t2 = lane_bcast<S>((S)n2, 63);
                    bacc = (lane == k) ? t0 : ((lane == CP_KMAX + k) ? t2 : bacc);
                }
                maxs[DM_Y1 - CP_NSUM] = fmax(maxs[DM_Y1 - CP_NSUM], valid ? fabs(y1n) : 0.0);
                maxs[DM_S1 - CP_NSUM] = fmax(maxs[DM_S1 - CP_NSUM], valid ? s1v : 0.0);
                sums[DS_UX] += vmask * ux;
                sums[DS_UY] += vmask * uy;
            }
            if (SHF)   // one 256-byte store per wave and step (lanes of a column past the slab / of another wave's images: dropped)
                bst<S>(Rbout, (colv && bnd_mine && bnd_on) ? (unsigned)(bnd_block(sg + 1, j + 1, P.nl) * sizeof(S)) + bnd_lane : 0xFFFFFFFFu, 0u, (double)bacc);
            wave_lds_sync();   // hand-over and tiles of this step are complete before the next step touches them
        }
    }
    if (HS) hs_report(ctrl, hs_fb, lane);
    if (KS && wave != 0) { acc1 = double4_t{0.0, 0.0, 0.0, 0.0}; acc2 = double4_t{0.0, 0.0, 0.0, 0.0}; }   // phase 1 ran four times
    double *out = part + (size_t)blockIdx.x * CP_PSTRIDE;
    {
        constexpr int WS = 2 * 16 * CP_TS;
        static_assert(WS >= CP_PSTRIDE, "tile region too small for the staged partial row");
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

// Boundary array of a dual buffer that some other kernel wrote (k_mf_dual at start-up, a halo message): block (sg, jc) <-
// the Dx components of y2 at row 64 sg - 1, storage columns jc0 <= jc < jc1 (0 and nl + 1 are the halo columns).
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_mf_fill_bnd(int m, int nl, int K, int fs, int jc0, int jc1, const S *__restrict__ p,
                                                          S *__restrict__ bnd) {
    const int nseg = (m + 63) / 64, ncol = jc1 - jc0;
    const long long tot = (long long)nseg * ncol * CP_BND_BLOCK;
    for (long long gid = (long long)blockIdx.x * CP_BLOCK + threadIdx.x; gid < tot; gid += (long long)gridDim.x * CP_BLOCK) {
        const int e = (int)(gid % CP_BND_BLOCK);
        const long long r = gid / CP_BND_BLOCK;
        const int jc = jc0 + (int)(r % ncol), sg = 1 + (int)(r / ncol);     // sg = 1 .. nseg
        const int c = e / CP_KMAX, k = e % CP_KMAX, row = 64 * sg - 1;
        S v = (S)0;
        if (k < K && row < m) v = p[((size_t)K + 4 * (size_t)k + 2 * (size_t)c) * fs + (size_t)jc * m + row];
        bnd[bnd_block(sg, jc, nl) + e] = v;
    }
}

struct EFusedArgs {
    int warm;            // start both eigensolves from the previous iteration's eigenvectors (vwarm)
    int nblk, K;
    int n;               // iteration: completes primal row n and dual row n, starts dual row n+1
    int check_stop;
    int norm21;          // F2 = mu*||.||_{2,1}: its conjugate's violation is max||v|| - mu (NNE/norm21.m:81-85), not relative
    int reg;             // 1: NN-curvature (k_mf_fused_curv): F2, F2* and the stop rule belong to k_curv_fin, which runs after k_curv_y2
    double tol, mu1, mu2, sigma, tau, Ntot;
};

__global__ __launch_bounds__(CP_RED_THREADS) void k_mf_efused(EFusedArgs A, const double *__restrict__ part,
                                                              CpCtrl *ctrl, double *__restrict__ Wout, double *hist_p,
                                                              double *hist_d, double *__restrict__ vwarm) {
    // GG[0] = Gram(y3~), GG[1] = deflated Gram(B*L); VV: accumulated rotations; V0: warm-start bases; Tm: scratch
    __shared__ double GG[2][16][17], VV[2][16][17], V0[2][16][17], Tm[2][16][17], G2[16][17], sc[CP_NSCAL],
        dvec[16], stage[CP_RED_THREADS], stage2[CP_RED_THREADS];
    double (*G)[17] = GG[0], (*V)[17] = VV[0], (*G2h)[17] = GG[1];
    __shared__ double zv[16], zs[16];   // l1-ball projection of the singular values
    __shared__ double svl[16];          // singular values of B*L_n
    __shared__ double res[4];           // largest singular value of the prox, of the input
    const int t = threadIdx.x;
    // both control words are requested before the reduction's loads (one global round trip less on the critical path);
    // a stopped solve leaves after the reduction's barriers (it reads a stale partial buffer, nothing is written)
    const int stopped = ctrl->stop;
    const double nu = ctrl->nu;
#ifdef CP_EPROF
    if (t == 0) g_eprof[20] = 0;
#endif
    EPROF(0);
    reduce_partials2(part, A.nblk, G, G2, sc, stage, stage2);   // Gram(y3~_{n+1}), Gram(B*L_n) and the scalars
    if (stopped) return;
    EPROF(1);
    if (t >= 512) return;        // the eigensolves need 2 x 256 threads; ended waves do not take part in barriers
    // Gram(B*L_n) restricted to the complement of (1,..,1) (see helmert_rows)
    if (t < 256) helmert_rows(G2, Tm[1], A.K, t >> 4, t & 15);
    __syncthreads();
    if (t < 256) helmert_cols(Tm[1], G2h, A.K, t >> 4, t & 15);
    __syncthreads();
    EPROF(2);
    eig16_warm<2>(GG, VV, V0, Tm, vwarm, A.warm, A.K);   // both Gram matrices side by side, warm-started
    EPROF(5);
    if (t < 64) l1ball_scale16(G, A.K, nu * A.sigma, zv, zs, dvec, res, ctrl->sv);   // NNE/nuclear_norm_constraint.m:72
    else if (t < 80) {   // singular values of B*L_n (K-1 after the deflation), one lane each
        const int i = t - 64;
        const double sL = (i < A.K - 1) ? sqrt(fmax(G2h[i][i], 0.0)) : 0.0;
        svl[i] = sL;
        if (i < A.K) ctrl->svL[i] = sL;
    }
    __syncthreads();
    EPROF(6);
    if (t < 256) {   // W_{n+1} = I - V diag(d) V'
        const int k = t >> 4, kp = t & 15;
        double a = (k == kp) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) a -= V[k][i] * dvec[i] * V[kp][i];
        Wout[k * CP_KMAX + kp] = a;
    }
    if (t == 0) {
        double sumS = 0.0;
        for (int i = 0; i < A.K - 1; ++i) sumS += svl[i];
        const int n = A.n;
        double *hp = hist_p + (size_t)n * CP_HCOLS_MF;
        double *hd = hist_d + (size_t)n * CP_HCOLS_MF, *hd1 = hist_d + (size_t)(n + 1) * CP_HCOLS_MF;
        // primal row n                                        chambolle_pock.m:168-171
        const double F1 = A.mu1 * sc[DS_F1], F2 = A.mu2 * sc[DS_F2];
        hp[1] = F1; hp[3] = 0.0; hp[4] = 0.0;
        if (A.reg == 0) { hp[2] = F2; hp[0] = hp[4] + (F1 + F2 + 0.0); }
        hp[5] = (sumS > nu) ? (sumS - nu) / nu : 0.0;          // nuclear_norm_constraint.m:55-59
        hp[6] = fmax(fabs(sc[DS_UX] / A.Ntot), fabs(sc[DS_UY] / A.Ntot));   // mean_zero_indicator.m:65
        // dual row n: G* part (F* part was stored one iteration earlier)     :173-176
        const double mw = sc[PS_WX] / A.Ntot;
        const double dx = fmax(sc[PM_WXMAX] - mw, mw - (-sc[PM_WXMIN]));
        const double dy = fmax(sc[PM_WYMAX] - mw, mw - (-sc[PM_WYMIN]));
        hd[4] = 0.0;
        hd[6] = fmax(fmax(dx, dy), sc[PM_VL]);
        hd[0] = -(hd[4] + (hd[1] + hd[2] + hd[3]));
        // dual row n+1: F* part of y_{n+1}
        hd1[1] = sc[DS_F1S];
        if (A.reg == 0) hd1[2] = 0.0;
        hd1[3] = nu * res[0];
        const double con1 = (sc[DM_Y1] > A.mu1) ? sc[DM_Y1] - A.mu1 : 0.0;
        const double con2 = (A.reg != 0) ? 0.0                                           // NNE/SSD.m: no constraint
                            : A.norm21 ? ((sc[DM_S1] > A.mu2) ? sc[DM_S1] - A.mu2 : 0.0)   // NNE/norm21.m:81-85
                                       : fmax((sc[DM_S1] - A.mu2) / A.mu2, 0.0);           // NNE/pointwise_2x2_nn.m:52-57
        hd1[5] = fmax(con1, con2);
        // mean of the next un-projected u
        ctrl->cn[0] = sc[DS_UX] / A.Ntot + A.tau * (sc[DS_SGX] / A.Ntot);
        ctrl->cn[1] = sc[DS_UY] / A.Ntot + A.tau * (sc[DS_SGY] / A.Ntot);
        if (A.check_stop && n >= 4 && A.reg == 0) {            // chambolle_pock.m:142-153
            bool ok = true;
            for (int r = n - 4; r <= n; ++r) {
                const double *p_ = hist_p + (size_t)r * CP_HCOLS_MF, *d_ = hist_d + (size_t)r * CP_HCOLS_MF;
                const double gap = fabs((p_[0] - d_[0]) / d_[0]);
                if (!(gap <= A.tol) || !(p_[5] <= A.tol) || !(p_[6] <= A.tol) || !(d_[5] <= A.tol) || !(d_[6] <= A.tol))
                    ok = false;
            }
            if (ok) { ctrl->stop = 1; ctrl->stop_iter = n; }
        }
    }
    EPROF(7);
}
