// Host side of libcpengine.so: handle, device-resident state, solver loops, C ABI (include/cpengine.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cpengine.h"
#include "cp_mfnn_fused.cuh"
#include "cp_mfnn_big.cuh"
#include "cp_mfnn_curv.cuh"
#include "cp_var.cuh"
#include "cp_var_fused.cuh"
#include "cp_pair.cuh"
#include "cp_nnl.cuh"

static thread_local std::string g_err;

#define CP_FAIL(code, ...)                                   \
    do {                                                     \
        char buf_[512];                                      \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);            \
        g_err = buf_;                                        \
        return (code);                                       \
    } while (0)

// grid cap of the fused field kernels. 4096^2 has 8192 tasks (8 parts x 1024 column groups of 4): ONE task per workgroup,
// dealt by the dispatcher as slots free up, measured 9.23 ms per launch of k_mf_fused_tv against 9.87-9.90 ms with 4096
// workgroups walking two tasks each (same box); 16 parts (16384 workgroups of 4 steps) 9.64 ms, 4 parts 9.85 ms.
#ifndef CP_FUSED_MAXGRID
#define CP_FUSED_MAXGRID 8192
#endif
#define CP_S1_GROUP 64

#define CP_NOT_ITERATING(h) \
    if ((h)->it_n >= 0) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState an iteration is open: call cp_iterate_end / cp_mg_end first")

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            CP_FAIL(CP_ERR_DEVICE, "cpengine:device %s failed: %s (%s:%d)", #expr,               \
                    hipGetErrorString(e_), __FILE__, __LINE__);                                  \
    } while (0)

#define CPCHK(expr)                  \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != CP_OK) return rc_; \
    } while (0)

enum { T_DUAL = 0, T_EIG, T_PRIMAL, T_FIN, T_ITER, T_COUNT };
// for the fused NN-TV path "dual" times k_mf_fused_tv and "eig" times k_mf_efused
static const char *kPassNames[T_COUNT] = {"dual", "eig", "primal", "fin", "iterate"};

struct PassTimer {
    std::vector<hipEvent_t> ev;   // start/stop pairs
    size_t used = 0;              // events in use
};

struct cp_engine {
    cp_problem_desc d;
    bool mf = false;     // multi-image model (mf_nn or var): K images, CpCtrl, MfParams
    bool fp32 = false;   // CP_FLAG_FP32: fields stored as float inside the double-sized allocations
    double *vwarm = nullptr;   // NN-TV: eigenvector bases of the previous iteration (2 x 256), warm start of k_mf_efused
    double *stage = nullptr;   // fp32 mode: one field of doubles for host <-> device conversion
    bool var = false;    // variance model (no L block, no nuclear norm): cp_var.cuh
    int reg = 0;         // mf: 0 tv, 1 curvature
    int gt = 0;          // pairwise: GT_*
    int nc = 1;          // pairwise: components of u
    int m = 0, nl = 0, K = 1, npix = 0, fs = 0, batch = 1;
    int parts8 = 0;      // CP_TUNE_PARTS8: parts per column of the 8-wave kernels (0: fused2_cap's rule)
    int km = CP_KMAX;    // extent the K x K kernels of this handle are built for: 16, or CP_KBIG (32) for mf_nn with K > 16 (cp_mfnn_big.cuh)
    int nxf = 0, npf = 0;   // fields per problem
    size_t xN = 0, pN = 0;  // doubles per x / p buffer (all problems)
    double *X[2] = {nullptr, nullptr}, *P[2] = {nullptr, nullptr};
    double *g = nullptr, *b = nullptr, *Tu = nullptr, *img = nullptr, *imgR = nullptr;
    double *W[3] = {nullptr, nullptr, nullptr};   // W[2] = scratch
    double *xtmp = nullptr;
    double *bnd[2] = {nullptr, nullptr};     // NN-TV: boundary arrays of P[0], P[1] (cp_mfnn_fused.cuh: bnd_block), in the storage type
    double *vlazy[2] = {nullptr, nullptr};   // variance-TV, k_var_lazy_tv: [abar | beta] per pixel, by iteration parity
    bool lazy_live = false;   // the y1 fields hold alpha (y1 = alpha - beta) until var_end
    double *part = nullptr, *part2 = nullptr;     // per-block partials (mf: one; pw: dual / primal)
    double *svout = nullptr;                      // k_mf_sv_eig output (16 + 256) + identity (256)
    void *ctrl = nullptr;
    double *hist_p = nullptr, *hist_d = nullptr;
    int hist_rows = 0;   // rows allocated (per problem)
    int hcols = 0;
    int nblk = 0;
    int pw_strip = 0;        // columns per wave of k_pw_march (0: k_pw_fused), pw_groups row groups of 63, pw_waves waves
    int pw_groups = 0, pw_waves = 0, pw_grid = 0;
    bool pw_two = false;     // k_pw_march2: two rows per lane, row groups of 128 aligned rows (m even)
    bool time_all = false;   // CP_TUNE_TIME_ALL=1: HIP events around the small kernels of the fused paths too (each pair of
                             // events costs a few microseconds of stream time; by default only the dominant kernel is timed)
    bool pw_split = false;   // tuning aid (CP_TUNE_PW_SPLIT=1): pairwise models on the split passes instead of k_pw_fused
    bool var_fused = false;   // variance-TV: one k_var_lazy_tv + k_var_efused per iteration (CP_TUNE_VAR_SPLIT=1: split passes)
    bool fused_ks = false;   // k_mf_fused_tv<..., KS>: one column per workgroup, the images dealt to its waves (coarsest levels; CP_TUNE_FUSED_KS_TASKS)
    int fused_cw = 1;    // image columns per wavefront of k_mf_fused_tv (template parameter CW; CP_TUNE_FUSED_CW)
    int fused_waves = CP_WAVES;   // wavefronts (adjacent columns) per workgroup of k_mf_fused_tv (NWV; CP_TUNE_FUSED_WAVES = 4 | 8)
    int nblk_tv = 0;              // its grid
    int var_waves = CP_WAVES, nblk_var = 0;   // the same for k_var_lazy_tv (CP_TUNE_VAR_WAVES = 4 | 8)
    int fused_ka = 1;    // use the K <= 4 / 8 / 12 instantiations of k_mf_fused_tv (CP_TUNE_FUSED_KA=0: always 16)
    int nblk_fused = 0;  // grid of k_mf_fused_tv / k_mf_fused_curv (2 workgroups per CU resident: two LDS tiles per wave)
    int nblk_y2 = 0;     // grid of k_curv_y2
    int nblk_pow = 0;    // grid of the power iteration's kernels (cp_norm_estimate)
    bool y2_r2 = false;  // k_curv_y2_r2 (two rows per lane; m even)
    int curv_waves = CP_WAVES, nblk_curv = 0;   // wavefronts (adjacent columns) per workgroup of k_mf_fused_curv and its grid (CP_TUNE_CURV_WAVES = 4 | 8)
    bool curv_fused = false;   // NN-curvature on k_mf_fused_curv + k_curv_y2 (CP_TUNE_CURV_SPLIT=1: the split passes)
    bool linearized = false, has_nu = false, has_images = false;
    hipStream_t stream = nullptr;      // stream all work is enqueued on (own_stream unless cp_set_stream)
    int it_n = -1;                     // >= 0 between cp_iterate_begin / cp_mg_begin and the matching end: iterations done
    bool it_mg = false;                // the open iteration was started by cp_mg_begin
    hipStream_t own_stream = nullptr;
    double mg_theta = 1.0, mg_tau = 0.0, mg_sigma = 0.0;
    double *red = nullptr;             // reduced partial vector of this rank (CP_PSTRIDE doubles)
    std::vector<cp_result> last;   // per problem
    bool profile = false;
    PassTimer timers[T_COUNT];
};

// ------------------------------------------------------------------------------------------------
static const double *staged_partials(cp_engine *h, int nrows, int *rows, bool scalars_only = false);   // below, with the reductions

static MfParams mf_params(const cp_engine *h, double theta, double tau, double sigma) {
    MfParams P;
    P.m = h->m; P.nl = h->nl; P.K = h->K; P.npix = h->npix;
    P.n_global = h->d.n_global; P.j_off = h->d.j_offset; P.bc = h->d.bc; P.fs = h->fs;
    P.inv_h1 = 1.0 / h->d.h[0]; P.inv_h2 = 1.0 / h->d.h[1];
    P.mu1 = h->d.h[0] * h->d.h[1];              // prod(h_grid), NNE/mf_nn_tv...:257
    P.mu2 = h->d.mu * (h->d.h[0] * h->d.h[1]);  // mu * prod(h_grid), :264
    P.norm21 = (h->d.flags & CP_FLAG_MFNN_NORM21) ? 1 : 0;
    P.parts8 = h->parts8;
    if (P.norm21) { P.mu1 = 1.0; P.mu2 = h->d.mu; }   // SAD(y1, b, 1, ..), norm21(y2_j, mu, ..): NNE/mf_nn_tv_registration_no_ref.m:206,213
    P.sigma = sigma; P.tau = tau; P.theta = theta;
    P.ssd_swap = (h->d.flags & CP_FLAG_VAR_SSD_SWAP) ? 1 : 0;
    P.invK = 1.0 / (double)h->K;
    P.Ntot = (double)h->m * (double)h->d.n_global * (double)h->K;
    return P;
}

static PwParams pw_params(const cp_engine *h, double theta, double tau, double sigma) {
    PwParams P;
    P.m = h->m; P.n = h->nl; P.npix = h->npix; P.fs = h->fs; P.nxf = h->nxf; P.nvf = h->npf;
    P.inv_h1 = 1.0 / h->d.h[0]; P.inv_h2 = 1.0 / h->d.h[1];
    P.sigma = sigma; P.tau = tau; P.theta = theta; P.lambda = h->d.lambda;
    return P;
}

static int timer_begin(cp_engine *h, int which) {
    if (!h->profile) return CP_OK;
    PassTimer &t = h->timers[which];
    if (t.used + 2 > t.ev.size()) {
        for (int a = 0; a < 2; ++a) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            t.ev.push_back(e);
        }
    }
    HIPCHK(hipEventRecord(t.ev[t.used], h->stream));
    return CP_OK;
}
static int timer_end(cp_engine *h, int which) {
    if (!h->profile) return CP_OK;
    PassTimer &t = h->timers[which];
    HIPCHK(hipEventRecord(t.ev[t.used + 1], h->stream));
    t.used += 2;
    return CP_OK;
}

// copy between the host vector layout (fields contiguous, m x nl each) and the device field layout
static int upload_fields(cp_engine *h, double *dev, const double *host, int nfields) {
    if (h->fp32) {   // element (f, q) lives at float index f*fs + m + q of the same allocation
        float *devf = (float *)dev;
        for (int f = 0; f < nfields; ++f) {
            float *dst = devf + (size_t)f * h->fs + h->m;
            if (host) {
                HIPCHK(hipMemcpyAsync(h->stage, host + (size_t)f * h->npix, sizeof(double) * h->npix, hipMemcpyHostToDevice,
                                      h->stream));
                hipLaunchKernelGGL(k_cvt_d2f, dim3(256), dim3(256), 0, h->stream, h->npix, h->stage, dst);
            } else {
                HIPCHK(hipMemsetAsync(dst, 0, sizeof(float) * h->npix, h->stream));
            }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        return CP_OK;
    }
    for (int f = 0; f < nfields; ++f) {
        double *dst = dev + (size_t)f * h->fs + h->m;
        if (host)
            HIPCHK(hipMemcpyAsync(dst, host + (size_t)f * h->npix, sizeof(double) * h->npix, hipMemcpyHostToDevice,
                                  h->stream));
        else
            HIPCHK(hipMemsetAsync(dst, 0, sizeof(double) * h->npix, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return CP_OK;
}
static int download_fields(cp_engine *h, const double *dev, double *host, int nfields) {
    if (h->fp32) {
        const float *devf = (const float *)dev;
        for (int f = 0; f < nfields; ++f) {
            hipLaunchKernelGGL(k_cvt_f2d, dim3(256), dim3(256), 0, h->stream, h->npix, devf + (size_t)f * h->fs + h->m, h->stage);
            HIPCHK(hipMemcpyAsync(host + (size_t)f * h->npix, h->stage, sizeof(double) * h->npix, hipMemcpyDeviceToHost,
                                  h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        HIPCHK(hipGetLastError());
        return CP_OK;
    }
    for (int f = 0; f < nfields; ++f)
        HIPCHK(hipMemcpyAsync(host + (size_t)f * h->npix, dev + (size_t)f * h->fs + h->m, sizeof(double) * h->npix,
                              hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return CP_OK;
}

static int ensure_hist(cp_engine *h, int rows) {
    if (rows <= h->hist_rows) return CP_OK;
    if (h->hist_p) HIPCHK(hipFree(h->hist_p));
    if (h->hist_d) HIPCHK(hipFree(h->hist_d));
    h->hist_p = h->hist_d = nullptr;
    const size_t bytes = sizeof(double) * (size_t)rows * h->hcols * h->batch;
    HIPCHK(hipMalloc(&h->hist_p, bytes));
    HIPCHK(hipMalloc(&h->hist_d, bytes));
    HIPCHK(hipMemsetAsync(h->hist_p, 0, bytes, h->stream));
    HIPCHK(hipMemsetAsync(h->hist_d, 0, bytes, h->stream));
    h->hist_rows = rows;
    return CP_OK;
}

// grid of the full-domain passes: grid-stride over 256-pixel chunks with at most one resident wave of blocks
// (256 CUs x blocks per CU admitted by the kernels' register/LDS budget), so that no partial last wave of blocks
// leaves CUs idle and the partial buffers stay small.
static int grid_for(const cp_engine *h) {
    const int need = (h->npix + CP_BLOCK - 1) / CP_BLOCK;
    const int resident = 256 * (h->mf ? 3 : 8);
    return std::max(1, std::min(need, resident));
}

// ------------------------------------------------------------------------------------------------
extern "C" const char *cp_last_error(void) { return g_err.c_str(); }
#ifdef CP_EPROF
extern "C" int cp_debug_eprof(unsigned long long *out) {   // experiments only: phase timestamps of the last k_mf_efused
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eprof), 32 * sizeof(unsigned long long));
}
#endif
extern "C" const char *cp_version(void) { return "cpengine 0.1 (gfx950, fp64)"; }

extern "C" int cp_destroy(cp_handle h) {
    if (!h) return CP_OK;
    hipSetDevice(h->d.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->red) hipFree(h->red);
    if (h->stage) hipFree(h->stage);
    if (h->vwarm) hipFree(h->vwarm);
    double *ptrs[] = {h->X[0], h->X[1], h->P[0], h->P[1], h->g, h->b, h->Tu, h->img, h->imgR, h->W[0], h->W[1],
                      h->W[2], h->xtmp, h->vlazy[0], h->vlazy[1], h->bnd[0], h->bnd[1], h->part, h->part2, h->svout, h->hist_p, h->hist_d};
    for (double *p : ptrs)
        if (p) hipFree(p);
    if (h->ctrl) hipFree(h->ctrl);
    for (auto &t : h->timers)
        for (hipEvent_t e : t.ev) hipEventDestroy(e);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return CP_OK;
}

extern "C" int cp_create(const cp_problem_desc *desc, cp_handle *out) {
    if (!desc || !out) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null descriptor or output");
    *out = nullptr;
    const cp_problem_desc &d = *desc;
    if (d.m < 2 || d.n < 2) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg image size %d x %d", d.m, d.n);
    if (!(d.h[0] > 0.0) || !(d.h[1] > 0.0)) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg grid spacing must be positive");
    const bool var = (d.model == CP_MODEL_VAR_TV || d.model == CP_MODEL_VAR_CURV);
    const bool mf = (d.model == CP_MODEL_MF_NN_TV || d.model == CP_MODEL_MF_NN_CURV) || var;
    const bool pw = (d.model == CP_MODEL_TV_L1_DENOISE || d.model == CP_MODEL_TV_L1_REG || d.model == CP_MODEL_TV_L2_REG);
    if (!mf && !pw) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg unknown model %d", d.model);
    if (mf) {
        // the nuclear-norm models do their K x K spectral work on one 16 x 16 MFMA tile / one-wave Jacobi; the variance models
        // have none and take up to CP_VAR_KMAX images (the reference's real data sets have k = size(data, 3) frames)
        // nuclear-norm models: up to 16 images on one MFMA tile / one-wave Jacobi with the fused kernels; 17..32 on the general
        // path (split passes built for 32 images, cp_mfnn_big.cuh) -- one GPU, fp64 storage
        const int kmax = var ? CP_VAR_KMAX : CP_KBIG;
        if (d.k < 1 || d.k > kmax) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported k = %d (1..%d)", d.k, kmax);
        if (!var && d.k > CP_KMAX && (d.flags & CP_FLAG_FP32))
            CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported k = %d > %d images in fp32 storage (the general path is fp64)", d.k, CP_KMAX);
        if (!var && d.k > CP_KMAX && d.n_global > 0 && d.n_global != d.n)
            CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported row slabs with k = %d > %d images", d.k, CP_KMAX);
        if ((d.model == CP_MODEL_MF_NN_TV || d.model == CP_MODEL_VAR_TV) && d.bc != CP_BC_NEUMANN && d.bc != CP_BC_LINEAR)
            CP_FAIL(CP_ERR_BAD_BC, "cpengine:badBC Unknown boundary condition!");   // finite_difference_operator.m:35
        if (d.bc < 0 || d.bc > 2) CP_FAIL(CP_ERR_BAD_BC, "cpengine:badBC Unknown boundary condition!");
        if (d.batch > 1) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported batch > 1 for mf_nn models");
        if ((d.flags & CP_FLAG_MFNN_NORM21) && d.model != CP_MODEL_MF_NN_TV)
            CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg CP_FLAG_MFNN_NORM21 belongs to CP_MODEL_MF_NN_TV");
        if (d.n_global > 0 && d.n_global != d.n) {
            // the linear boundary row couples every column with the last one (finite_difference_operator.m:31)
            if (var || d.bc == CP_BC_LINEAR)
                CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported row slabs need an mf_nn model with neumann or dirichlet boundary");
            if (d.j_offset < 0 || d.j_offset + d.n > d.n_global)
                CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg slab [%d, %d) outside [0, %d)", d.j_offset, d.j_offset + d.n, d.n_global);
        }
    }
    // the field kernels address a field with 32-bit byte offsets (ldo / sto, cp_device.cuh) and count pixels in int
    if (((long long)d.m * ((long long)d.n + 2) + 8192) * (long long)sizeof(double) >= (1LL << 32))   // + the field stagger (below)
        CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported a field of %d x %d doubles reaches 4 GiB (32-bit offsets inside a field)", d.m, d.n);
    // the buffer-addressed kernels of the multi-image models reach the second field of a pair through the descriptor's
    // uniform offset: lane offset + field size stays below 2^32 (whichever way the range check counts that offset)
    if (mf && ((long long)d.m * ((long long)d.n + 2) + 8192) * (long long)sizeof(double) >= (1LL << 31))
        CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported a field of %d x %d doubles reaches 2 GiB (multi-image models)", d.m, d.n);
    cp_engine *h = new cp_engine();
    h->d = d;
    if (h->d.batch < 1) h->d.batch = 1;
    if (h->d.n_global <= 0) { h->d.n_global = d.n; h->d.j_offset = 0; }
    if (h->d.omega[0] == 0.0 && h->d.omega[1] == 0.0 && h->d.omega[2] == 0.0 && h->d.omega[3] == 0.0) {
        h->d.omega[1] = d.m * d.h[0];                     // IRE/evaluate_displacement.m:34-36
        h->d.omega[3] = h->d.n_global * d.h[1];
    }
    if (pw && (h->d.n_global != d.n)) { delete h; CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported slabs for pairwise models"); }
    h->fp32 = (d.flags & CP_FLAG_FP32) != 0;
    h->mf = mf;
    h->var = var;
    h->m = d.m; h->nl = d.n; h->npix = d.m * d.n; h->fs = d.m * (d.n + 2);
    // Field stagger (round 3, profiles/r03_field_pad.txt). m x (n + 2) doubles of the benchmark shapes are multiples of 64 KB, so
    // every field of a buffer started at the same phase of the memory system's address interleave and the ~21 streams a wave
    // reads and writes for one pixel met on the same channels: k_mf_fused_tv on a 4096 x 512 slab 1.25-1.30 ms, with fields
    // staggered by 1-3 KB 1.07 (-15 %); 4096 x 1024 -7 %, 2048^2 -6 %, k_mf_fused_curv on the slab -16 %; multiples of 4 KB bring
    // the loss back. Large multi-image fields are therefore padded so that the field stride is 3 KB modulo 64 KB (a multiple
    // of 512 B: the 64-row segments of a wave stay aligned). At most 64 KB per field. CP_TUNE_FIELD_PAD=<elements> overrides.
    if (mf && (long long)h->fs * (long long)sizeof(double) >= (256LL << 10)) {
        const long long base = (long long)h->fs * (long long)sizeof(double), target = 3072, period = 65536;
        h->fs += (int)((((target - base) % period + period) % period) / (long long)sizeof(double));
    }
    if (const char *ev = getenv("CP_TUNE_FIELD_PAD")) h->fs = d.m * (d.n + 2) + std::max(0, atoi(ev));
    h->batch = h->d.batch;
    if (mf) {
        h->K = d.k;
        h->km = (!var && d.k > CP_KMAX) ? CP_KBIG : CP_KMAX;
        h->reg = (d.model == CP_MODEL_MF_NN_TV || d.model == CP_MODEL_VAR_TV) ? 0 : 1;
        if (var) {
            h->nxf = 2 * h->K;                               // x = u only, NNE/var_tv...:102
            h->npf = (h->reg == 0 ? 5 : 3) * h->K;           // :117, var_curvature...:116
            h->hcols = CP_HCOLS_VAR;
            h->has_nu = true;                                // no nuclear-norm ball in these models
        } else {
            h->nxf = 3 * h->K;
            h->npf = (h->reg == 0 ? 6 : 4) * h->K;
            h->hcols = CP_HCOLS_MF;
        }
    } else {
        h->K = 1;
        h->gt = (d.model == CP_MODEL_TV_L1_REG) ? GT_SAD_REG : (d.model == CP_MODEL_TV_L2_REG ? GT_SSD_REG : GT_SAD_DEN);
        h->nc = (h->gt == GT_SAD_DEN) ? 1 : 2;
        h->nxf = h->nc;
        h->npf = 2 * h->nc;
        h->hcols = CP_HCOLS_PW;
    }
    h->last.assign(h->batch, cp_result{0, 0, 0, 0});
    *out = h;   // from here on cp_destroy cleans up
    int rc = CP_OK;
    auto body = [&]() -> int {
        HIPCHK(hipSetDevice(d.device));
        HIPCHK(hipStreamCreate(&h->own_stream));
        h->stream = h->own_stream;
        h->xN = (size_t)h->nxf * h->fs * h->batch;
        h->pN = (size_t)h->npf * h->fs * h->batch;
        const size_t gN = (size_t)2 * h->K * h->fs * h->batch, bN = (size_t)h->K * h->fs * h->batch;
        for (int a = 0; a < 2; ++a) {
            HIPCHK(hipMalloc(&h->X[a], sizeof(double) * h->xN));
            HIPCHK(hipMalloc(&h->P[a], sizeof(double) * h->pN));
        }
        HIPCHK(hipMalloc(&h->g, sizeof(double) * gN));
        HIPCHK(hipMalloc(&h->b, sizeof(double) * bN));
        for (int a = 0; a < 2; ++a) {
            HIPCHK(hipMemsetAsync(h->X[a], 0, sizeof(double) * h->xN, h->stream));
            HIPCHK(hipMemsetAsync(h->P[a], 0, sizeof(double) * h->pN, h->stream));
        }
        HIPCHK(hipMalloc(&h->Tu, sizeof(double) * bN));
        HIPCHK(hipMemsetAsync(h->g, 0, sizeof(double) * gN, h->stream));
        HIPCHK(hipMemsetAsync(h->b, 0, sizeof(double) * bN, h->stream));
        HIPCHK(hipMemsetAsync(h->Tu, 0, sizeof(double) * bN, h->stream));
        const size_t imgN = (size_t)h->K * h->m * h->d.n_global * h->batch;
        HIPCHK(hipMalloc(&h->img, sizeof(double) * imgN));
        if (!mf && h->gt != GT_SAD_DEN) HIPCHK(hipMalloc(&h->imgR, sizeof(double) * imgN));
        h->nblk = grid_for(h);
        if (const char *ev = getenv("CP_TUNE_BIG_GRID")) { if (h->km == CP_KBIG) h->nblk = std::max(1, std::min(h->nblk, atoi(ev))); }
        // measured at 4096^2 x 16 (NN-curvature, ms per power step): 768 blocks 13.1, 2048 11.6, 4096 9.7, 8192 9.4; 1024^2 x 16: 0.75 / 0.67 / 0.68
        h->nblk_pow = std::max(h->nblk, std::min((h->npix + CP_BLOCK - 1) / CP_BLOCK, CP_FUSED_MAXGRID));
        if (const char *ev = getenv("CP_TUNE_POW_GRID")) h->nblk_pow = std::max(1, atoi(ev));
        // k_mf_fused_tv: one workgroup task per (part of a column, group of 4 columns), up to 4096 workgroups (512 are
        // resident: 2 per CU), tasks strided by the grid; a multiple of 8 workgroups keeps a part on one XCD. Measured at
        // 4096^2 x 16: 10.4-10.7 ms for grids of 4096, 2048, 1024; 10.8 ms for 512 (persistent).
        if (const char *ev = getenv("CP_TUNE_FUSED_CW")) { const int v = atoi(ev); if (v == 1 || v == 2 || v == 4) h->fused_cw = v; }
        if (!(mf && !var && h->reg == 0) || h->d.bc == CP_BC_LINEAR || (h->d.flags & CP_FLAG_MFNN_NORM21)) h->fused_cw = 1;
        h->nblk_fused = (int)std::min<long long>((fused2_tasks(h->m, h->nl, h->fused_cw) + 7) / 8 * 8, CP_FUSED_MAXGRID);
        // 8 adjacent columns per workgroup with the LDS hand-over between neighbouring waves (cp_mfnn_fused.cuh) where that
        // instantiation exists (K = 16, neumann / dirichlet rows) and the problem still gives every CU a workgroup
        h->fused_waves = 8;
        bool force8 = false;   // CP_TUNE_FUSED_WAVES=8: also for small problems and fp32 storage (tests of the hand-over at small sizes)
        if (const char *ev = getenv("CP_TUNE_FUSED_WAVES")) { if (atoi(ev) == 4) h->fused_waves = CP_WAVES; force8 = atoi(ev) == 8; }
        if (const char *ev = getenv("CP_TUNE_PARTS8")) h->parts8 = std::max(0, atoi(ev));
        if (h->parts8 == 0 && mf) {
            // parts per column of the hand-over kernels: the fewest (longest tasks) for which the tasks fill whole rounds of the 256 CUs
            // to >= 95 % -- 1536^2 x 16: 2 parts (384 tasks, 1.5 rounds) 1.51 ms, 4 parts (768 tasks) 1.31, the 4-wave kernel 1.39;
            // if no count does (768^2: 288 / 384 / 576 tasks), fused2_cap's rule stays and handover_fits() below turns the kernel down
            const int cg8 = (h->nl + 7) / 8;
            for (int pc = 2; pc <= CP_FUSED_PARTS; ++pc) {
                const long long t = (long long)fused2_geom(h->m, 64, pc).parts * cg8;
                if (t >= 256 && t * 20 >= (t + 255) / 256 * 256 * 19) { h->parts8 = pc; break; }
            }
        }
        // Round 3: with the boundary array, the compiler-only hand-over fences and the field stagger the hand-over kernels win wherever
        // their tasks load the 256 CUs evenly (one workgroup per CU): 2048^2 x 16 2.38 -> 2.22 ms, 1024^2 x 16 0.634 -> 0.564, 1024^2 x 10
        // 0.422 -> 0.365, 512^2 x 16 0.181 -> 0.157 (var-TV 2.04 -> 1.78, 0.559 -> 0.475, 0.160 -> 0.134); with 1.5 rounds of tasks they
        // lose: 1536^2 1.39 -> 1.51 (384 tasks), 768^2 0.409 -> 0.423, 384^2 0.138 -> 0.156, and 256^2 (128 tasks) 0.064 -> 0.081.
        // Round 2's rule asked for >= 8 segments per task (m >= 3585).
        auto handover_fits = [&]() {
            const long long t = fused2_tasks(h->m, h->nl, 1, 8, h->parts8);
            return t >= 256 && t * 20 >= (t + 255) / 256 * 256 * 19;   // every CU a task, and the last round at least 95 % full
        };
        if (!(mf && !var && h->reg == 0) || h->d.bc == CP_BC_LINEAR || (h->d.flags & CP_FLAG_MFNN_NORM21) || h->fused_cw != 1 ||
            h->K <= 4 || (!force8 && (!handover_fits() || (h->d.flags & CP_FLAG_FP32))))
            // fp32 storage: 5.5 vs 6.9 ms with 4 waves. K = 5..15 (KA = 8, 12, 16 instantiations): 4096^2 x 12 7.39 vs 7.47 ms, x 8 4.83 vs 4.93
            h->fused_waves = CP_WAVES;
        h->nblk_tv = (int)std::min<long long>((fused2_tasks(h->m, h->nl, h->fused_cw, h->fused_waves, h->parts8) + 7) / 8 * 8, CP_FUSED_MAXGRID);
        {
            h->var_waves = 8;
            bool force8v = false;
            if (const char *ev = getenv("CP_TUNE_VAR_WAVES")) { if (atoi(ev) == 4) h->var_waves = CP_WAVES; force8v = atoi(ev) == 8; }
            if (!(var && h->reg == 0) || h->d.bc == CP_BC_LINEAR || (!force8v && (!handover_fits() || h->fp32))) h->var_waves = CP_WAVES;
            h->nblk_var = (int)std::min<long long>((fused2_tasks(h->m, h->nl, 1, h->var_waves, h->parts8) + 7) / 8 * 8, CP_FUSED_MAXGRID);
        }
        h->curv_fused = mf && !var && h->reg == 1;
        {
            if (const char *ev = getenv("CP_TUNE_CURV_WAVES")) { if (atoi(ev) == 8) h->curv_waves = 8; if (atoi(ev) == 4) h->curv_waves = CP_WAVES; }
            if (!h->curv_fused) h->curv_waves = CP_WAVES;
            const long long t = (long long)fused2_geom(h->m).parts * ((h->nl + h->curv_waves - 1) / h->curv_waves);
            h->nblk_curv = (int)std::min<long long>((t + 7) / 8 * 8, CP_FUSED_MAXGRID);
        }
        h->var_fused = var && h->nl == h->d.n_global;
        if (const char *ev = getenv("CP_TUNE_VAR_SPLIT")) { if (atoi(ev) != 0) h->var_fused = false; }
        if (const char *ev = getenv("CP_TUNE_CURV_SPLIT")) { if (atoi(ev) != 0) h->curv_fused = false; }
        if (mf) {
            // two rows per lane (16-byte accesses) where a column still gives every wave of a workgroup its 128 rows
            h->y2_r2 = (h->m % 2 == 0) && h->m >= 512;
            if (const char *ev = getenv("CP_TUNE_Y2_R2")) h->y2_r2 = (atoi(ev) != 0) && (h->m % 2 == 0);
            h->nblk_y2 = (int)std::min<long long>((curv_y2_tasks(h->m, h->nl, h->K, h->y2_r2 ? 512 : 256) + 7) / 8 * 8, CP_FUSED_MAXGRID);
        }
        if (const char *ev = getenv("CP_TUNE_FUSED_KA")) h->fused_ka = atoi(ev) != 0;
        if (const char *ev = getenv("CP_TUNE_PW_SPLIT")) h->pw_split = atoi(ev) != 0;
        if (const char *ev = getenv("CP_TUNE_TIME_ALL")) h->time_all = atoi(ev) != 0;
        if (const char *ev = getenv("CP_TUNE_FUSED_GRID")) h->nblk_fused = std::max(8, std::min(h->nblk_fused, atoi(ev) / 8 * 8));
        h->nblk_tv = std::min(h->nblk_tv, h->nblk_fused);
        {
            // measured (K = 10): 32^2 31.5 -> 22.9 us per launch, 128^2 (256 tasks, with the stage-1 reduction) 41.0 -> 27.6, 256^2 (1024 tasks) 48 -> 62
            long long ks_tasks = 256;
            if (const char *ev = getenv("CP_TUNE_FUSED_KS_TASKS")) ks_tasks = atoll(ev);
            const long long t = (long long)fused2_geom(h->m, 64).parts * h->nl;
            h->fused_ks = mf && !var && h->reg == 0 && h->d.bc != CP_BC_LINEAR && !(h->d.flags & CP_FLAG_MFNN_NORM21) &&
                          h->fused_cw == 1 && h->fused_waves == CP_WAVES && h->K >= 2 && t <= ks_tasks;
            if (h->fused_ks) h->nblk_tv = (int)t;
        }
        h->nblk_var = std::min(h->nblk_var, h->nblk_fused);
        if (mf) {
            for (int a = 0; a < 3; ++a) HIPCHK(hipMalloc(&h->W[a], sizeof(double) * h->km * h->km));
            HIPCHK(hipMalloc(&h->xtmp, sizeof(double) * h->xN));
            if (!var && h->reg == 0)
                for (int a = 0; a < 2; ++a) {
                    HIPCHK(hipMalloc(&h->bnd[a], sizeof(double) * bnd_elems(h->m, h->nl)));
                    HIPCHK(hipMemsetAsync(h->bnd[a], 0, sizeof(double) * bnd_elems(h->m, h->nl), h->stream));
                }
            if (h->var_fused)
                for (int a = 0; a < 2; ++a) {
                    HIPCHK(hipMalloc(&h->vlazy[a], sizeof(double) * 2 * (size_t)h->fs));
                    HIPCHK(hipMemsetAsync(h->vlazy[a], 0, sizeof(double) * 2 * (size_t)h->fs, h->stream));
                }
            HIPCHK(hipMalloc(&h->part, sizeof(double) * (size_t)std::max(std::max(std::max(std::max(h->nblk, h->nblk_fused), h->nblk_tv), h->nblk_y2), h->nblk_pow) *
                                           (h->km == CP_KBIG ? cp_pstride<CP_KBIG>() : CP_PSTRIDE)));
            HIPCHK(hipMalloc(&h->part2, sizeof(double) * (size_t)(CP_FUSED_MAXGRID / CP_S1_GROUP) * CP_PSTRIDE));
            HIPCHK(hipMalloc(&h->svout, sizeof(double) * (h->km + 2 * h->km * h->km)));
            HIPCHK(hipMalloc(&h->red, sizeof(double) * CP_PSTRIDE));
            if (h->fp32) HIPCHK(hipMalloc(&h->stage, sizeof(double) * h->npix));
            HIPCHK(hipMalloc(&h->vwarm, sizeof(double) * 2 * h->km * h->km));
            HIPCHK(hipMemsetAsync(h->vwarm, 0, sizeof(double) * 2 * h->km * h->km, h->stream));
            HIPCHK(hipMalloc(&h->ctrl, sizeof(CpCtrl)));
            HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(CpCtrl), h->stream));
            hipLaunchKernelGGL(k_mf_reset_ctrl, dim3(1), dim3(256), 0, h->stream, (CpCtrl *)h->ctrl, h->W[0], h->W[1], h->km);
            hipLaunchKernelGGL(k_mf_reset_ctrl, dim3(1), dim3(256), 0, h->stream, (CpCtrl *)h->ctrl, h->W[2],
                               h->svout + h->km + h->km * h->km, h->km);
        } else {
            if (h->fp32) HIPCHK(hipMalloc(&h->stage, sizeof(double) * h->npix));
            // marching kernel when it yields enough waves to fill the chip (8 per SIMD slot would be 8192); small problems stay
            // on k_pw_fused (one 256-pixel chunk per workgroup). Strip width, measured on 256 x 512^2 (ms per iteration of the
            // batch, fp64 / fp32 storage): 16 columns 2.04 / 1.03, 8: 1.94 / 0.99, 6: 1.79 / -, 4: 1.81 / 1.08, 3: 1.84, 2: 1.98
            // -- narrow strips recompute more ((C+1)/C evaluations per pixel) but start more waves, and a strip of 8 columns of
            // 512 rows puts the waves 32 KB apart; 2048^2: 4, 6 and 8 alike (0.135), 16: 0.143
            h->pw_two = (h->m % 2 == 0) && h->m >= 128;
            if (const char *ev = getenv("CP_TUNE_PW_TWO")) h->pw_two = (atoi(ev) != 0) && (h->m % 2 == 0);
            h->pw_groups = h->pw_two ? (h->m + 127) / 128 : (h->m + 62) / 63;
            // k_pw_march2 (256 x 512^2): fp64 strips of 4 / 6 / 8 / 12 / 16 / 32 columns 1.64 / 1.617 / 1.70 / 1.69 / 1.73 / 1.68 ms
            // (k_pw_march with 6: 1.785); fp32 storage 6 / 8 / 16 / 32: 1.10 / 1.07 / 1.02 / 1.01 (k_pw_march with 8: 1.03)
            const std::vector<int> cands = h->pw_two ? (h->fp32 ? std::vector<int>{32, 16, 8, 4} : std::vector<int>{6, 4})
                                                     : std::vector<int>{h->fp32 ? 8 : 6, 4};
            for (int strip : cands) {
                const long long waves = (long long)h->pw_groups * ((h->nl + strip - 1) / strip) * h->batch;
                if (waves >= (h->pw_two ? 4096 : 8192)) { h->pw_strip = strip; break; }   // a wave of k_pw_march2 carries twice the rows
            }
            if (const char *ev = getenv("CP_TUNE_PW_STRIP")) h->pw_strip = std::max(0, atoi(ev));
            if (h->pw_strip > 0) {
                h->pw_waves = h->pw_groups * ((h->nl + h->pw_strip - 1) / h->pw_strip);
                h->pw_grid = (h->pw_waves + CP_WAVES - 1) / CP_WAVES;
            }
            HIPCHK(hipMalloc(&h->part, sizeof(double) * (size_t)std::max(h->nblk, h->pw_grid) * PWF_PSTRIDE * h->batch));
            HIPCHK(hipMalloc(&h->part2, sizeof(double) * (size_t)h->nblk * PW_PSTRIDE * h->batch));
            HIPCHK(hipMalloc(&h->ctrl, sizeof(PwCtrl) * h->batch));
            HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(PwCtrl) * h->batch, h->stream));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipGetLastError());
        return CP_OK;
    };
    rc = body();
    if (rc != CP_OK) { cp_destroy(h); *out = nullptr; }
    return rc;
}

// ------------------------------------------------------------------------------------------------
extern "C" int cp_set_images(cp_handle h, const double *a, const double *b) {
    if (!h || !a) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle or image");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    const size_t imgN = (size_t)h->K * h->m * h->d.n_global * h->batch;
    if (!h->mf && h->gt == GT_SAD_DEN) {
        // the noisy image g plays the role of b (field layout), PDO/test_TV_L1_denoising.m:58
        CPCHK(upload_fields(h, h->b, a, h->batch));     // one field per problem
        h->linearized = true;
    } else {
        HIPCHK(hipMemcpyAsync(h->img, a, sizeof(double) * imgN, hipMemcpyHostToDevice, h->stream));
        if (!h->mf) {
            if (!b) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg reference image R missing");
            HIPCHK(hipMemcpyAsync(h->imgR, b, sizeof(double) * imgN, hipMemcpyHostToDevice, h->stream));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        h->linearized = false;
    }
    h->has_images = true;
    return CP_OK;
}

static int set_state(cp_handle h, double *dev, const double *host, int nfields) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    return upload_fields(h, dev, host, nfields * h->batch);
}
extern "C" int cp_set_primal(cp_handle h, const double *x0) { return set_state(h, h ? h->X[0] : nullptr, x0, h ? h->nxf : 0); }
extern "C" int cp_set_dual(cp_handle h, const double *y0) { return set_state(h, h ? h->P[0] : nullptr, y0, h ? h->npf : 0); }
extern "C" int cp_get_primal(cp_handle h, double *x) {
    if (!h || !x) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle or buffer");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    return download_fields(h, h->X[0], x, h->nxf * h->batch);
}
extern "C" int cp_get_dual(cp_handle h, double *y) {
    if (!h || !y) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle or buffer");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    return download_fields(h, h->P[0], y, h->npf * h->batch);
}

// ------------------------------------------------------------------------------------------------
extern "C" int cp_linearize(cp_handle h, const double *u0) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    if (!h->mf && h->gt == GT_SAD_DEN) return CP_OK;   // nothing to linearise
    if (!h->has_images) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_linearize before cp_set_images");
    const double *xsrc = h->X[0];
    if (u0) {
        // development point given explicitly: stage it in the spare x buffer
        const int nuf = h->mf ? 2 * h->K : 2;
        if (h->mf) {
            CPCHK(upload_fields(h, h->X[1], u0, nuf));
        } else {
            CPCHK(upload_fields(h, h->X[1], u0, 2 * h->batch));   // nxf == 2 for the registration models
        }
        xsrc = h->X[1];
    }
    if (h->mf) {
        WarpParams W;
        W.m = h->m; W.nl = h->nl; W.K = h->K; W.npix = h->npix; W.n_global = h->d.n_global; W.j_off = h->d.j_offset;
        W.fs = h->fs; W.mi = h->m; W.ni = h->d.n_global; W.ifs = h->m * h->d.n_global;
        W.om0 = h->d.omega[0]; W.om2 = h->d.omega[2];
        W.hx = (h->d.omega[1] - h->d.omega[0]) / h->m;              // IRE/cell_centered_grid.m:21-22
        W.hy = (h->d.omega[3] - h->d.omega[2]) / h->d.n_global;
        W.inv_h1 = 1.0 / h->d.h[0]; W.inv_h2 = 1.0 / h->d.h[1];
        // slabs: also linearise the halo columns that lie inside the image (the fused pass recomputes the primal
        // update at j+1); the caller has exchanged the halo columns of u before
        W.jlo = (h->d.j_offset > 0) ? -1 : 0;
        const int jhi = (h->d.j_offset + h->nl < h->d.n_global) ? h->nl : h->nl - 1;
        W.npix = h->m * (jhi - W.jlo + 1);
        const long long tot = (long long)W.npix * h->K;
        const int grid = (int)((tot + CP_BLOCK - 1) / CP_BLOCK);
        if (h->fp32)
            hipLaunchKernelGGL((k_mf_warp<float>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, W, h->img, (const float *)xsrc,
                               (float *)h->Tu, (float *)h->g, (float *)h->b);
        else
            hipLaunchKernelGGL((k_mf_warp<double>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, W, h->img, xsrc, h->Tu, h->g, h->b);
        if (h->var)   // b <- B*b, NNE/var_tv_registration_no_ref_ml.m:153-154
            { if (h->fp32) hipLaunchKernelGGL(k_var_center_b<float>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, mf_params(h, 1.0, 1.0, 1.0), (float *)h->b);
              else hipLaunchKernelGGL(k_var_center_b<double>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, mf_params(h, 1.0, 1.0, 1.0), h->b); }
    } else {
        PwWarpParams W;
        W.m = h->m; W.n = h->nl; W.npix = h->npix; W.fs = h->fs; W.nxf = h->nxf;
        W.hx = (h->d.omega[1] - h->d.omega[0]) / h->m;
        W.hy = (h->d.omega[3] - h->d.omega[2]) / h->nl;
        W.inv_h1 = 1.0 / h->d.h[0]; W.inv_h2 = 1.0 / h->d.h[1];
        const int grid = (h->npix + CP_BLOCK - 1) / CP_BLOCK;
        if (h->fp32)
            hipLaunchKernelGGL((k_pw_warp<float>), dim3(grid, h->batch), dim3(CP_BLOCK), 0, h->stream, W, h->img, h->imgR,
                               (const float *)xsrc, (float *)h->Tu, (float *)h->g, (float *)h->b);
        else
            hipLaunchKernelGGL((k_pw_warp<double>), dim3(grid, h->batch), dim3(CP_BLOCK), 0, h->stream, W, h->img, h->imgR,
                               xsrc, h->Tu, h->g, h->b);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->linearized = true;
    return CP_OK;
}

extern "C" int cp_get_linearization(cp_handle h, double *T_u, double *grad, double *b) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    HIPCHK(hipSetDevice(h->d.device));
    if (!h->linearized) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState no linearisation yet");
    if (T_u) CPCHK(download_fields(h, h->Tu, T_u, h->K * h->batch));
    if (grad) CPCHK(download_fields(h, h->g, grad, 2 * h->K * h->batch));
    if (b) CPCHK(download_fields(h, h->b, b, h->K * h->batch));
    return CP_OK;
}

extern "C" int cp_set_linearization(cp_handle h, const double *T_u, const double *grad, const double *b) {
    if (!h || !grad || !b) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_set_linearization needs a handle, grad and b");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    if (!h->mf && h->gt == GT_SAD_DEN) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported the denoiser has no linearisation");
    if (h->d.n_global != h->nl) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported cp_set_linearization on a slab");
    if (T_u) CPCHK(upload_fields(h, h->Tu, T_u, h->K * h->batch));
    CPCHK(upload_fields(h, h->g, grad, 2 * h->K * h->batch));
    CPCHK(upload_fields(h, h->b, b, h->K * h->batch));
    if (h->var) {
        { if (h->fp32) hipLaunchKernelGGL(k_var_center_b<float>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, mf_params(h, 1.0, 1.0, 1.0), (float *)h->b);
              else hipLaunchKernelGGL(k_var_center_b<double>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, mf_params(h, 1.0, 1.0, 1.0), h->b); }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h->linearized = true;
    return CP_OK;
}

extern "C" int cp_set_nu(cp_handle h, double nu) {
    if (!h || !h->mf) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_set_nu needs an mf_nn handle");
    if (!(nu > 0.0)) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg nu must be positive");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    HIPCHK(hipMemcpyAsync(&((CpCtrl *)h->ctrl)->nu, &nu, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->has_nu = true;
    return CP_OK;
}

// ------------------------------------------------------------------------------------------------
static void launch_gram_field(cp_engine *h, const MfParams &P, const double *field, const double *Vm) {
    if (h->km == CP_KBIG)
        hipLaunchKernelGGL((k_mf_gram_field<double, CP_KBIG>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, field, Vm, h->part);
    else if (h->fp32)
        hipLaunchKernelGGL((k_mf_gram_field<float>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, (const float *)field, Vm, h->part);
    else
        hipLaunchKernelGGL((k_mf_gram_field<double>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, field, Vm, h->part);
}
// y = A*(scale*x) and xo = A'*y of the power iteration; its vectors are doubles in every mode (the spare buffers
// are double-sized), only the gradient field follows the storage mode
static void launch_mf_applyA(cp_engine *h, const MfParams &P, const double *x, double scale, double *y, int grid) {
    if (h->fp32) {
        if (h->reg == 0) hipLaunchKernelGGL((k_mf_applyA<float, 0>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, x, scale, (const float *)h->g, y);
        else hipLaunchKernelGGL((k_mf_applyA<float, 1>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, x, scale, (const float *)h->g, y);
    } else {
        if (h->reg == 0) hipLaunchKernelGGL((k_mf_applyA<double, 0>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, x, scale, h->g, y);
        else hipLaunchKernelGGL((k_mf_applyA<double, 1>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, x, scale, h->g, y);
    }
}
static void launch_mf_applyAT(cp_engine *h, const MfParams &P, const double *y, double *xo, int grid) {
    if (h->fp32) {
        if (h->reg == 0) hipLaunchKernelGGL((k_mf_applyAT<float, 0>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, y, (const float *)h->g, xo, h->part);
        else hipLaunchKernelGGL((k_mf_applyAT<float, 1>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, y, (const float *)h->g, xo, h->part);
    } else {
        if (h->reg == 0) hipLaunchKernelGGL((k_mf_applyAT<double, 0>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, y, h->g, xo, h->part);
        else hipLaunchKernelGGL((k_mf_applyAT<double, 1>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, y, h->g, xo, h->part);
    }
}
static void launch_any_applyA(cp_engine *h, const MfParams &P, const double *x, double scale, double *y, int grid = 0) {
    if (grid <= 0) grid = h->nblk;
    if (h->var && h->fp32) {
        if (h->reg == 0) hipLaunchKernelGGL((k_var_applyA<float, 0>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, x, scale, (const float *)h->g, y);
        else hipLaunchKernelGGL((k_var_applyA<float, 1>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, x, scale, (const float *)h->g, y);
    } else if (h->var && h->reg == 0) hipLaunchKernelGGL((k_var_applyA<double, 0>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, x, scale, h->g, y);
    else if (h->var) hipLaunchKernelGGL((k_var_applyA<double, 1>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, x, scale, h->g, y);
    else launch_mf_applyA(h, P, x, scale, y, grid);
}
static void launch_any_applyAT(cp_engine *h, const MfParams &P, const double *y, double *xo, int grid = 0) {
    if (grid <= 0) grid = h->nblk;
    if (h->var && h->fp32) {
        if (h->reg == 0) hipLaunchKernelGGL((k_var_applyAT<float, 0>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, y, (const float *)h->g, xo, h->part);
        else hipLaunchKernelGGL((k_var_applyAT<float, 1>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, y, (const float *)h->g, xo, h->part);
    } else if (h->var && h->reg == 0) hipLaunchKernelGGL((k_var_applyAT<double, 0>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, y, h->g, xo, h->part);
    else if (h->var) hipLaunchKernelGGL((k_var_applyAT<double, 1>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, P, y, h->g, xo, h->part);
    else launch_mf_applyAT(h, P, y, xo, grid);
}
// the power iteration's vectors move between host and device as doubles in every mode
struct Fp64Transfers {
    cp_engine *h; bool saved;
    explicit Fp64Transfers(cp_engine *e) : h(e), saved(e->fp32) { h->fp32 = false; }
    ~Fp64Transfers() { h->fp32 = saved; }
};

// singular values of reshape(B*field, mn, k): Gram -> Jacobi, then a second Gram in the rotated basis
// (column norms are sums of squares, so small singular values keep their relative accuracy).
extern "C" int cp_singular_values(cp_handle h, int which, double *sv) {
    if (!h || !h->mf || !sv) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_singular_values needs an mf_nn handle");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    if (h->d.n_global != h->nl) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported cp_singular_values on a slab");
    const double *field;
    if (which == 0) {
        if (!h->linearized) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_singular_values(0) before cp_linearize");
        field = h->Tu;
    } else if (which == 1) {
        if (h->var) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported the variance models have no L block");
        field = h->fp32 ? (const double *)((const float *)h->X[0] + (size_t)2 * h->K * h->fs) : h->X[0] + (size_t)2 * h->K * h->fs;
    } else {
        CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg which = %d", which);
    }
    MfParams P = mf_params(h, 1.0, 1.0, 1.0);
    const int km = h->km;
    double *ident = h->svout + km + km * km, *vout = h->svout + km;
    auto sv_eig = [&](const double *Vin) {
        if (km == CP_KBIG) {
            int nr = 0;
            const double *rows = staged_partials(h, h->nblk, &nr);   // only the Gram block of the rows is meaningful here
            hipLaunchKernelGGL(k_big_sv_eig, dim3(1), dim3(CP_BIG_THREADS), 0, h->stream, nr, rows, Vin, h->svout);
        }
        else hipLaunchKernelGGL(k_mf_sv_eig, dim3(1), dim3(CP_BLOCK), 0, h->stream, h->nblk, h->part, Vin, h->svout);
    };
    launch_gram_field(h, P, field, ident);
    sv_eig(ident);
    // second pass in the rotated basis; W[2] (scratch) receives the rotation
    HIPCHK(hipMemcpyAsync(h->W[2], vout, sizeof(double) * km * km, hipMemcpyDeviceToDevice, h->stream));
    launch_gram_field(h, P, field, h->W[2]);
    sv_eig(h->W[2]);
    HIPCHK(hipGetLastError());
    double lam[CP_KBIG];
    HIPCHK(hipMemcpyAsync(lam, h->svout, sizeof(double) * km, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> s(h->K);
    for (int i = 0; i < h->K; ++i) s[i] = std::sqrt(std::max(lam[i], 0.0));
    std::sort(s.begin(), s.end(), [](double a, double b) { return a > b; });
    for (int i = 0; i < h->K; ++i) sv[i] = s[i];
    return CP_OK;
}

// ------------------------------------------------------------------------------------------------
// sum of squares over the pixels of `nfields` fields of doubles (start vector of the power iteration)
__global__ __launch_bounds__(CP_BLOCK) void k_sumsq_fields(int npix, int m, int fs, int nfields, const double *__restrict__ x,
                                                           double *__restrict__ part) {
    __shared__ double red[CP_WAVES * 2];
    double ss[1] = {0.0}, mm[1] = {0.0};
    const long long tot = (long long)npix * nfields;
    for (long long gid = (long long)blockIdx.x * CP_BLOCK + threadIdx.x; gid < tot; gid += (long long)gridDim.x * CP_BLOCK) {
        const int f = (int)(gid / npix), q = (int)(gid - (long long)f * npix);
        const double v = x[(size_t)f * fs + m + q];
        ss[0] += v * v;
    }
    block_reduce_store<1, 1>(ss, mm, red, part + (size_t)blockIdx.x * CP_PSTRIDE);
}

// sum of the first entry of `rows` partial rows (fixed order: thread t takes rows t, t + 256, ...; then a tree over the 256 threads)
__global__ __launch_bounds__(256) void k_pow_rowsum(const double *__restrict__ part, int rows, int stride, double *__restrict__ out) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) a += part[(size_t)r * stride];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

extern "C" int cp_norm_estimate(cp_handle h, const double *x0, double *est, int32_t *iters) {
    if (!h || !h->mf || !est) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_norm_estimate needs an mf_nn handle");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    if (!h->linearized) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_norm_estimate before cp_linearize");
    if (h->d.n_global != h->nl) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported cp_norm_estimate on a slab");
    MfParams P = mf_params(h, 1.0, 1.0, 1.0);
    double *xa = h->X[1], *xb = h->xtmp, *yb = h->P[1];   // spare buffers (state lives in X[0], P[0])
    HIPCHK(hipMemsetAsync(xa, 0, sizeof(double) * h->xN, h->stream));
    HIPCHK(hipMemsetAsync(xb, 0, sizeof(double) * h->xN, h->stream));
    if (x0) { Fp64Transfers t(h); CPCHK(upload_fields(h, xa, x0, h->nxf)); }
    else hipLaunchKernelGGL(k_mf_hash_start, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->nxf, xa);
    // ||x0||. Round 3: the partial sums of a step are added on the device (k_pow_rowsum) and ONE double comes back to the host; before,
    // whole partial rows (768 x 532 doubles through pageable memory) came back every step, half of a step's 26 ms at 4096^2 x 16.
    // The two kernels of a step have no LDS tiles, so their grid may exceed that of the field passes (CP_TUNE_POW_GRID).
    const int pgrid = h->nblk_pow;
    auto sumsq_partials = [&](int rows, double &out) -> int {
        hipLaunchKernelGGL(k_pow_rowsum, dim3(1), dim3(256), 0, h->stream, (const double *)h->part, rows, (int)CP_PSTRIDE, h->red);
        HIPCHK(hipMemcpyAsync(&out, h->red, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return CP_OK;
    };
    double nrm2 = 0.0;
    // on the device (k_sumsq_fields, as cp_mg_power_start): downloading the start vector to sum it on the host cost 2-3 s
    // per call at 4096^2 x 16 (6.4 GB through pageable memory)
    hipLaunchKernelGGL(k_sumsq_fields, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, h->npix, h->m, h->fs, h->nxf, xa, h->part);
    HIPCHK(hipGetLastError());
    CPCHK(sumsq_partials(h->nblk, nrm2));
    double scale = 1.0 / std::sqrt(nrm2);                       // NNE/matrix_norm.m:23
    const double tol = 1e-6;
    const int maxIter = 100;
    std::vector<double> e;
    int i = 0;
    while (i < 2 || std::fabs(e[i - 1] - e[i - 2]) > tol * e[i - 1]) {   // :25
        ++i;
        launch_any_applyA(h, P, xa, scale, yb, pgrid);
        launch_any_applyAT(h, P, yb, xb, pgrid);
        HIPCHK(hipGetLastError());
        double ss;
        CPCHK(sumsq_partials(pgrid, ss));
        const double normY = std::sqrt(ss);                     // :35
        scale = 1.0 / normY;                                    // :38
        std::swap(xa, xb);
        e.push_back(std::sqrt(normY));                          // :40
        if (i == maxIter) break;
    }
    // the spare buffers are scratch again; the dual spare must be zero-free of halo garbage -> clear
    HIPCHK(hipMemsetAsync(h->P[1], 0, sizeof(double) * h->pN, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *est = e.back();
    if (iters) *iters = i;
    return CP_OK;
}

// y = A*x / x = A'*y on caller-supplied vectors, staged in the spare buffers as the power iteration does
static int apply_operator(cp_handle h, bool transpose, const double *in, double *out) {
    if (!h || !h->mf || !in || !out) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_apply_A / cp_apply_At need an mf_nn or var handle and two buffers");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    if (h->d.n_global != h->nl) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported cp_apply_A / cp_apply_At on a slab");
    const MfParams P = mf_params(h, 1.0, 1.0, 1.0);
    double *xa = h->X[1], *yb = h->P[1];   // spare buffers (state lives in X[0], P[0]); their halo columns stay zero
    Fp64Transfers t(h);
    if (!transpose) {
        HIPCHK(hipMemsetAsync(xa, 0, sizeof(double) * h->xN, h->stream));
        CPCHK(upload_fields(h, xa, in, h->nxf));
        launch_any_applyA(h, P, xa, 1.0, yb);
        HIPCHK(hipGetLastError());
        CPCHK(download_fields(h, yb, out, h->npf));
    } else {
        HIPCHK(hipMemsetAsync(yb, 0, sizeof(double) * h->pN, h->stream));
        CPCHK(upload_fields(h, yb, in, h->npf));
        launch_any_applyAT(h, P, yb, xa);
        HIPCHK(hipGetLastError());
        CPCHK(download_fields(h, xa, out, h->nxf));
    }
    HIPCHK(hipMemsetAsync(h->X[1], 0, sizeof(double) * h->xN, h->stream));
    HIPCHK(hipMemsetAsync(h->P[1], 0, sizeof(double) * h->pN, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return CP_OK;
}
extern "C" int cp_apply_A(cp_handle h, const double *x, double *y) { return apply_operator(h, false, x, y); }
extern "C" int cp_apply_At(cp_handle h, const double *y, double *x) { return apply_operator(h, true, y, x); }

// ------------------------------------------------------------------------------------------------
template <int MODE>
static void launch_mf_dual(cp_engine *h, const MfParams &P, const double *xc, const double *xo, const double *pin,
                           double *pout, const double *W, int cc, int co) {
    if (h->km == CP_KBIG) {   // K > 16: the same kernel built for 32 images (fp64 storage only)
        if (h->reg == 0)
            hipLaunchKernelGGL((k_mf_dual<double, MODE, 0, CP_KBIG>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xc, xo, pin, pout,
                               h->g, h->b, W, (const CpCtrl *)h->ctrl, cc, co, h->part);
        else
            hipLaunchKernelGGL((k_mf_dual<double, MODE, 1, CP_KBIG>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xc, xo, pin, pout,
                               h->g, h->b, W, (const CpCtrl *)h->ctrl, cc, co, h->part);
        return;
    }
    if (h->fp32) {
        const float *xcf = (const float *)xc, *xof = (const float *)xo, *pinf = (const float *)pin;
        const float *gf = (const float *)h->g, *bf = (const float *)h->b;
        if (h->reg == 0)
            hipLaunchKernelGGL((k_mf_dual<float, MODE, 0>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xcf, xof, pinf,
                               (float *)pout, gf, bf, W, (const CpCtrl *)h->ctrl, cc, co, h->part);
        else
            hipLaunchKernelGGL((k_mf_dual<float, MODE, 1>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xcf, xof, pinf,
                               (float *)pout, gf, bf, W, (const CpCtrl *)h->ctrl, cc, co, h->part);
        return;
    }
    if (h->reg == 0)
        hipLaunchKernelGGL((k_mf_dual<double, MODE, 0>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xc, xo, pin, pout, h->g,
                           h->b, W, (const CpCtrl *)h->ctrl, cc, co, h->part);
    else
        hipLaunchKernelGGL((k_mf_dual<double, MODE, 1>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xc, xo, pin, pout, h->g,
                           h->b, W, (const CpCtrl *)h->ctrl, cc, co, h->part);
}
template <int MODE>
static void launch_mf_primal(cp_engine *h, const MfParams &P, const double *xc, double *xn, const double *p,
                             const double *W, int cc) {
    if (h->km == CP_KBIG) {
        if (h->reg == 0)
            hipLaunchKernelGGL((k_mf_primal<double, MODE, 0, CP_KBIG>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xc, xn, p, h->g, W,
                               (const CpCtrl *)h->ctrl, cc, h->part);
        else
            hipLaunchKernelGGL((k_mf_primal<double, MODE, 1, CP_KBIG>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xc, xn, p, h->g, W,
                               (const CpCtrl *)h->ctrl, cc, h->part);
        return;
    }
    if (h->fp32) {
        const float *xcf = (const float *)xc, *pf = (const float *)p, *gf = (const float *)h->g;
        if (h->reg == 0)
            hipLaunchKernelGGL((k_mf_primal<float, MODE, 0>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xcf, (float *)xn, pf,
                               gf, W, (const CpCtrl *)h->ctrl, cc, h->part);
        else
            hipLaunchKernelGGL((k_mf_primal<float, MODE, 1>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xcf, (float *)xn, pf,
                               gf, W, (const CpCtrl *)h->ctrl, cc, h->part);
        return;
    }
    if (h->reg == 0)
        hipLaunchKernelGGL((k_mf_primal<double, MODE, 0>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xc, xn, p, h->g, W,
                           (const CpCtrl *)h->ctrl, cc, h->part);
    else
        hipLaunchKernelGGL((k_mf_primal<double, MODE, 1>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xc, xn, p, h->g, W,
                           (const CpCtrl *)h->ctrl, cc, h->part);
}

// warm start of the eigensolves from the previous iteration's bases; a cold start every 128 iterations (and at the first
// one) keeps the accumulated product of rotations orthogonal to rounding (the drift is ~16 eps per warm iteration)
static inline int efused_warm(int n) { return (n > 1 && (n & 127) != 1) ? 1 : 0; }
// split passes: k_mf_eig keeps its basis in vwarm[0..255], k_mf_fin in vwarm[256..511]
static inline EigArgs with_warm(EigArgs A, const cp_engine *h, int n, int slot) {
    A.vwarm = h->vwarm + h->km * h->km * slot;
    A.warm = efused_warm(n);
    return A;
}

static EigArgs eig_args(const cp_engine *h, const MfParams &P, int rp, int rd, int check, int update, int newbuf,
                        double tol, int monitor = 0, int compute_cn = 0) {
    EigArgs A;
    A.monitor = monitor;
    A.compute_cn = compute_cn;
    A.tau = P.tau;
    A.nblk = h->nblk; A.K = h->K; A.reg = h->reg; A.rp = rp; A.rd = rd; A.check_stop = check; A.update = update;
    A.newbuf = newbuf; A.tol = tol; A.mu1 = P.mu1; A.mu2 = P.mu2; A.sigma = P.sigma; A.Ntot = P.Ntot;
    A.norm21 = P.norm21;
    A.vwarm = nullptr; A.warm = 0;   // cold eigensolve, nothing kept (monitor rows, start-up); see eig_args_warm
    return A;
}

// First stage of the reduction of a field kernel's partial rows: output row r = fixed-order sum (slots < CP_NSUM and
// the Gram entries) / maximum (slots CP_NSUM..CP_NSCAL-1) of input rows [r*CP_S1_GROUP, (r+1)*CP_S1_GROUP).
// blockIdx.y selects a chunk of 256 columns; 1024 threads = 4 row quarters x 256 columns, a thread sums 16 rows of one
// column (row index clamped, loads unconditional), the quarters are combined in fixed order through LDS. The kernel is
// a few dependent round trips long: many waves in flight matter, not the work per thread (17 MB of partial rows at
// 4096 workgroups; one workgroup per row group walking its three column chunks in turn took 18 us).
__global__ __launch_bounds__(1024) void k_mf_reduce_stage1(int nrows, const double *__restrict__ part, double *__restrict__ out) {
    __shared__ double stage[4][256];
    const int tx = threadIdx.x & 255, sub = threadIdx.x >> 8;
    const int r1 = min(nrows, (int)(blockIdx.x + 1) * CP_S1_GROUP);
    const int r0 = blockIdx.x * CP_S1_GROUP + sub * (CP_S1_GROUP / 4);
    const int c = min((int)blockIdx.y * 256 + tx, CP_PSTRIDE - 1);
    const bool is_max = (c >= CP_NSUM && c < CP_NSCAL);
    double a[CP_S1_GROUP / 4];
#pragma unroll
    for (int u = 0; u < CP_S1_GROUP / 4; ++u) {
        const double o = part[(size_t)min(r0 + u, nrows - 1) * CP_PSTRIDE + c];
        a[u] = (r0 + u < r1) ? o : (is_max ? -1.0e300 : 0.0);
    }
#pragma unroll
    for (int w = CP_S1_GROUP / 8; w >= 1; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) a[u] = is_max ? fmax(a[u], a[u + w]) : a[u] + a[u + w];
    stage[sub][tx] = a[0];
    __syncthreads();
    if (sub == 0 && (int)blockIdx.y * 256 + tx < CP_PSTRIDE) {
        const double v = is_max ? fmax(fmax(stage[0][tx], stage[1][tx]), fmax(stage[2][tx], stage[3][tx]))
                                : (stage[0][tx] + stage[1][tx]) + (stage[2][tx] + stage[3][tx]);
        out[(size_t)blockIdx.x * CP_PSTRIDE + c] = v;
    }
}

// the same for the partial rows of the KM = 32 field kernels (cp_pstride<32>() doubles per row): groups of 32 rows, fixed order
__global__ __launch_bounds__(256) void k_big_reduce_stage1(int nrows, const double *__restrict__ part, double *__restrict__ out) {
    constexpr int PS = cp_pstride<CP_KBIG>();
    const int col = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (col >= PS) return;
    const bool is_max = (col >= CP_NSUM && col < CP_NSCAL);
    const int r1 = min(nrows, g * 32 + 32);
    double v[4] = {is_max ? -1.0e300 : 0.0, is_max ? -1.0e300 : 0.0, is_max ? -1.0e300 : 0.0, is_max ? -1.0e300 : 0.0};
    int r = g * 32;
    for (; r + 3 < r1; r += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double o = part[(size_t)(r + u) * PS + col];
            v[u] = is_max ? fmax(v[u], o) : v[u] + o;
        }
    }
    for (int u = 0; r < r1; ++r, ++u) {
        const double o = part[(size_t)r * PS + col];
        v[u] = is_max ? fmax(v[u], o) : v[u] + o;
    }
    out[(size_t)g * PS + col] = is_max ? fmax(fmax(v[0], v[1]), fmax(v[2], v[3])) : (v[0] + v[1]) + (v[2] + v[3]);
}

// partial rows of the last field kernel as the single-workgroup kernels read them (after stage 1 if there are more than
// 128: a single workgroup reads about 18 rows per microsecond)
static const double *staged_partials(cp_engine *h, int nrows, int *rows, bool scalars_only) {
    if (h->km == CP_KBIG) {   // K > 16: rows of cp_pstride<32>() doubles; one workgroup reads 16.5 KB per row (0.2 ms for 256 rows)
        if (nrows <= 32) { *rows = nrows; return h->part; }
        const int r2 = (nrows + 31) / 32;
        hipLaunchKernelGGL(k_big_reduce_stage1, dim3((cp_pstride<CP_KBIG>() + 255) / 256, r2), dim3(256), 0, h->stream, nrows, h->part, h->part2);
        *rows = r2;
        return h->part2;
    }
    if (nrows <= 128) { *rows = nrows; return h->part; }
    const int r2 = (nrows + CP_S1_GROUP - 1) / CP_S1_GROUP;
    hipLaunchKernelGGL(k_mf_reduce_stage1, dim3(r2, scalars_only ? 1 : (CP_PSTRIDE + 255) / 256), dim3(1024), 0, h->stream, nrows, h->part,
                       h->part2);
    *rows = r2;
    return h->part2;
}
static const double *fused_partials(cp_engine *h, int *rows) {
    return staged_partials(h, (h->reg == 0 && !h->var) ? h->nblk_tv : ((h->reg == 1 && !h->var) ? h->nblk_curv : h->nblk_fused), rows);
}

// the single-workgroup kernels of the split passes for this handle's extent (16: cp_mfnn_small.cuh, 32: cp_mfnn_big.cuh)
static void launch_eig(cp_engine *h, const EigArgs &A, const double *rows, double *Wout) {
    if (h->km == CP_KBIG)
        hipLaunchKernelGGL(k_big_eig, dim3(1), dim3(CP_BIG_THREADS), 0, h->stream, A, rows, (CpCtrl *)h->ctrl, Wout, h->hist_p, h->hist_d);
    else
        hipLaunchKernelGGL(k_mf_eig, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, rows, (CpCtrl *)h->ctrl, Wout, h->hist_p, h->hist_d);
}
static void launch_fin(cp_engine *h, const EigArgs &A, const double *rows) {
    if (h->km == CP_KBIG)
        hipLaunchKernelGGL(k_big_fin, dim3(1), dim3(CP_BIG_THREADS), 0, h->stream, A, rows, (CpCtrl *)h->ctrl, h->hist_p, h->hist_d);
    else
        hipLaunchKernelGGL(k_mf_fin, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, rows, (CpCtrl *)h->ctrl, h->hist_p, h->hist_d);
}

// iterations n = n0+1 .. n0+count of PDO/chambolle_pock.m:156-179 (state parity follows n)
static int mf_iterations(cp_engine *h, const MfParams &P, int n0, int count, int check, double tol) {
    CpCtrl *ctrl = (CpCtrl *)h->ctrl;
    for (int n = n0 + 1; n <= n0 + count; ++n) {
        const int cur = (n - 1) & 1, nxt = n & 1;
        CPCHK(timer_begin(h, T_DUAL));
        launch_mf_dual<0>(h, P, h->X[cur], h->X[nxt], h->P[cur], h->P[nxt], h->W[cur], cur, nxt);
        CPCHK(timer_end(h, T_DUAL));
        CPCHK(timer_begin(h, T_EIG));
        EigArgs Ae = with_warm(eig_args(h, P, n - 1, n, check, 0, 0, tol), h, n, 0);
        const double *rows_e = staged_partials(h, h->nblk, &Ae.nblk);
        launch_eig(h, Ae, rows_e, h->W[nxt]);
        CPCHK(timer_end(h, T_EIG));
        CPCHK(timer_begin(h, T_PRIMAL));
        launch_mf_primal<0>(h, P, h->X[cur], h->X[nxt], h->P[nxt], h->W[nxt], cur);
        CPCHK(timer_end(h, T_PRIMAL));
        CPCHK(timer_begin(h, T_FIN));
        EigArgs Af = with_warm(eig_args(h, P, n, n, 0, 1, nxt, tol), h, n, 1);
        const double *rows_f = staged_partials(h, h->nblk, &Af.nblk);
        launch_fin(h, Af, rows_f);
        CPCHK(timer_end(h, T_FIN));
    }
    HIPCHK(hipGetLastError());
    return CP_OK;
}

// one launch of k_mf_fused_tv: x_{n-1} in xp, x_n written to xc; y_n in pc, y_{n+1} written to pn
template <typename S>
static void launch_fused_t(cp_engine *h, const MfParams &P, const double *xp, double *xc, const double *pc, double *pn,
                           const double *W) {
    const bool lin = (h->d.bc == CP_BC_LINEAR);
    auto kern = lin ? k_mf_fused_tv<S, true, false> : k_mf_fused_tv<S, false, false>;
    if (P.norm21) kern = lin ? k_mf_fused_tv<S, true, false, CP_KMAX, true> : k_mf_fused_tv<S, false, false, CP_KMAX, true>;   // single-level driver
    else if (h->K == CP_KMAX) kern = lin ? k_mf_fused_tv<S, true, true> : k_mf_fused_tv<S, false, true>;
    else if (h->K <= 4 && h->fused_ka) kern = lin ? k_mf_fused_tv<S, true, false, 4> : k_mf_fused_tv<S, false, false, 4>;
    else if (h->K <= 8 && h->fused_ka) kern = lin ? k_mf_fused_tv<S, true, false, 8> : k_mf_fused_tv<S, false, false, 8>;
    else if (h->K <= 12 && h->fused_ka) kern = lin ? k_mf_fused_tv<S, true, false, 12> : k_mf_fused_tv<S, false, false, 12>;
    if (h->fused_cw == 2) kern = (h->K == CP_KMAX) ? k_mf_fused_tv<S, false, true, CP_KMAX, false, 2> : k_mf_fused_tv<S, false, false, CP_KMAX, false, 2>;
    if (h->fused_cw == 4) kern = (h->K == CP_KMAX) ? k_mf_fused_tv<S, false, true, CP_KMAX, false, 4> : k_mf_fused_tv<S, false, false, CP_KMAX, false, 4>;
    if (h->fused_waves == 8) {
        if (h->K == CP_KMAX) kern = k_mf_fused_tv<S, false, true, CP_KMAX, false, 1, 8>;
        else if (h->K <= 8 && h->fused_ka) kern = k_mf_fused_tv<S, false, false, 8, false, 1, 8>;
        else if (h->K <= 12 && h->fused_ka) kern = k_mf_fused_tv<S, false, false, 12, false, 1, 8>;
        else kern = k_mf_fused_tv<S, false, false, CP_KMAX, false, 1, 8>;
    }
    if (h->fused_ks) {    // coarsest levels (cp_create): one column per workgroup, images dealt to its four waves
        if (h->K == CP_KMAX) kern = k_mf_fused_tv<S, false, true, CP_KMAX, false, 1, CP_WAVES, true>;
        else if (h->K <= 4 && h->fused_ka) kern = k_mf_fused_tv<S, false, false, 4, false, 1, CP_WAVES, true>;
        else if (h->K <= 8 && h->fused_ka) kern = k_mf_fused_tv<S, false, false, 8, false, 1, CP_WAVES, true>;
        else if (h->K <= 12 && h->fused_ka) kern = k_mf_fused_tv<S, false, false, 12, false, 1, CP_WAVES, true>;
        else kern = k_mf_fused_tv<S, false, false, CP_KMAX, false, 1, CP_WAVES, true>;
    }
    // boundary arrays travel with the dual buffers: read the one of y_n (pc), write the one of y_{n+1} (pn)
    const S *bin = (const S *)h->bnd[pc == h->P[1] ? 1 : 0];
    S *bout = (S *)h->bnd[pn == h->P[1] ? 1 : 0];
    hipLaunchKernelGGL(kern, dim3(h->nblk_tv), dim3(64 * h->fused_waves), 0, h->stream, P, (const S *)xp, (S *)xc, (const S *)pc,
                       (S *)pn, (const S *)h->g, (const S *)h->b, W, (const CpCtrl *)h->ctrl, h->part, bin, bout);
}
// boundary array of dual buffer `pb` (0 / 1) for the storage columns [jc0, jc1) from the buffer's y2 fields: after a kernel
// other than k_mf_fused_tv wrote them (k_mf_dual at start-up, a halo message)
static void fill_bnd(cp_engine *h, int pb, int jc0, int jc1) {
    if (!h->bnd[pb] || jc1 <= jc0) return;
    const long long tot = (long long)((h->m + 63) / 64) * (jc1 - jc0) * CP_BND_BLOCK;
    const int grid = (int)std::max<long long>(1, std::min<long long>((tot + CP_BLOCK - 1) / CP_BLOCK, 2048));
    if (h->fp32)
        hipLaunchKernelGGL((k_mf_fill_bnd<float>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, h->m, h->nl, h->K, h->fs, jc0, jc1,
                           (const float *)h->P[pb], (float *)h->bnd[pb]);
    else
        hipLaunchKernelGGL((k_mf_fill_bnd<double>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, h->m, h->nl, h->K, h->fs, jc0, jc1,
                           (const double *)h->P[pb], h->bnd[pb]);
}
static void launch_fused(cp_engine *h, const MfParams &P, const double *xp, double *xc, const double *pc, double *pn,
                         const double *W) {
    if (h->fp32) launch_fused_t<float>(h, P, xp, xc, pc, pn, W);
    else launch_fused_t<double>(h, P, xp, xc, pc, pn, W);
}

// NN-curvature: k_mf_fused_curv (x_{n-1} in xp, x_n -> xc; y_n in pc; y1_{n+1}, y3~_{n+1} -> pn) and k_curv_y2 (y2_{n+1} -> pn)
template <typename S>
static void launch_curv_t(cp_engine *h, const MfParams &P, const double *xp, double *xc, const double *pc, double *pn,
                          const double *W) {
    const bool neu = (h->d.bc == CP_BC_NEUMANN);
    auto kern = neu ? k_mf_fused_curv<S, true, false> : k_mf_fused_curv<S, false, false>;
    if (h->curv_waves == 8) {
        kern = neu ? k_mf_fused_curv<S, true, false, CP_KMAX, 8> : k_mf_fused_curv<S, false, false, CP_KMAX, 8>;
        if (h->K == CP_KMAX) kern = neu ? k_mf_fused_curv<S, true, true, CP_KMAX, 8> : k_mf_fused_curv<S, false, true, CP_KMAX, 8>;
        else if (h->K <= 8 && h->fused_ka) kern = neu ? k_mf_fused_curv<S, true, false, 8, 8> : k_mf_fused_curv<S, false, false, 8, 8>;
        else if (h->K <= 12 && h->fused_ka) kern = neu ? k_mf_fused_curv<S, true, false, 12, 8> : k_mf_fused_curv<S, false, false, 12, 8>;
    }
    else if (h->K == CP_KMAX) kern = neu ? k_mf_fused_curv<S, true, true> : k_mf_fused_curv<S, false, true>;
    else if (h->K <= 4 && h->fused_ka) kern = neu ? k_mf_fused_curv<S, true, false, 4> : k_mf_fused_curv<S, false, false, 4>;
    else if (h->K <= 8 && h->fused_ka) kern = neu ? k_mf_fused_curv<S, true, false, 8> : k_mf_fused_curv<S, false, false, 8>;
    else if (h->K <= 12 && h->fused_ka) kern = neu ? k_mf_fused_curv<S, true, false, 12> : k_mf_fused_curv<S, false, false, 12>;
    hipLaunchKernelGGL(kern, dim3(h->nblk_curv), dim3(64 * h->curv_waves), 0, h->stream, P, (const S *)xp, (S *)xc, (const S *)pc,
                       (S *)pn, (const S *)h->g, (const S *)h->b, W, (const CpCtrl *)h->ctrl, h->part);
}
template <typename S>
static void launch_curv_y2_t(cp_engine *h, const MfParams &P, const double *xc, const double *xp, const double *pc, double *pn) {
    auto kern = (h->d.bc == CP_BC_NEUMANN) ? k_curv_y2<S, true> : k_curv_y2<S, false>;
    if (h->y2_r2) kern = (h->d.bc == CP_BC_NEUMANN) ? k_curv_y2_r2<S, true> : k_curv_y2_r2<S, false>;
    hipLaunchKernelGGL(kern, dim3(h->nblk_y2), dim3(CP_BLOCK), 0, h->stream, P, (const S *)xc, (const S *)xp, (const S *)pc, (S *)pn,
                       (const CpCtrl *)h->ctrl, h->part);
}

// ---- fused NN-TV path: F_0 = dual step 1 alone, then one k_mf_fused_tv + k_mf_efused per iteration -------------
static int mf_fused_start(cp_engine *h, const MfParams &P, double tol) {
    CpCtrl *ctrl = (CpCtrl *)h->ctrl;
    launch_mf_dual<0>(h, P, h->X[0], h->X[1], h->P[0], h->P[1], h->W[0], 0, 1);
    if (h->reg == 0) fill_bnd(h, 1, 1, h->nl + 1);   // y_1 was written by k_mf_dual: its boundary array for the first fused launch
    hipLaunchKernelGGL(k_mf_eig, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, eig_args(h, P, -1, 1, 0, 0, 0, tol, 0, 1),
                       h->part, ctrl, h->W[1], h->hist_p, h->hist_d);
    HIPCHK(hipGetLastError());
    return CP_OK;
}
static int mf_fused_iterations(cp_engine *h, const MfParams &P, int n0, int count, int check, double tol) {
    CpCtrl *ctrl = (CpCtrl *)h->ctrl;
    for (int n = n0 + 1; n <= n0 + count; ++n) {
        const int cur = n & 1, prv = (n - 1) & 1, nxt = (n + 1) & 1;   // x_{n-1} in X[prv], y_n in P[cur]
        CPCHK(timer_begin(h, T_DUAL));
        if (h->reg == 0) launch_fused(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[nxt], h->W[cur]);
        else if (h->fp32) launch_curv_t<float>(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[nxt], h->W[cur]);
        else launch_curv_t<double>(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[nxt], h->W[cur]);
        CPCHK(timer_end(h, T_DUAL));
        EFusedArgs A;
        A.K = h->K; A.n = n; A.check_stop = check; A.tol = tol; A.mu1 = P.mu1; A.mu2 = P.mu2;
        A.sigma = P.sigma; A.tau = P.tau; A.Ntot = P.Ntot; A.norm21 = P.norm21; A.reg = h->reg;
        A.warm = efused_warm(n);
        if (h->time_all) CPCHK(timer_begin(h, T_EIG));
        const double *rows = fused_partials(h, &A.nblk);
        hipLaunchKernelGGL(k_mf_efused, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, rows, ctrl, h->W[nxt], h->hist_p,
                           h->hist_d, h->vwarm);
        if (h->time_all) CPCHK(timer_end(h, T_EIG));
        if (h->reg != 0) {   // NN-curvature: y2_{n+1}, then F2 / F2* / the boundary term of the next mean / the stop rule
            CPCHK(timer_begin(h, T_PRIMAL));
            if (h->fp32) launch_curv_y2_t<float>(h, P, h->X[cur], h->X[prv], h->P[cur], h->P[nxt]);
            else launch_curv_y2_t<double>(h, P, h->X[cur], h->X[prv], h->P[cur], h->P[nxt]);
            CPCHK(timer_end(h, T_PRIMAL));
            CurvFinArgs F;
            F.n = n; F.check_stop = check; F.tol = tol; F.mu2 = P.mu2; F.tau = P.tau; F.Ntot = P.Ntot;
            if (h->time_all) CPCHK(timer_begin(h, T_FIN));
            const double *rows2 = staged_partials(h, h->nblk_y2, &F.nblk, true);
            hipLaunchKernelGGL(k_curv_fin, dim3(1), dim3(256), 0, h->stream, F, rows2, ctrl, h->hist_p, h->hist_d);
            if (h->time_all) CPCHK(timer_end(h, T_FIN));
        }
    }
    HIPCHK(hipGetLastError());
    return CP_OK;
}

static int mf_begin(cp_engine *h) {
    CpCtrl *ctrl = (CpCtrl *)h->ctrl;
    hipLaunchKernelGGL(k_mf_reset_ctrl, dim3(1), dim3(256), 0, h->stream, ctrl, h->W[0], h->W[1], h->km);
    HIPCHK(hipMemcpyAsync(h->X[1], h->X[0], sizeof(double) * h->xN, hipMemcpyDeviceToDevice, h->stream));   // x_old = x0
    return CP_OK;
}

static int mf_end(cp_engine *h, const MfParams &P, int iterations) {
    const int fb = iterations & 1;
    if (h->km == CP_KBIG) {
        if (h->reg == 0)
            hipLaunchKernelGGL((k_mf_materialize<double, 0, CP_KBIG>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->X[fb], h->X[0],
                               h->P[fb], h->P[0], h->W[fb], (const CpCtrl *)h->ctrl, fb);
        else
            hipLaunchKernelGGL((k_mf_materialize<double, 1, CP_KBIG>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->X[fb], h->X[0],
                               h->P[fb], h->P[0], h->W[fb], (const CpCtrl *)h->ctrl, fb);
    } else if (h->fp32) {
        const float *xs = (const float *)h->X[fb], *ps = (const float *)h->P[fb];
        float *xd = (float *)h->X[0], *pd = (float *)h->P[0];
        if (h->reg == 0)
            hipLaunchKernelGGL((k_mf_materialize<float, 0>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xs, xd, ps, pd,
                               h->W[fb], (const CpCtrl *)h->ctrl, fb);
        else
            hipLaunchKernelGGL((k_mf_materialize<float, 1>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, xs, xd, ps, pd,
                               h->W[fb], (const CpCtrl *)h->ctrl, fb);
    } else if (h->reg == 0)
        hipLaunchKernelGGL((k_mf_materialize<double, 0>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->X[fb], h->X[0],
                           h->P[fb], h->P[0], h->W[fb], (const CpCtrl *)h->ctrl, fb);
    else
        hipLaunchKernelGGL((k_mf_materialize<double, 1>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->X[fb], h->X[0],
                           h->P[fb], h->P[0], h->W[fb], (const CpCtrl *)h->ctrl, fb);
    hipLaunchKernelGGL(k_mf_reset_ctrl, dim3(1), dim3(256), 0, h->stream, (CpCtrl *)h->ctrl, h->W[0], h->W[1], h->km);
    HIPCHK(hipGetLastError());
    return CP_OK;
}

// ---- variance models (cp_var.cuh) -----------------------------------------------------------------------------
template <typename S, int MODE>
static void launch_var_dual_t(cp_engine *h, const MfParams &P, const double *xc, const double *xo, const double *pin,
                              double *pout, int cc, int co) {
    if (h->reg == 0)
        hipLaunchKernelGGL((k_var_dual<S, MODE, 0>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, (const S *)xc, (const S *)xo,
                           (const S *)pin, (S *)pout, (const S *)h->g, (const S *)h->b, (const CpCtrl *)h->ctrl, cc, co, h->part);
    else
        hipLaunchKernelGGL((k_var_dual<S, MODE, 1>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, (const S *)xc, (const S *)xo,
                           (const S *)pin, (S *)pout, (const S *)h->g, (const S *)h->b, (const CpCtrl *)h->ctrl, cc, co, h->part);
}
template <int MODE>
static void launch_var_dual(cp_engine *h, const MfParams &P, const double *xc, const double *xo, const double *pin,
                            double *pout, int cc, int co) {
    if (h->fp32) launch_var_dual_t<float, MODE>(h, P, xc, xo, pin, pout, cc, co);
    else launch_var_dual_t<double, MODE>(h, P, xc, xo, pin, pout, cc, co);
}
template <typename S, int MODE>
static void launch_var_primal_t(cp_engine *h, const MfParams &P, const double *xc, double *xn, const double *p, int cc) {
    if (h->reg == 0)
        hipLaunchKernelGGL((k_var_primal<S, MODE, 0>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, (const S *)xc, (S *)xn,
                           (const S *)p, (const S *)h->g, (const CpCtrl *)h->ctrl, cc, h->part);
    else
        hipLaunchKernelGGL((k_var_primal<S, MODE, 1>), dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, (const S *)xc, (S *)xn,
                           (const S *)p, (const S *)h->g, (const CpCtrl *)h->ctrl, cc, h->part);
}
template <int MODE>
static void launch_var_primal(cp_engine *h, const MfParams &P, const double *xc, double *xn, const double *p, int cc) {
    if (h->fp32) launch_var_primal_t<float, MODE>(h, P, xc, xn, p, cc);
    else launch_var_primal_t<double, MODE>(h, P, xc, xn, p, cc);
}
static VarFinArgs var_fin_args(const cp_engine *h, const MfParams &P, int rp, int rd, int check, int update, int newbuf,
                               double tol) {
    VarFinArgs A;
    A.nblk = h->nblk; A.reg = h->reg; A.rp = rp; A.rd = rd; A.check_stop = check; A.update = update; A.newbuf = newbuf;
    A.tol = tol; A.mu1 = P.mu1; A.mu2 = P.mu2; A.Ntot = P.Ntot;
    return A;
}
// iterations n = n0+1 .. n0+count of PDO/chambolle_pock.m:156-179 (state parity follows n)
static int var_iterations(cp_engine *h, const MfParams &P, int n0, int count, int check, double tol) {
    CpCtrl *ctrl = (CpCtrl *)h->ctrl;
    for (int n = n0 + 1; n <= n0 + count; ++n) {
        const int cur = (n - 1) & 1, nxt = n & 1;
        CPCHK(timer_begin(h, T_DUAL));
        launch_var_dual<0>(h, P, h->X[cur], h->X[nxt], h->P[cur], h->P[nxt], cur, nxt);
        CPCHK(timer_end(h, T_DUAL));
        CPCHK(timer_begin(h, T_EIG));
        hipLaunchKernelGGL(k_var_fin_dual, dim3(1), dim3(256), 0, h->stream, var_fin_args(h, P, n - 1, n, check, 0, 0, tol),
                           h->part, ctrl, h->hist_p, h->hist_d);
        CPCHK(timer_end(h, T_EIG));
        CPCHK(timer_begin(h, T_PRIMAL));
        launch_var_primal<0>(h, P, h->X[cur], h->X[nxt], h->P[nxt], cur);
        CPCHK(timer_end(h, T_PRIMAL));
        CPCHK(timer_begin(h, T_FIN));
        hipLaunchKernelGGL(k_var_fin_primal, dim3(1), dim3(256), 0, h->stream, var_fin_args(h, P, -1, n, 0, 1, nxt, tol),
                           h->part, ctrl, h->hist_p, h->hist_d);
        CPCHK(timer_end(h, T_FIN));
    }
    HIPCHK(hipGetLastError());
    return CP_OK;
}
// ---- fused variance-TV path: dual step 1 alone, then one k_var_lazy_tv + k_var_efused per iteration -------------------
static VarEFusedArgs var_efused_args(const MfParams &P, int n, int start, int check, double tol) {
    VarEFusedArgs A;
    A.nblk = 0; A.n = n; A.start = start; A.check_stop = check; A.tol = tol; A.mu1 = P.mu1; A.mu2 = P.mu2; A.tau = P.tau;
    A.Ntot = P.Ntot; A.reg = 0;
    return A;
}
static int var_fused_start(cp_engine *h, const MfParams &P) {
    launch_var_dual<0>(h, P, h->X[0], h->X[1], h->P[0], h->P[1], 0, 1);                 // y_1 from xbar_0 = x_0
    VarEFusedArgs A = var_efused_args(P, 0, 1, 0, 0.0);
    A.nblk = h->nblk; A.reg = h->reg;
    hipLaunchKernelGGL(k_var_efused, dim3(1), dim3(256), 0, h->stream, A, h->part, (CpCtrl *)h->ctrl, h->hist_p, h->hist_d);
    // y_1 = alpha_1 - beta_1 with beta_1 = 0, abar_1 = mean_k y1
    if (h->fp32) hipLaunchKernelGGL(k_var_lazy_init<float>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, (const float *)h->P[1], h->vlazy[1]);
    else hipLaunchKernelGGL(k_var_lazy_init<double>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, (const double *)h->P[1], h->vlazy[1]);
    h->lazy_live = true;
    HIPCHK(hipGetLastError());
    return CP_OK;
}
template <typename S>
static void launch_var_fused_t(cp_engine *h, const MfParams &P, const double *xp, double *xc, const double *pc, double *pn) {
    const int cur = (pc == h->P[1]) ? 1 : 0;
    const S *xps = (const S *)xp, *pcs = (const S *)pc, *gs = (const S *)h->g, *bs = (const S *)h->b;
    const double *lzi = h->vlazy[cur];
    double *lzo = h->vlazy[cur ^ 1];
    if (h->reg != 0) {
        const bool neu = (h->d.bc == CP_BC_NEUMANN);
        auto kern = neu ? k_var_lazy_curv<S, true> : k_var_lazy_curv<S, false>;
        hipLaunchKernelGGL(kern, dim3(h->nblk_fused), dim3(CP_BLOCK), 0, h->stream, P, xps, (S *)xc, pcs, (S *)pn, gs, bs, lzi, lzo,
                           (const CpCtrl *)h->ctrl, h->part);
        return;
    }
    const bool lin = (h->d.bc == CP_BC_LINEAR);
    auto kern = lin ? k_var_lazy_tv<S, true> : k_var_lazy_tv<S, false>;
    if (h->var_waves == 8) kern = k_var_lazy_tv<S, false, 8>;
    hipLaunchKernelGGL(kern, dim3(h->nblk_var), dim3(64 * h->var_waves), 0, h->stream, P, xps, (S *)xc, pcs, (S *)pn, gs, bs, lzi, lzo,
                       (const CpCtrl *)h->ctrl, h->part);
}
static void launch_var_fused(cp_engine *h, const MfParams &P, const double *xp, double *xc, const double *pc, double *pn) {
    if (h->fp32) launch_var_fused_t<float>(h, P, xp, xc, pc, pn);
    else launch_var_fused_t<double>(h, P, xp, xc, pc, pn);
}
static int var_fused_iterations(cp_engine *h, const MfParams &P, int n0, int count, int check, double tol) {
    for (int n = n0 + 1; n <= n0 + count; ++n) {
        const int cur = n & 1, prv = (n - 1) & 1, nxt = (n + 1) & 1;   // x_{n-1} in X[prv], y_n in P[cur]
        CPCHK(timer_begin(h, T_DUAL));
        launch_var_fused(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[nxt]);
        CPCHK(timer_end(h, T_DUAL));
        VarEFusedArgs A = var_efused_args(P, n, 0, check, tol);
        A.reg = h->reg;
        if (h->time_all) CPCHK(timer_begin(h, T_EIG));
        const double *rows = staged_partials(h, h->reg == 0 ? h->nblk_var : h->nblk_fused, &A.nblk, true);
        hipLaunchKernelGGL(k_var_efused, dim3(1), dim3(256), 0, h->stream, A, rows, (CpCtrl *)h->ctrl, h->hist_p, h->hist_d);
        if (h->time_all) CPCHK(timer_end(h, T_EIG));
        if (h->reg != 0) {   // variance-curvature: y2_{n+1} (the kernel of the NN-curvature model), then F2 / F2* / boundary term / stop rule
            CPCHK(timer_begin(h, T_PRIMAL));
            if (h->fp32) launch_curv_y2_t<float>(h, P, h->X[cur], h->X[prv], h->P[cur], h->P[nxt]);
            else launch_curv_y2_t<double>(h, P, h->X[cur], h->X[prv], h->P[cur], h->P[nxt]);
            CPCHK(timer_end(h, T_PRIMAL));
            CurvFinArgs F;
            F.n = n; F.check_stop = check; F.tol = tol; F.mu2 = P.mu2; F.tau = P.tau; F.Ntot = P.Ntot;
            if (h->time_all) CPCHK(timer_begin(h, T_FIN));
            const double *rows2 = staged_partials(h, h->nblk_y2, &F.nblk, true);
            hipLaunchKernelGGL(k_var_curv_fin, dim3(1), dim3(256), 0, h->stream, F, rows2, (CpCtrl *)h->ctrl, h->hist_p, h->hist_d);
            if (h->time_all) CPCHK(timer_end(h, T_FIN));
        }
    }
    HIPCHK(hipGetLastError());
    return CP_OK;
}
static int var_end(cp_engine *h, const MfParams &P, int iterations) {
    const int fb = iterations & 1;
    if (h->lazy_live) {   // y1 = alpha - beta in place, before the copy into the canonical buffers
        if (h->fp32) hipLaunchKernelGGL(k_var_lazy_finish<float>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, (float *)h->P[fb], (const double *)h->vlazy[fb]);
        else hipLaunchKernelGGL(k_var_lazy_finish<double>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->P[fb], (const double *)h->vlazy[fb]);
        h->lazy_live = false;
    }
    if (h->fp32)
        hipLaunchKernelGGL(k_var_materialize<float>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->npf, (const float *)h->X[fb],
                           (float *)h->X[0], (const float *)h->P[fb], (float *)h->P[0], (const CpCtrl *)h->ctrl, fb);
    else
        hipLaunchKernelGGL(k_var_materialize<double>, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->npf, (const double *)h->X[fb],
                           h->X[0], (const double *)h->P[fb], h->P[0], (const CpCtrl *)h->ctrl, fb);
    hipLaunchKernelGGL(k_mf_reset_ctrl, dim3(1), dim3(256), 0, h->stream, (CpCtrl *)h->ctrl, h->W[0], h->W[1], h->km);
    HIPCHK(hipGetLastError());
    return CP_OK;
}
static int solve_var(cp_engine *h, double theta, double tau, double sigma, int maxIter, double tol, cp_result *res) {
    if (!h->linearized) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_solve before cp_linearize");
    CPCHK(ensure_hist(h, maxIter + 2));
    const MfParams P = mf_params(h, theta, tau, sigma);
    CpCtrl *ctrl = (CpCtrl *)h->ctrl;
    HIPCHK(hipMemsetAsync(h->hist_p, 0, sizeof(double) * (size_t)h->hist_rows * h->hcols, h->stream));
    HIPCHK(hipMemsetAsync(h->hist_d, 0, sizeof(double) * (size_t)h->hist_rows * h->hcols, h->stream));
    CPCHK(mf_begin(h));
    // row 0: primal_objective(x0), dual_objective(y0)          chambolle_pock.m:89-106
    launch_var_dual<1>(h, P, h->X[0], h->X[1], h->P[0], nullptr, 0, 1);
    hipLaunchKernelGGL(k_var_fin_dual, dim3(1), dim3(256), 0, h->stream, var_fin_args(h, P, 0, 0, 0, 0, 0, tol), h->part, ctrl,
                       h->hist_p, h->hist_d);
    launch_var_primal<1>(h, P, h->X[0], nullptr, h->P[0], 0);
    hipLaunchKernelGGL(k_var_fin_primal, dim3(1), dim3(256), 0, h->stream, var_fin_args(h, P, -1, 0, 0, 0, 0, tol), h->part,
                       ctrl, h->hist_p, h->hist_d);
    HIPCHK(hipGetLastError());
    int done = 0, stopped = 0, stop_iter = -1;
    const bool fused = h->var_fused && maxIter > 0;
    if (fused) CPCHK(var_fused_start(h, P));
    while (done < maxIter && !stopped) {
        const int cnt = std::min(32, maxIter - done);
        if (fused) CPCHK(var_fused_iterations(h, P, done, cnt, 1, tol));
        else CPCHK(var_iterations(h, P, done, cnt, 1, tol));
        done += cnt;
        int hdr[2];
        HIPCHK(hipMemcpyAsync(hdr, ctrl, sizeof(hdr), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        stopped = hdr[0];
        stop_iter = hdr[1];
    }
    if (!stopped && !fused) {   // primal row maxIter trails the last primal pass by one kernel
        const int cur = maxIter & 1;
        launch_var_dual<1>(h, P, h->X[cur], h->X[cur ^ 1], h->P[cur], nullptr, cur, cur ^ 1);
        hipLaunchKernelGGL(k_var_fin_dual, dim3(1), dim3(256), 0, h->stream, var_fin_args(h, P, maxIter, -1, 0, 0, 0, tol),
                           h->part, ctrl, h->hist_p, h->hist_d);
        HIPCHK(hipGetLastError());
    }
    const int iterations = stopped ? stop_iter : maxIter;
    if (stopped) {
        int zero = 0;
        HIPCHK(hipMemcpyAsync(ctrl, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    CPCHK(var_end(h, P, iterations));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->last[0] = cp_result{iterations, stopped ? CP_STOP_TOLERANCE : CP_STOP_MAXITER, iterations + 1, h->hcols};
    if (res) *res = h->last[0];
    return CP_OK;
}

static int solve_mf(cp_engine *h, double theta, double tau, double sigma, int maxIter, double tol, cp_result *res) {
    if (!h->linearized) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_solve before cp_linearize");
    if (!h->has_nu) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_solve before cp_set_nu");
    CPCHK(ensure_hist(h, maxIter + 2));
    const MfParams P = mf_params(h, theta, tau, sigma);
    CpCtrl *ctrl = (CpCtrl *)h->ctrl;
    HIPCHK(hipMemsetAsync(h->hist_p, 0, sizeof(double) * (size_t)h->hist_rows * h->hcols, h->stream));
    HIPCHK(hipMemsetAsync(h->hist_d, 0, sizeof(double) * (size_t)h->hist_rows * h->hcols, h->stream));
    CPCHK(mf_begin(h));
    // row 0: primal_objective(x0), dual_objective(y0)          chambolle_pock.m:89-106
    launch_mf_dual<1>(h, P, h->X[0], h->X[1], h->P[0], nullptr, h->W[0], 0, 1);
    launch_eig(h, eig_args(h, P, 0, 0, 0, 0, 0, tol, 1), h->part, h->W[2]);
    launch_mf_primal<1>(h, P, h->X[0], nullptr, h->P[0], h->W[0], 0);
    launch_fin(h, eig_args(h, P, 0, 0, 0, 0, 0, tol), h->part);
    HIPCHK(hipGetLastError());
    int done = 0, stopped = 0, stop_iter = -1;
    const int chunk = 32;
    const bool fused = ((h->reg == 0) || h->curv_fused) && h->km == CP_KMAX;   // K > 16: the split passes
    if (fused && maxIter > 0) CPCHK(mf_fused_start(h, P, tol));
    while (done < maxIter && !stopped) {
        const int cnt = std::min(chunk, maxIter - done);
        if (fused) CPCHK(mf_fused_iterations(h, P, done, cnt, 1, tol));
        else CPCHK(mf_iterations(h, P, done, cnt, 1, tol));
        done += cnt;
        int hdr[2];
        HIPCHK(hipMemcpyAsync(hdr, ctrl, sizeof(hdr), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        stopped = hdr[0];
        stop_iter = hdr[1];
    }
    if (!stopped && !(fused && maxIter > 0)) {
        // primal row maxIter (trails by one kernel) -- and the stop rule may still fire for i = maxIter? No:
        // the reference breaks on i == maxIter before evaluating the rule (chambolle_pock.m:137).
        const int cur = maxIter & 1;
        launch_mf_dual<1>(h, P, h->X[cur], h->X[cur ^ 1], h->P[cur], nullptr, h->W[cur], cur, cur ^ 1);
        launch_eig(h, eig_args(h, P, maxIter, -1, 0, 0, 0, tol), h->part, h->W[2]);
        HIPCHK(hipGetLastError());
    }
    const int iterations = stopped ? stop_iter : maxIter;
    // un-stop so that the canonicalisation kernels run
    if (stopped) {
        int zero = 0;
        HIPCHK(hipMemcpyAsync(ctrl, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    CPCHK(mf_end(h, P, iterations));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->last[0] = cp_result{iterations, stopped ? CP_STOP_TOLERANCE : CP_STOP_MAXITER, iterations + 1, h->hcols};
    if (res) *res = h->last[0];
    return CP_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename S, int MODE>
static void launch_pw_dual_t(cp_engine *h, const PwParams &P, const double *xc, const double *xo, const double *vin,
                             double *vout) {
    dim3 grid(h->nblk, h->batch);
    if (h->nc == 2)
        hipLaunchKernelGGL((k_pw_dual<S, MODE, 2>), grid, dim3(CP_BLOCK), 0, h->stream, P, (const S *)xc, (const S *)xo,
                           (const S *)vin, (S *)vout, (const PwCtrl *)h->ctrl, h->part);
    else
        hipLaunchKernelGGL((k_pw_dual<S, MODE, 1>), grid, dim3(CP_BLOCK), 0, h->stream, P, (const S *)xc, (const S *)xo,
                           (const S *)vin, (S *)vout, (const PwCtrl *)h->ctrl, h->part);
}
template <int MODE>
static void launch_pw_dual(cp_engine *h, const PwParams &P, const double *xc, const double *xo, const double *vin,
                           double *vout) {
    if (h->fp32) launch_pw_dual_t<float, MODE>(h, P, xc, xo, vin, vout);
    else launch_pw_dual_t<double, MODE>(h, P, xc, xo, vin, vout);
}
template <typename S, int MODE>
static void launch_pw_primal_t(cp_engine *h, const PwParams &P, const double *xc_, double *xn_, const double *v_) {
    dim3 grid(h->nblk, h->batch);
    const PwCtrl *ctrl = (const PwCtrl *)h->ctrl;
    const S *xc = (const S *)xc_, *v = (const S *)v_, *g = (const S *)h->g, *b = (const S *)h->b;
    S *xn = (S *)xn_;
    if (h->gt == GT_SAD_REG)
        hipLaunchKernelGGL((k_pw_primal<S, MODE, GT_SAD_REG>), grid, dim3(CP_BLOCK), 0, h->stream, P, xc, xn, v, g, b, ctrl,
                           h->part2);
    else if (h->gt == GT_SSD_REG)
        hipLaunchKernelGGL((k_pw_primal<S, MODE, GT_SSD_REG>), grid, dim3(CP_BLOCK), 0, h->stream, P, xc, xn, v, g, b, ctrl,
                           h->part2);
    else
        hipLaunchKernelGGL((k_pw_primal<S, MODE, GT_SAD_DEN>), grid, dim3(CP_BLOCK), 0, h->stream, P, xc, xn, v, b, b, ctrl,
                           h->part2);
}
template <int MODE>
static void launch_pw_primal(cp_engine *h, const PwParams &P, const double *xc, double *xn, const double *v) {
    if (h->fp32) launch_pw_primal_t<float, MODE>(h, P, xc, xn, v);
    else launch_pw_primal_t<double, MODE>(h, P, xc, xn, v);
}
static PwFinArgs pw_fin_args(const cp_engine *h, int rpF, int rpG, int rd, int check, double tol) {
    PwFinArgs A;
    A.nblk = h->nblk; A.gt = h->gt; A.rpF = rpF; A.rpG = rpG; A.rd = rd; A.check_stop = check;
    A.hist_rows = h->hist_rows; A.tol = tol; A.lambda = h->d.lambda;
    return A;
}
static int pw_iterations(cp_engine *h, const PwParams &P, int n0, int count, int check, double tol) {
    PwCtrl *ctrl = (PwCtrl *)h->ctrl;
    for (int n = n0 + 1; n <= n0 + count; ++n) {
        const int cur = (n - 1) & 1, nxt = n & 1;
        CPCHK(timer_begin(h, T_DUAL));
        launch_pw_dual<0>(h, P, h->X[cur], h->X[nxt], h->P[cur], h->P[nxt]);
        CPCHK(timer_end(h, T_DUAL));
        CPCHK(timer_begin(h, T_PRIMAL));
        launch_pw_primal<0>(h, P, h->X[cur], h->X[nxt], h->P[nxt]);
        CPCHK(timer_end(h, T_PRIMAL));
        CPCHK(timer_begin(h, T_FIN));
        hipLaunchKernelGGL(k_pw_fin, dim3(h->batch), dim3(PW_FIN_THREADS), 0, h->stream, pw_fin_args(h, n - 1, n, n, check, tol),
                           h->part, h->part2, ctrl, h->hist_p, h->hist_d);
        CPCHK(timer_end(h, T_FIN));
    }
    HIPCHK(hipGetLastError());
    return CP_OK;
}

// ---- fused pairwise path: dual step 1 alone, then one k_pw_fused + k_pw_fin_fused per iteration ---------------------
// state parity: x_n in X[n & 1], v_n in P[n & 1] (as in the split passes); launch n also leaves v_{n+1} in the other P
template <typename S>
static void launch_pw_fused_t(cp_engine *h, const PwParams &P, const double *xo_, double *xn_, const double *vin_,
                              double *vout_) {
    const PwCtrl *ctrl = (const PwCtrl *)h->ctrl;
    const S *xo = (const S *)xo_, *vin = (const S *)vin_, *g = (const S *)h->g, *b = (const S *)h->b;
    S *xn = (S *)xn_, *vout = (S *)vout_;
    if (h->pw_strip > 0) {
        dim3 mg(h->pw_grid, h->batch);
        const int st = h->pw_strip, ng = h->pw_groups, nw = h->pw_waves;
        if (h->pw_two) {
            if (h->gt == GT_SAD_REG)
                hipLaunchKernelGGL((k_pw_march2<S, GT_SAD_REG>), mg, dim3(CP_BLOCK), 0, h->stream, P, st, ng, nw, xo, xn, vin, vout, g, b, ctrl, h->part);
            else if (h->gt == GT_SSD_REG)
                hipLaunchKernelGGL((k_pw_march2<S, GT_SSD_REG>), mg, dim3(CP_BLOCK), 0, h->stream, P, st, ng, nw, xo, xn, vin, vout, g, b, ctrl, h->part);
            else
                hipLaunchKernelGGL((k_pw_march2<S, GT_SAD_DEN>), mg, dim3(CP_BLOCK), 0, h->stream, P, st, ng, nw, xo, xn, vin, vout, b, b, ctrl, h->part);
            return;
        }
        if (h->gt == GT_SAD_REG)
            hipLaunchKernelGGL((k_pw_march<S, GT_SAD_REG>), mg, dim3(CP_BLOCK), 0, h->stream, P, st, ng, nw, xo, xn, vin, vout, g, b, ctrl, h->part);
        else if (h->gt == GT_SSD_REG)
            hipLaunchKernelGGL((k_pw_march<S, GT_SSD_REG>), mg, dim3(CP_BLOCK), 0, h->stream, P, st, ng, nw, xo, xn, vin, vout, g, b, ctrl, h->part);
        else
            hipLaunchKernelGGL((k_pw_march<S, GT_SAD_DEN>), mg, dim3(CP_BLOCK), 0, h->stream, P, st, ng, nw, xo, xn, vin, vout, b, b, ctrl, h->part);
        return;
    }
    dim3 grid(h->nblk, h->batch);
    if (h->gt == GT_SAD_REG)
        hipLaunchKernelGGL((k_pw_fused<S, GT_SAD_REG>), grid, dim3(CP_BLOCK), 0, h->stream, P, xo, xn, vin, vout, g, b, ctrl, h->part);
    else if (h->gt == GT_SSD_REG)
        hipLaunchKernelGGL((k_pw_fused<S, GT_SSD_REG>), grid, dim3(CP_BLOCK), 0, h->stream, P, xo, xn, vin, vout, g, b, ctrl, h->part);
    else
        hipLaunchKernelGGL((k_pw_fused<S, GT_SAD_DEN>), grid, dim3(CP_BLOCK), 0, h->stream, P, xo, xn, vin, vout, b, b, ctrl, h->part);
}
static int pw_fused_start(cp_engine *h, const PwParams &P) {
    // v_1 = prox_{sigma F*}(v_0 + sigma K xbar_0), xbar_0 = x_0 (X[1] holds the copy of x_0); its partials give the F*
    // violation of dual row 1 (the other entries k_pw_fin writes into that row are overwritten by the fused launch 1)
    launch_pw_dual<0>(h, P, h->X[0], h->X[1], h->P[0], h->P[1]);
    hipLaunchKernelGGL(k_pw_fin, dim3(h->batch), dim3(PW_FIN_THREADS), 0, h->stream, pw_fin_args(h, -1, -1, 1, 0, 0.0), h->part,
                       h->part2, (PwCtrl *)h->ctrl, h->hist_p, h->hist_d);
    HIPCHK(hipGetLastError());
    return CP_OK;
}
static int pw_fused_iterations(cp_engine *h, const PwParams &P, int n0, int count, int check, double tol) {
    PwCtrl *ctrl = (PwCtrl *)h->ctrl;
    for (int n = n0 + 1; n <= n0 + count; ++n) {
        const int cur = n & 1, prv = (n - 1) & 1;   // x_{n-1} in X[prv], v_n in P[cur]
        CPCHK(timer_begin(h, T_DUAL));
        if (h->fp32) launch_pw_fused_t<float>(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[prv]);
        else launch_pw_fused_t<double>(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[prv]);
        CPCHK(timer_end(h, T_DUAL));
        PwFusedFinArgs A;
        A.nblk = h->pw_strip > 0 ? h->pw_grid : h->nblk;
        A.gt = h->gt; A.n = n; A.check_stop = check; A.hist_rows = h->hist_rows; A.tol = tol;
        A.lambda = h->d.lambda;
        if (h->time_all) CPCHK(timer_begin(h, T_FIN));
        hipLaunchKernelGGL(k_pw_fin_fused, dim3(h->batch), dim3(PW_FINF_THREADS), 0, h->stream, A, h->part, ctrl, h->hist_p,
                           h->hist_d);
        if (h->time_all) CPCHK(timer_end(h, T_FIN));
    }
    HIPCHK(hipGetLastError());
    return CP_OK;
}

static int solve_pw(cp_engine *h, double theta, double tau, double sigma, int maxIter, double tol, cp_result *res) {
    if (!h->linearized) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_solve before cp_linearize / cp_set_images");
    CPCHK(ensure_hist(h, maxIter + 2));
    const PwParams P = pw_params(h, theta, tau, sigma);
    PwCtrl *ctrl = (PwCtrl *)h->ctrl;
    const size_t hbytes = sizeof(double) * (size_t)h->hist_rows * h->hcols * h->batch;
    HIPCHK(hipMemsetAsync(h->hist_p, 0, hbytes, h->stream));
    HIPCHK(hipMemsetAsync(h->hist_d, 0, hbytes, h->stream));
    HIPCHK(hipMemsetAsync(ctrl, 0, sizeof(PwCtrl) * h->batch, h->stream));
    HIPCHK(hipMemcpyAsync(h->X[1], h->X[0], sizeof(double) * h->xN, hipMemcpyDeviceToDevice, h->stream));
    // row 0
    launch_pw_dual<1>(h, P, h->X[0], h->X[1], h->P[0], nullptr);
    launch_pw_primal<1>(h, P, h->X[0], nullptr, h->P[0]);
    hipLaunchKernelGGL(k_pw_fin, dim3(h->batch), dim3(PW_FIN_THREADS), 0, h->stream, pw_fin_args(h, 0, 0, 0, 0, tol), h->part,
                       h->part2, ctrl, h->hist_p, h->hist_d);
    HIPCHK(hipGetLastError());
    std::vector<PwCtrl> hc(h->batch);
    int done = 0;
    bool all_stopped = false;
    const int chunk = 32;
    if (maxIter > 0 && !h->pw_split) CPCHK(pw_fused_start(h, P));
    while (done < maxIter && !all_stopped) {
        const int cnt = std::min(chunk, maxIter - done);
        if (h->pw_split) CPCHK(pw_iterations(h, P, done, cnt, 1, tol));
        else CPCHK(pw_fused_iterations(h, P, done, cnt, 1, tol));
        done += cnt;
        HIPCHK(hipMemcpyAsync(hc.data(), ctrl, sizeof(PwCtrl) * h->batch, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        all_stopped = true;
        for (auto &c : hc) all_stopped = all_stopped && c.stop;
    }
    // F part of primal row maxIter for the problems that ran to the end (stopped ones skip the kernels); the fused
    // launch maxIter has produced it already
    if (h->pw_split || maxIter == 0) {
        const int cur = maxIter & 1;
        launch_pw_dual<1>(h, P, h->X[cur], h->X[cur ^ 1], h->P[cur], nullptr);
        hipLaunchKernelGGL(k_pw_fin, dim3(h->batch), dim3(PW_FIN_THREADS), 0, h->stream, pw_fin_args(h, maxIter, -1, -1, 0, tol),
                           h->part, h->part2, ctrl, h->hist_p, h->hist_d);
        HIPCHK(hipGetLastError());
    }
    // canonical state: result of problem pr lives in buffer (iterations & 1) -> copy into buffer 0
    for (int pr = 0; pr < h->batch; ++pr) {
        const int it = hc[pr].stop ? hc[pr].stop_iter : maxIter;
        h->last[pr] = cp_result{it, hc[pr].stop ? CP_STOP_TOLERANCE : CP_STOP_MAXITER, it + 1, h->hcols};
        if (it & 1) {
            const size_t esz = h->fp32 ? sizeof(float) : sizeof(double);
            const size_t xo = (size_t)pr * h->nxf * h->fs * esz, po = (size_t)pr * h->npf * h->fs * esz;
            HIPCHK(hipMemcpyAsync((char *)h->X[0] + xo, (char *)h->X[1] + xo, esz * h->nxf * h->fs, hipMemcpyDeviceToDevice,
                                  h->stream));
            HIPCHK(hipMemcpyAsync((char *)h->P[0] + po, (char *)h->P[1] + po, esz * h->npf * h->fs, hipMemcpyDeviceToDevice,
                                  h->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (res) *res = h->last[0];
    return CP_OK;
}

extern "C" int cp_solve(cp_handle h, double theta, double tau, double sigma, int32_t max_iter, double tol,
                        cp_result *res) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    if (max_iter < 0) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg max_iter < 0");
    if (!(tau > 0.0) || !(sigma > 0.0)) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg step sizes must be positive");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    if (h->var) return solve_var(h, theta, tau, sigma, max_iter, tol, res);
    return h->mf ? solve_mf(h, theta, tau, sigma, max_iter, tol, res) : solve_pw(h, theta, tau, sigma, max_iter, tol, res);
}

extern "C" int cp_get_result(cp_handle h, int32_t problem, cp_result *res) {
    if (!h || !res || problem < 0 || problem >= h->batch) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_get_result");
    *res = h->last[problem];
    return CP_OK;
}

extern "C" int cp_get_history(cp_handle h, double *primal_history, double *dual_history) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    HIPCHK(hipSetDevice(h->d.device));
    if (h->hist_rows == 0) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState no solve yet");
    // device: [problem][row][col]; host: per problem rows x cols column-major, problems stacked
    std::vector<double> tmp((size_t)h->hist_rows * h->hcols * h->batch);
    for (int which = 0; which < 2; ++which) {
        double *dst = which ? dual_history : primal_history;
        if (!dst) continue;
        HIPCHK(hipMemcpy(tmp.data(), which ? h->hist_d : h->hist_p, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
        size_t off = 0;
        for (int pr = 0; pr < h->batch; ++pr) {
            const int rows = h->last[pr].rows;
            const double *src = tmp.data() + (size_t)pr * h->hist_rows * h->hcols;
            for (int c = 0; c < h->hcols; ++c)
                for (int r = 0; r < rows; ++r) dst[off + (size_t)c * rows + r] = src[(size_t)r * h->hcols + c];
            off += (size_t)rows * h->hcols;
        }
    }
    return CP_OK;
}

// ------------------------------------------------------------------------------------------------
// joff: global index of the fine slab's first column minus twice that of the coarse slab's (0 on a full domain)
template <typename S>
__global__ void k_prolong(int m, int n, int fs, int ml, int nlw, int fsl, int nfields, double factor, int jf_off, int jc_off,
                          const S *__restrict__ xl, S *__restrict__ xh) {
    const long long tot = (long long)m * n * nfields;
    for (long long gid = (long long)blockIdx.x * CP_BLOCK + threadIdx.x; gid < tot; gid += (long long)gridDim.x * CP_BLOCK) {
        const int f = (int)(gid / ((long long)m * n)), q = (int)(gid - (long long)f * m * n);
        const int j = q / m, i = q - j * m;
        const int jc = ((jf_off + j) >> 1) - jc_off;           // local column of the coarse handle
        // kron(x_low, ones(2)) cropped to (1:m, 1:n)          NNE/mf_nn_tv...:113-119
        xh[(size_t)f * fs + m + q] = (S)(factor * (double)xl[(size_t)f * fsl + ml + (size_t)jc * ml + (i >> 1)]);
    }
}

extern "C" int cp_prolong(cp_handle coarse, cp_handle fine, double factor) {
    if (!coarse || !fine) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    if (coarse->nxf != fine->nxf || coarse->batch != fine->batch || coarse->d.device != fine->d.device)
        CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg handles do not match");
    if (fine->m > 2 * coarse->m || fine->d.n_global > 2 * coarse->d.n_global)
        CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg fine level larger than twice the coarse level");
    // row slabs: every owned column of the fine handle needs its parent column among the coarse handle's owned columns
    // (a full coarse handle serves any fine slab; slab -> slab needs fine boundaries = twice the coarse ones)
    {
        const int c_lo = fine->d.j_offset >> 1, c_hi = (fine->d.j_offset + fine->nl - 1) >> 1;
        if (c_lo < coarse->d.j_offset || c_hi >= coarse->d.j_offset + coarse->nl)
            CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg coarse columns [%d, %d] needed, the coarse handle owns [%d, %d)", c_lo, c_hi,
                    coarse->d.j_offset, coarse->d.j_offset + coarse->nl);
    }
    HIPCHK(hipSetDevice(fine->d.device));
    CP_NOT_ITERATING(coarse);
    CP_NOT_ITERATING(fine);
    if (coarse->fp32 != fine->fp32) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg handles differ in storage mode");
    HIPCHK(hipStreamSynchronize(coarse->stream));
    const int nf = fine->nxf * fine->batch;
    if (fine->fp32)
        hipLaunchKernelGGL((k_prolong<float>), dim3(2048), dim3(CP_BLOCK), 0, fine->stream, fine->m, fine->nl, fine->fs, coarse->m,
                           coarse->nl, coarse->fs, nf, factor, fine->d.j_offset, coarse->d.j_offset, (const float *)coarse->X[0],
                           (float *)fine->X[0]);
    else
        hipLaunchKernelGGL((k_prolong<double>), dim3(2048), dim3(CP_BLOCK), 0, fine->stream, fine->m, fine->nl, fine->fs, coarse->m,
                           coarse->nl, coarse->fs, nf, factor, fine->d.j_offset, coarse->d.j_offset, coarse->X[0], fine->X[0]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(fine->P[0], 0, sizeof(double) * fine->pN, fine->stream));   // p = zeros (:122)
    HIPCHK(hipStreamSynchronize(fine->stream));
    return CP_OK;
}

__global__ void k_downsample(int m, int n, int mc, int ncw, int nimg, const double *__restrict__ fimg,
                             double *__restrict__ cimg) {
    const long long tot = (long long)mc * ncw * nimg;
    for (long long gid = (long long)blockIdx.x * CP_BLOCK + threadIdx.x; gid < tot; gid += (long long)gridDim.x * CP_BLOCK) {
        const int f = (int)(gid / ((long long)mc * ncw)), q = (int)(gid - (long long)f * mc * ncw);
        const int j = q / mc, i = q - j * mc;
        const double *src = fimg + (size_t)f * m * n;
        const int i0 = 2 * i, j0 = 2 * j;
        // conv2(., 0.25*ones(2), 'same')(1:2:end, 1:2:end): zero padding past the end    NNE/mf_nn_tv...:75-76
        const double a = src[(size_t)j0 * m + i0];
        const double b = (i0 + 1 < m) ? src[(size_t)j0 * m + i0 + 1] : 0.0;
        const double c = (j0 + 1 < n) ? src[(size_t)(j0 + 1) * m + i0] : 0.0;
        const double d = (i0 + 1 < m && j0 + 1 < n) ? src[(size_t)(j0 + 1) * m + i0 + 1] : 0.0;
        cimg[(size_t)f * mc * ncw + q] = 0.25 * a + 0.25 * b + 0.25 * c + 0.25 * d;
    }
}

extern "C" int cp_downsample(cp_handle fine, cp_handle coarse) {
    if (!coarse || !fine) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    if (!fine->has_images) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState fine level has no images");
    const int mc = (fine->m + 1) / 2, ncw = (fine->d.n_global + 1) / 2;
    if (coarse->m != mc || coarse->d.n_global != ncw || coarse->K != fine->K || coarse->batch != fine->batch)
        CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg coarse level must be ceil(size/2)");
    HIPCHK(hipSetDevice(fine->d.device));
    const int nimg = fine->K * fine->batch;
    hipLaunchKernelGGL(k_downsample, dim3(2048), dim3(CP_BLOCK), 0, fine->stream, fine->m, fine->d.n_global, mc, ncw, nimg,
                       fine->img, coarse->img);
    if (fine->imgR && coarse->imgR)
        hipLaunchKernelGGL(k_downsample, dim3(2048), dim3(CP_BLOCK), 0, fine->stream, fine->m, fine->d.n_global, mc, ncw,
                           nimg, fine->imgR, coarse->imgR);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(fine->stream));
    coarse->has_images = true;
    coarse->linearized = false;
    return CP_OK;
}

extern "C" int cp_get_images(cp_handle h, double *a) {
    if (!h || !a) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle or buffer");
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipMemcpy(a, h->img, sizeof(double) * (size_t)h->K * h->m * h->d.n_global * h->batch, hipMemcpyDeviceToHost));
    return CP_OK;
}

// ------------------------------------------------------------------------------------------------
// benchmark / streaming hooks: begin (start-up: x_old = x0, first dual step of the fused path), any number of
// cp_iterate_steps (steady-state iterations, nothing else), end (lazy state -> canonical X[0], P[0])
static int iter_timer_slot(cp_engine *h) {
    PassTimer &t = h->timers[T_ITER];
    if (t.used + 2 > t.ev.size())
        for (int a = 0; a < 2; ++a) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); t.ev.push_back(e); }
    return CP_OK;
}

extern "C" int cp_iterate_begin(cp_handle h, double theta, double tau, double sigma, int32_t max_iters) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    if (max_iters < 0) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg max_iters < 0");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    if (!h->linearized) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_iterate before cp_linearize");
    if (h->mf && !h->has_nu) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_iterate before cp_set_nu");
    CPCHK(ensure_hist(h, max_iters + 2));
    h->mg_theta = theta; h->mg_tau = tau; h->mg_sigma = sigma;
    if (h->mf) {
        const MfParams P = mf_params(h, theta, tau, sigma);
        CPCHK(mf_begin(h));
        if ((h->reg == 0 || h->curv_fused) && !h->var && h->km == CP_KMAX) CPCHK(mf_fused_start(h, P, 0.0));
        if (h->var && h->var_fused) CPCHK(var_fused_start(h, P));
    } else {
        HIPCHK(hipMemsetAsync(h->ctrl, 0, sizeof(PwCtrl) * h->batch, h->stream));
        HIPCHK(hipMemcpyAsync(h->X[1], h->X[0], sizeof(double) * h->xN, hipMemcpyDeviceToDevice, h->stream));
        if (!h->pw_split) CPCHK(pw_fused_start(h, pw_params(h, theta, tau, sigma)));
    }
    h->it_n = 0;
    return CP_OK;
}

extern "C" int cp_iterate_steps(cp_handle h, int32_t iters) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    HIPCHK(hipSetDevice(h->d.device));
    if (h->it_n < 0 || h->it_mg) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_iterate_steps without cp_iterate_begin");
    if (iters < 0 || h->it_n + iters + 2 > h->hist_rows)
        CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg %d more iterations exceed max_iters of cp_iterate_begin", iters);
    h->profile = true;
    CPCHK(iter_timer_slot(h));
    PassTimer &t = h->timers[T_ITER];
    HIPCHK(hipEventRecord(t.ev[t.used], h->stream));
    if (h->mf) {
        const MfParams P = mf_params(h, h->mg_theta, h->mg_tau, h->mg_sigma);
        if (h->var && h->var_fused) CPCHK(var_fused_iterations(h, P, h->it_n, iters, 0, 0.0));
        else if (h->var) CPCHK(var_iterations(h, P, h->it_n, iters, 0, 0.0));
        else if ((h->reg == 0 || h->curv_fused) && h->km == CP_KMAX) CPCHK(mf_fused_iterations(h, P, h->it_n, iters, 0, 0.0));
        else CPCHK(mf_iterations(h, P, h->it_n, iters, 0, 0.0));
    } else {
        const PwParams P = pw_params(h, h->mg_theta, h->mg_tau, h->mg_sigma);
        if (h->pw_split) CPCHK(pw_iterations(h, P, h->it_n, iters, 0, 0.0));
        else CPCHK(pw_fused_iterations(h, P, h->it_n, iters, 0, 0.0));
    }
    HIPCHK(hipEventRecord(t.ev[t.used + 1], h->stream));
    t.used += 2;
    h->profile = false;
    h->it_n += iters;
    return CP_OK;
}

extern "C" int cp_iterate_end(cp_handle h) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    HIPCHK(hipSetDevice(h->d.device));
    if (h->it_n < 0 || h->it_mg) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_iterate_end without cp_iterate_begin");
    const int iters = h->it_n;
    h->it_n = -1;
    if (h->mf) {
        const MfParams P = mf_params(h, h->mg_theta, h->mg_tau, h->mg_sigma);
        if (h->var) CPCHK(var_end(h, P, iters));
        else CPCHK(mf_end(h, P, iters));
    } else if (iters & 1) {
        HIPCHK(hipMemcpyAsync(h->X[0], h->X[1], sizeof(double) * h->xN, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->P[0], h->P[1], sizeof(double) * h->pN, hipMemcpyDeviceToDevice, h->stream));
    }
    for (int pI = 0; pI < h->batch; ++pI) h->last[pI] = cp_result{iters, CP_STOP_MAXITER, iters + 1, h->hcols};
    return CP_OK;
}

extern "C" int cp_iterate(cp_handle h, double theta, double tau, double sigma, int32_t iters) {
    CPCHK(cp_iterate_begin(h, theta, tau, sigma, iters));
    int rc = cp_iterate_steps(h, iters);
    if (rc != CP_OK) { h->it_n = -1; return rc; }
    return cp_iterate_end(h);
}

extern "C" int cp_sync(cp_handle h) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return CP_OK;
}

extern "C" int cp_kernel_time_ms(cp_handle h, const char *pass, double *ms, int64_t *launches) {
    if (!h || !pass || !ms) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_kernel_time_ms");
    HIPCHK(hipSetDevice(h->d.device));
    int which = -1;
    for (int a = 0; a < T_COUNT; ++a)
        if (strcmp(pass, kPassNames[a]) == 0) which = a;
    if (which < 0) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg unknown pass '%s'", pass);
    HIPCHK(hipStreamSynchronize(h->stream));
    PassTimer &t = h->timers[which];
    double tot = 0.0;
    for (size_t a = 0; a + 1 < t.used; a += 2) {
        float f = 0.f;
        HIPCHK(hipEventElapsedTime(&f, t.ev[a], t.ev[a + 1]));
        tot += f;
    }
    *ms = tot;
    if (launches) *launches = (int64_t)(t.used / 2);
    t.used = 0;
    return CP_OK;
}

extern "C" int cp_handover_fallbacks(cp_handle h, int64_t *count) {
    if (!h || !count) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_handover_fallbacks");
    *count = 0;
    if (!h->mf) return CP_OK;   // the pairwise kernels have no hand-over
    HIPCHK(hipSetDevice(h->d.device));
    int v = 0;
    HIPCHK(hipMemcpyAsync(&v, &((CpCtrl *)h->ctrl)->hs_fallbacks, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *count = v;
    return CP_OK;
}

// the kernel name as rocprofv3 prints it, from the same fields the launchers branch on (launch_fused_t, launch_curv_t,
// launch_var_fused_t, launch_pw_fused_t): keep the two in step
extern "C" int cp_dominant_kernel(cp_handle h, char *buf, int32_t len) {
    if (!h || !buf || len < 1) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_dominant_kernel");
    const char *S = h->fp32 ? "float" : "double";
    const char *tf[2] = {"false", "true"};
    char name[256];
    const int K = h->K;
    if (h->mf && !h->var && h->km == CP_KBIG) {
        snprintf(name, sizeof name, "k_mf_dual<double, 0, %d, 32>", h->reg);
    } else if (h->mf && !h->var && h->reg == 0) {
        const bool lin = h->d.bc == CP_BC_LINEAR, n21 = (h->d.flags & CP_FLAG_MFNN_NORM21) != 0;
        int ka = CP_KMAX, cw = 1, nwv = CP_WAVES;
        bool fullk = false, ks = false;
        if (n21) { }
        else if (K == CP_KMAX) fullk = true;
        else if (K <= 4 && h->fused_ka) ka = 4;
        else if (K <= 8 && h->fused_ka) ka = 8;
        else if (K <= 12 && h->fused_ka) ka = 12;
        if (h->fused_cw == 2 || h->fused_cw == 4) { cw = h->fused_cw; ka = CP_KMAX; fullk = (K == CP_KMAX); }
        if (h->fused_waves == 8) {
            nwv = 8; cw = 1; fullk = (K == CP_KMAX);
            ka = (K == CP_KMAX) ? CP_KMAX : ((K <= 8 && h->fused_ka) ? 8 : ((K <= 12 && h->fused_ka) ? 12 : CP_KMAX));
        }
        if (h->fused_ks) {
            ks = true; nwv = CP_WAVES; cw = 1; fullk = (K == CP_KMAX);
            ka = (K == CP_KMAX) ? CP_KMAX : ((K <= 4 && h->fused_ka) ? 4 : ((K <= 8 && h->fused_ka) ? 8 : ((K <= 12 && h->fused_ka) ? 12 : CP_KMAX)));
        }
        const bool lin_eff = lin && nwv == CP_WAVES && cw == 1 && !ks;
        snprintf(name, sizeof name, "k_mf_fused_tv<%s, %s, %s, %d, %s, %d, %d, %s>", S, tf[lin_eff], tf[fullk], ka, tf[n21 && nwv == CP_WAVES && cw == 1 && !ks], cw, nwv, tf[ks]);
    } else if (h->mf && !h->var) {
        if (!h->curv_fused) snprintf(name, sizeof name, "k_mf_dual<%s, 0, 1, 16>", S);
        else {
            const bool neu = h->d.bc == CP_BC_NEUMANN;
            int ka = CP_KMAX;
            if (K != CP_KMAX && h->fused_ka) {
                if (h->curv_waves == 8) ka = K <= 8 ? 8 : (K <= 12 ? 12 : CP_KMAX);
                else ka = K <= 4 ? 4 : (K <= 8 ? 8 : (K <= 12 ? 12 : CP_KMAX));
            }
            if (h->curv_waves == 8) snprintf(name, sizeof name, "k_mf_fused_curv<%s, %s, %s, %d, 8>", S, tf[neu], tf[K == CP_KMAX], ka);
            else snprintf(name, sizeof name, "k_mf_fused_curv<%s, %s, %s, %d, 4>", S, tf[neu], tf[K == CP_KMAX], ka);
        }
    } else if (h->mf) {
        if (!h->var_fused) snprintf(name, sizeof name, "k_var_dual<%s, 0, %d>", S, h->reg);
        else if (h->reg != 0) snprintf(name, sizeof name, "k_var_lazy_curv<%s, %s>", S, tf[h->d.bc == CP_BC_NEUMANN]);
        else snprintf(name, sizeof name, "k_var_lazy_tv<%s, %s, %d>", S, tf[h->d.bc == CP_BC_LINEAR && h->var_waves != 8], h->var_waves);
    } else {
        if (h->pw_split) snprintf(name, sizeof name, "k_pw_dual<%s, 0, %d>", S, h->nc);
        else if (h->pw_strip > 0) snprintf(name, sizeof name, "%s<%s, %d>", h->pw_two ? "k_pw_march2" : "k_pw_march", S, h->gt);
        else snprintf(name, sizeof name, "k_pw_fused<%s, %d>", S, h->gt);
    }
    snprintf(buf, (size_t)len, "%s", name);
    return CP_OK;
}

// ================================================================================================
// row-slab multi-GPU hooks
__global__ __launch_bounds__(CP_RED_THREADS) void k_mf_reduce_local(int nblk, const double *__restrict__ part,
                                                                    double *__restrict__ out) {
    __shared__ double G[16][17], G2[16][17], sc[CP_NSCAL], stage[CP_RED_THREADS], stage2[CP_RED_THREADS];
    const int t = threadIdx.x;
    reduce_partials2(part, nblk, G, G2, sc, stage, stage2);
    if (t < CP_NSCAL) out[t] = sc[t];
    if (t < 256) {
        out[CP_NSCAL + t] = G[t >> 4][t & 15];
        out[CP_NSCAL + 256 + t] = G2[t >> 4][t & 15];
    }
}

// S = element type of the buffer: the storage type of the state (double, or float in the fp32 mode), double for the
// vectors of the power iteration
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_halo_copy(int m, int fs, int nsel, int sel, int K, int col, int pack,
                                                        S *__restrict__ buf, S *__restrict__ lin) {
    const int tot = nsel * m;
    for (int gid = blockIdx.x * CP_BLOCK + threadIdx.x; gid < tot; gid += gridDim.x * CP_BLOCK) {
        const int s = gid / m, i = gid - s * m;
        int f = s;                                                  // CP_SEL_ALL, CP_SEL_X_U, CP_SEL_P_Y12: leading fields
        if (sel == CP_SEL_P_DY) f = K + 4 * (s >> 1) + 1 + 2 * (s & 1);   // Dy u_x, Dy u_y of image s/2
        if (sel == CP_SEL_P_Y2) f = K + s;                                 // all y2 fields
        S *p = buf + (size_t)f * fs + (size_t)col * m + i;
        if (pack) lin[gid] = *p; else *p = lin[gid];
    }
}

// The per-iteration halo traffic of the fused NN-TV path in one launch per direction of data flow:
//   pack:   to_left  = [u of xbuf | y1, y2 of pbuf] of the first owned column (7K fields), to_right = Dy parts of y2 of the
//           last owned column (2K fields);
//   unpack: from_right (a neighbour's to_left) -> halo column nl+1, from_left (a neighbour's to_right) -> halo column 0.
// A null buffer skips that side (ends of the chain).
template <typename S>
__global__ __launch_bounds__(CP_BLOCK) void k_halo_state(int m, int fs, int K, int nl, int pack, S *__restrict__ xbuf,
                                                         S *__restrict__ pbuf, S *__restrict__ a7, S *__restrict__ a2) {
    const int n7 = a7 ? 7 * K * m : 0, n2 = a2 ? 2 * K * m : 0;
    const int col7 = pack ? 1 : nl + 1, col2 = pack ? nl : 0;
    for (int gid = blockIdx.x * CP_BLOCK + threadIdx.x; gid < n7 + n2; gid += gridDim.x * CP_BLOCK) {
        S *p, *q;
        if (gid < n7) {
            const int s = gid / m, i = gid - s * m;
            p = (s < 2 * K ? xbuf + (size_t)s * fs : pbuf + (size_t)(s - 2 * K) * fs) + (size_t)col7 * m + i;
            q = a7 + gid;
        } else {
            const int g2 = gid - n7, s = g2 / m, i = g2 - s * m;
            const int f = K + 4 * (s >> 1) + 1 + 2 * (s & 1);          // Dy u_x, Dy u_y of image s/2
            p = pbuf + (size_t)f * fs + (size_t)col2 * m + i;
            q = a2 + g2;
        }
        if (pack) *q = *p; else *p = *q;
    }
}

// selections CP_SEL_VEC_*: the same fields as CP_SEL_X_U / CP_SEL_P_DY / CP_SEL_P_Y2, but the buffer holds doubles in
// every storage mode (vectors of the power iteration in the spare buffers)
static int mg_base_sel(int sel) {
    switch (sel) {
        case CP_SEL_VEC_X_U: return CP_SEL_X_U;
        case CP_SEL_VEC_P_DY: return CP_SEL_P_DY;
        case CP_SEL_VEC_P_Y2: return CP_SEL_P_Y2;
    }
    return sel;
}
static int mg_nsel(const cp_engine *h, int buf, int sel) {
    const bool isx = (buf == CP_BUF_X0 || buf == CP_BUF_X1 || buf == CP_BUF_XTMP);
    switch (mg_base_sel(sel)) {
        case CP_SEL_ALL: return isx ? h->nxf : h->npf;
        case CP_SEL_X_U: return isx ? 2 * h->K : -1;
        case CP_SEL_P_Y12: return (isx || h->reg != 0) ? -1 : 5 * h->K;
        case CP_SEL_P_DY: return (isx || h->reg != 0) ? -1 : 2 * h->K;
        case CP_SEL_P_Y2: return isx ? -1 : (h->reg == 0 ? 4 : 2) * h->K;
    }
    return -1;
}
static double *mg_buf(cp_engine *h, int buf) {
    switch (buf) {
        case CP_BUF_X0: return h->X[0];
        case CP_BUF_X1: return h->X[1];
        case CP_BUF_P0: return h->P[0];
        case CP_BUF_P1: return h->P[1];
        case CP_BUF_XTMP: return h->xtmp;
    }
    return nullptr;
}

extern "C" int cp_set_stream(cp_handle h, void *hip_stream) {
    if (!h) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null handle");
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return CP_OK;
}
extern "C" int64_t cp_mg_partial_len(void) { return CP_PSTRIDE; }
// doubles the caller's message buffer needs for one packed halo column (fp32 mode: two floats per double)
extern "C" int64_t cp_mg_halo_len(cp_handle h, int sel) {
    if (!h || !h->mf) return -1;
    const int base = mg_base_sel(sel);
    const int n = (base == CP_SEL_ALL) ? std::max(h->nxf, h->npf) : mg_nsel(h, base == CP_SEL_X_U ? CP_BUF_X0 : CP_BUF_P0, sel);
    if (n < 0) return -1;
    const bool as_float = h->fp32 && base == sel && sel != CP_SEL_ALL;
    return as_float ? ((int64_t)n * h->m + 1) / 2 : (int64_t)n * h->m;
}
static int mg_halo(cp_handle h, int buf, int sel, int col, int pack, double *lin) {
    if (!h || !h->mf || !lin) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg halo exchange needs an mf_nn handle and a buffer");
    HIPCHK(hipSetDevice(h->d.device));
    double *b = mg_buf(h, buf);
    const int nsel = mg_nsel(h, buf, sel), base = mg_base_sel(sel);
    if (!b || nsel <= 0) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg halo buffer %d / selection %d", buf, sel);
    if (h->fp32 && sel == CP_SEL_ALL) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported CP_SEL_ALL in fp32 mode");
    const int grid = std::max(1, std::min((nsel * h->m + CP_BLOCK - 1) / CP_BLOCK, 1024));
    if (h->fp32 && base == sel)
        hipLaunchKernelGGL((k_halo_copy<float>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, h->m, h->fs, nsel, base, h->K, col, pack,
                           (float *)b, (float *)lin);
    else
        hipLaunchKernelGGL((k_halo_copy<double>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, h->m, h->fs, nsel, base, h->K, col, pack, b, lin);
    // a halo column of y2 arrived in a state buffer of the NN-TV model: its block of the boundary array (the fused kernel
    // evaluates the primal step at (i, nl) from it)
    if (!pack && base == sel && (buf == CP_BUF_P0 || buf == CP_BUF_P1) && (sel == CP_SEL_P_Y12 || sel == CP_SEL_ALL || sel == CP_SEL_P_Y2))
        fill_bnd(h, buf == CP_BUF_P1 ? 1 : 0, col, col + 1);
    HIPCHK(hipGetLastError());
    return CP_OK;
}
extern "C" int cp_mg_halo_pack(cp_handle h, int buf, int sel, int column, double *dev_out) {
    return mg_halo(h, buf, sel, column ? (h ? h->nl : 0) : 1, 1, dev_out);          // owned columns are 1..nl
}
extern "C" int cp_mg_halo_unpack(cp_handle h, int buf, int sel, int side, const double *dev_in) {
    return mg_halo(h, buf, sel, side ? (h ? h->nl + 1 : 0) : 0, 0, (double *)dev_in);   // halo columns 0 and nl+1
}

static int mg_state(cp_handle h, int xbuf, int pbuf, int pack, double *a7, double *a2) {
    if (!h || !h->mf || h->var || h->reg != 0) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg the state halo needs an NN-TV handle");
    HIPCHK(hipSetDevice(h->d.device));
    double *xb = mg_buf(h, xbuf), *pb = mg_buf(h, pbuf);
    const bool xok = (xbuf == CP_BUF_X0 || xbuf == CP_BUF_X1), pok = (pbuf == CP_BUF_P0 || pbuf == CP_BUF_P1);
    if (!xb || !pb || !xok || !pok) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg state halo buffers %d / %d", xbuf, pbuf);
    if (!a7 && !a2) return CP_OK;
    const int tot = ((a7 ? 7 : 0) + (a2 ? 2 : 0)) * h->K * h->m;
    const int grid = std::max(1, std::min((tot + CP_BLOCK - 1) / CP_BLOCK, 1024));
    if (h->fp32)
        hipLaunchKernelGGL((k_halo_state<float>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, h->m, h->fs, h->K, h->nl, pack, (float *)xb,
                           (float *)pb, (float *)a7, (float *)a2);
    else
        hipLaunchKernelGGL((k_halo_state<double>), dim3(grid), dim3(CP_BLOCK), 0, h->stream, h->m, h->fs, h->K, h->nl, pack, xb, pb, a7, a2);
    if (!pack && a7) fill_bnd(h, pbuf == CP_BUF_P1 ? 1 : 0, h->nl + 1, h->nl + 2);   // y2 of the right halo column arrived
    HIPCHK(hipGetLastError());
    return CP_OK;
}
extern "C" int cp_mg_state_pack(cp_handle h, int xbuf, int pbuf, double *dev_to_left, double *dev_to_right) {
    return mg_state(h, xbuf, pbuf, 1, dev_to_left, dev_to_right);
}
extern "C" int cp_mg_state_unpack(cp_handle h, int xbuf, int pbuf, const double *dev_from_right, const double *dev_from_left) {
    return mg_state(h, xbuf, pbuf, 0, (double *)dev_from_right, (double *)dev_from_left);
}

extern "C" int cp_mg_begin(cp_handle h, double theta, double tau, double sigma, int32_t max_iters) {
    if (!h || !h->mf || h->var) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_mg_begin needs an mf_nn handle");
    if (h->km != CP_KMAX) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported the staged (multi-GPU) iteration with k = %d > %d images", h->K, CP_KMAX);
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    if (!h->linearized || !h->has_nu) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_mg_begin before linearize / set_nu");
    h->mg_theta = theta; h->mg_tau = tau; h->mg_sigma = sigma;
    if (max_iters < 0) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg max_iters < 0");
    CPCHK(ensure_hist(h, max_iters + 2));
    HIPCHK(hipMemsetAsync(h->hist_p, 0, sizeof(double) * (size_t)h->hist_rows * h->hcols, h->stream));
    HIPCHK(hipMemsetAsync(h->hist_d, 0, sizeof(double) * (size_t)h->hist_rows * h->hcols, h->stream));
    h->profile = true;     // per-launch HIP events of the fused kernel (cp_kernel_time_ms)
    h->it_n = 0;
    h->it_mg = true;
    // the spare dual buffer may hold a vector of the power iteration (doubles): no stale bit patterns in its halo columns
    HIPCHK(hipMemsetAsync(h->P[1], 0, sizeof(double) * h->pN, h->stream));
    return mf_begin(h);
}

extern "C" int cp_mg_stage(cp_handle h, int stage, int32_t n, double *dev_buf, int32_t world, double scale,
                           int32_t check_stop, double tol) {
    if (!h || !h->mf || h->var || !dev_buf) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_mg_stage");
    if (h->reg != 0 && stage == CP_STAGE_FUSED) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg CP_STAGE_FUSED needs the NN-TV model");
    if (h->reg == 0 && stage >= CP_STAGE_CURV_P && stage <= CP_STAGE_CURV_FIN) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg curvature stages need the NN-curvature model");
    HIPCHK(hipSetDevice(h->d.device));
    const bool power = (stage == CP_STAGE_APPLY_A || stage == CP_STAGE_APPLY_AT);
    if (!power && !(h->it_n >= 0 && h->it_mg)) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_mg_stage without cp_mg_begin");
    if (power && h->it_n >= 0) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState power-iteration stage while an iteration is open");
    const MfParams P = mf_params(h, h->mg_theta, h->mg_tau, h->mg_sigma);
    CpCtrl *ctrl = (CpCtrl *)h->ctrl;
    switch (stage) {
        case CP_STAGE_DUAL0:
            launch_mf_dual<0>(h, P, h->X[0], h->X[1], h->P[0], h->P[1], h->W[0], 0, 1);
            if (h->reg == 0) fill_bnd(h, 1, 1, h->nl + 1);   // cf. mf_fused_start
            {
                int nrows_l = 0;
                const double *rows_l = staged_partials(h, h->nblk, &nrows_l);
                hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows_l, rows_l, dev_buf);
            }
            break;
        case CP_STAGE_EIG0: {
            EigArgs A = eig_args(h, P, -1, 1, 0, 0, 0, tol, 0, 1);
            A.nblk = world;
            hipLaunchKernelGGL(k_mf_eig, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, dev_buf, ctrl, h->W[1], h->hist_p,
                               h->hist_d);
            break;
        }
        case CP_STAGE_FUSED: {
            if (n + 2 > h->hist_rows) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg iteration %d beyond the history capacity", n);
            const int cur = n & 1, prv = (n - 1) & 1, nxt = (n + 1) & 1;
            CPCHK(timer_begin(h, T_DUAL));
            launch_fused(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[nxt], h->W[cur]);
            CPCHK(timer_end(h, T_DUAL));
            int nrows = 0;
            const double *rows = fused_partials(h, &nrows);
            hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows, rows, dev_buf);
            break;
        }
        case CP_STAGE_EFUSED: {
            EFusedArgs A;
            A.nblk = world; A.K = h->K; A.n = n; A.check_stop = check_stop; A.tol = tol; A.mu1 = P.mu1; A.mu2 = P.mu2;
            A.sigma = P.sigma; A.tau = P.tau; A.Ntot = P.Ntot; A.norm21 = P.norm21; A.reg = h->reg;
            A.warm = efused_warm(n);
            if (h->time_all) CPCHK(timer_begin(h, T_EIG));
            hipLaunchKernelGGL(k_mf_efused, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, dev_buf, ctrl, h->W[(n + 1) & 1],
                               h->hist_p, h->hist_d, h->vwarm);
            if (h->time_all) CPCHK(timer_end(h, T_EIG));
            break;
        }
        case CP_STAGE_SPLIT_DUAL: {
            if (n < 1 || n + 2 > h->hist_rows) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg iteration %d beyond the history capacity", n);
            const int cur = (n - 1) & 1, nxt = n & 1;
            CPCHK(timer_begin(h, T_DUAL));
            launch_mf_dual<0>(h, P, h->X[cur], h->X[nxt], h->P[cur], h->P[nxt], h->W[cur], cur, nxt);
            CPCHK(timer_end(h, T_DUAL));
            {
                int nrows_l = 0;
                const double *rows_l = staged_partials(h, h->nblk, &nrows_l);
                hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows_l, rows_l, dev_buf);
            }
            break;
        }
        case CP_STAGE_SPLIT_EIG: {
            EigArgs A = with_warm(eig_args(h, P, n - 1, n, check_stop, 0, 0, tol), h, n, 0);
            A.nblk = world;
            hipLaunchKernelGGL(k_mf_eig, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, dev_buf, ctrl, h->W[n & 1], h->hist_p,
                               h->hist_d);
            break;
        }
        case CP_STAGE_SPLIT_PRIMAL: {
            const int cur = (n - 1) & 1, nxt = n & 1;
            CPCHK(timer_begin(h, T_PRIMAL));
            launch_mf_primal<0>(h, P, h->X[cur], h->X[nxt], h->P[nxt], h->W[nxt], cur);
            CPCHK(timer_end(h, T_PRIMAL));
            {
                int nrows_l = 0;
                const double *rows_l = staged_partials(h, h->nblk, &nrows_l);
                hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows_l, rows_l, dev_buf);
            }
            break;
        }
        case CP_STAGE_SPLIT_FIN: {
            EigArgs A = with_warm(eig_args(h, P, n, n, 0, 1, n & 1, tol), h, n, 1);
            A.nblk = world;
            hipLaunchKernelGGL(k_mf_fin, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, dev_buf, ctrl, h->hist_p, h->hist_d);
            break;
        }
        case CP_STAGE_ROW0_DUAL: {   // history row 0 (PDO/chambolle_pock.m:89-106): monitors of (x0, y0), nothing is written but the partials
            launch_mf_dual<1>(h, P, h->X[0], h->X[1], h->P[0], nullptr, h->W[0], 0, 1);
            int nrows_l = 0;
            const double *rows_l = staged_partials(h, h->nblk, &nrows_l);
            hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows_l, rows_l, dev_buf);
            break;
        }
        case CP_STAGE_ROW0_EIG: {
            EigArgs A = eig_args(h, P, 0, 0, 0, 0, 0, tol, 1);
            A.nblk = world;
            hipLaunchKernelGGL(k_mf_eig, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, dev_buf, ctrl, h->W[2], h->hist_p, h->hist_d);
            break;
        }
        case CP_STAGE_ROW0_PRIMAL: {
            launch_mf_primal<1>(h, P, h->X[0], nullptr, h->P[0], h->W[0], 0);
            int nrows_l = 0;
            const double *rows_l = staged_partials(h, h->nblk, &nrows_l);
            hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows_l, rows_l, dev_buf);
            break;
        }
        case CP_STAGE_ROW0_FIN: {
            EigArgs A = eig_args(h, P, 0, 0, 0, 0, 0, tol);
            A.nblk = world;
            hipLaunchKernelGGL(k_mf_fin, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, A, dev_buf, ctrl, h->hist_p, h->hist_d);
            break;
        }
        case CP_STAGE_CURV_P: {      // NN-curvature in two field passes: k_mf_fused_curv(n) + local reduction -> dev_buf
            if (n < 1 || n + 2 > h->hist_rows) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg iteration %d beyond the history capacity", n);
            const int cur = n & 1, prv = (n - 1) & 1, nxt = (n + 1) & 1;
            CPCHK(timer_begin(h, T_DUAL));
            if (h->fp32) launch_curv_t<float>(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[nxt], h->W[cur]);
            else launch_curv_t<double>(h, P, h->X[prv], h->X[cur], h->P[cur], h->P[nxt], h->W[cur]);
            CPCHK(timer_end(h, T_DUAL));
            int nrows = 0;
            const double *rows = fused_partials(h, &nrows);
            hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows, rows, dev_buf);
            break;
        }
        case CP_STAGE_CURV_Y2: {     // k_curv_y2(n) (needs the halo columns of u_n) + local reduction of its scalars -> dev_buf
            const int cur = n & 1, prv = (n - 1) & 1, nxt = (n + 1) & 1;
            CPCHK(timer_begin(h, T_PRIMAL));
            if (h->fp32) launch_curv_y2_t<float>(h, P, h->X[cur], h->X[prv], h->P[cur], h->P[nxt]);
            else launch_curv_y2_t<double>(h, P, h->X[cur], h->X[prv], h->P[cur], h->P[nxt]);
            CPCHK(timer_end(h, T_PRIMAL));
            int nrows = 0;
            const double *rows = staged_partials(h, h->nblk_y2, &nrows, true);
            hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows, rows, dev_buf);
            break;
        }
        case CP_STAGE_CURV_FIN: {    // k_curv_fin(n) on the gathered rows: F2, F2*, boundary term of the next mean, stop rule
            CurvFinArgs F;
            F.nblk = world; F.n = n; F.check_stop = check_stop; F.tol = tol; F.mu2 = P.mu2; F.tau = P.tau; F.Ntot = P.Ntot;
            hipLaunchKernelGGL(k_curv_fin, dim3(1), dim3(256), 0, h->stream, F, dev_buf, ctrl, h->hist_p, h->hist_d);
            break;
        }
        case CP_STAGE_APPLY_A:
            launch_mf_applyA(h, P, h->X[1], scale, h->P[1], h->nblk);
            break;
        case CP_STAGE_APPLY_AT:
            launch_mf_applyAT(h, P, h->P[1], h->xtmp, h->nblk);
            {
                int nrows_l = 0;
                const double *rows_l = staged_partials(h, h->nblk, &nrows_l);
                hipLaunchKernelGGL(k_mf_reduce_local, dim3(1), dim3(CP_RED_THREADS), 0, h->stream, nrows_l, rows_l, dev_buf);
            }
            std::swap(h->X[1], h->xtmp);
            break;
        default:
            CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg unknown stage %d", stage);
    }
    HIPCHK(hipGetLastError());
    return CP_OK;
}

extern "C" int cp_mg_end(cp_handle h, int32_t iterations) {
    if (!h || !h->mf) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_mg_end");
    HIPCHK(hipSetDevice(h->d.device));
    if (!(h->it_n >= 0 && h->it_mg)) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_mg_end without cp_mg_begin");
    if (iterations < 0 || iterations + 1 > h->hist_rows)
        CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_mg_end(%d) beyond the %d history rows of cp_mg_begin", iterations, h->hist_rows);
    const MfParams P = mf_params(h, h->mg_theta, h->mg_tau, h->mg_sigma);
    int hdr[2] = {0, -1}, zero = 0;
    HIPCHK(hipMemcpyAsync(hdr, h->ctrl, sizeof(hdr), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    // the stop rule fired at iteration `iterations` (PDO/chambolle_pock.m:145-153): un-stop so that the canonicalisation runs
    const bool stopped = hdr[0] != 0 && hdr[1] == iterations;
    HIPCHK(hipMemcpyAsync(h->ctrl, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
    CPCHK(mf_end(h, P, iterations));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->profile = false;
    h->it_n = -1;
    h->it_mg = false;
    h->last[0] = cp_result{iterations, stopped ? CP_STOP_TOLERANCE : CP_STOP_MAXITER, iterations + 1, h->hcols};
    return CP_OK;
}

extern "C" int cp_mg_stop_state(cp_handle h, int32_t *stopped, int32_t *stop_iter) {
    if (!h || !h->mf || !stopped || !stop_iter) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_mg_stop_state");
    HIPCHK(hipSetDevice(h->d.device));
    int hdr[2];
    HIPCHK(hipMemcpyAsync(hdr, h->ctrl, sizeof(hdr), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *stopped = hdr[0];
    *stop_iter = hdr[1];
    return CP_OK;
}

extern "C" int cp_mg_gram(cp_handle h, int which, const double *V, double *gram) {
    if (!h || !h->mf || h->var || !gram) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_mg_gram");
    if (h->km != CP_KMAX) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported cp_mg_gram with k = %d > %d images", h->K, CP_KMAX);
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    const double *field = (which == 0) ? h->Tu
                          : (h->fp32 ? (const double *)((const float *)h->X[0] + (size_t)2 * h->K * h->fs)
                                     : h->X[0] + (size_t)2 * h->K * h->fs);
    if (which == 0 && !h->linearized) CP_FAIL(CP_ERR_BAD_STATE, "cpengine:badState cp_mg_gram(0) before cp_linearize");
    MfParams P = mf_params(h, 1.0, 1.0, 1.0);
    double *ident = h->svout + 16 + 256, *Vd = ident;
    if (V) {
        HIPCHK(hipMemcpyAsync(h->W[2], V, sizeof(double) * 256, hipMemcpyHostToDevice, h->stream));
        Vd = h->W[2];
    }
    launch_gram_field(h, P, field, Vd);
    // only the first Gram block of the partials is written by k_mf_gram_field: clear the rest first
    HIPCHK(hipGetLastError());
    std::vector<double> hp((size_t)h->nblk * CP_PSTRIDE);
    HIPCHK(hipMemcpyAsync(hp.data(), h->part, sizeof(double) * hp.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int e = 0; e < 256; ++e) {
        double s = 0.0;
        for (int bI = 0; bI < h->nblk; ++bI) s += hp[(size_t)bI * CP_PSTRIDE + CP_NSCAL + e];
        gram[e] = s;
    }
    return CP_OK;
}

extern "C" int cp_mg_power_start(cp_handle h, double *local_sumsq) {
    if (!h || !h->mf || h->var || !local_sumsq) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_mg_power_start");
    HIPCHK(hipSetDevice(h->d.device));
    CP_NOT_ITERATING(h);
    MfParams P = mf_params(h, 1.0, 1.0, 1.0);
    HIPCHK(hipMemsetAsync(h->X[1], 0, sizeof(double) * h->xN, h->stream));
    HIPCHK(hipMemsetAsync(h->xtmp, 0, sizeof(double) * h->xN, h->stream));
    hipLaunchKernelGGL(k_mf_hash_start, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, P, h->nxf, h->X[1]);
    hipLaunchKernelGGL(k_sumsq_fields, dim3(h->nblk), dim3(CP_BLOCK), 0, h->stream, h->npix, h->m, h->fs, h->nxf, h->X[1],
                       h->part);
    HIPCHK(hipGetLastError());
    std::vector<double> hp((size_t)h->nblk * CP_PSTRIDE);
    HIPCHK(hipMemcpyAsync(hp.data(), h->part, sizeof(double) * hp.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (int bI = 0; bI < h->nblk; ++bI) s += hp[(size_t)bI * CP_PSTRIDE];
    *local_sumsq = s;
    return CP_OK;
}

// ------------------------------------------------------------------------------------------------
// IRE/evaluate_displacement.m in full generality: image m x n with spacing h, evaluated on a k x l cell-centred grid over
// omega displaced by u -- the output resolution need not be the image's (s_grid, :27-31), which is how the reference's
// resampling facility is used (e.g. an image upsampled with u = 0). One thread per output sample; coordinate arithmetic
// in explicitly rounded operations as in k_mf_warp / k_pw_warp (floor() is the index logic).
__global__ __launch_bounds__(CP_BLOCK) void k_eval_displacement(int mi, int ni, int k, int l, double om0, double om2, double hx,
                                                                double hy, double inv_h1, double inv_h2,
                                                                const double *__restrict__ img, const double *__restrict__ u,
                                                                double *__restrict__ Tu, double *__restrict__ grad) {
    const long long tot = (long long)k * l;
    for (long long q = (long long)blockIdx.x * CP_BLOCK + threadIdx.x; q < tot; q += (long long)gridDim.x * CP_BLOCK) {
        const int j = (int)(q / k), i = (int)(q - (long long)j * k);
        const double ux = u ? u[q] : 0.0, uy = u ? u[tot + q] : 0.0;                       // u = [u_x(:), u_y(:)]  (:15)
        const double px = __dadd_rn(__dadd_rn(om0, __dmul_rn(hx, (double)i + 0.5)), ux);   // IRE/cell_centered_grid.m:21-29, :43
        const double py = __dadd_rn(__dadd_rn(om2, __dmul_rn(hy, (double)j + 0.5)), uy);
        double val, dx, dy;
        bilinear_tap(img, mi, ni, px, py, inv_h1, inv_h2, val, dx, dy);
        Tu[q] = val;
        if (grad) { grad[q] = dx; grad[tot + q] = dy; }
    }
}

extern "C" int cp_evaluate_displacement(int32_t device, const double *img, int32_t m, int32_t n, const double *h,
                                        const double *u, const double *omega, int32_t k, int32_t l, double *img_u,
                                        double *grad) {
    if (!img || !h || !img_u) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_evaluate_displacement needs img, h and img_u");
    if (m < 1 || n < 1 || k < 1 || l < 1) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg image %d x %d, grid %d x %d", m, n, k, l);
    if (!(h[0] > 0.0) || !(h[1] > 0.0)) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg grid spacing must be positive");
    HIPCHK(hipSetDevice(device));
    double om[4] = {0.0, m * h[0], 0.0, n * h[1]};                      // default region (:33-36)
    if (omega) memcpy(om, omega, sizeof om);
    const size_t ni_ = (size_t)m * n, no = (size_t)k * l;
    double *d_img = nullptr, *d_u = nullptr, *d_T = nullptr, *d_g = nullptr;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc(&d_img, ni_ * sizeof(double)));
        HIPCHK(hipMalloc(&d_T, no * sizeof(double)));
        if (u) HIPCHK(hipMalloc(&d_u, 2 * no * sizeof(double)));
        if (grad) HIPCHK(hipMalloc(&d_g, 2 * no * sizeof(double)));
        HIPCHK(hipMemcpy(d_img, img, ni_ * sizeof(double), hipMemcpyHostToDevice));
        if (u) HIPCHK(hipMemcpy(d_u, u, 2 * no * sizeof(double), hipMemcpyHostToDevice));
        const int grid = (int)std::min<size_t>((no + CP_BLOCK - 1) / CP_BLOCK, 65536);
        hipLaunchKernelGGL(k_eval_displacement, dim3(grid), dim3(CP_BLOCK), 0, nullptr, m, n, k, l, om[0], om[2],
                           (om[1] - om[0]) / k, (om[3] - om[2]) / l, 1.0 / h[0], 1.0 / h[1], d_img, d_u, d_T, d_g);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(img_u, d_T, no * sizeof(double), hipMemcpyDeviceToHost));
        if (grad) HIPCHK(hipMemcpy(grad, d_g, 2 * no * sizeof(double), hipMemcpyDeviceToHost));
        return CP_OK;
    };
    const int rc = body();
    for (double *p : {d_img, d_u, d_T, d_g})
        if (p) hipFree(p);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Evaluation utilities: NNE/landmark_transform.m and NNE/landmark_accuracy.m.
// k_landmark_transform: one workgroup per landmark. The nearest point of the transformed grid g = p + u (:38-45, MATLAB's
// min: the first minimum in column-major order) is found by all 256 threads; thread 0 then runs the fixed-point iteration
// x <- x + (y - F(x)), F(z) = z + [u_x(z), u_y(z)] (bilinear, zero padding, :32-33), at most 100 steps, with the
// reference's literal termination expression sum(F(x) - LM(i,:).^2) < 1e-12 (:50, :54 -- not a squared distance) and its
// fallback to the initial guess (:55-57; failed[l] = 1 is the warning).
__global__ __launch_bounds__(CP_BLOCK) void k_landmark_transform(int m, int n, double om0, double om2, double hx, double hy,
                                                                 const double *__restrict__ u, const double *__restrict__ LM,
                                                                 int num_lm, double *__restrict__ out, int *__restrict__ failed) {
    __shared__ double sd[CP_BLOCK];
    __shared__ long long sq[CP_BLOCK];
    const int l = blockIdx.x, tid = threadIdx.x;
    const double y1 = LM[l], y2 = LM[num_lm + l];
    const long long tot = (long long)m * n;
    double best = 1.0e300 * 1.0e300;   // +inf
    long long bq = tot;
    for (long long q = tid; q < tot; q += CP_BLOCK) {
        const int j = (int)(q / m), i = (int)(q - (long long)j * m);
        const double g1 = __dadd_rn(__dadd_rn(om0, __dmul_rn(hx, (double)i + 0.5)), u[q]);          // IRE/cell_centered_grid.m:21-29
        const double g2 = __dadd_rn(__dadd_rn(om2, __dmul_rn(hy, (double)j + 0.5)), u[tot + q]);
        const double e1 = g1 - y1, e2 = g2 - y2;
        const double d = __dadd_rn(__dmul_rn(e1, e1), __dmul_rn(e2, e2));
        if (d < best) { best = d; bq = q; }           // q ascends within a thread: the first minimum is kept
    }
    sd[tid] = best; sq[tid] = bq;
    __syncthreads();
    for (int s = CP_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const double od = sd[tid + s];
            const long long oq = sq[tid + s];
            if (od < sd[tid] || (od == sd[tid] && oq < sq[tid])) { sd[tid] = od; sq[tid] = oq; }
        }
        __syncthreads();
    }
    if (tid != 0) return;
    const long long q0 = sq[0] < tot ? sq[0] : 0;     // all-NaN input: MATLAB's min returns index 1
    const int j0 = (int)(q0 / m), i0 = (int)(q0 - (long long)j0 * m);
    const double p1 = __dadd_rn(om0, __dmul_rn(hx, (double)i0 + 0.5)), p2 = __dadd_rn(om2, __dmul_rn(hy, (double)j0 + 0.5));
    const double inv1 = 1.0 / hx, inv2 = 1.0 / hy;
    const double *ux = u, *uy = u + tot;
    auto F = [&](double z1, double z2, double &f1, double &f2) {
        double v1, v2, dx, dy;
        bilinear_tap(ux, m, n, z1, z2, inv1, inv2, v1, dx, dy);
        bilinear_tap(uy, m, n, z1, z2, inv1, inv2, v2, dx, dy);
        f1 = z1 + v1; f2 = z2 + v2;
    };
    const double ysq1 = y1 * y1, ysq2 = y2 * y2;
    double x1 = p1, x2 = p2, f1, f2, crit = 0.0;
    for (int it = 0; it < 100; ++it) {
        F(x1, x2, f1, f2);
        x1 = x1 + (y1 - f1);
        x2 = x2 + (y2 - f2);
        F(x1, x2, f1, f2);
        crit = (f1 - ysq1) + (f2 - ysq2);
        if (crit < 1e-12) break;
    }
    const bool fail = !(crit < 1e-12);
    if (fail) { x1 = p1; x2 = p2; }
    out[l] = x1; out[num_lm + l] = x2;
    if (failed) failed[l] = fail ? 1 : 0;
}

// LM_acc(l) = sum_i ||LM_i(l,:) - mean_i LM_i(l,:)|| / k   (NNE/landmark_accuracy.m:19-27); LM stacked num_lm x 2 x k
__global__ void k_landmark_accuracy(const double *__restrict__ LM, int num_lm, int k, double *__restrict__ acc) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= num_lm) return;
    double m1 = 0.0, m2 = 0.0;
    for (int i = 0; i < k; ++i) { m1 += LM[(size_t)i * 2 * num_lm + l]; m2 += LM[(size_t)i * 2 * num_lm + num_lm + l]; }
    m1 /= (double)k; m2 /= (double)k;
    double s = 0.0;
    for (int i = 0; i < k; ++i) {
        const double e1 = LM[(size_t)i * 2 * num_lm + l] - m1, e2 = LM[(size_t)i * 2 * num_lm + num_lm + l] - m2;
        s += sqrt(__dadd_rn(__dmul_rn(e1, e1), __dmul_rn(e2, e2)));
    }
    acc[l] = s / (double)k;
}

extern "C" int cp_landmark_transform(int32_t device, const double *LM, int32_t num_lm, const double *u, int32_t m, int32_t n,
                                     const double *omega, double *LM_transformed, int32_t *failed) {
    if (!LM || !u || !omega || !LM_transformed) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_landmark_transform needs LM, u, omega and LM_transformed");
    if (m < 1 || n < 1 || num_lm < 0) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg field %d x %d, %d landmarks", m, n, num_lm);
    if (!(omega[1] > omega[0]) || !(omega[3] > omega[2])) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg omega must span a region");
    if (num_lm == 0) return CP_OK;
    HIPCHK(hipSetDevice(device));
    const size_t tot = (size_t)m * n;
    double *d_u = nullptr, *d_lm = nullptr, *d_out = nullptr;
    int *d_f = nullptr;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc(&d_u, 2 * tot * sizeof(double)));
        HIPCHK(hipMalloc(&d_lm, 2 * (size_t)num_lm * sizeof(double)));
        HIPCHK(hipMalloc(&d_out, 2 * (size_t)num_lm * sizeof(double)));
        HIPCHK(hipMalloc(&d_f, (size_t)num_lm * sizeof(int)));
        HIPCHK(hipMemcpy(d_u, u, 2 * tot * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_lm, LM, 2 * (size_t)num_lm * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_landmark_transform, dim3(num_lm), dim3(CP_BLOCK), 0, nullptr, m, n, omega[0], omega[2],
                           (omega[1] - omega[0]) / m, (omega[3] - omega[2]) / n, d_u, d_lm, num_lm, d_out, d_f);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(LM_transformed, d_out, 2 * (size_t)num_lm * sizeof(double), hipMemcpyDeviceToHost));
        if (failed) {
            static_assert(sizeof(int) == sizeof(int32_t), "int32_t flags");
            HIPCHK(hipMemcpy(failed, d_f, (size_t)num_lm * sizeof(int), hipMemcpyDeviceToHost));
        }
        return CP_OK;
    };
    const int rc = body();
    for (void *p : {(void *)d_u, (void *)d_lm, (void *)d_out, (void *)d_f})
        if (p) hipFree(p);
    return rc;
}

extern "C" int cp_landmark_accuracy(int32_t device, const double *LM, int32_t num_lm, int32_t k, double *LM_acc) {
    if (!LM || !LM_acc || num_lm < 0 || k < 1) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_landmark_accuracy");
    if (num_lm == 0) return CP_OK;
    HIPCHK(hipSetDevice(device));
    double *d_lm = nullptr, *d_acc = nullptr;
    const size_t len = 2 * (size_t)num_lm * k;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc(&d_lm, len * sizeof(double)));
        HIPCHK(hipMalloc(&d_acc, (size_t)num_lm * sizeof(double)));
        HIPCHK(hipMemcpy(d_lm, LM, len * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_landmark_accuracy, dim3((num_lm + 63) / 64), dim3(64), 0, nullptr, d_lm, num_lm, k, d_acc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(LM_acc, d_acc, (size_t)num_lm * sizeof(double), hipMemcpyDeviceToHost));
        return CP_OK;
    };
    const int rc = body();
    for (double *p : {d_lm, d_acc})
        if (p) hipFree(p);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Soft nuclear norm and the L-only problems of the landscape experiment (cp_nnl.cuh)
struct NnlWork {
    size_t npix = 0;
    int K = 0, grid = 0;
    double *I = nullptr, *x = nullptr, *xt = nullptr, *xbar = nullptr, *y = nullptr, *part = nullptr, *W = nullptr,
           *hist_p = nullptr, *hist_d = nullptr;
    NnlCtrl *ctrl = nullptr;
    ~NnlWork() {
        for (void *p : {(void *)I, (void *)x, (void *)xt, (void *)xbar, (void *)y, (void *)part, (void *)W, (void *)hist_p,
                        (void *)hist_d, (void *)ctrl})
            if (p) hipFree(p);
    }
    int init(size_t npix_, int K_, int hist_rows, bool full) {
        npix = npix_; K = K_;
        grid = (int)std::min<size_t>((npix + CP_BLOCK - 1) / CP_BLOCK, 768);
        const size_t fb = npix * K * sizeof(double);
        HIPCHK(hipMalloc(&x, fb));
        HIPCHK(hipMalloc(&xt, fb));
        if (full) {
            HIPCHK(hipMalloc(&I, fb));
            HIPCHK(hipMalloc(&xbar, fb));
            HIPCHK(hipMalloc(&y, fb));
            HIPCHK(hipMalloc(&hist_p, sizeof(double) * CP_HCOLS_NNL * hist_rows));
            HIPCHK(hipMalloc(&hist_d, sizeof(double) * CP_HCOLS_NNL * hist_rows));
            HIPCHK(hipMemset(hist_p, 0, sizeof(double) * CP_HCOLS_NNL * hist_rows));
            HIPCHK(hipMemset(hist_d, 0, sizeof(double) * CP_HCOLS_NNL * hist_rows));
        }
        HIPCHK(hipMalloc(&part, sizeof(double) * CP_PSTRIDE * grid));
        HIPCHK(hipMemset(part, 0, sizeof(double) * CP_PSTRIDE * grid));
        HIPCHK(hipMalloc(&W, sizeof(double) * 256));
        HIPCHK(hipMalloc(&ctrl, sizeof(NnlCtrl)));
        HIPCHK(hipMemset(ctrl, 0, sizeof(NnlCtrl)));
        return CP_OK;
    }
    void lin(const double *a, double ca, const double *b, double cb, int useB, double *out, bool gram) {
        hipLaunchKernelGGL(k_nnl_lin, dim3(grid), dim3(CP_BLOCK), 0, nullptr, ctrl, npix, K, a, ca, b, cb, useB, out,
                           gram ? part : nullptr);
    }
    void spec(int mode, double param) {
        hipLaunchKernelGGL(k_nnl_spec, dim3(1), dim3(CP_RED_THREADS), 0, nullptr, ctrl, grid, K, mode, param, part, W);
    }
    void applyW(const double *in, double *out) {
        hipLaunchKernelGGL(k_nnl_applyW, dim3(grid), dim3(CP_BLOCK), 0, nullptr, ctrl, npix, K, in, W, out);
    }
    void sad(double *v, double s, int conj) {
        hipLaunchKernelGGL(k_nnl_sad, dim3(grid), dim3(CP_BLOCK), 0, nullptr, ctrl, npix * K, v, I, s, conj);
    }
};

static int nnl_check(const void *L, int64_t npix, int32_t k) {
    if (!L) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg null matrix");
    if (npix < 1 || k < 1) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg matrix %lld x %d", (long long)npix, k);
    if (k > CP_KMAX) CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported at most %d images (got %d)", CP_KMAX, k);
    return CP_OK;
}

extern "C" int cp_nuclear_norm(int32_t device, const double *L, int64_t npix, int32_t k, double tau, double mu,
                               int32_t conjugate_flag, double *res1, double *res2, double *prox) {
    CPCHK(nnl_check(L, npix, k));
    if (!res1 || !res2) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_nuclear_norm needs res1 and res2");
    if (!(mu > 0.0) || !(tau > 0.0)) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg mu and tau must be positive");
    HIPCHK(hipSetDevice(device));
    NnlWork w;
    CPCHK(w.init((size_t)npix, k, 0, false));
    const size_t fb = (size_t)npix * k * sizeof(double);
    HIPCHK(hipMemcpy(w.xt, L, fb, hipMemcpyHostToDevice));
    NnlCtrl c;
    if (!conjugate_flag) {
        w.lin(nullptr, 0.0, w.xt, 1.0, 0, w.x, true);                       // Gram(L)
        w.spec(NNL_SOFT, mu * tau);                                         // NNE/nuclear_norm.m:43-52
        if (prox) w.applyW(w.x, w.x);                                       // U max(S - mu tau, 0) V' = L W (:54)
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(&c, w.ctrl, sizeof c, hipMemcpyDeviceToHost));
        double s = 0.0;
        for (int i = 0; i < k; ++i) s += c.sv[i];
        *res1 = mu * s;                                                     // (:47)
        *res2 = 0.0;
    } else {
        w.lin(nullptr, 0.0, w.xt, 1.0 / mu, 0, w.xt, true);                 // L / mu (:63) and its Gram
        w.spec(NNL_VALUES, 0.0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(&c, w.ctrl, sizeof c, hipMemcpyDeviceToHost));
        double mx = 0.0;
        for (int i = 0; i < k; ++i) mx = std::max(mx, c.sv[i]);
        *res1 = mu * 0.0;                                                   // (:71)
        *res2 = (mx > 1.0) ? mx - 1.0 : 0.0;                                // (:72-74)
        if (prox) {   // Moreau (:78-80): (L/mu) - mu tau * prox_{(1/mu)(1/tau)||.||_*}((L/mu) / (mu tau))
            w.lin(nullptr, 0.0, w.xt, 1.0 / (mu * tau), 0, w.x, true);
            w.spec(NNL_SOFT, (1.0 / mu) * (1.0 / tau));
            w.applyW(w.x, w.x);
            w.lin(w.xt, 1.0, w.x, -(mu * tau), 0, w.x, false);
            HIPCHK(hipGetLastError());
        }
    }
    if (prox) HIPCHK(hipMemcpy(prox, w.x, fb, hipMemcpyDeviceToHost));
    return CP_OK;
}

extern "C" int cp_nn_measure(int32_t device, int32_t version, const double *I, int64_t npix, int32_t k, double param,
                             const double *L0, const double *P0, double theta, double tau, double sigma,
                             int32_t max_iter, double tol, double *L_star, double *P_star, double *primal_history,
                             double *dual_history, cp_result *res) {
    CPCHK(nnl_check(I, npix, k));
    if (version < 1 || version > 3) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg version %d (1 soft, 2 hard, 3 mean-free hard)", version);
    if (version == 3 && k < 2) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg the mean-free problem needs at least 2 images");
    if (max_iter < 0) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg max_iter < 0");
    if (!(tau > 0.0) || !(sigma > 0.0) || !(param > 0.0)) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg tau, sigma and mu|nu must be positive");
    HIPCHK(hipSetDevice(device));
    NnlWork w;
    CPCHK(w.init((size_t)npix, k, max_iter + 1, true));
    const size_t fb = (size_t)npix * k * sizeof(double), ntot = (size_t)npix * k;
    HIPCHK(hipMemcpy(w.I, I, fb, hipMemcpyHostToDevice));
    if (L0) HIPCHK(hipMemcpy(w.x, L0, fb, hipMemcpyHostToDevice)); else HIPCHK(hipMemset(w.x, 0, fb));
    if (P0) HIPCHK(hipMemcpy(w.y, P0, fb, hipMemcpyHostToDevice)); else HIPCHK(hipMemset(w.y, 0, fb));
    HIPCHK(hipMemcpy(w.xbar, w.x, fb, hipMemcpyDeviceToDevice));           // PDO/chambolle_pock.m:82-84
    const int useB = (version == 3) ? 1 : 0;
    auto row = [&](int r, int use_fsv) {
        hipLaunchKernelGGL(k_nnl_stats, dim3(w.grid), dim3(CP_BLOCK), 0, nullptr, w.ctrl, w.npix, k, useB, w.x, w.y, w.I, w.part);
        NnlHistArgs A{w.grid, k, version, r, use_fsv, 1, param, tol};
        hipLaunchKernelGGL(k_nnl_hist, dim3(1), dim3(CP_RED_THREADS), 0, nullptr, A, w.ctrl, w.part, w.hist_p, w.hist_d);
    };
    row(0, 0);
    for (int i = 1; i <= max_iter; ++i) {
        // y <- prox_{sigma F*}(y + sigma K xbar)  (:162)
        w.lin(w.y, 1.0, w.xbar, sigma, useB, w.y, version == 3);
        if (version == 3) {
            w.spec(NNL_CONJ_BALL, param * tau);        // nuclear_norm_constraint(., k, tau, nu, true): mu_ = nu * tau (:72)
            w.applyW(w.y, w.y);
        } else {
            w.sad(w.y, sigma, 1);                      // SAD(., I, 1, sigma, true)
        }
        // x <- prox_{tau G}(x - tau K' y)  (:165); B is symmetric
        w.lin(w.x, 1.0, w.y, -tau, useB, w.xt, version != 3);
        if (version == 3) {
            hipLaunchKernelGGL(k_nnl_sad, dim3(w.grid), dim3(CP_BLOCK), 0, nullptr, w.ctrl, ntot, w.xt, w.I, sigma, 0);   // SAD(., I, 1, sigma, false)
        } else {
            w.spec(version == 1 ? NNL_SOFT : NNL_BALL, version == 1 ? param * tau : param);
            w.applyW(w.xt, w.xt);
        }
        hipLaunchKernelGGL(k_nnl_extrap, dim3(w.grid), dim3(CP_BLOCK), 0, nullptr, w.ctrl, ntot, w.xt, w.x, w.xbar, theta);   // (:179)
        row(i, 1);
    }
    HIPCHK(hipGetLastError());
    NnlCtrl c;
    HIPCHK(hipMemcpy(&c, w.ctrl, sizeof c, hipMemcpyDeviceToHost));
    const int iterations = c.stop ? c.stop_iter : max_iter;
    const int rows = iterations + 1;
    if (res) {
        res->iterations = iterations;
        res->stop_reason = (c.stop && iterations < max_iter) ? CP_STOP_TOLERANCE : CP_STOP_MAXITER;   // (:137 is tested first)
        res->rows = rows;
        res->cols = CP_HCOLS_NNL;
    }
    if (L_star) HIPCHK(hipMemcpy(L_star, w.x, fb, hipMemcpyDeviceToHost));
    if (P_star) HIPCHK(hipMemcpy(P_star, w.y, fb, hipMemcpyDeviceToHost));
    std::vector<double> tmp((size_t)rows * CP_HCOLS_NNL);
    for (int which = 0; which < 2; ++which) {
        double *dst = which ? dual_history : primal_history;
        if (!dst) continue;
        HIPCHK(hipMemcpy(tmp.data(), which ? w.hist_d : w.hist_p, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
        for (int cc = 0; cc < CP_HCOLS_NNL; ++cc)
            for (int r = 0; r < rows; ++r) dst[(size_t)cc * rows + r] = tmp[(size_t)r * CP_HCOLS_NNL + cc];
    }
    return CP_OK;
}

// ------------------------------------------------------------------------------------------------
// Streaming copy: the practical HBM ceiling of this box next to the 8 TB/s of the data sheet (bench.py's
// roofline.measured_copy_GBps). ONE element of VEC doubles per thread, no loop (the grid covers the buffer), nontemporal on
// both sides. Round 3 (scratch/bw_copy.hip, profiles/r03_copy_shapes.txt): this shape reaches 6.50 TB/s at 16 B per lane and
// 5.74 at 8 B per lane; the grid-stride loop of round 2 (one dependent load -> store per thread and trip, 4096 workgroups)
// reached 4.9 / 5.1 on the same box -- it measured its own loop, not the memory system. The guide's figure for a float4 copy
// is 6.29 TB/s (/opt/skills/guides/MI355X_MICROARCH.md:36).
template <int VEC>
__global__ __launch_bounds__(CP_BLOCK) void k_stream_copy(size_t nv, const double *__restrict__ src, double *__restrict__ dst) {
    typedef double vec_t __attribute__((ext_vector_type(VEC)));
    const size_t q = (size_t)blockIdx.x * CP_BLOCK + threadIdx.x;
    if (q < nv)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const vec_t *>(src) + q),
                                    reinterpret_cast<vec_t *>(dst) + q);
}

extern "C" int cp_stream_copy_gbps(int32_t device, int64_t bytes, int32_t lane_bytes, int32_t repeats, double *gbps) {
    if (!gbps || bytes < (1 << 20) || repeats < 1 || (lane_bytes != 8 && lane_bytes != 16))
        CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_stream_copy_gbps(bytes >= 1 MiB, lane_bytes 8 | 16, repeats >= 1)");
    HIPCHK(hipSetDevice(device));
    const size_t n = (size_t)bytes / 16 * 2;   // doubles, a multiple of 2
    double *a = nullptr, *bq = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = CP_OK;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc(&a, n * sizeof(double)));
        HIPCHK(hipMalloc(&bq, n * sizeof(double)));
        HIPCHK(hipMemset(a, 0, n * sizeof(double)));
        HIPCHK(hipMemset(bq, 0, n * sizeof(double)));
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        double best = 0.0;
        const size_t nv = (lane_bytes == 16) ? n / 2 : n;
        const unsigned grid = (unsigned)((nv + CP_BLOCK - 1) / CP_BLOCK);
        for (int r = 0; r < repeats + 1; ++r) {   // the first pass warms up
            HIPCHK(hipEventRecord(e0, nullptr));
            if (lane_bytes == 16) hipLaunchKernelGGL((k_stream_copy<2>), dim3(grid), dim3(CP_BLOCK), 0, nullptr, nv, a, bq);
            else hipLaunchKernelGGL((k_stream_copy<1>), dim3(grid), dim3(CP_BLOCK), 0, nullptr, nv, a, bq);
            HIPCHK(hipEventRecord(e1, nullptr));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms > 0.f) best = std::max(best, 2.0 * (double)n * sizeof(double) / (ms * 1e-3) / 1e9);
        }
        HIPCHK(hipGetLastError());
        *gbps = best;
        return CP_OK;
    };
    rc = body();
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (a) hipFree(a);
    if (bq) hipFree(bq);
    return rc;
}

// ================================================================================================
// Allocation-free variant of cp_singular_values (include/cpengine.h: cp_singular_values_static). Same kernels, same
// two passes (Gram of B*field -> Jacobi, second Gram in the rotated basis -> Jacobi), but the per-workgroup partial
// rows, the rotation and the eigenvalues live in a file-scope device array instead of the handle's hipMalloc'ed
// buffers, the input is a caller-owned device field and the sorted result goes to a caller-owned device buffer.
#define CP_NA_MAXBLK 768                                     // = the grid cap of grid_for() for the multi-image models
#define CP_NA_PART ((size_t)CP_NA_MAXBLK * CP_PSTRIDE)        // partial rows
#define CP_NA_SCRATCH (CP_NA_PART + 16 + 256 + 256 + 256)     // + eigenvalues, rotation out, identity, rotation in
static __device__ __align__(256) double scratch[CP_NA_SCRATCH];   // 3.3 MB on every GPU the library runs on
#define CP_NA_MAXDEV 64
static std::atomic<double *> g_na_ptr[CP_NA_MAXDEV];           // device address of `scratch` per device (null: not looked up yet)

__global__ void k_na_identity(double *ident) {
    const int t = threadIdx.x;
    ident[t] = ((t >> 4) == (t & 15)) ? 1.0 : 0.0;
}
// sv[rank] = sqrt(max(lambda, 0)), descending (rank counting; ties keep their order) -- what the host does in cp_singular_values
__global__ void k_na_sort(const double *__restrict__ lam, int K, double *__restrict__ sv) {
    __shared__ double s[16];
    const int i = threadIdx.x;
    if (i < 16) s[i] = (i < K) ? sqrt(fmax(lam[i], 0.0)) : -1.0;
    __syncthreads();
    if (i < K) {
        int rank = 0;
        for (int j = 0; j < K; ++j) rank += (s[j] > s[i] || (s[j] == s[i] && j < i)) ? 1 : 0;
        sv[rank] = s[i];
    }
}

extern "C" int cp_singular_values_static(int32_t device, const double *dev_field, int32_t m, int32_t n, int32_t k,
                                         double *dev_sv, void *hip_stream) {
    if (!dev_field || !dev_sv) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg cp_singular_values_static needs a device field and a device output");
    if (device < 0 || device >= CP_NA_MAXDEV) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg device %d", device);
    if (m < 1 || n < 1) CP_FAIL(CP_ERR_BAD_ARG, "cpengine:badArg image size %d x %d", m, n);
    if (k < 1 || k > CP_KMAX)
        CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported cp_singular_values_static: k = %d exceeds the static scratch (1..%d images)", k, CP_KMAX);
    if ((long long)m * n > (1LL << 30))
        CP_FAIL(CP_ERR_UNSUPPORTED, "cpengine:unsupported cp_singular_values_static: %d x %d pixels exceed the limit of 2^30 per image", m, n);
    // the caller's current device is restored on every path out (a serving loop must not find its device changed)
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard;
    int cur = -1;
    HIPCHK(hipGetDevice(&cur));
    if (cur != device) { HIPCHK(hipSetDevice(device)); guard.prev = cur; }
    double *base = g_na_ptr[device].load(std::memory_order_acquire);
    if (!base) {   // once per device, with that device current: every GPU holds its own copy of the array. Two threads that
                   // race here look up the same address and store the same value. This first call per device must happen
                   // OUTSIDE a stream capture (hipGetSymbolAddress is not a stream operation); later calls only enqueue.
        double *ptr = nullptr;
        HIPCHK(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) CP_FAIL(CP_ERR_DEVICE, "cpengine:device hipGetSymbolAddress(scratch) returned a null address");
        g_na_ptr[device].store(ptr, std::memory_order_release);
        base = ptr;
    }
    double *part = base, *svout = base + CP_NA_PART, *vout = svout + 16, *ident = svout + 16 + 256, *vin = ident + 256;
    hipStream_t st = (hipStream_t)hip_stream;
    MfParams P;
    memset(&P, 0, sizeof P);
    P.m = 0;                       // the field has no halo column: element q of image i at i*npix + q
    P.nl = n; P.K = k; P.npix = m * n; P.n_global = n; P.fs = m * n; P.invK = 1.0 / (double)k;
    const int nblk = std::max(1, std::min((P.npix + CP_BLOCK - 1) / CP_BLOCK, CP_NA_MAXBLK));
    // every buffer below is written before it is read within this call; rows of `part` beyond nblk keep an earlier
    // call's contents and are not read
    hipLaunchKernelGGL(k_na_identity, dim3(1), dim3(256), 0, st, ident);
    hipLaunchKernelGGL((k_mf_gram_field<double>), dim3(nblk), dim3(CP_BLOCK), 0, st, P, dev_field, ident, part);
    hipLaunchKernelGGL(k_mf_sv_eig, dim3(1), dim3(CP_BLOCK), 0, st, nblk, part, ident, svout);
    HIPCHK(hipMemcpyAsync(vin, vout, sizeof(double) * 256, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL((k_mf_gram_field<double>), dim3(nblk), dim3(CP_BLOCK), 0, st, P, dev_field, vin, part);
    hipLaunchKernelGGL(k_mf_sv_eig, dim3(1), dim3(CP_BLOCK), 0, st, nblk, part, vin, svout);
    hipLaunchKernelGGL(k_na_sort, dim3(1), dim3(64), 0, st, svout, k, dev_sv);
    HIPCHK(hipGetLastError());
    return CP_OK;
}
