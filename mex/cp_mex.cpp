// cp_mex -- MEX gateway between the reference's MATLAB entry points (matlab/*.m) and libcpengine.so (include/cpengine.h).
//
//     out = cp_mex('command', args...)
//
// One command per C-ABI entry point; handles travel as uint64 scalars. Rules of the MEX contract kept here:
//   * prhs[] is never written (inputs are read through const double*), outputs are mxCreate* arrays owned by MATLAB;
//   * mexErrMsgIdAndTxt longjmps out of the function: no C++ object with a destructor is alive when it is called
//     (error text is copied into a static buffer first), and no exception crosses the boundary (catch (...));
//   * the library keeps device state between calls: mexLock() at the first create, every live handle is destroyed by the
//     mexAtExit hook (clear mex / exit), mexUnlock() when the last handle goes.
// Build:  mex -I../include cp_mex.cpp -L../ma_amd -lcpengine      (MATLAB)
//         mkoctfile --mex -I../include cp_mex.cpp -L../ma_amd -lcpengine   (Octave)
// Syntax check in this repository (no MEX toolchain in the image): g++ -fsyntax-only -I mex/proto -I include mex/cp_mex.cpp
#include "mex.h"

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "cpengine.h"

namespace {

const int kMaxHandles = 256;
cp_handle g_live[kMaxHandles];
int g_nlive = 0;
char g_msg[600];

void at_exit_hook() {
    for (int i = 0; i < kMaxHandles; ++i)
        if (g_live[i]) { cp_destroy(g_live[i]); g_live[i] = nullptr; }
    g_nlive = 0;
}

[[noreturn]] void fail_id(const char *id, const char *text) {
    snprintf(g_msg, sizeof g_msg, "%s", text);
    mexErrMsgIdAndTxt(id, "%s", g_msg);
    for (;;) {}   // not reached: mexErrMsgIdAndTxt does not return
}

// status code of the library -> MATLAB error with the library's message id (cpengine:badArg, cpengine:badBC, ...)
void check(int rc) {
    if (rc == CP_OK) return;
    const char *e = cp_last_error();
    char id[64] = "cpengine:error";
    if (e && !strncmp(e, "cpengine:", 9)) {
        size_t n = 0;
        while (e[n] && e[n] != ' ' && n + 1 < sizeof id) { id[n] = e[n]; ++n; }
        id[n] = 0;
    }
    fail_id(id, e ? e : "cpengine: unknown error");
}

void need(bool ok, const char *what) {
    if (!ok) fail_id("cpengine:badArg", what);
}

cp_handle handle_of(const mxArray *a) {
    need(a && mxGetClassID(a) == mxUINT64_CLASS && mxGetNumberOfElements(a) == 1, "cp_mex: handle must be a uint64 scalar");
    const uint64_t v = *static_cast<const uint64_t *>(mxGetData(a));
    cp_handle h = reinterpret_cast<cp_handle>(static_cast<uintptr_t>(v));
    for (int i = 0; i < kMaxHandles; ++i)
        if (g_live[i] == h && h) return h;
    fail_id("cpengine:badArg", "cp_mex: stale or foreign handle");
}

const double *vec_or_null(const mxArray *a, size_t expect, const char *name) {
    if (!a || mxIsEmpty(a)) return nullptr;
    need(mxIsDouble(a) && !mxIsComplex(a), "cp_mex: real double array expected");
    if (expect && mxGetNumberOfElements(a) != expect) {
        snprintf(g_msg, sizeof g_msg, "cp_mex: %s has %zu elements, expected %zu", name, mxGetNumberOfElements(a), expect);
        mexErrMsgIdAndTxt("cpengine:badArg", "%s", g_msg);
    }
    return mxGetPr(a);
}

double scalar(const mxArray *a, const char *name) {
    need(a && mxGetNumberOfElements(a) == 1, name);
    return mxGetScalar(a);
}

// problem sizes of a handle as the gateway needs them for output allocation; kept beside the handle
struct Dims { cp_handle h; int model, m, n, k, batch, n_global; };
Dims g_dims[kMaxHandles];

const Dims &dims_of(cp_handle h) {
    for (int i = 0; i < kMaxHandles; ++i)
        if (g_dims[i].h == h) return g_dims[i];
    fail_id("cpengine:badArg", "cp_mex: unknown handle");
}
size_t nx_of(const Dims &d) {
    const size_t mn = (size_t)d.m * d.n;
    switch (d.model) {
        case CP_MODEL_TV_L1_DENOISE: return mn * d.batch;
        case CP_MODEL_TV_L1_REG: case CP_MODEL_TV_L2_REG: return 2 * mn * d.batch;
        case CP_MODEL_VAR_TV: case CP_MODEL_VAR_CURV: return 2 * mn * d.k;
        default: return 3 * mn * d.k;
    }
}
size_t ny_of(const Dims &d) {
    const size_t mn = (size_t)d.m * d.n;
    switch (d.model) {
        case CP_MODEL_TV_L1_DENOISE: return 2 * mn * d.batch;
        case CP_MODEL_TV_L1_REG: case CP_MODEL_TV_L2_REG: return 4 * mn * d.batch;
        case CP_MODEL_MF_NN_TV: return 6 * mn * d.k;
        case CP_MODEL_MF_NN_CURV: return 4 * mn * d.k;
        case CP_MODEL_VAR_TV: return 5 * mn * d.k;
        default: return 3 * mn * d.k;
    }
}

void dispatch(const char *cmd, int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (!strcmp(cmd, "version")) {
        plhs[0] = mxCreateString(cp_version());
        return;
    }
    if (!strcmp(cmd, "create")) {
        // h = cp_mex('create', model, m, n, k, bc, h_grid, omega, lambda, mu [, batch, flags, n_global, j_offset, device])
        need(nrhs >= 10, "cp_mex('create', model, m, n, k, bc, h_grid, omega, lambda, mu, ...)");
        cp_problem_desc d;
        memset(&d, 0, sizeof d);
        d.model = (int32_t)scalar(prhs[1], "model"); d.m = (int32_t)scalar(prhs[2], "m"); d.n = (int32_t)scalar(prhs[3], "n");
        d.k = (int32_t)scalar(prhs[4], "k"); d.bc = (int32_t)scalar(prhs[5], "bc");
        const double *hg = vec_or_null(prhs[6], 2, "h_grid");
        need(hg != nullptr, "cp_mex: h_grid = [h1, h2]");
        d.h[0] = hg[0]; d.h[1] = hg[1];
        if (const double *om = vec_or_null(prhs[7], 4, "omega")) memcpy(d.omega, om, sizeof d.omega);   // [] -> default region
        d.lambda = scalar(prhs[8], "lambda"); d.mu = scalar(prhs[9], "mu");
        d.batch = nrhs > 10 ? (int32_t)scalar(prhs[10], "batch") : 1;
        d.flags = nrhs > 11 ? (int32_t)scalar(prhs[11], "flags") : 0;
        d.n_global = nrhs > 12 ? (int32_t)scalar(prhs[12], "n_global") : 0;
        d.j_offset = nrhs > 13 ? (int32_t)scalar(prhs[13], "j_offset") : 0;
        d.device = nrhs > 14 ? (int32_t)scalar(prhs[14], "device") : 0;
        need(g_nlive < kMaxHandles, "cp_mex: too many live handles");
        cp_handle h = nullptr;
        check(cp_create(&d, &h));
        for (int i = 0; i < kMaxHandles; ++i)
            if (!g_live[i]) {
                g_live[i] = h;
                g_dims[i] = Dims{h, d.model, d.m, d.n, d.k < 1 ? 1 : d.k, d.batch < 1 ? 1 : d.batch, d.n_global > 0 ? d.n_global : d.n};
                break;
            }
        if (g_nlive++ == 0) { mexLock(); mexAtExit(at_exit_hook); }
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *static_cast<uint64_t *>(mxGetData(plhs[0])) = static_cast<uint64_t>(reinterpret_cast<uintptr_t>(h));
        return;
    }
    if (!strcmp(cmd, "evaluate_displacement")) {
        // [img_u, g] = cp_mex('evaluate_displacement', img, h, u, omega, s_grid)      IRE/evaluate_displacement.m:1
        need(nrhs >= 3, "cp_mex('evaluate_displacement', img, h, u, omega, s_grid)");
        const int m = (int)mxGetM(prhs[1]), n = (int)mxGetN(prhs[1]);
        int k = m, l = n;
        if (nrhs > 5 && !mxIsEmpty(prhs[5])) { const double *sg = vec_or_null(prhs[5], 2, "s_grid"); k = (int)sg[0]; l = (int)sg[1]; }
        const double *u = nrhs > 3 ? vec_or_null(prhs[3], 0, "u") : nullptr;
        if (u && mxGetNumberOfElements(prhs[3]) == 1) { need(u[0] == 0.0, "cp_mex: scalar u must be 0"); u = nullptr; }
        if (u) need(mxGetNumberOfElements(prhs[3]) == (size_t)2 * k * l, "cp_mex: u must be (k*l) x 2");
        const double *om = nrhs > 4 ? vec_or_null(prhs[4], 4, "omega") : nullptr;
        plhs[0] = mxCreateDoubleMatrix(k, l, mxREAL);
        mxArray *g = mxCreateDoubleMatrix((mwSize)k * l, 2, mxREAL);
        check(cp_evaluate_displacement(0, vec_or_null(prhs[1], 0, "img"), m, n, vec_or_null(prhs[2], 2, "h"), u, om, k, l,
                                       mxGetPr(plhs[0]), mxGetPr(g)));
        if (nlhs > 1) plhs[1] = g;
        return;
    }
    if (!strcmp(cmd, "landmark_transform")) {
        // [LM_transformed, failed] = cp_mex('landmark_transform', LM, u, omega)      NNE/landmark_transform.m:1
        need(nrhs >= 4, "cp_mex('landmark_transform', LM, u, omega)");
        const int num_lm = (int)mxGetM(prhs[1]);
        need(mxGetN(prhs[1]) == 2, "cp_mex: LM must be numLM x 2");
        const mwSize *ud = mxGetDimensions(prhs[2]);
        need(mxGetNumberOfDimensions(prhs[2]) == 3 && ud[2] == 2, "cp_mex: u must be m x n x 2");
        plhs[0] = mxCreateDoubleMatrix(num_lm, 2, mxREAL);
        mxArray *fl = mxCreateNumericMatrix(num_lm, 1, mxINT32_CLASS, mxREAL);
        check(cp_landmark_transform(0, vec_or_null(prhs[1], 0, "LM"), num_lm, vec_or_null(prhs[2], 0, "u"), (int)ud[0], (int)ud[1],
                                    vec_or_null(prhs[3], 4, "omega"), mxGetPr(plhs[0]), static_cast<int32_t *>(mxGetData(fl))));
        if (nlhs > 1) plhs[1] = fl;
        return;
    }
    if (!strcmp(cmd, "landmark_accuracy")) {
        // LM_acc = cp_mex('landmark_accuracy', LM3)  with LM3 = cat(3, LM{:}) (numLM x 2 x k)      NNE/landmark_accuracy.m:1
        need(nrhs >= 2, "cp_mex('landmark_accuracy', cat(3, LM{:}))");
        const mwSize nd = mxGetNumberOfDimensions(prhs[1]);
        const mwSize *ld = mxGetDimensions(prhs[1]);
        need(ld[1] == 2, "cp_mex: LM must be numLM x 2 x k");
        const int k = nd > 2 ? (int)ld[2] : 1;
        plhs[0] = mxCreateDoubleMatrix(ld[0], 1, mxREAL);
        check(cp_landmark_accuracy(0, vec_or_null(prhs[1], 0, "LM"), (int)ld[0], k, mxGetPr(plhs[0])));
        return;
    }
    if (!strcmp(cmd, "nuclear_norm")) {
        // [res1, res2, res3] = cp_mex('nuclear_norm', L, numImg, tau, mu, conjugate_flag)      NNE/nuclear_norm.m:1-2
        need(nrhs >= 5, "cp_mex('nuclear_norm', L, numImg, tau, mu, conjugate_flag)");
        const int k = (int)scalar(prhs[2], "numImg");
        const size_t tot = mxGetNumberOfElements(prhs[1]);
        need(k >= 1 && tot % (size_t)k == 0, "cp_mex: numel(L) must be a multiple of numImg");
        double r1 = 0.0, r2 = 0.0;
        mxArray *px = (nlhs > 2) ? mxCreateDoubleMatrix(tot, 1, mxREAL) : nullptr;
        check(cp_nuclear_norm(0, vec_or_null(prhs[1], 0, "L"), (int64_t)(tot / k), k, scalar(prhs[3], "tau"), scalar(prhs[4], "mu"),
                              nrhs > 5 && scalar(prhs[5], "conjugate_flag") != 0.0, &r1, &r2, px ? mxGetPr(px) : nullptr));
        plhs[0] = mxCreateDoubleScalar(r1);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(r2);
        if (nlhs > 2) plhs[2] = px;
        return;
    }
    if (!strcmp(cmd, "nn_measure")) {
        // [x_star, y_star, primal_history, dual_history] = cp_mex('nn_measure', version, I, numImg, param, L0, P0, theta,
        //     tau, sigma, maxIter, tol)                                     NNE/test_nuclear_norm_measure.m:212-242
        need(nrhs >= 12, "cp_mex('nn_measure', version, I, numImg, param, L0, P0, theta, tau, sigma, maxIter, tol)");
        const int k = (int)scalar(prhs[3], "numImg");
        const size_t tot = mxGetNumberOfElements(prhs[2]);
        need(k >= 1 && tot % (size_t)k == 0, "cp_mex: numel(I) must be a multiple of numImg");
        const int maxIter = (int)scalar(prhs[10], "maxIter");
        need(maxIter >= 0, "cp_mex: maxIter >= 0");
        plhs[0] = mxCreateDoubleMatrix(tot, 1, mxREAL);
        mxArray *ys = mxCreateDoubleMatrix(tot, 1, mxREAL);
        mxArray *php = mxCreateDoubleMatrix((mwSize)(maxIter + 1) * 5, 1, mxREAL), *dhp = mxCreateDoubleMatrix((mwSize)(maxIter + 1) * 5, 1, mxREAL);
        cp_result r;
        check(cp_nn_measure(0, (int)scalar(prhs[1], "version"), vec_or_null(prhs[2], 0, "I"), (int64_t)(tot / k), k,
                            scalar(prhs[4], "param"), vec_or_null(prhs[5], tot, "L0"), vec_or_null(prhs[6], tot, "P0"),
                            scalar(prhs[7], "theta"), scalar(prhs[8], "tau"), scalar(prhs[9], "sigma"), maxIter,
                            scalar(prhs[11], "tol"), mxGetPr(plhs[0]), mxGetPr(ys), mxGetPr(php), mxGetPr(dhp), &r));
        if (nlhs > 1) plhs[1] = ys;
        for (int which = 0; which < 2 && nlhs > 2 + which; ++which) {   // the first rows*5 entries are the rows x 5 matrix
            plhs[2 + which] = mxCreateDoubleMatrix(r.rows, r.cols, mxREAL);
            memcpy(mxGetPr(plhs[2 + which]), mxGetPr(which ? dhp : php), sizeof(double) * (size_t)r.rows * r.cols);
        }
        return;
    }
    need(nrhs >= 2, "cp_mex: handle expected");
    cp_handle h = handle_of(prhs[1]);
    const Dims &D = dims_of(h);
    const size_t mn = (size_t)D.m * D.n, kk = (size_t)D.k * D.batch;
    if (!strcmp(cmd, "destroy")) {
        for (int i = 0; i < kMaxHandles; ++i)
            if (g_live[i] == h) { g_live[i] = nullptr; g_dims[i].h = nullptr; }
        cp_destroy(h);
        if (--g_nlive == 0) mexUnlock();
    } else if (!strcmp(cmd, "set_images")) {             // cp_mex('set_images', h, a [, b])
        need(nrhs >= 3, "cp_mex('set_images', h, a, b)");
        check(cp_set_images(h, vec_or_null(prhs[2], 0, "a"), nrhs > 3 ? vec_or_null(prhs[3], 0, "b") : nullptr));
    } else if (!strcmp(cmd, "set_primal")) {
        check(cp_set_primal(h, nrhs > 2 ? vec_or_null(prhs[2], nx_of(D), "x0") : nullptr));
    } else if (!strcmp(cmd, "set_dual")) {
        check(cp_set_dual(h, nrhs > 2 ? vec_or_null(prhs[2], ny_of(D), "y0") : nullptr));
    } else if (!strcmp(cmd, "linearize")) {              // cp_mex('linearize', h [, u0])
        check(cp_linearize(h, nrhs > 2 ? vec_or_null(prhs[2], 0, "u0") : nullptr));
    } else if (!strcmp(cmd, "get_linearization")) {      // [T_u, g, b] = cp_mex('get_linearization', h)
        plhs[0] = mxCreateDoubleMatrix(mn * kk, 1, mxREAL);
        mxArray *g = mxCreateDoubleMatrix(2 * mn * kk, 1, mxREAL), *b = mxCreateDoubleMatrix(mn * kk, 1, mxREAL);
        check(cp_get_linearization(h, mxGetPr(plhs[0]), mxGetPr(g), mxGetPr(b)));
        if (nlhs > 1) plhs[1] = g;
        if (nlhs > 2) plhs[2] = b;
    } else if (!strcmp(cmd, "set_linearization")) {      // cp_mex('set_linearization', h, T_u | [], g, b)
        need(nrhs >= 5, "cp_mex('set_linearization', h, T_u, g, b)");
        check(cp_set_linearization(h, vec_or_null(prhs[2], mn * kk, "T_u"), vec_or_null(prhs[3], 2 * mn * kk, "g"),
                                   vec_or_null(prhs[4], mn * kk, "b")));
    } else if (!strcmp(cmd, "set_nu")) {
        need(nrhs >= 3, "cp_mex('set_nu', h, nu)");
        check(cp_set_nu(h, scalar(prhs[2], "nu")));
    } else if (!strcmp(cmd, "singular_values")) {        // s = cp_mex('singular_values', h, which)
        need(nrhs >= 3, "cp_mex('singular_values', h, which)");
        plhs[0] = mxCreateDoubleMatrix(D.k, 1, mxREAL);
        check(cp_singular_values(h, (int)scalar(prhs[2], "which"), mxGetPr(plhs[0])));
    } else if (!strcmp(cmd, "norm_estimate")) {          // [e_end, iters] = cp_mex('norm_estimate', h [, x0])
        double est = 0.0;
        int32_t it = 0;
        check(cp_norm_estimate(h, nrhs > 2 ? vec_or_null(prhs[2], nx_of(D), "x0") : nullptr, &est, &it));
        plhs[0] = mxCreateDoubleScalar(est);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)it);
    } else if (!strcmp(cmd, "apply_A")) {
        need(nrhs >= 3, "cp_mex('apply_A', h, x)");
        plhs[0] = mxCreateDoubleMatrix(ny_of(D), 1, mxREAL);
        check(cp_apply_A(h, vec_or_null(prhs[2], nx_of(D), "x"), mxGetPr(plhs[0])));
    } else if (!strcmp(cmd, "apply_At")) {
        need(nrhs >= 3, "cp_mex('apply_At', h, y)");
        plhs[0] = mxCreateDoubleMatrix(nx_of(D), 1, mxREAL);
        check(cp_apply_At(h, vec_or_null(prhs[2], ny_of(D), "y"), mxGetPr(plhs[0])));
    } else if (!strcmp(cmd, "solve")) {                  // [iterations, stop_reason] = cp_mex('solve', h, theta, tau, sigma, maxIter, tol)
        need(nrhs >= 7, "cp_mex('solve', h, theta, tau, sigma, maxIter, tol)");
        cp_result r;
        check(cp_solve(h, scalar(prhs[2], "theta"), scalar(prhs[3], "tau"), scalar(prhs[4], "sigma"),
                       (int32_t)scalar(prhs[5], "maxIter"), scalar(prhs[6], "tol"), &r));
        plhs[0] = mxCreateDoubleScalar((double)r.iterations);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)r.stop_reason);
    } else if (!strcmp(cmd, "get_history")) {            // [primal_history, dual_history] = cp_mex('get_history', h [, problem])
        const int prob = nrhs > 2 ? (int)scalar(prhs[2], "problem") : 0;
        need(D.batch == 1 || nrhs > 2, "cp_mex('get_history', h, problem) for a batch handle");
        cp_result r;
        check(cp_get_result(h, prob, &r));
        if (D.batch == 1) {
            plhs[0] = mxCreateDoubleMatrix(r.rows, r.cols, mxREAL);
            mxArray *dh = mxCreateDoubleMatrix(r.rows, r.cols, mxREAL);
            check(cp_get_history(h, mxGetPr(plhs[0]), mxGetPr(dh)));
            if (nlhs > 1) plhs[1] = dh;
        } else {   // the library returns the problems stacked: copy out the requested one
            size_t tot = 0, off = 0;
            for (int p = 0; p < D.batch; ++p) { cp_result q; check(cp_get_result(h, p, &q)); if (p < prob) off += (size_t)q.rows * q.cols; tot += (size_t)q.rows * q.cols; }
            mxArray *allp = mxCreateDoubleMatrix(tot, 1, mxREAL), *alld = mxCreateDoubleMatrix(tot, 1, mxREAL);
            check(cp_get_history(h, mxGetPr(allp), mxGetPr(alld)));
            plhs[0] = mxCreateDoubleMatrix(r.rows, r.cols, mxREAL);
            memcpy(mxGetPr(plhs[0]), mxGetPr(allp) + off, sizeof(double) * r.rows * r.cols);
            if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(r.rows, r.cols, mxREAL); memcpy(mxGetPr(plhs[1]), mxGetPr(alld) + off, sizeof(double) * r.rows * r.cols); }
        }
    } else if (!strcmp(cmd, "get_primal")) {
        plhs[0] = mxCreateDoubleMatrix(nx_of(D), 1, mxREAL);
        check(cp_get_primal(h, mxGetPr(plhs[0])));
    } else if (!strcmp(cmd, "get_dual")) {
        plhs[0] = mxCreateDoubleMatrix(ny_of(D), 1, mxREAL);
        check(cp_get_dual(h, mxGetPr(plhs[0])));
    } else if (!strcmp(cmd, "get_images")) {
        plhs[0] = mxCreateDoubleMatrix((size_t)D.m * D.n_global * kk, 1, mxREAL);
        check(cp_get_images(h, mxGetPr(plhs[0])));
    } else if (!strcmp(cmd, "prolong")) {                // cp_mex('prolong', h_coarse, h_fine, factor)
        need(nrhs >= 4, "cp_mex('prolong', h_coarse, h_fine, factor)");
        check(cp_prolong(h, handle_of(prhs[2]), scalar(prhs[3], "factor")));
    } else if (!strcmp(cmd, "downsample")) {             // cp_mex('downsample', h_fine, h_coarse)
        need(nrhs >= 3, "cp_mex('downsample', h_fine, h_coarse)");
        check(cp_downsample(h, handle_of(prhs[2])));
    } else if (!strcmp(cmd, "dominant_kernel")) {        // name = cp_mex('dominant_kernel', h)
        char name[256];
        check(cp_dominant_kernel(h, name, (int32_t)sizeof name));
        plhs[0] = mxCreateString(name);
    } else if (!strcmp(cmd, "handover_fallbacks")) {     // n = cp_mex('handover_fallbacks', h): must be 0
        int64_t cnt = 0;
        check(cp_handover_fallbacks(h, &cnt));
        plhs[0] = mxCreateDoubleScalar((double)cnt);
    } else if (!strcmp(cmd, "iterate")) {                // cp_mex('iterate', h, theta, tau, sigma, iters): no stop rule, asynchronous
        need(nrhs >= 6, "cp_mex('iterate', h, theta, tau, sigma, iters)");
        check(cp_iterate(h, scalar(prhs[2], "theta"), scalar(prhs[3], "tau"), scalar(prhs[4], "sigma"), (int32_t)scalar(prhs[5], "iters")));
        check(cp_sync(h));
    } else {
        snprintf(g_msg, sizeof g_msg, "cp_mex: unknown command '%s'", cmd);
        mexErrMsgIdAndTxt("cpengine:badArg", "%s", g_msg);
    }
}

}   // namespace

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("cpengine:badArg", "cp_mex('command', ...)");
    char cmd[48];
    if (mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("cpengine:badArg", "cp_mex: command string too long");
    try {
        dispatch(cmd, nlhs, plhs, nrhs, prhs);
    } catch (...) {
        mexErrMsgIdAndTxt("cpengine:error", "cp_mex: C++ exception in command '%s'", cmd);
    }
}
