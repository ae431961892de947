/*
 * cpengine.h -- C ABI of the MI355X-native Chambolle-Pock registration engine.
 *
 * This is the drop-in boundary for the primal-dual hot path of roland1993/MA: the
 * reference's MATLAB entry points keep their signatures and call these functions
 * through a MEX gateway.  Plain C, `int` status return
 * (0 = ok), no exceptions cross the boundary, no torch types.
 *
 * Conventions
 *   - host buffers are caller-owned, column-major (MATLAB layout), double;
 *     inputs are never written, outputs go into caller-allocated buffers;
 *   - device memory is owned by the handle and stays resident across CP
 *     iterations and across outer (re-linearisation) iterations;
 *   - a handle is not thread-safe: one handle per host thread;
 *   - every entry point is synchronous on return unless stated otherwise.
 *
 * Vector layouts (identical to the reference):
 *   pairwise models   x = [u_x(:); u_y(:)] (2mn)  or  u(:) (mn, denoise)
 *                     y = [Dx u_x; Dy u_x; Dx u_y; Dy u_y] (4mn) or [Dx u; Dy u] (2mn)
 *                     (PDO/test_TV_L1_registration.m:67-79)
 *   mf_nn models      x = [u; L], u = per image [u_x,i(:); u_y,i(:)], L = k stacked images (3kmn)
 *                     y = [y1 (kmn); y2 (4kmn TV | 2kmn curvature); y3 (kmn)]
 *                     (NNE/mf_nn_tv_registration_no_ref_ml.m:150,246-248)
 */
#ifndef CPENGINE_H
#define CPENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cp_engine *cp_handle;

/* model = which (F, G, K) triple of the reference the handle solves */
enum {
    CP_MODEL_TV_L1_DENOISE = 1, /* F=TV_denoise,      G=SAD_denoise       PDO/test_TV_L1_denoising.m:58-59   */
    CP_MODEL_TV_L1_REG     = 2, /* F=TV_registration, G=SAD_registration  PDO/test_TV_L1_registration.m:84-85 */
    CP_MODEL_TV_L2_REG     = 3, /* F=TV_registration, G=SSD_registration  PDO/test_TV_L2_registration.m:84-85 */
    CP_MODEL_MF_NN_TV      = 4, /* NNE/mf_nn_tv_registration_no_ref_ml.m:183-203,235-346          */
    CP_MODEL_MF_NN_CURV    = 5, /* NNE/mf_nn_curvature_registration_no_ref_ml.m:182-202,230-332   */
    CP_MODEL_VAR_TV        = 6, /* NNE/var_tv_registration_no_ref_ml.m:154-176,201-260: x = u (2K fields),
                                   y = [y1 (K); y2 (4K)], histories have 6 columns, b is stored centred (B*b) */
    CP_MODEL_VAR_CURV      = 7  /* NNE/var_curvature_registration_no_ref_ml.m:116-175,199-252: y2 has 2K fields */
};

/* boundary condition strings of the reference */
enum {
    CP_BC_NEUMANN   = 0, /* NNE/finite_difference_operator.m:33, NNE/discrete_laplacian.m:25-27 */
    CP_BC_LINEAR    = 1, /* NNE/finite_difference_operator.m:31 (last row [-1 1], unscaled)      */
    CP_BC_DIRICHLET = 2  /* NNE/discrete_laplacian.m:28                                          */
};

enum { CP_STOP_MAXITER = 0, CP_STOP_TOLERANCE = 1 }; /* PDO/chambolle_pock.m:137,145-153 */

/* status codes; cp_last_error() holds the text (message ids cpengine:<name>) */
enum {
    CP_OK = 0,
    CP_ERR_BAD_ARG = 1,      /* cpengine:badArg  */
    CP_ERR_BAD_BC = 2,       /* cpengine:badBC   -- 'Unknown boundary condition!' finite_difference_operator.m:35 */
    CP_ERR_BAD_STATE = 3,    /* cpengine:badState (e.g. solve before linearize) */
    CP_ERR_DEVICE = 4,       /* cpengine:device  (HIP runtime error) */
    CP_ERR_UNSUPPORTED = 5   /* cpengine:unsupported */
};

/* CP_FLAG_VAR_SSD_SWAP: the single-level driver NNE/var_tv_registration_no_ref.m:173 calls the prox of the data
 * term as SSD(y1, b, sigma, 1, flag) (lambda and tau exchanged) while its value calls (:192) use (1, sigma); set
 * this flag to reproduce that driver, leave it clear for the *_ml drivers. */
enum cp_flag { CP_FLAG_VAR_SSD_SWAP = 1,
               /* fp32 mode (every model; row slabs too): state (x, y)
                * and linearisation (T(u0), grad T, b) are stored as float on the device -- half the HBM traffic
                * per iteration; arithmetic, reductions, eigensolves, the power iteration and histories stay fp64. The host interface is unchanged (double buffers in and out). Tolerance vs the fp64 mode:
                * 1e-4 relative on the displacement field and the energies (north_star). */
               CP_FLAG_FP32 = 2,
               /* CP_MODEL_MF_NN_TV only: the single-level driver NNE/mf_nn_tv_registration_no_ref.m composes F as
                * F1 = SAD(y1, b, 1, sigma) (:206,230) and F2 = sum_j norm21(y2_j, mu, sigma) (:209-216,232-241) --
                * weights 1 and mu without prod(h_grid), and mu*||.||_{2,1} (NNE/norm21.m) in place of the pointwise
                * 2x2 nuclear norm of the _ml driver (NNE/mf_nn_tv_registration_no_ref_ml.m:263-264,289-290). */
               CP_FLAG_MFNN_NORM21 = 4 };

typedef struct cp_problem_desc {
    int32_t model;      /* CP_MODEL_*                                                        */
    int32_t m, n;       /* image rows, columns (column-major, rows contiguous)               */
    int32_t k;          /* number of images: mf_nn models 1..32 (the reference has no limit: NNE/nuclear_norm_constraint.m:34-37;
                         * 1..16 run the fused kernels, 17..32 the general path -- one GPU, fp64 storage), variance models
                         * 1..64; 1 for pairwise */
    int32_t bc;         /* CP_BC_* ; pairwise models always use neumann (script form)        */
    int32_t device;     /* HIP device ordinal                                                */
    int32_t batch;      /* number of independent problems of this shape in the handle (>=1)  */
    int32_t flags;      /* CP_FLAG_* (0 for the reference's current drivers)                 */
    double  h[2];       /* grid spacing h_grid = [h1, h2]                                    */
    double  omega[4];   /* image region [w1,w2]x[w3,w4]; all zeros -> [0, m*h1, 0, n*h2]     */
    double  lambda;     /* data-term weight of the pairwise models                           */
    double  mu;         /* regulariser weight of the mf_nn models (optPara.mu)               */
    /* row-slab decomposition along j (image columns); single GPU: n_global = n, j_offset = 0 */
    int32_t n_global;   /* total number of image columns over all slabs                      */
    int32_t j_offset;   /* global index of this slab's first column                          */
} cp_problem_desc;

typedef struct cp_result {
    int32_t iterations;   /* i at exit of the loop, PDO/chambolle_pock.m:134-193            */
    int32_t stop_reason;  /* CP_STOP_*                                                       */
    int32_t rows;         /* history rows = iterations + 1                                   */
    int32_t cols;         /* history columns (5 pairwise, 7 mf_nn), PDO/chambolle_pock.m:224-228 */
} cp_result;

/* --- life cycle ---------------------------------------------------------------------- */
int cp_create(const cp_problem_desc *desc, cp_handle *out);
int cp_destroy(cp_handle h);
const char *cp_last_error(void);
/* version string, also proves the library loads without touching a GPU */
const char *cp_version(void);

/* --- data in --------------------------------------------------------------------------
 * cp_set_images: pairwise registration: a = template T, b = reference R (m x n each, already
 *   normalised as PDO/test_TV_L1_registration.m:57-58); denoise: a = noisy image, b = NULL;
 *   mf_nn: a = k stacked m x n images at this level (already normalised), b = NULL.
 *   For batch > 1 the problems are stacked along the last dimension. In slab mode the FULL
 *   images (n_global columns) are passed on every rank (warps need no communication).      */
int cp_set_images(cp_handle h, const double *a, const double *b);
/* start vectors x0 / y0 (chambolle_pock.m:82-84); NULL = zeros                             */
int cp_set_primal(cp_handle h, const double *x0);
int cp_set_dual(cp_handle h, const double *y0);

/* --- linearisation  (replaces the evaluate_displacement call inside G / the driver) -----
 * Warps the template(s) at u0 (NULL = the handle's current primal u, i.e. the warm start of
 * the outer loop, PDO/test_TV_L1_registration.m:119-120 / NNE/mf_nn_tv...:164-169), stores
 * grad T(u0) and b = T(u0) - grad T * u0 [- R].
 * IRE/evaluate_displacement.m:38-49, IRE/bilinear_interpolation.m:29-105.                   */
int cp_linearize(cp_handle h, const double *u0);
/* copies of the linearisation for unit parity: T(u0) (m*n*k), grad (m*n x 2 per image), b  */
int cp_get_linearization(cp_handle h, double *T_u, double *grad, double *b);
/* the operators' data arguments given directly instead of through a warp: grad = g and b = the I / b argument of
 * NNE/SAD.m:1 (SAD(L, I, mu, sigma, flag)), PDO/SAD_registration.m:40 / PDO/SSD_registration.m:44 (b = T(u0) -
 * grad T * u0 - R), NNE/mf_nn_tv...:167-168; T_u (may be NULL) only feeds cp_singular_values(0). Layout as returned
 * by cp_get_linearization; the variance models centre b (B*b, NNE/var_tv...:153-154) on the device as cp_linearize
 * does. Full-domain handles, every model but the denoiser. Lets a caller evaluate the operators on chosen inputs
 * (unit parity of a8-a18) and re-use a linearisation computed elsewhere.                  */
int cp_set_linearization(cp_handle h, const double *T_u, const double *grad, const double *b);

/* [img_u, dimg_u] = evaluate_displacement(img, h, u, omega, s_grid)  (IRE/evaluate_displacement.m:1-50) on its own, in
 * full generality: img is m x n with spacing h; the k x l cell-centred grid over omega (NULL: [0, m*h1, 0, n*h2], :33-36)
 * is displaced by u ((k*l) x 2, column-major; NULL: zero) and img interpolated there (bilinear, zero padding). k x l need
 * not equal m x n: this is the reference's resampling facility (s_grid, :27-31; e.g. an upsampled copy of an image with
 * u = NULL). img_u: k*l doubles; grad: (k*l) x 2 = the two diagonals of dimg_u (:47-48), or NULL. Host buffers in and
 * out; stateless (allocates and frees its own device buffers). */
int cp_evaluate_displacement(int32_t device, const double *img, int32_t m, int32_t n, const double *h, const double *u,
                             const double *omega, int32_t k, int32_t l, double *img_u, double *grad);

/* --- soft nuclear norm and the landscape experiment (NNE/nuclear_norm.m, NNE/test_nuclear_norm_measure.m) -------------
 * [res1, res2, res3] = nuclear_norm(L, numImg, tau, mu, conjugate_flag)  (NNE/nuclear_norm.m:1-2): L is npix x k, column
 * major, host memory. conjugate_flag 0: res1 = mu*||L||_*, res2 = 0, prox = U max(S - mu*tau, 0) V' (:43-55);
 * 1: res1 = 0, res2 = max(max(svd(L/mu)) - 1, 0), prox by Moreau's identity exactly as :78-80 writes it. prox may be NULL
 * (nargout 2). Singular values come from the eigenvalues of the k x k Gram matrix: values below sqrt(eps)*s_max are noise. */
int cp_nuclear_norm(int32_t device, const double *L, int64_t npix, int32_t k, double tau, double mu, int32_t conjugate_flag,
                    double *res1, double *res2, double *prox);
/* The L-only Chambolle-Pock problems of NNE/test_nuclear_norm_measure.m:122-243, solved by PDO/chambolle_pock.m:75-228:
 *   version 1: min_L mu*||L||_* + ||L - I||_1       K = identity, F = SAD(., I, 1, sigma), G = nuclear_norm(., k, tau, mu)  (:216-217)
 *   version 2: min_L delta{||L||_* <= nu} + ||L - I||_1           G = nuclear_norm_constraint(., k, tau, nu)               (:222-226)
 *   version 3: min_L delta{||B*L||_* <= nu} + ||L - I||_1   K = B, F = nuclear_norm_constraint(., k, tau, nu), G = SAD     (:231-236)
 * param = mu (1) or nu (2, 3). The script's calls SAD(L, I(:), sigma, c_flag) predate the mu argument of NNE/SAD.m:1; the
 * engine computes what they mean, SAD(L, I, 1, sigma, c_flag). I, L0, P0 (NULL: zeros, :132-133), L_star, P_star: npix x k
 * column major, host memory. Histories: caller buffers of (max_iter+1)*5 doubles, filled column major with res->rows rows:
 * [p, F, G, F_con, G_con] and [q, F*, G*, F*_con, G*_con] (:225-228). Stateless (allocates and frees its device buffers). */
int cp_nn_measure(int32_t device, int32_t version, const double *I, int64_t npix, int32_t k, double param, const double *L0,
                  const double *P0, double theta, double tau, double sigma, int32_t max_iter, double tol, double *L_star,
                  double *P_star, double *primal_history, double *dual_history, cp_result *res);

/* --- evaluation utilities (NNE/landmark_transform.m, NNE/landmark_accuracy.m) ------------------
 * LM_transformed = landmark_transform(LM, u, omega)  (NNE/landmark_transform.m:1-67): LM is num_lm x 2 (column-major,
 * host), u the m x n x 2 displacement field (host), omega the image region. For every landmark y the point x with
 * x + u(x) = y: start at the grid point whose displaced position is closest to y (:38-45, first minimum), at most 100
 * fixed-point steps x <- x + (y - F(x)) with the reference's literal termination expression (:50,:54) and its fallback
 * to the initial guess (:55-57). failed (may be NULL): 1 where the reference would print 'Fixed-point iteration
 * failed!'. One workgroup per landmark; stateless (allocates and frees its device buffers). */
int cp_landmark_transform(int32_t device, const double *LM, int32_t num_lm, const double *u, int32_t m, int32_t n,
                          const double *omega, double *LM_transformed, int32_t *failed);
/* LM_acc = landmark_accuracy(LM)  (NNE/landmark_accuracy.m:1-28): LM = the k cells stacked, num_lm x 2 x k column-major;
 * LM_acc(l) = mean over the k images of the distance of landmark l to its mean position. */
int cp_landmark_accuracy(int32_t device, const double *LM, int32_t num_lm, int32_t k, double *LM_acc);

/* --- mf_nn only --------------------------------------------------------------------------
 * radius nu of the nuclear-norm ball (NNE/mf_nn_tv...:171-180 keeps the schedule on the host) */
int cp_set_nu(cp_handle h, double nu);
/* singular values (descending, k of them) of reshape(B*field, mn, k):
 *   which = 0: field = T(u0) of the last cp_linearize   (nu_0, NNE/mf_nn_tv...:173-175)
 *   which = 1: field = L part of the current primal x    (SV_history, :214-215)             */
int cp_singular_values(cp_handle h, int which, double *sv);
/* power iteration of NNE/matrix_norm.m:15-48 on the current block operator A. x0 = start
 * vector (length 3kmn) or NULL for the engine's deterministic start
 * x_i = ((i*2654435761) mod 2^32)/2^31 - 1. Returns e(end) and the number of iterations.    */
int cp_norm_estimate(cp_handle h, const double *x0, double *est, int32_t *iters);
/* y = A*x and x = A'*y for the handle's block operator (mf_nn: NNE/mf_nn_tv...:183-188, [-grad T, I; D | Lap, 0;
 * 0, B]; var: NNE/var_tv...:154-161) -- the products of PDO/chambolle_pock.m:162,165 on caller-supplied vectors
 * (x: 3kmn or 2kmn, y: the dual length). grad T = 0 until the first linearisation, so that D, D', the Laplacian and
 * B (NNE/finite_difference_operator.m:29-50, NNE/discrete_laplacian.m:23-46, NNE/mean_free_operator.m:20) can be
 * applied on their own. fp64 in every storage mode; the solver state (x, y) is left untouched.                  */
int cp_apply_A(cp_handle h, const double *x, double *y);
int cp_apply_At(cp_handle h, const double *y, double *x);

/* --- solve: PDO/chambolle_pock.m:54-228 for the handle's (F, G, K) ----------------------- */
int cp_solve(cp_handle h, double theta, double tau, double sigma, int32_t max_iter, double tol,
             cp_result *res);
/* primal_history / dual_history of the last solve, rows x cols column-major
 * (PDO/chambolle_pock.m:224-228)                                                            */
int cp_get_history(cp_handle h, double *primal_history, double *dual_history);

/* result record of problem `problem` of a batch after the last cp_solve (0 for single problems)  */
int cp_get_result(cp_handle h, int32_t problem, cp_result *res);

/* --- data out ------------------------------------------------------------------------------ */
int cp_get_primal(cp_handle h, double *x);
int cp_get_dual(cp_handle h, double *y);
/* the handle's template image(s) at this level (after cp_set_images / cp_downsample)           */
int cp_get_images(cp_handle h, double *a);

/* --- multi-level -----------------------------------------------------------------------------
 * cp_prolong: x_fine = kron(x_coarse, ones(2)) cropped, all fields, times `factor`
 * (NNE/mf_nn_tv...:113-119, factor 1; var drivers use 2, NNE/var_tv...:112); the dual of
 * `fine` is reset to zero (:122).
 * cp_downsample: 2x2 block mean with zero padding past the end, odd samples (:75-76); writes the
 * images of `coarse` from those of `fine`.                                                   */
int cp_prolong(cp_handle coarse, cp_handle fine, double factor);
int cp_downsample(cp_handle fine, cp_handle coarse);

/* --- benchmarking hooks (no counterpart in the reference) -----------------------------------
 * cp_iterate runs `iters` full CP iterations (dual pass, eigensolve, primal pass, history row)
 * asynchronously on the handle's stream with the stop rule disabled, without resetting the
 * state; cp_sync waits for them. cp_kernel_time_ms returns the accumulated HIP-event time and
 * launch count of the named pass ("dual", "primal", "eig", "fin") since the last call.       */
int cp_iterate(cp_handle h, double theta, double tau, double sigma, int32_t iters);
/* The same in three parts, for streaming callers and steady-state measurements:
 *   cp_iterate_begin  start-up of a run of at most max_iters iterations (x_old = x0; NN-TV: the first dual step),
 *   cp_iterate_steps  `iters` more iterations of PDO/chambolle_pock.m:156-179 and nothing else (may be called repeatedly),
 *   cp_iterate_end    brings the state back to its canonical form (projected y3, mean-free u) for the accessors.
 * Between begin and end every entry point that touches the state fails with CP_ERR_BAD_STATE.      */
int cp_iterate_begin(cp_handle h, double theta, double tau, double sigma, int32_t max_iters);
int cp_iterate_steps(cp_handle h, int32_t iters);
int cp_iterate_end(cp_handle h);
int cp_sync(cp_handle h);
int cp_kernel_time_ms(cp_handle h, const char *pass, double *ms, int64_t *launches);
/* diagnostics of the field kernels (no counterpart in the reference):
 *   cp_dominant_kernel   the name, as rocprofv3 prints it, of the kernel the handle's steady-state iteration spends its time
 *                        in (template arguments as instantiated for this handle: storage type, boundary rows, K extent, waves
 *                        per workgroup, ...), NUL-terminated into buf[len]; bench.py keys committed PMC traffic by it.
 *   cp_handover_fallbacks  number of times, since cp_create, that a wave of an 8-wave kernel gave up on the LDS hand-over
 *                        between neighbouring waves (time-out) and loaded the neighbour position itself. Results are
 *                        correct either way; anything but 0 is a performance bug (tests and bench.py assert 0). */
int cp_dominant_kernel(cp_handle h, char *buf, int32_t len);
int cp_handover_fallbacks(cp_handle h, int64_t *count);
/* device-to-device streaming copy rate of this GPU in GB/s (read + write bytes / time, best of `repeats` passes over
 * two buffers of `bytes` each, 8 or 16 bytes per lane, nontemporal): the practical HBM ceiling next to the data sheet's
 * peak. Allocates and frees its own buffers. */
int cp_stream_copy_gbps(int32_t device, int64_t bytes, int32_t lane_bytes, int32_t repeats, double *gbps);

/* --- row-slab multi-GPU hooks (no counterpart in the reference; NN-TV model) -----------------------
 * One process per GPU, each with a handle created with n = owned columns, n_global, j_offset. The
 * library launches kernels and packs/unpacks halo columns; the CALLER moves the packed buffers and the
 * reduced partials between ranks (RCCL through torch.distributed, see ma_amd/slab_mg.py). All device
 * pointers passed in are caller-owned device memory; kernels run on the stream given to cp_set_stream
 * (e.g. torch's current stream), so the caller's collectives on that stream are ordered with them.   */
enum { CP_BUF_X0 = 0, CP_BUF_X1 = 1, CP_BUF_P0 = 2, CP_BUF_P1 = 3, CP_BUF_XTMP = 4 };
enum { CP_SEL_ALL = 0, CP_SEL_X_U = 1, CP_SEL_P_Y12 = 2, CP_SEL_P_DY = 3,
       CP_SEL_P_Y2 = 4,     /* all y2 fields (4K TV, 2K curvature): both directions */
       /* the same fields of a VECTOR of the power iteration (CP_BUF_X1 / CP_BUF_P1 / CP_BUF_XTMP between cp_mg_power_start
        * and the next solve): doubles in every storage mode, where the state selections above move floats in the fp32 mode */
       CP_SEL_VEC_X_U = 5, CP_SEL_VEC_P_DY = 6, CP_SEL_VEC_P_Y2 = 7 };
enum { CP_STAGE_DUAL0 = 0,   /* dual step 1 alone (first kernel of a solve) + local reduction -> red      */
       CP_STAGE_EIG0 = 1,    /* eigensolve on the gathered partials of DUAL0                               */
       CP_STAGE_FUSED = 2,   /* k_mf_fused_tv for iteration n + local reduction -> red                     */
       CP_STAGE_EFUSED = 3,  /* k_mf_efused for iteration n on the gathered partials                       */
       CP_STAGE_APPLY_A = 4, /* power iteration: P1 = A * (scale * X1)                                     */
       CP_STAGE_APPLY_AT = 5, /* power iteration: XTMP = A' * P1, local sum of squares -> red[0]; swaps X1/XTMP */
       /* split passes (NN-curvature; also NN-TV): iteration n = DUAL -> gather -> EIG -> halo(y2) -> PRIMAL -> gather
        * -> FIN -> halo(u)                                                                                    */
       CP_STAGE_SPLIT_DUAL = 6, CP_STAGE_SPLIT_EIG = 7, CP_STAGE_SPLIT_PRIMAL = 8, CP_STAGE_SPLIT_FIN = 9,
       /* history row 0 = primal_objective(x0), dual_objective(y0) (PDO/chambolle_pock.m:89-106) on slabs:
        * ROW0_DUAL -> gather -> ROW0_EIG, ROW0_PRIMAL -> gather -> ROW0_FIN (monitor passes: the state is not touched) */
       CP_STAGE_ROW0_DUAL = 10, CP_STAGE_ROW0_EIG = 11, CP_STAGE_ROW0_PRIMAL = 12, CP_STAGE_ROW0_FIN = 13,
       /* NN-curvature in two field passes (cp_mfnn_curv.cuh): iteration n = CURV_P -> gather -> EFUSED, halo(u_n, both
        * directions) -> CURV_Y2 -> gather -> CURV_FIN (stop rule), halo(y2_{n+1}, both directions) */
       CP_STAGE_CURV_P = 14, CP_STAGE_CURV_Y2 = 15, CP_STAGE_CURV_FIN = 16 };
int cp_set_stream(cp_handle h, void *hip_stream);          /* NULL = the handle's own stream            */
int64_t cp_mg_partial_len(void);                           /* doubles per reduced partial vector        */
int64_t cp_mg_halo_len(cp_handle h, int sel);              /* doubles the message buffer of one packed halo column needs
                                                              (fp32 mode: the state selections pack floats, two per double) */
/* column = 0: first owned column, 1: last owned column -> dev_out                                      */
int cp_mg_halo_pack(cp_handle h, int buf, int sel, int column, double *dev_out);
/* side = 0: left halo column, 1: right halo column <- dev_in                                           */
int cp_mg_halo_unpack(cp_handle h, int buf, int sel, int side, const double *dev_in);
/* the whole per-iteration halo of the fused NN-TV path in one launch: pack [u of xbuf | y1,y2 of pbuf] of the first
 * owned column into dev_to_left (7*k*m doubles) and the Dy parts of y2 of the last owned column into dev_to_right
 * (2*k*m doubles); unpack a right neighbour's dev_to_left / a left neighbour's dev_to_right into the halo columns.
 * NULL skips a side (ends of the chain). Same layout as CP_SEL_X_U + CP_SEL_P_Y12 / CP_SEL_P_DY back to back.
 * (No counterpart in the reference: its matrices are never partitioned.)                                   */
int cp_mg_state_pack(cp_handle h, int xbuf, int pbuf, double *dev_to_left, double *dev_to_right);
int cp_mg_state_unpack(cp_handle h, int xbuf, int pbuf, const double *dev_from_right, const double *dev_from_left);
/* max_iters: capacity of the history (rows 0..max_iters) for this run */
int cp_mg_begin(cp_handle h, double theta, double tau, double sigma, int32_t max_iters);
/* dev_buf: stage output (own reduced partials) or input (world gathered partial vectors)               */
int cp_mg_stage(cp_handle h, int stage, int32_t n, double *dev_buf, int32_t world, double scale,
                int32_t check_stop, double tol);
int cp_mg_end(cp_handle h, int32_t iterations);
/* stop flag / iteration of the stop rule after the last EFUSED stage (synchronises)                    */
int cp_mg_stop_state(cp_handle h, int32_t *stopped, int32_t *stop_iter);
/* local Gram matrix (16x16, row-major, host) of (B*field)*V over the owned pixels; which as in
 * cp_singular_values, V = 16x16 row-major host matrix or NULL for the identity                         */
int cp_mg_gram(cp_handle h, int which, const double *V, double *gram);
/* power iteration start: fills X1 with the deterministic start vector (global indices), zeroes XTMP    */
int cp_mg_power_start(cp_handle h, double *local_sumsq);

/* --- allocation-free entry point -------------------------------------------------------------------
 * No counterpart in the reference. For callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc. A variant of cp_singular_values(h, 0, sv): singular values
 * (descending, k of them) of reshape(B*field, m*n, k), B = NNE/mean_free_operator.m:20 -- the quantity behind nu_0 and
 * SV_history (NNE/mf_nn_tv_registration_no_ref_ml.m:173-175,214-215) -- for a caller-owned DEVICE field of k images,
 * m x n each, column-major, stacked (no halo columns), written to the caller-owned DEVICE buffer dev_sv (k doubles).
 * Same kernels and arithmetic as cp_singular_values; the per-workgroup partial sums and the 16 x 16 work matrices
 * live in a static device array of the library (3.3 MB per GPU, held until the process ends) instead of a handle.
 * Calls no hipMalloc / hipFree and no allocating routine; enqueues on hip_stream (NULL: the default stream) and
 * returns WITHOUT synchronising -- dev_sv is valid once the stream has reached that point.
 * Limits: 1 <= k <= 16, m*n <= 2^30; larger inputs are rejected with CP_ERR_UNSUPPORTED.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (different streams or threads) race.
 * The caller's current device is left as it was found. The FIRST call per device looks up the array's device address
 * (hipGetSymbolAddress) and must therefore happen outside a stream capture; every later call only enqueues kernels and
 * one device-to-device copy. */
int cp_singular_values_static(int32_t device, const double *dev_field, int32_t m, int32_t n, int32_t k,
                              double *dev_sv, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* CPENGINE_H */
