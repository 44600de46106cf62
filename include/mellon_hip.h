/*
 * mellon_hip.h -- C ABI of libmellon_hip.so, the MI355X (gfx950) sparse-GP density core.
 *
 * The reference (settylab/Mellon v1.7.1) has NO native code: its hot path is Python on
 * JAX/XLA:CPU.  Each entry point below replaces the arithmetic of the reference function(s)
 * cited next to it (paths relative to the reference checkout), so that the reference's
 * estimator classes -- or this repo's mirror of them, mellon_amd/ -- can bind it with ctypes.
 * INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *   - all matrices are row-major float64; sizes are int64_t, feature counts int32_t;
 *   - every data pointer may be a HOST pointer or a DEVICE pointer obtained from mln_malloc():
 *     the library detects which (hipPointerGetAttributes) and stages host buffers itself;
 *   - the caller owns every buffer it passes; the library owns what sits behind mln_ctx /
 *     mln_fit handles; no pointer is retained after a call returns except inside a handle;
 *   - every function returns an mln_status (0 = ok); no C++ exception crosses the ABI;
 *     mln_last_error() gives the text.  MLN_ERR_NOT_PD maps to the reference's
 *     ValueError("Covariance not positively definite with jitter=...") (decomposition.py:116-122);
 *   - a mln_ctx is bound to one GPU and is not thread-safe; calls are synchronous on return;
 *   - there is no CPU fallback: without a usable gfx950 device mln_ctx_create fails.
 */
#ifndef MELLON_HIP_H
#define MELLON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mln_ctx mln_ctx;
typedef struct mln_fit mln_fit;

typedef enum {
  MLN_OK = 0,
  MLN_ERR_NOT_PD = 1,      /* non-positive / NaN pivot in a Cholesky factorisation          */
  MLN_ERR_SHAPE = 2,       /* inconsistent sizes                                            */
  MLN_ERR_HIP = 3,         /* HIP runtime failure (text in mln_last_error)                  */
  MLN_ERR_RCCL = 4,        /* RCCL failure                                                  */
  MLN_ERR_ARG = 5,         /* null pointer / bad enum / unsupported descriptor              */
  MLN_ERR_UNSUPPORTED = 6, /* valid request this build cannot serve (e.g. m too large)      */
  MLN_ERR_NOCONV = 7       /* an iterative kernel (Jacobi eigensolver) did not converge     */
} mln_status;

/* ---- kernel-plugin surface: mellon/base_cov.py:17-497 + mellon/cov.py ------------------- */
typedef enum {
  MLN_K_MATERN32 = 1,    /* cov.py:62-66   */
  MLN_K_MATERN52 = 2,    /* cov.py:157-161 */
  MLN_K_EXPQUAD = 3,     /* cov.py:255-259 */
  MLN_K_EXPONENTIAL = 4, /* cov.py:352-356 */
  MLN_K_RATQUAD = 5,     /* cov.py:453-457 */
  MLN_K_LINEAR = 6,      /* cov.py:551-556 */
  MLN_K_DISTANCE = 7     /* util.py:351-366: the pairwise distance itself, dist / ls (values only, no gradients) */
} mln_kind;

/* A covariance tree (Covariance / Add / Mul / Pow, base_cov.py:301-453) is lowered to a postfix
 * program over leaves.  Each leaf carries the FINAL column indices it sees after every enclosing
 * `active_dims` selection has been composed (util.py:150-171).                                 */
typedef struct {
  int32_t kind;        /* mln_kind                                        */
  int32_t ndims;       /* number of active columns                        */
  double ls;           /* length scale                                    */
  double alpha;        /* RatQuad only                                    */
  const int32_t* dims; /* ndims column indices into the d input columns   */
} mln_leaf;

typedef enum { MLN_OP_LEAF = 0, MLN_OP_CONST = 1, MLN_OP_ADD = 2, MLN_OP_MUL = 3, MLN_OP_POW = 4 } mln_op;

typedef struct {
  int32_t op;   /* mln_op                                   */
  int32_t leaf; /* MLN_OP_LEAF: index into leaves           */
  double value; /* MLN_OP_CONST: the scalar                 */
} mln_tok;

#define MLN_MAX_LEAVES 4
#define MLN_MAX_TOKS 16
#define MLN_MAX_DIMS 256 /* total over all leaves */

typedef struct {
  int32_t n_leaves;
  int32_t n_toks;
  const mln_leaf* leaves;
  const mln_tok* toks;
} mln_kernel_desc;

/* ---- context / memory ------------------------------------------------------------------- */
int mln_ctx_create(int device, mln_ctx** out);
void mln_ctx_destroy(mln_ctx* ctx);
const char* mln_last_error(mln_ctx* ctx); /* ctx may be NULL: last error of the calling thread */
int mln_device_info(mln_ctx* ctx, char* name, int name_cap, int* n_cu, int64_t* mem_bytes);
int mln_device_count(int* count_out); /* GPUs visible to this process (0 and MLN_OK when there are none) */
int mln_synchronize(mln_ctx* ctx);

int mln_malloc(mln_ctx* ctx, int64_t bytes, void** dev_ptr);
int mln_free(mln_ctx* ctx, void* dev_ptr);
int mln_memcpy(mln_ctx* ctx, void* dst, const void* src, int64_t bytes); /* any host/device mix */
/* Page-lock a HOST array of the caller (hipHostRegister) so that the library's uploads from it are DMA transfers that run
 * under its kernels: the reference's timed region starts from X in host memory (density_estimator.py:542-581), and from
 * pageable memory ~11 ms of C3's 0.4 GB upload stay exposed.  Registration itself costs ~0.1 ms per MB: once per array,
 * not per fit.  The caller unregisters before freeing the array.  Both return MLN_ERR_HIP when the runtime refuses.   */
int mln_host_register(mln_ctx* ctx, const void* host_ptr, int64_t bytes);
int mln_host_unregister(mln_ctx* ctx, const void* host_ptr);
/* Internal buffers (the 8*n*m-byte factor, Gram partials, ...) are recycled by a caching
 * allocator so that repeated fits do not pay hipMalloc / page-mapping again; this returns all
 * cached blocks to the driver.                                                                */
int mln_release_cached_memory(void);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------------------------------
 * Cells (rows of x) are sharded; landmarks, Lp and z are replicated.  Collectives: all-reduce
 * of (loss, grad) per objective evaluation and of the m x m Ridge Gram once per fit.          */
#define MLN_UNIQUE_ID_BYTES 128
int mln_comm_unique_id(void* id_out /* MLN_UNIQUE_ID_BYTES */);
int mln_comm_init(mln_ctx* ctx, const void* id, int n_ranks, int rank);
int mln_comm_allreduce_sum(mln_ctx* ctx, double* buf, int64_t count); /* host or device buffer */
/* Loopback communicator: n_ranks contexts of ONE process (one host thread per rank; the contexts may all sit
 * on the same GPU) exchange through device memory and a host barrier, every rank summing the contributions in
 * rank order.  Same call sites, same results as RCCL -- it runs the N-rank sharded path on a single GPU.
 * Every rank must enter each collective; the group outlives the contexts attached to it.                       */
typedef struct mln_loopback mln_loopback;
int mln_loopback_create(int n_ranks /* <= 16 */, mln_loopback** out);
void mln_loopback_destroy(mln_loopback* group);
int mln_comm_init_loopback(mln_ctx* ctx, mln_loopback* group, int rank);
/* Host-staged collectives: the fall-back for a box where RCCL cannot be initialised, and the way to run the real
 * multi-PROCESS path on fewer GPUs than ranks (RCCL refuses two ranks on one device).  The library copies the buffer to
 * pinned host memory, calls `fn`, copies the result back, all in stream order.  op 0: all-reduce (sum over ranks, in rank
 * order) of buf[count] in place; op 1: buf[count] <- rank 0's; op 2: all-gather, buf[count] of every rank in rank order
 * into buf2[n_ranks * count].  fn returns 0 on success. */
typedef int (*mln_host_collective_fn)(void* user, int op, double* buf, double* buf2, int64_t count);
int mln_comm_init_host(mln_ctx* ctx, int n_ranks, int rank, mln_host_collective_fn fn, void* user);
void mln_loopback_abort(mln_loopback* group); /* a rank failed outside a collective: wake the others with MLN_ERR_RCCL */
/* What the communicator of `ctx` is and what it has carried (no counterpart in the reference, which has no distributed
 * code: this is how a multi-GPU bench line proves which transport ran -- SURVEY.md S8(e)).
 *   info[4]  (nullable): transport 0 none / 1 RCCL / 2 in-process loopback / 3 host-staged; the rank count and the rank
 *            AS THE TRANSPORT REPORTS THEM (ncclCommCount / ncclCommUserRank for RCCL; -1 if it cannot say); RCCL's
 *            version code (0: not loaded).
 *   stats[10] (nullable): calls, bytes of all-reduce | broadcast | all-gather since the last reset; the all-reduces of
 *            <= 64 KB among them (the per-evaluation [grad ; loss]); event-timed milliseconds of the LARGE all-reduces,
 *            of broadcasts + all-gathers, of the small all-reduces (0 unless timing is on; reading them synchronises).
 *   flags:   1 = timing on (a pair of stream events around every collective), 2 = off, 4 = reset counters afterwards. */
int mln_comm_info(mln_ctx* ctx, int32_t* info, double* stats, int32_t flags);

/* ---- a-1..a-3: K = cov(x, y)   (util.py:351-366 distance, cov.py k(), base_cov.py Add/Mul/Pow)
 * x: n x d, y: m x d, out: n x m.                                                              */
int mln_kernel_matrix(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n,
                      const double* y, int64_t m, int32_t d, double* out);

/* ---- feeding the path (S8f rank 1): exact Euclidean nearest-neighbour distance of each row of x
 * (n x d) among the rows of y (m x d), excluding the pair (i, i + self_offset) -- pass y = x and
 * self_offset = 0 for the estimator's nn_distances (parameters.py:408-433; the reference's
 * pynndescent search is approximate), or y = all cells and self_offset = shard start when sharded. */
/* cov(x, xu)^T cov(x, xu) (m x m), summed over the ranks' shards: the B^T B of the landmark leverage
 * diag(B M^-1 B^T), M = sigma^2 K_uu + B^T B + jitter I (conditional.py:593-601,660-685). */
int mln_kernel_gram(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local, int32_t d,
                    const double* xu, int64_t m, double* out /* m x m */);
int mln_nn_distances(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int32_t d,
                     int64_t self_offset, double* out /* n */);

/* k-means landmarks (S8f rank 1): k-means++ seeding + Lloyd sweeps on the device, the algorithm
 * family of sklearn.cluster.k_means(x, m, n_init=1, random_state) that parameters.compute_landmarks
 * calls (parameters.py:275-291).  Own RNG stream / summation order: centroids are equivalent in
 * quality, not bit-compatible with sklearn.  d <= 64.  centers: m x d.  The sweeps carry Hamerly's
 * distance bounds (same assignments, searched only where the bounds do not decide).  n_iter_out and
 * inertia_out may be NULL; the inertia is a full fp64 assignment of all cells to the final centres. */
int mln_kmeans(mln_ctx* ctx, const double* x, int64_t n, int32_t d, int64_t m, int64_t seed,
               int32_t max_iter, double tol, double* centers, int32_t* n_iter_out, double* inertia_out);
/* The same with sklearn's OWN seeding (sklearn.cluster._kmeans._kmeans_plusplus behind k_means(x, m, n_init=1,
 * random_state), parameters.py:275-291): the first centre is cell `first_id`, centre c is the best of n_local_trials
 * (= 2 + int(log m)) candidates drawn at uniforms[(c - 1) * n_local_trials + l] * current_pot on the cumulative sums of the
 * squared distances.  The caller draws first_id and the (m - 1) * n_local_trials uniforms with numpy's
 * RandomState(random_state) -- exactly the numbers sklearn consumes -- so the seeds are the cells sklearn picks (up to
 * summation order: see csrc/kmeans.hip) and the landmarks comparable with the reference's at any size.  Lloyd's sweeps
 * then run over all cells to sklearn's stopping rule; max_iter = 0 returns the seeds.  indices_out (m, may be NULL): the
 * seeded cells. */
int mln_kmeans_sklearn(mln_ctx* ctx, const double* x, int64_t n, int32_t d, int64_t m, int64_t first_id,
                       const double* uniforms, int32_t n_local_trials, int32_t max_iter, double tol, double* centers,
                       int64_t* indices_out, int32_t* n_iter_out, double* inertia_out);

/* ---- a-4/a-5: in-place lower Cholesky of (A + add_diag * I), A m x m symmetric (lower read).
 * decomposition.py:111-123 (`stabilize` util.py:269-293 + jnp.linalg.cholesky).  Strict upper
 * triangle of the result is zero.  MLN_ERR_NOT_PD on a non-positive or NaN pivot.              */
int mln_chol_lower(mln_ctx* ctx, double* A, int64_t m, double add_diag);

/* triangular solves with a lower factor Lf (m x m):  B (m x p) <- op(Lf)^-1 B.
 * trans = 0: Lf^-1 B (solve_triangular(L, B, lower=True)); trans = 1: Lf^-T B
 * (solve_triangular(L.T, B)), conditional.py:63-65,264,818.                                    */
int mln_trsm_lower(mln_ctx* ctx, const double* Lf, int64_t m, int32_t trans, double* B, int64_t p);

/* ---- fit handle: device-resident shard state --------------------------------------------------
 * mln_fit_prepare follows parameters.compute_Lp (parameters.py:648-714) and compute_L
 * (parameters.py:783-874):
 *   xu != NULL (sparse_cholesky): Lp = chol(cov(xu,xu) + max(sigma^2,jitter) I)  decomposition.py:111-123
 *                                 L  = cov(x,xu) Lp^-T                           decomposition.py:205-210
 *   xu == NULL (full):            Lp = chol(cov(x,x) + jitter I);  L = Lp        parameters.py:847-850
 * Lp_in (optional, m x m) skips the factorisation (the estimator's `Lp=` ctor argument).
 * x is THIS RANK's shard (n_local x d); with a communicator the full-GP branch is refused.    */
#define MLN_FIT_IMPLICIT 1 /* flags: keep K = cov(x,xu) in the n x m buffer and fold Lp^-T into the
                             m-vectors (L z = K (Lp^-T z), L^T v = Lp^-1 (K^T v)): no n x m triangular
                             solve; mln_fit_get_L materialises rows on demand; no Hessian diagonal.
                             For n_local * m >= 2^27 the kernel-matrix pass also keeps a 32-bit copy of K
                             (n_local x ld x 4 bytes: fixed point round(v 2^32) when the covariance is a stationary
                             kernel or a product of such, i.e. bounded by 1; fp32 otherwise) for the warm-up passes of
                             mln_map_solve, which finishes on the fp64 buffer at the same tolerances; environment
                             MELLON_AMD_MIXED=0 disables, MELLON_AMD_MIXED_MIN_ELEMS moves the threshold. */
int mln_fit_prepare(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                    int32_t d, const double* xu, int64_t m, double jitter, const double* Lp_in,
                    int32_t flags, mln_fit** out);
/* Adopt factors computed elsewhere (the estimator's `L=` / `Lp=` ctor arguments,
 * density_estimator.py:180-205): L is n_local x m; Lp (m x m) may be NULL, in which case the
 * predictor-weight entries are unavailable on this handle.                                      */
int mln_fit_from_L(mln_ctx* ctx, const double* L, int64_t n_local, int64_t m, const double* Lp,
                   mln_fit** out);
/* ---- kernel-plugin surface, user-defined kernels: a Covariance subclass whose k(x, y) is Python code
 * (base_cov.py:17-69: the ABC's only contract is `k`) cannot be lowered to an mln_kernel_desc.  The binding then
 * evaluates the USER'S function itself -- cov(xu, xu) once, cov(x_block, xu) in row blocks -- and hands the values
 * over; everything after the kernel matrix (Cholesky, triangular solves, Ridge, MAP solve, weights) is the same
 * device path as for the built-in kernels.  The same entries serve covariance trees too large for one device
 * program (MLN_MAX_LEAVES / MLN_MAX_TOKS), whose blocks the binding assembles on the device with mln_ewise.
 *   mln_fit_prepare_from_K  Kuu = cov(xu, xu) (m x m; jitter is added here; ignored when Lp_in is given);
 *                           flags: MLN_FIT_IMPLICIT as above; MLN_FIT_FULL: the full GP (n_local == m, L = Lp,
 *                           no n x m buffer, nothing to upload)
 *   mln_fit_set_K_rows      rows [row0, row0 + n_rows) of cov(x, xu), n_rows x m, host or device
 *   mln_fit_finish_K        after the last block: L = K Lp^-T unless implicit; the handle is then a normal fit     */
#define MLN_FIT_FULL 2
/* MLN_FIT_DEFER_LP (with MLN_FIT_IMPLICIT, no Lp given): Lp = chol(cov(xu, xu) + jitter I) is not factored inside
 * mln_fit_prepare but together with the preconditioner's matrix in mln_precond_build / mln_ridge_init (two
 * factorisations in one chain of launches), or by the first call that needs it; MLN_ERR_NOT_PD then comes from
 * that call, its message starting with "cov(xu, xu)". */
#define MLN_FIT_DEFER_LP 4
int mln_fit_prepare_from_K(mln_ctx* ctx, const double* Kuu, int64_t n_local, int64_t m, double jitter,
                           const double* Lp_in, int32_t flags, mln_fit** out);
int mln_fit_set_K_rows(mln_fit* fit, int64_t row0, int64_t n_rows, const double* K_rows);
int mln_fit_finish_K(mln_fit* fit);
/* C (M x N) = alpha op(A) op(B) + beta C on the fp64 matrix cores; row-major, host or device pointers.
 * ta = 0: A is M x K (lda >= K); ta = 1: A is stored K x M.  tb likewise.  The mean of a predictor with a
 * user-defined kernel is mu + cov(Xnew, centers) W with the kernel block evaluated by the user's k
 * (conditional.py:366-373,651-658,899-906).                                                                       */
int mln_gemm(mln_ctx* ctx, int32_t ta, int32_t tb, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
             int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
/* out[i] = a[i] op (b ? b[i] : scalar), op in {MLN_OP_ADD, MLN_OP_MUL, MLN_OP_POW} (base_cov.py:309-315,375-381,
 * 449-453 on whole blocks); out may alias a.  Device or host pointers.                                            */
int mln_ewise(mln_ctx* ctx, int32_t op, const double* a, const double* b, double scalar, double* out, int64_t count);

void mln_fit_destroy(mln_fit* fit);
int mln_fit_get_Lp(mln_fit* fit, double* out /* m x m */);
int mln_fit_get_L(mln_fit* fit, int64_t row0, int64_t n_rows, double* out /* n_rows x m */);
int mln_fit_rank(mln_fit* fit, int64_t* m_out); /* number of columns of L */

/* ---- analytic gradients ---------------------------------------------------------------------------
 * mln_kernel_grad: Covariance.k_grad (cov.py:68-100,163-202,261-299,358-396,459-499,558-596 and the
 * Add/Mul/Pow rules of base_cov.py:317-497): out[i][j][:] = d cov(x_i, y_j) / d y_j, zero on inactive
 * dims, with util.distance_grad's denominator (dist + 1e-12) (util.py:416-423).
 * mln_predict_gradient: Predictor.gradient (base_predictor.py:490-505 -> derivatives.gradient, the
 * jax.jacrev of `_mean`): out[i][:] = sum_j W[j] d cov(x_i, c_j) / d x_i, the exact derivative of the
 * mean mln_predict_mean evaluates (denominator dist); the n x m x d tensor is never formed.          */
int mln_kernel_grad(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n, const double* y,
                    int64_t m, int32_t d, double* out /* n x m x d */);
int mln_predict_gradient(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new,
                         int32_t d, const double* centers, int64_t m, const double* W /* m */,
                         double* out /* n_new x d */);
/* Hessian of the mean at every row of xnew, out: n_new x d x d (base_predictor.py:507-521: jacfwd(jacrev(_mean));
 * here the closed-form second derivatives of the kernels, contracted with the weights on the matrix cores).      */
int mln_predict_hessian(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new, int32_t d,
                        const double* centers, int64_t m, const double* W /* m */, double* out);


/* ---- Nystroem rank reduction (decomposition.py:23-76,126-171,213-266) ---------------------------
 * mln_eigh replaces jax.numpy.linalg.eigh at decomposition.py:50 (_eigendecomposition): A (m x m,
 * symmetrised as (A + A^T)/2 like jax's symmetrize_input default) -> w (m, ascending) and V (m x m
 * row-major, COLUMN j = eigenvector j; sign of each column is arbitrary as in LAPACK).  Block
 * one-sided Jacobi on the device; MLN_ERR_NOCONV if 40 sweeps do not orthogonalise.
 *
 * _modified_low_rank (:213-266: QR of C = cov(x,xu), eigh of W = cov(xu,xu)+sigma2 I, eigh of
 * R W^-1 R^T, L = Q V sqrt(S)) is computed through the identity  L L^T = B U_p U_p^T B^T  with
 * B = C Lp^-T (the factor of an explicit mln_fit_prepare handle, Lp Lp^T = W) and (S, U) the
 * eigenpairs of the m x m Gram B^T B = Lp^-1 C^T C Lp^-T  (same non-zero spectrum as R W^-1 R^T):
 *   mln_fit_gram_eigh : S (m, ascending) to the host, U kept in the handle   [all-reduce of the Gram]
 *   mln_fit_project   : new handle with L = B U[:, m-p:]  (n_local x p; columns ordered like the
 *                       reference's s[-p:], v[:, -p:]; column signs arbitrary)
 * The rank rule of _eigendecomposition (:51-76) is host logic on S (mellon_amd/decomposition.py).   */
int mln_eigh(mln_ctx* ctx, const double* A, int64_t m, double* w, double* V, int32_t* n_sweeps);
int mln_fit_gram_eigh(mln_fit* fit, double* w /* m */, int32_t* n_sweeps);
/* util.test_rank (util.py:429-483) = numpy.linalg.matrix_rank(L, rtol=tol): the number of singular values of L above
 * tol * the largest, i.e. of eigenvalues of L^T L (all cells, all ranks) above tol^2 * lambda_max -- counted by Sturm
 * sequences on the Householder-tridiagonalised Gram, no eigendecomposition.  sigma_max_out may be NULL. */
int mln_fit_gram_rank(mln_fit* fit, double tol, int64_t* rank_out, double* sigma_max_out);
int mln_fit_project(mln_fit* fit, int64_t p, mln_fit** out);

/* a-9: Ridge initial value  z0 = (L^T L + I)^-1 L^T target   (parameters.py:877-896;
 * sklearn Ridge(alpha=1, fit_intercept=False)).  target: n_local.  All-reduced over ranks.     */
int mln_ridge_init(mln_fit* fit, const double* target, double* z0 /* m */);

/* Preconditioned MAP variable (no counterpart in the reference, which runs L-BFGS-B on z directly):
 * with C C^T = L^T L + I -- the Ridge matrix above, which equals the MAP Hessian wherever
 * exp(f + V) = 1 -- the substitution z = C^-T u makes the strictly convex objective well conditioned,
 * so the same optimiser reaches the same unique optimum in ~10x fewer passes over L.
 *   mln_precond_build      factor C and C^-1 (done implicitly, from all cells, by mln_ridge_init).  Any SPD
 *                          matrix preconditions a strictly convex problem: estimating the Gram from every
 *                          row_stride-th cell (~8 m rows suffice) costs 1/row_stride of the n m^2 flops
 *   mln_precond_apply      mode 0: u = C^T z;  mode 1: z = C^-T u;  mode 2: g_u = C^-1 g_z   (host m-vectors)
 *   mln_objective_precond  loss(C^-T u) and its gradient in u; optionally also z = C^-T u      */
int mln_precond_build(mln_fit* fit, int64_t row_stride /* Gram from every row_stride-th cell; 1 = all */);
/* Cell-sharded fits: the global index of this shard's first cell (default 0).  The row subsample above takes the cells
 * whose GLOBAL index is a multiple of row_stride, so the preconditioner does not depend on the sharding.             */
int mln_fit_set_row_offset(mln_fit* fit, int64_t global_row0);
int mln_precond_apply(mln_fit* fit, int32_t mode, const double* in, double* out);
int mln_objective_precond(mln_fit* fit, const double* u, double* loss, double* grad_u /* m */,
                          double* z_out /* m or NULL */);

/* a-7: nearest-neighbour likelihood constants of this shard (inference.py:83-85), computed by
 * the caller from nn_distances and d:  V = d log r + c,  Vdr = log d + (d-1) log r + c.        */
int mln_fit_set_likelihood(mln_fit* fit, const double* V, const double* Vdr, double mu);

/* a-7 (+ a-14): loss(z) = 1/2 |z|^2 + (m/2) log 2pi - sum_i [f_i + Vdr_i - exp(f_i + V_i)],
 * f = L z + mu (inference.py:35-92,167-192), its gradient z + L^T (exp(f+V) - 1), and optionally
 * the diagonal of the Hessian 1 + sum_i L_ij^2 exp(f_i+V_i) (inference.py:291-338 in closed form).
 * One pass over L.  Sums are all-reduced over ranks; prior terms are added once.               */
int mln_objective(mln_fit* fit, const double* z, double* loss, double* grad /* m */,
                  double* hess_diag /* m or NULL */);

/* a-8: the MAP solve.  inference.minimize_lbfgsb (inference.py:272-288) is SciPy's L-BFGS-B without
 * bounds behind jaxopt.ScipyMinimize; this is the same limited-memory BFGS with the same stopping
 * tests (relative loss decrease <= ftol, max|grad| <= gtol, maxiter), run inside the library on the
 * preconditioned variable so that one evaluation = one device pass with no host framework in between.
 * status: 0 converged, 1 maxiter, 2 line search failed; + 4 when a solve that did NOT converge stopped with cells above the
 * likelihood cap (its loss / gradient are then the capped objective's, a minorant of inference.py:35-92's).   */
typedef struct {
  int32_t maxiter; /* 5000  */
  int32_t maxcor;  /* 10    L-BFGS memory (<= 64)    */
  int32_t maxls;   /* 30    line-search evaluations  */
  double ftol;     /* 1e-13 (SciPy default 2.2e-9 leaves the log-density ~5e-5 off the optimum) */
  double gtol;     /* 1e-7  on the preconditioned gradient */
} mln_solver_opts;
int mln_map_solve(mln_fit* fit, const double* z0, const mln_solver_opts* opts /* NULL = defaults */,
                  double* z_out /* m */, double* loss_out, int32_t* n_eval_out, int32_t* n_iter_out,
                  int32_t* status_out);

/* a-11: f = L z + mu on this shard (inference.py:51-69,341-354).                                */
int mln_transform(mln_fit* fit, const double* z, double mu, double* f_out /* n_local */);

/* a-12: predictor weights.
 *   sparse-Cholesky:  w = Lp^-T z                       conditional.py:818
 *   full:             w = Lp^-T Lp^-1 (y - mu)          conditional.py:263-264                 */
int mln_weights_cholesky(mln_fit* fit, const double* z, double* w /* m */);
int mln_weights_full(mln_fit* fit, const double* y, int64_t p, double mu, double* w /* m x p */);

/* a-12 (FunctionEstimator, scalar sigma): conditional.py:513-547 + _sparse_solve :57-66
 *   A = Lp^-1 cov(xu,x);  L_B = chol(A A^T / sigma^2 + I);
 *   W = Lp^-T L_B^-T L_B^-1 A (y - mu) / sigma^2          (m x p)
 * x, y are this rank's shard; A A^T and A y are all-reduced.                                   */
int mln_sparse_solve(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                     int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                     double sigma, double jitter, double* W /* m x p */);
/* Same solve, additionally returning the with_uncertainty state of the noisy landmark conditional
 * (conditional.py:571-577): Lp (m x m) and Cs = Lp L_B (m x m, lower), either may be NULL.          */
int mln_sparse_solve_factors(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                             int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                             double sigma, double jitter, double* W, double* Lp_out, double* Cs_out);

/* Noise models of the landmark conditional beyond one scalar (conditional.py:13-43,140-159,529-545):
 *   MLN_SIGMA_SCALAR      sigma[1]        the solve above
 *   MLN_SIGMA_PER_OUTPUT  sigma[p]        one noise level per output column ("per-gene" sigma, the vmap of
 *                                         conditional.py:529-545); A A^T and A (y - mu) are formed once; adjacent
 *                                         columns with equal sigma share one L_B, and beyond 32 such runs one
 *                                         eigendecomposition A A^T = U diag(lam) U^T serves every level:
 *                                         (A A^T / s^2 + I)^-1 = U diag(1 / (lam / s^2 + 1)) U^T
 *   MLN_SIGMA_PER_CELL    sigma[n_local]  element-wise standard deviation of this rank's cells
 *                                         (`sigma.shape == r.shape`, conditional.py:155-159):
 *                                         L_B L_B^T = A diag(1/sigma^2) A^T + I, c = L_B^-1 A ((y - mu) / sigma^2)   */
enum { MLN_SIGMA_SCALAR = 0, MLN_SIGMA_PER_OUTPUT = 1, MLN_SIGMA_PER_CELL = 2 };
int mln_sparse_solve_noise(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                           int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                           const double* sigma, int32_t sigma_kind, double jitter, double* W /* m x p */);

/* Full GP conditioned on p outputs with one noise level each (conditional.py:239-251; leverage :313-323,385-403;
 * HC3 residuals :330-333; variance weights :338-350) from ONE eigendecomposition K = U diag(lam) U^T of the n x n
 * kernel matrix: (K + (sigma_j^2 + jitter) I)^-1 = U diag(1 / (lam + sigma_j^2 + jitter)) U^T.
 * W: n x p.  leverage, corrected_r2, variance_W: n x p or NULL (each needs the one before it).                         */
int mln_full_conditional_noise(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n, int32_t d,
                               const double* y, int64_t p, double mu, const double* sigma /* p */, double jitter,
                               double* W, double* leverage, double* corrected_r2, double* variance_W);

/* Leverage of the landmark conditional for p noise levels at once (conditional.py:660-685 applied per level by the
 * vmap of :672-680):  out[i][j] = b_i^T (sigma_j^2 K_uu + B^T B + jitter I)^-1 b_i  with B = cov(x, xu) over the cells
 * of all ranks, b_i the rows of this rank, and K_uu = Lk Lk^T given by its lower factor Lk (m x m; the predictor's L).
 * One eigendecomposition of L^T L + jitter Lk^-1 Lk^-T (L = B Lk^-T) serves every level.                              */
int mln_landmark_leverage(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local, int32_t d,
                          const double* xu, int64_t m, const double* Lk, const double* sigma, int64_t p,
                          double jitter, double* out /* n_local x p */);

/* a-13: mean(Xnew) = mu + cov(Xnew, centers) W   (conditional.py:366-373,651-658,899-906).
 * centers: m x d (landmarks or, full GP, the training cells); W: m x p; out: n_new x p.
 * cov(Xnew, centers) is never materialised for p == 1.                                         */
int mln_predict_mean(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new,
                     int32_t d, const double* centers, int64_t m, const double* W, int64_t p,
                     double mu, double* out);

/* ---- diagnostics: measured rooflines of this device and the GEMM kernel in isolation ----------
 *   mln_diag_peak   what = 0: fp64 MFMA issue peak (TFLOP/s); 1: HBM streaming read, 4: streaming write (GB/s)
 *   mln_diag_dgemm  milliseconds per call of C = op(A) op(B) (same arguments as the internal GEMM) */
int mln_diag_peak(mln_ctx* ctx, int32_t what, int64_t bytes, double* result);
int mln_diag_dgemm(mln_ctx* ctx, int32_t ta, int32_t tb, int64_t M, int64_t N, int64_t K,
                   int32_t lower_only, int32_t split_k, int32_t reps, double* ms_out);
/* The GEMM's mixed-tile pipelined kernel against its single-size kernels on the same pseudo-random operands
 * (alpha = 0.75, the given beta; lower_only 0..3 and kmode 0..7 as the internal GEMM takes them: whole matrix / a triangle of
 * tiles, full / block-diagonal / triangular K ranges -- decomposition.py:111-123,205-210 and conditional.py:57-66 reach
 * it in every one of them).  out[0] = largest absolute difference (0 expected: the same summation order per element),
 * out[1] = largest absolute value.  any_size != 0: the mixed kernel also below the size where the policy selects it. */
int mln_diag_dgemm_compare(mln_ctx* ctx, int32_t ta, int32_t tb, int64_t M, int64_t N, int64_t K, int32_t lower_only,
                           int32_t kmode, double beta, int32_t any_size, double* out);
/* The integer Gram of the preconditioner in isolation (csrc/gram_i8.hip): out (m x m) = Q^T Q / 8355711^2 with
 * Q = round(A * 8355711), A rows x m with values in [0, 1] (host or device); terms below 2^-23 relative are dropped.
 * ms_out (may be NULL): milliseconds per call over `reps` calls.  Not a reference operation: the reference forms the
 * fp64 Gram of all cells (parameters.py:895-896); this one only ever feeds the preconditioner. */
int mln_diag_gram_i8(mln_ctx* ctx, const double* A, int64_t rows, int64_t m, double* out, int32_t reps, double* ms_out);
/* ms of: kernel-matrix pass alone, Gram GEMM alone, both on two streams, Cholesky alone, kernel matrix ||
 * Cholesky, kernel matrix || (Cholesky then Gram) -- the measurement behind the two-stream set-up phase. */
int mln_diag_overlap(mln_ctx* ctx, int64_t n, int64_t m, int32_t d, int64_t gram_rows, double* out /* 6 */);

/* ---- predictive uncertainty (S8f rank 2) ----------------------------------------------------------
 * covariance:       k(x*,x*) - A A^T with A = cov(x*, centers) Lf^-T      conditional.py:409-422,930-945
 * mean covariance:  (K W)(K W)^T with K = cov(x*, centers), W m x q        conditional.py:423-440,947-963
 * diag != 0: out has n_new entries (the variances); diag == 0: out is n_new x n_new (n_new <= 32768). */
int mln_predict_covariance(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new,
                           int32_t d, const double* centers, int64_t m, const double* Lf, int32_t diag,
                           double* out);
int mln_predict_mean_covariance(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new,
                                int32_t d, const double* centers, int64_t m, const double* W, int64_t q,
                                int32_t diag, double* out);

/* wall-clock seconds of the stages of the last mln_fit_prepare / mln_ridge_init and counters
 * of mln_objective: [0] kernel matrix, [1] cholesky, [2] trsm, [3] ridge gram, [4] ridge solve,
 * [5] objective kernel time (sum, HIP events), [6] objective launches, [7] bytes of L streamed
 * per objective launch; [8] / [9] the same time / launch count for the warm-up passes of mln_map_solve on the
 * 32-bit copy (mixed precision: they stream 4 bytes per element, half of [7]); [10] the format of that copy:
 * 0 none, 1 fp32 values, 2 32-bit fixed point round(v 2^32) (covariances bounded by 1); [11] seconds spent on
 * other ranks' column blocks under MELLON_AMD_EMULATE_RANKS (tools/emulate_rank.py; 0 otherwise), already excluded
 * from [3] and [4].
 * [12] / [13] kernel seconds / launches of the solver's subsample passes (every [14]-th row: the cells of the preconditioner's
 * Gram); [15] wall seconds and [16] count of preconditioner rebuilds inside mln_map_solve; [17] passes over the n x m buffer in
 * full-fp64-pass equivalents ([6] + [9] / 2 + [13] / [14]); [18] rebuilds that declined (the sample's weights span more
 * than 1e5) or lost positive definiteness -- the solve went on with the first preconditioner; [19] rebuilt preconditioners
 * that failed their trial (no convergence within 60 iterations) and were replaced by the first again; [20] halvings of a
 * start whose loss was not finite or above 1e30; [21] how the last mln_fit_gram_rank counted: 1 = inertia of G - x I
 * (csrc/ldl_inertia.hip), 2 = tridiagonalisation + Sturm counts (csrc/tridiag.hip), 0 = not called.                       */
#define MLN_N_STAGE_TIMES 22
int mln_stage_times(mln_fit* fit, double* out /* MLN_N_STAGE_TIMES */);

#ifdef __cplusplus
}
#endif
#endif /* MELLON_HIP_H */
