// Dense m x m factorisation and triangular solves on the device (fp64).
//
// Reference arithmetic: jnp.linalg.cholesky (decomposition.py:115, conditional.py:73) and
// jax.scipy.linalg.solve_triangular (decomposition.py:209, conditional.py:63-65,264,818), i.e.
// LAPACK potrf/trtrs semantics.  Here: right-looking blocked Cholesky (128-wide block columns: the
// diagonal block and its inverse by one workgroup in LDS (potrf.hip), then the panel solve and the trailing
// SYRK as fp64 MFMA GEMMs), and triangular solves as sequences of GEMMs against row-/column-scaled copies of
// the factor whose 128x128 diagonal blocks are inverted explicitly (the standard GPU TRSM; only
// diagonal blocks are ever inverted, so the conditioning that enters is that of a 128-block).
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "mln_internal.h"
#include "linalg.h"
#include "mln_options.h"

namespace {

constexpr int PB = 64;    // Cholesky panel width
constexpr int TB = 128;   // triangular-solve block

// inverse of every 64 x 64 diagonal block of a lower-triangular factor, written into the same
// position of W (blocks past m are padded with identity rows)
__global__ __launch_bounds__(256) void k_trtri64(const double* __restrict__ Lf, int64_t m, int64_t ld,
                                                 double* __restrict__ W, int64_t ldw) {
  __shared__ double T[PB][PB + 1];
  __shared__ double X[PB][PB + 1];
  const int tid = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * PB;
  const int nb = (int)((m - j0 < PB) ? (m - j0) : PB);
  for (int e = tid; e < PB * PB; e += 256) {
    int i = e / PB, j = e % PB;
    T[i][j] = (i < nb && j <= i) ? Lf[(j0 + i) * ld + j0 + j] : ((i == j) ? 1.0 : 0.0);
  }
  __syncthreads();
  if (tid < PB) {
    const int j = tid;
    for (int i = 0; i < PB; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      if (i < j) { X[i][j] = 0.0; continue; }
      for (int k = j; k < i; ++k) s = fma(-T[i][k], X[k][j], s);
      X[i][j] = s / T[i][i];
    }
  }
  __syncthreads();
  for (int e = tid; e < PB * PB; e += 256) {
    int i = e / PB, j = e % PB;
    if (i < nb && j < nb) W[(j0 + i) * ldw + j0 + j] = X[i][j];
  }
}

// completes the inverse of each 128 x 128 diagonal block from its two 64-block inverses:
// X21 = -X22 * T21 * X11
__global__ __launch_bounds__(256) void k_trtri_merge128(const double* __restrict__ Lf, int64_t m, int64_t ld,
                                                        double* W, int64_t ldw) {
  __shared__ double Ta[PB][PB + 1];
  __shared__ double Xa[PB][PB + 1];
  __shared__ double Xb[PB][PB + 1];
  const int tid = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * TB;
  const int64_t r0 = j0 + PB;
  if (r0 >= m) return;
  const int nb2 = (int)((m - r0 < PB) ? (m - r0) : PB);
  for (int e = tid; e < PB * PB; e += 256) {
    int i = e / PB, j = e % PB;
    Ta[i][j] = (i < nb2) ? Lf[(r0 + i) * ld + j0 + j] : 0.0;                 // T21
    Xa[i][j] = W[(j0 + i) * ldw + j0 + j];                                   // X11 (full 64 block exists)
    Xb[i][j] = (i < nb2 && j < nb2) ? W[(r0 + i) * ldw + r0 + j] : 0.0;      // X22
  }
  __syncthreads();
  // tmp = T21 * X11 ; each thread 16 outputs: row i = tid / 4, cols (tid % 4) * 16 ..
  const int ti = tid >> 2, tc = (tid & 3) * 16;
  double tmp[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) tmp[q] = 0.0;
  for (int k = 0; k < PB; ++k) {
    const double a = Ta[ti][k];
#pragma unroll
    for (int q = 0; q < 16; ++q) tmp[q] = fma(a, Xa[k][tc + q], tmp[q]);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) Ta[ti][tc + q] = tmp[q];
  __syncthreads();
  double out[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) out[q] = 0.0;
  for (int k = 0; k < PB; ++k) {
    const double a = Xb[ti][k];
#pragma unroll
    for (int q = 0; q < 16; ++q) out[q] = fma(a, Ta[k][tc + q], out[q]);
  }
  if (ti < nb2) {
#pragma unroll
    for (int q = 0; q < 16; ++q) W[(r0 + ti) * ldw + j0 + tc + q] = -out[q];
  }
}

__global__ void k_add_diag(double* A, int64_t m, int64_t lda, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) A[i * lda + i] += v;
}

__global__ void k_zero_upper(double* A, int64_t m, int64_t lda) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t i = blockIdx.y;
  if (j < m && j > i) A[i * lda + j] = 0.0;
}

__global__ void k_sym_from_lower(double* A, int64_t m, int64_t lda) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t i = blockIdx.y;
  if (j < m && j > i) A[i * lda + j] = A[j * lda + i];
}

__global__ void k_axpby(int64_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + ((b != 0.0) ? b * y[i] : 0.0);
}

__global__ void k_copy_block(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd,
                             int64_t rows, int64_t cols) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t i = blockIdx.y;
  if (i < rows && j < cols) dst[i * ldd + j] = src[i * lds + j];
}

__global__ void k_transpose(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd,
                            int64_t m) {
  __shared__ double tile[32][33];
  const int64_t bx = (int64_t)blockIdx.x * 32, by = (int64_t)blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    int64_t i = by + r, j = bx + threadIdx.x;
    tile[r][threadIdx.x] = (i < m && j < m) ? src[i * lds + j] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    int64_t i = bx + r, j = by + threadIdx.x;
    if (i < m && j < m) dst[i * ldd + j] = tile[threadIdx.x][r];
  }
}

}  // namespace

int launch_transpose(mln_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t m) {
  if (m <= 0) return MLN_OK;
  const unsigned t = (unsigned)((m + 31) / 32);
  hipLaunchKernelGGL(k_transpose, dim3(t, t), dim3(32, 8), 0, ctx->stream, src, lds, dst, ldd, m);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_add_diag(mln_ctx* ctx, double* A, int64_t m, int64_t lda, double v) {
  if (m <= 0) return MLN_OK;
  hipLaunchKernelGGL(k_add_diag, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, A, m, lda, v);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_symmetrize_from_lower(mln_ctx* ctx, double* A, int64_t m, int64_t lda) {
  if (m <= 0) return MLN_OK;
  hipLaunchKernelGGL(k_sym_from_lower, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, ctx->stream, A, m, lda);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_axpby(mln_ctx* ctx, int64_t n, double a, const double* x, double b, double* y) {
  if (n <= 0) return MLN_OK;
  int64_t nb = (n + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(k_axpby, dim3((unsigned)nb), dim3(256), 0, ctx->stream, n, a, x, b, y);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_copy_block(mln_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows,
                      int64_t cols) {
  if (rows <= 0 || cols <= 0) return MLN_OK;
  if (rows > 65535) {
    for (int64_t r = 0; r < rows; r += 65535) {
      int64_t nr = (rows - r < 65535) ? rows - r : 65535;
      MLN_TRY(launch_copy_block(ctx, src + r * lds, lds, dst + r * ldd, ldd, nr, cols));
    }
    return MLN_OK;
  }
  hipLaunchKernelGGL(k_copy_block, dim3((unsigned)((cols + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream,
                     src, lds, dst, ldd, rows, cols);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

namespace {
// dst <- src on the tiles strictly below the 128-block diagonal
__global__ void k_copy_below_blocks(const double* __restrict__ src, double* __restrict__ dst, int64_t m, int64_t ld) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j < m && (j >> 7) < (i >> 7)) dst[i * ld + j] = src[i * ld + j];
}

}  // namespace

// In-place lower Cholesky of the m x m matrix at A (only the lower triangle is read): right-looking over 128-wide
// block columns.  Per step: k_potrf128 (potrf.hip: the diagonal block and its inverse, one workgroup), the rest of
// the block column as one GEMM  P = A_panel T^-T  written to a side matrix (no aliasing -> the 64 x 64 latency tiles
// apply: with <= 39 row tiles a 128-tile launch is 14 us of serial work per CU), the trailing update as one
// lower-tiles-only GEMM  A22 -= P P^T per PAIR of block columns.  The side matrix is copied under the block diagonal of
// A at the end.
// Right-looking blocked Cholesky, two 128-wide block columns per round.  The critical path of a round is five dependent
// launches -- diagonal block, panel, the narrow update of the second block column, its diagonal block, its panel: 2 x 45 us
// of single-workgroup work and three ~10 us GEMMs that are launch + pipeline latency -- followed by the trailing update,
// the only launch with work for the whole chip (150 us at the start of a 5000 x 5000 factorisation, falling).
// Timeline of chol(5000): 20 x 121 us of critical path + 1.1 ms of trailing updates = 3.5 ms (profiles/r04_step_timeline.txt).
// (Round 4 measured LOOK-AHEAD across kernels -- the trailing update split into the next round's two block columns and the
//  rest on a CU-masked side stream: each factorisation got ~1.2 ms SLOWER, the diagonal-block kernel 65 us instead of 50 next
//  to the big GEMM; taken out in round 5, profiles/HISTORY.md.)
// Round 5: nbatch (1 or 2) matrices of one shape go through the SAME chain of launches (grid z of the GEMMs, workgroup b of
// the diagonal-block kernel): the chain is latency, so the second factorisation costs only its trailing-update flops --
// chol(Kj) rides along with the preconditioner's chol(M) instead of taking 4.1 ms of its own before the kernel-matrix pass.
static int cholesky_lower_batched(mln_ctx* ctx, double* A, int64_t m, int64_t lda, int nbatch, int64_t a_bs, int* bad) {
  if (bad) *bad = 0;
  if (m <= 0) return MLN_OK;
  constexpr int CB = 128;
  double* Dinv = nullptr;
  double* Ls = nullptr;
  const size_t mat = (size_t)m * (size_t)lda;
  MLN_HIP(ctx, mln_dmalloc((void**)&Dinv, sizeof(double) * CB * CB * nbatch));
  MLN_HIP(ctx, hipMemsetAsync(Dinv, 0, sizeof(double) * CB * CB * nbatch, ctx->stream));
  MLN_HIP(ctx, hipMemsetAsync(ctx->d_info, 0, sizeof(int) * 2, ctx->stream));
  if (m > CB) MLN_HIP(ctx, mln_dmalloc((void**)&Ls, sizeof(double) * mat * nbatch));
  int rc = MLN_OK;
  auto batched = [&](GemmArgs& g, int64_t bsa, int64_t bsb, int64_t bsc) { if (nbatch > 1) { g.batch = nbatch; g.bsa = bsa; g.bsb = bsb; g.bsc = bsc; } };
  // Two block columns per round: the second one is brought up to date by a narrow GEMM (K = 128, 128 columns), and the
  // big trailing update then runs ONCE with K = 256 on the two panels side by side in the side matrix -- the same flops
  // as two K = 128 updates over nearly the same area, at half the per-tile prologue/epilogue cost.
  auto panel = [&](int64_t j, int nb, int64_t rem) -> int {          // P = A[j+nb.., j..j+nb] Dinv^T -> side matrix
    GemmArgs g{};
    g.A = A + (j + nb) * lda + j; g.lda = lda; g.B = Dinv; g.ldb = CB; g.C = Ls + (j + nb) * lda + j; g.ldc = lda;
    g.M = rem; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1;
    batched(g, a_bs, (int64_t)CB * CB, (int64_t)mat);
    return launch_dgemm(ctx, g);
  };
  for (int64_t j0 = 0; j0 < m && rc == MLN_OK; j0 += 2 * CB) {
    const int nb1 = (int)((m - j0 < CB) ? (m - j0) : CB);
    rc = launch_potrf128(ctx, A + j0 * lda + j0, lda, nb1, Dinv, ctx->d_info, j0, nbatch, a_bs);
    const int64_t rem1 = m - j0 - nb1;
    if (rc != MLN_OK || rem1 <= 0) break;
    rc = panel(j0, nb1, rem1);
    if (rc != MLN_OK) break;
    const int64_t j1 = j0 + nb1;
    const int nb2 = (int)((rem1 < CB) ? rem1 : CB);
    const double* P1 = Ls + j1 * lda + j0;
    {
      GemmArgs u{};   // block column j1 only: A[j1.., j1..j1+nb2] -= P1 P1[0:nb2]^T
      u.A = P1; u.lda = lda; u.B = P1; u.ldb = lda; u.C = A + j1 * lda + j1; u.ldc = lda;
      u.M = rem1; u.N = nb2; u.K = nb1; u.alpha = -1.0; u.beta = 1.0; u.ta = 0; u.tb = 1;
      batched(u, (int64_t)mat, (int64_t)mat, a_bs);
      rc = launch_dgemm(ctx, u);
      if (rc != MLN_OK) break;
    }
    rc = launch_potrf128(ctx, A + j1 * lda + j1, lda, nb2, Dinv, ctx->d_info, j1, nbatch, a_bs);
    const int64_t rem2 = rem1 - nb2;
    if (rc != MLN_OK || rem2 <= 0) break;
    rc = panel(j1, nb2, rem2);
    if (rc != MLN_OK) break;
    const int64_t j2 = j1 + nb2;
    const double* Pw = Ls + j2 * lda + j0;
    GemmArgs t{};     // A22 -= [P1' P2] [P1' P2]^T on lower tiles, K = nb1 + nb2
    t.A = Pw; t.lda = lda; t.B = Pw; t.ldb = lda; t.C = A + j2 * lda + j2; t.ldc = lda;
    t.M = rem2; t.N = rem2; t.K = nb1 + nb2; t.alpha = -1.0; t.beta = 1.0; t.ta = 0; t.tb = 1; t.lower_only = 1;
    batched(t, (int64_t)mat, (int64_t)mat, a_bs);
    rc = launch_dgemm(ctx, t);
  }
  int info[2] = {0, 0};
  if (rc == MLN_OK) {
    for (int b = 0; b < nbatch; ++b) {
      if (Ls) hipLaunchKernelGGL(k_copy_below_blocks, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, ctx->stream, Ls + (size_t)b * mat, A + (int64_t)b * a_bs, m, lda);
      hipLaunchKernelGGL(k_zero_upper, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, ctx->stream, A + (int64_t)b * a_bs, m, lda);
    }
    hipError_t e = hipMemcpyAsync(info, ctx->d_info, sizeof(int) * 2, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "cholesky sync", __FILE__, __LINE__);
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  (void)mln_dfree(Dinv);
  if (Ls) (void)mln_dfree(Ls);
  if (rc != MLN_OK) return rc;
  for (int b = nbatch - 1; b >= 0; --b)
    if (info[b] != 0) {
      if (bad) *bad |= 1 << b;
      mln_set_error(ctx, "Cholesky failed: non-positive or NaN pivot at index " + std::to_string(info[b] - 1));
      rc = MLN_ERR_NOT_PD;
    }
  return rc;
}

int dev_cholesky_lower(mln_ctx* ctx, double* A, int64_t m, int64_t lda) { return cholesky_lower_batched(ctx, A, m, lda, 1, 0, nullptr); }

int dev_cholesky_lower2(mln_ctx* ctx, double* A, double* A2, int64_t m, int64_t lda, int* bad) {
  return cholesky_lower_batched(ctx, A, m, lda, 2, (int64_t)(A2 - A), bad);
}

// ---- triangular solves through block-scaled copies of the factor ------------------------------
void triinv_free(TriInv* t) {
  if (t->W) (void)mln_dfree(t->W);
  if (t->W2) (void)mln_dfree(t->W2);
  t->W = t->W2 = nullptr;
}

// W  (row-scaled):    W[j, <=j]  = Dinv_j [ -Lf[j,<j] | I ]     -> forward left-looking solves, X Lf^-T, backward updates
// W2 (column-scaled): W2[>=j, j] = [ I ; -Lf[>j,j] ] Dinv_j     -> backward left-looking solves, forward updates
// Both are built whatever the flags say: each is ONE launch of the GEMM kernel in its block-diagonal mode
// (the tile (i, j), i > j, of W is -Dinv_i Lf_ij: K range = row block i; of W2 it is -Lf_ij Dinv_j: K range =
// column block j) -- 2 launches where the block-column loops took 78.
int triinv_build(mln_ctx* ctx, const double* Lf, int64_t m, int64_t ld, bool need_w, bool need_w2, TriInv* out) {
  (void)need_w; (void)need_w2;
  out->m = m;
  out->Lf = Lf;
  out->ldf = ld;
  out->ld = ((m + 15) / 16) * 16;
  const size_t bytes = sizeof(double) * (size_t)m * (size_t)out->ld;
  MLN_HIP(ctx, mln_dmalloc((void**)&out->W, bytes));
  hipError_t e = mln_dmalloc((void**)&out->W2, bytes);
  if (e != hipSuccess) { triinv_free(out); return mln_hip_fail(ctx, e, "alloc W2", __FILE__, __LINE__); }
  double* D = out->W;   // block-diagonal inverse first; the blocks under it follow
  MLN_HIP(ctx, hipMemsetAsync(D, 0, bytes, ctx->stream));
  const int64_t nb64 = (m + PB - 1) / PB, nb128 = (m + TB - 1) / TB;
  hipLaunchKernelGGL(k_trtri64, dim3((unsigned)nb64), dim3(256), 0, ctx->stream, Lf, m, ld, D, out->ld);
  hipLaunchKernelGGL(k_trtri_merge128, dim3((unsigned)nb128), dim3(256), 0, ctx->stream, Lf, m, ld, D, out->ld);
  MLN_HIP(ctx, hipGetLastError());
  MLN_HIP(ctx, hipMemcpyAsync(out->W2, D, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  int rc = MLN_OK;
  if (m > TB) {
    GemmArgs g{};   // W2[i, j] = -Lf[i, j] Dinv_j  (i > j): op(B) = the block-diagonal D, read from W2's own diagonal
    g.A = Lf; g.lda = ld; g.B = D; g.ldb = out->ld; g.C = out->W2; g.ldc = out->ld;
    g.M = m; g.N = m; g.K = m; g.alpha = -1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.lower_only = 2; g.kmode = 2;
    rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK) {
      GemmArgs h{};  // W[i, j] = -Dinv_i Lf[i, j]  (i > j): op(A) block-diagonal -- read from W2, whose diagonal
      h.A = out->W2; h.lda = out->ld; h.B = Lf; h.ldb = ld; h.C = out->W; h.ldc = out->ld;   // blocks this launch leaves alone
      h.M = m; h.N = m; h.K = m; h.alpha = -1.0; h.beta = 0.0; h.ta = 0; h.tb = 0; h.lower_only = 2; h.kmode = 1;
      rc = launch_dgemm(ctx, h);
    }
  }
  if (rc != MLN_OK) { (void)hipStreamSynchronize(ctx->stream); triinv_free(out); mln_set_error(ctx, "triinv_build failed"); }
  return rc;
}

// X (n x m, in place) <- X Lf^-T, left-looking over 128-wide block columns:
//   X_j <- [X_<j | X_j] W_j^T      (decomposition.py:209:  L = solve_triangular(Lp, C.T, lower=True).T)
int triinv_solve_right_T(mln_ctx* ctx, const TriInv& t, double* X, int64_t n, int64_t ldx) {
  if (n < 32768 && t.W && t.W2 && t.m > TB) {
    // Few rows (the per-rank Gram sample of a many-rank run): the left-looking form would give each of the n / 128
    // row tiles one workgroup with a K loop of up to m -- serial work of ~0.5 ms per launch, 40 launches.  Right-
    // looking with the diagonal blocks factored out (see triinv_solve_left): B_i += B~_j W2[i,j]^T for i > j is an
    // n x (m - j) x 128 product on many tiles, and X = B~ blockdiag(Dinv^T) is one launch at the end.
    for (int64_t j0 = 0; j0 < t.m; j0 += TB) {
      const int64_t nb = (t.m - j0 < TB) ? (t.m - j0) : TB;
      const int64_t rem = t.m - j0 - nb;
      if (rem <= 0) break;
      GemmArgs u{};
      u.A = X + j0; u.lda = ldx; u.B = t.W2 + (j0 + nb) * t.ld + j0; u.ldb = t.ld; u.C = X + j0 + nb; u.ldc = ldx;
      u.M = n; u.N = rem; u.K = nb; u.alpha = 1.0; u.beta = 1.0; u.ta = 0; u.tb = 1;
      MLN_TRY(launch_dgemm(ctx, u));
    }
    GemmArgs g{};   // in place: tile (i, j) reads columns of block j of its own rows only
    g.A = X; g.lda = ldx; g.B = t.W; g.ldb = t.ld; g.C = X; g.ldc = ldx;
    g.M = n; g.N = t.m; g.K = t.m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1; g.kmode = 2;
    return launch_dgemm(ctx, g);
  }
  for (int64_t j0 = 0; j0 < t.m; j0 += TB) {
    const int64_t nb = (t.m - j0 < TB) ? (t.m - j0) : TB;
    GemmArgs g{};
    g.A = X; g.lda = ldx; g.B = t.W + j0 * t.ld; g.ldb = t.ld; g.C = X + j0; g.ldc = ldx;
    g.M = n; g.N = nb; g.K = j0 + nb; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1;
    MLN_TRY(launch_dgemm(ctx, g));
  }
  return MLN_OK;
}

// B (m x p, in place) <- Lf^-1 B (forward substitution by 128-row blocks).
//   few right-hand sides: left-looking,  B_j <- W_j [B_<j ; B_j]            (one GEMM per block row)
//   many (p >= 256):      right-looking with the diagonal blocks factored out of the loop.  With B~_j the block row
//   after all updates from above, the solution is X_j = Dinv_j B~_j and the update it triggers is
//   B_>j -= Lf[>j,j] X_j = B_>j + W2[>j,j] B~_j: every step is ONE big GEMM on the un-multiplied block row, and the
//   40 diagonal products -- each a 14-45 us chain link when done one by one -- collapse into one final launch of
//   the GEMM kernel in its block-diagonal mode.
int triinv_solve_left(mln_ctx* ctx, const TriInv& t, double* B, int64_t p, int64_t ldb, bool tri_b) {
  const bool right_looking = (p >= 256) && t.W && t.W2;
  tri_b = tri_b && right_looking && p == t.m;
  if (!right_looking) {
    for (int64_t j0 = 0; j0 < t.m; j0 += TB) {
      const int64_t nb = (t.m - j0 < TB) ? (t.m - j0) : TB;
      GemmArgs g{};
      // in place: one row tile (nb <= 128), so each workgroup reads and writes only its own column tile
      g.A = t.W + j0 * t.ld; g.lda = t.ld; g.B = B; g.ldb = ldb; g.C = B + j0 * ldb; g.ldc = ldb;
      g.M = nb; g.N = p; g.K = j0 + nb; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0;
      MLN_TRY(launch_dgemm(ctx, g));
    }
    return MLN_OK;
  }
  for (int64_t j0 = 0; j0 < t.m; j0 += TB) {
    const int64_t nb = (t.m - j0 < TB) ? (t.m - j0) : TB;
    const int64_t rem = t.m - j0 - nb;
    if (rem <= 0) break;
    const int64_t ncol = tri_b ? (j0 + nb) : p;      // lower-triangular B: rows of this block end at column j0 + nb
    GemmArgs u{};
    u.A = t.W2 + (j0 + nb) * t.ld + j0; u.lda = t.ld; u.B = B + j0 * ldb; u.ldb = ldb;
    u.C = B + (j0 + nb) * ldb; u.ldc = ldb;
    u.M = rem; u.N = ncol; u.K = nb; u.alpha = 1.0; u.beta = 1.0; u.ta = 0; u.tb = 0;
    MLN_TRY(launch_dgemm(ctx, u));
  }
  GemmArgs g{};   // X = blockdiag(Dinv) B~, in place: a tile reads exactly the tile it overwrites
  g.A = t.W; g.lda = t.ld; g.B = B; g.ldb = ldb; g.C = B; g.ldc = ldb;
  g.M = t.m; g.N = p; g.K = t.m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.kmode = 1;
  g.lower_only = tri_b ? 1 : 0;
  return launch_dgemm(ctx, g);
}

// B (m x p, in place) <- Lf^-T B (backward substitution).
//   few right-hand sides: B_j <- W2[>=j, j]^T B[>=j]
//   many (p >= 256):      the mirror image of the forward solve: B_<j += W[j,<j]^T B~_j from the last block row up,
//   then X = blockdiag(Dinv^T) B~ in one launch.
int triinv_solve_left_T(mln_ctx* ctx, const TriInv& t, double* B, int64_t p, int64_t ldb, bool tri_b) {
  const bool right_looking = (p >= 256) && t.W && t.W2;
  tri_b = tri_b && right_looking && p == t.m;
  const int64_t nblk = (t.m + TB - 1) / TB;
  if (!right_looking) {
    for (int64_t jb = nblk - 1; jb >= 0; --jb) {
      const int64_t j0 = jb * TB;
      const int64_t nb = (t.m - j0 < TB) ? (t.m - j0) : TB;
      GemmArgs g{};
      g.A = t.W2 + j0 * t.ld + j0; g.lda = t.ld; g.B = B + j0 * ldb; g.ldb = ldb;
      g.C = B + j0 * ldb; g.ldc = ldb;
      g.M = nb; g.N = p; g.K = t.m - j0; g.alpha = 1.0; g.beta = 0.0; g.ta = 1; g.tb = 0;
      MLN_TRY(launch_dgemm(ctx, g));
    }
    return MLN_OK;
  }
  for (int64_t jb = nblk - 1; jb >= 1; --jb) {
    const int64_t j0 = jb * TB;
    const int64_t nb = (t.m - j0 < TB) ? (t.m - j0) : TB;
    const int64_t c0 = tri_b ? j0 : 0;                // upper-triangular B: rows of this block start at column j0
    GemmArgs u{};
    u.A = t.W + j0 * t.ld; u.lda = t.ld; u.B = B + j0 * ldb + c0; u.ldb = ldb; u.C = B + c0; u.ldc = ldb;
    u.M = j0; u.N = p - c0; u.K = nb; u.alpha = 1.0; u.beta = 1.0; u.ta = 1; u.tb = 0;   // W[j,<j]^T B~_j
    MLN_TRY(launch_dgemm(ctx, u));
  }
  GemmArgs g{};
  g.A = t.W; g.lda = t.ld; g.B = B; g.ldb = ldb; g.C = B; g.ldc = ldb;
  g.M = t.m; g.N = p; g.K = t.m; g.alpha = 1.0; g.beta = 0.0; g.ta = 1; g.tb = 0; g.kmode = 1;
  g.lower_only = tri_b ? 3 : 0;
  return launch_dgemm(ctx, g);
}
