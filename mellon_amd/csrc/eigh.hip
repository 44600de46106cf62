// Symmetric eigensolver for the Nystroem rank reduction (reference: jax.numpy.linalg.eigh at
// mellon/decomposition.py:50, reached from _full_decomposition_low_rank :126-171 and _modified_low_rank
// :213-266).
//
// Method: block two-sided Jacobi in one-sided form.  Y = [X | V] is kept row-wise ("row j" = [x_j ; v_j])
// with V <- I and X <- V A refreshed by one GEMM per sweep, so that B = V A V^T is available blockwise as
// B_pq = v_p . x_q without ever forming it.  One launch = one round of a round-robin tournament over
// blocks of 16 rows: each workgroup owns one block pair (32 rows), the nb/2 pairs of a round touch
// disjoint rows.  Per pair:
//   1. S (32 x 32) = V_rows X_rows^T, symmetrised     (LDS-tiled, 2 x 2 outputs per thread)
//   2. cyclic two-sided Jacobi on S in LDS -> Q       (16 disjoint rotations per step, 31 steps per sweep,
//                                                      until every |S_pq| <= thresh)
//   3. rows <- Q^T rows over all 2 m columns           (one column per thread, Q broadcast from LDS)
// nb - 1 rounds = one sweep (every pair of rows meets once); sweeps repeat until no |B_pq| exceeded
// thresh = 2 eps sqrt(m) |A|_F  (absolute criterion: the backward-stable accuracy of LAPACK syevd; a
// relative criterion cannot be met for the noise-level eigenvalues of a kernel matrix).  V is a product
// of plane rotations, orthogonal to rounding regardless of the spectrum; lambda_j = v_j . (A v_j).
// All traffic is coalesced along rows; the working set (2 m^2 doubles = 400 MB at m = 5000) streams
// from HBM once in phase 1 and once more (read + write) in phase 3 of every round.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <vector>
#include "mln_options.h"

#include "linalg.h"
#include "mln_internal.h"

namespace {

constexpr int EB = 16;       // rows per block
constexpr int ER = 2 * EB;   // rows per block pair
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
constexpr int SP = 34;       // LDS row stride of S / Q (even: 16-byte aligned pairs; 68 dwords: bank shift 4)

struct JacArgs {
  double* Y;
  int64_t ld;       // = 2 * mp
  int64_t mp;       // padded order (multiple of 16)
  int nb;           // real blocks
  int nbp;          // players (even, >= nb)
  int round;
  int max_local;    // local sweeps per visit
  double tol;       // absolute threshold on |B_pq|
  unsigned* rotations;   // device counter (rotations above threshold in this sweep)
};

__device__ __forceinline__ void circle_pair(int n_players, int round, int slot, int* a, int* b) {
  const int n1 = n_players - 1;
  if (slot == 0) { *a = n1; *b = round % n1; }
  else { *a = (round + slot) % n1; *b = (round - slot + n1) % n1; }
}

constexpr int JW = 8;            // waves per block pair (phases 1 and 3 split the columns over them)
constexpr int JT = 64 * JW;

__global__ __launch_bounds__(JT) void k_jacobi_round(JacArgs a) {
  __shared__ double part[JW][ER][SP];
  __shared__ double S[ER][SP];
  __shared__ double Q[ER][SP];
  __shared__ double rot_c[EB], rot_s[EB];
  __shared__ int rot_p[EB], rot_q[EB];
  __shared__ int sh_any;

  const int t = threadIdx.x;
  int bi, bj;
  circle_pair(a.nbp, a.round, blockIdx.x, &bi, &bj);
  const bool vi = bi < a.nb, vj = bj < a.nb;
  if (!vi && !vj) return;
  if (!vi) { bi = bj; bj = a.nb; }   // lone block first
  const bool have_j = vi && vj;
  const int64_t row_i = (int64_t)bi * EB, row_j = (int64_t)bj * EB;
  auto grow = [&](int r) -> int64_t { return r < EB ? row_i + r : row_j + (r - EB); };

  // ---- 1. S = V_rows . X_rows^T (block of V A V^T), symmetrised ------------------------------------
  // v_mfma_f64_16x16x4: A[m = li][k = lk], B[k = lk][n = li], D[row = lk + 4 reg][col = li].  The k index
  // of one instruction is any 4 columns as long as both operands agree: lane group lk owns columns
  // k0 + 4 lk .. + 3 of a 16-column chunk (two 16-byte loads per row: full 128-byte lines per 16 rows),
  // and the 4 components feed 4 successive MFMAs.  The 4 waves split the chunks; partial S meet in LDS.
  const int lane = t & 63, wave = t >> 6, li = lane & 15, lk = lane >> 4;
  {
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
    const double* r0 = a.Y + grow(li) * a.ld;
    const double* r1 = have_j ? a.Y + grow(EB + li) * a.ld : r0;
    for (int64_t k0 = (int64_t)wave * 16; k0 < a.mp; k0 += 16 * JW) {
      const int64_t k = k0 + 4 * lk;
      double x[2][4], v[2][4];
      const d2v xa = *reinterpret_cast<const d2v*>(r0 + k), xb = *reinterpret_cast<const d2v*>(r0 + k + 2);
      const d2v va = *reinterpret_cast<const d2v*>(r0 + a.mp + k), vb = *reinterpret_cast<const d2v*>(r0 + a.mp + k + 2);
      x[0][0] = xa.x; x[0][1] = xa.y; x[0][2] = xb.x; x[0][3] = xb.y;
      v[0][0] = va.x; v[0][1] = va.y; v[0][2] = vb.x; v[0][3] = vb.y;
      if (have_j) {
        const d2v xc = *reinterpret_cast<const d2v*>(r1 + k), xd = *reinterpret_cast<const d2v*>(r1 + k + 2);
        const d2v vc = *reinterpret_cast<const d2v*>(r1 + a.mp + k), vd = *reinterpret_cast<const d2v*>(r1 + a.mp + k + 2);
        x[1][0] = xc.x; x[1][1] = xc.y; x[1][2] = xd.x; x[1][3] = xd.y;
        v[1][0] = vc.x; v[1][1] = vc.y; v[1][2] = vd.x; v[1][3] = vd.y;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) { x[1][c] = 0.0; v[1][c] = 0.0; }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[i][c], x[j][c], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][i * EB + lk + 4 * r][j * EB + li] = acc[i][j][r];
  }
  __syncthreads();
  for (int idx = t; idx < ER * ER; idx += JT) {
    const int r = idx >> 5, c = idx & 31;
    double s_rc = 0.0, s_cr = 0.0;
#pragma unroll
    for (int w = 0; w < JW; ++w) { s_rc += part[w][r][c]; s_cr += part[w][c][r]; }
    S[r][c] = 0.5 * (s_rc + s_cr);
    Q[r][c] = (r == c) ? 1.0 : 0.0;
  }
  if (t == 0) sh_any = 0;
  __syncthreads();

  // ---- 2. local symmetric Jacobi: Q^T S Q -> diagonal ----------------------------------------------
  for (int sweep = 0; sweep < a.max_local; ++sweep) {
    int mine = 0;
    for (int idx = t; idx < ER * ER; idx += JT) {
      const int r = idx >> 5, c = idx & 31;
      if (r < c && fabs(S[r][c]) > a.tol) mine = 1;
    }
    if (!__syncthreads_or(mine)) break;
    if (t == 0) sh_any = 1;
    for (int step = 0; step < ER - 1; ++step) {
      if (t < EB) {
        int p, q;
        circle_pair(ER, step, t, &p, &q);
        if (p > q) { const int tmp = p; p = q; q = tmp; }
        const double app = S[p][p], aqq = S[q][q], apq = S[p][q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > a.tol) {
          const double zeta = (aqq - app) / (2.0 * apq);
          const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          c = 1.0 / sqrt(1.0 + tt * tt);
          s = c * tt;
        }
        rot_c[t] = c; rot_s[t] = s; rot_p[t] = p; rot_q[t] = q;
      }
      __syncthreads();
      // columns: S <- S J, Q <- Q J
#pragma unroll
      for (int it = t; it < ER * EB; it += JT) {
        const int i = it & 31, l = it >> 5;
        const double c = rot_c[l], s = rot_s[l];
        if (s != 0.0) {
          const int p = rot_p[l], q = rot_q[l];
          const double sp = S[i][p], sq = S[i][q];
          S[i][p] = c * sp - s * sq; S[i][q] = s * sp + c * sq;
          const double qp = Q[i][p], qq = Q[i][q];
          Q[i][p] = c * qp - s * qq; Q[i][q] = s * qp + c * qq;
        }
      }
      __syncthreads();
      // rows: S <- J^T S
#pragma unroll
      for (int it = t; it < ER * EB; it += JT) {
        const int j = it & 31, l = it >> 5;
        const double c = rot_c[l], s = rot_s[l];
        if (s != 0.0) {
          const int p = rot_p[l], q = rot_q[l];
          const double sp = S[p][j], sq = S[q][j];
          S[p][j] = c * sp - s * sq; S[q][j] = s * sp + c * sq;
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (!sh_any) return;   // all 32 rows already orthogonal: nothing to apply
  if (t == 0) atomicAdd(a.rotations, 1u);

  // ---- 3. rows <- Q^T rows over [X | V] -------------------------------------------------------------
  // D (32 x 16 columns) = Q^T (32 x 32) . rows (32 x 16 columns): the Q^T operand (2 x 8 registers) is
  // loaded once; each wave walks 16-column chunks, 16 MFMAs per chunk, in place (a chunk is read
  // completely before it is written, chunks are disjoint).
  double qa[2][8];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qa[mt][ks] = Q[ks * 4 + lk][mt * EB + li];
  const int n_ks = have_j ? 8 : 4;
  for (int64_t c0 = (int64_t)wave * 16; c0 < a.ld; c0 += 16 * JW) {
    double b[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) b[ks] = (ks < n_ks) ? a.Y[grow(ks * 4 + lk) * a.ld + c0 + li] : 0.0;
    v4d d0 = v4d{0.0, 0.0, 0.0, 0.0}, d1 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[0][ks], b[ks], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[1][ks], b[ks], d1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a.Y[grow(lk + 4 * r) * a.ld + c0 + li] = d0[r];
      if (have_j) a.Y[grow(EB + lk + 4 * r) * a.ld + c0 + li] = d1[r];
    }
  }
}

// Y = [ (A + A^T)/2 | I ], zero padded to mp
__global__ void k_eigh_init(const double* __restrict__ A, int64_t lda, int64_t m, double* __restrict__ Y, int64_t ld,
                            int64_t mp) {
  const int64_t j = blockIdx.y;   // row of Y = column of A
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < mp; i += (int64_t)gridDim.x * blockDim.x) {
    double x = 0.0;
    if (i < m && j < m) x = 0.5 * (A[i * lda + j] + A[j * lda + i]);
    Y[j * ld + i] = x;
    Y[j * ld + mp + i] = (i == j) ? 1.0 : 0.0;
  }
}

// lambda_j = v_j . x_j, one wave per row
__global__ __launch_bounds__(256) void k_eigh_values(const double* __restrict__ Y, int64_t ld, int64_t mp, int64_t m,
                                                     double* __restrict__ w) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= m) return;
  double s = 0.0;
  for (int64_t i = lane; i < mp; i += 64) s = fma(Y[j * ld + i], Y[j * ld + mp + i], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) w[j] = s;
}

// ss[j] = |x_j|^2, one wave per row (summed on the host in index order: deterministic |A|_F)
__global__ __launch_bounds__(256) void k_eigh_rowss(const double* __restrict__ Y, int64_t ld, int64_t mp, int64_t m,
                                                    double* __restrict__ ss) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= m) return;
  double s = 0.0;
  for (int64_t i = lane; i < mp; i += 64) { const double x = Y[j * ld + i]; s = fma(x, x, s); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) ss[j] = s;
}

// rows[jj][i] = V-part of Y[perm[jj]][i]   (eigenvector jj as a contiguous row)
__global__ void k_eigh_gather_rows(const double* __restrict__ Y, int64_t ld, int64_t mp, int64_t m,
                                   const int* __restrict__ perm, double* __restrict__ out, int64_t ldo) {
  const int64_t jj = blockIdx.y;
  const int64_t src = perm[jj];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ldo; i += (int64_t)gridDim.x * blockDim.x)
    out[jj * ldo + i] = (i < m) ? Y[src * ld + mp + i] : 0.0;
}

}  // namespace

// Eigen-decomposition of the symmetric m x m device matrix A (leading dimension lda).
//   w_host  : m eigenvalues, ascending (LAPACK order)
//   Vrows   : device, m x ldv row-major, row j = eigenvector j (same order), columns >= m zeroed
int dev_eigh(mln_ctx* ctx, const double* A, int64_t m, int64_t lda, double* w_host, double* Vrows, int64_t ldv,
             int* n_sweeps_out) {
  if (m <= 0) return MLN_OK;
  const int64_t mp = ((m + EB - 1) / EB) * EB;
  const int64_t ld = 2 * mp;
  const int nb = (int)(mp / EB);
  const int nbp = (nb % 2 == 0) ? nb : nb + 1;
  double* Y = nullptr;
  unsigned* d_rot = nullptr;
  double* d_w = nullptr;
  int* d_perm = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&Y, sizeof(double) * (size_t)mp * ld));
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(ctx->stream);
    if (Y) (void)mln_dfree(Y);
    if (d_rot) (void)mln_dfree(d_rot);
    if (d_w) (void)mln_dfree(d_w);
    if (d_perm) (void)mln_dfree(d_perm);
  };
  auto fail = [&](hipError_t e, const char* what) { cleanup(); return mln_hip_fail(ctx, e, what, __FILE__, __LINE__); };
  hipError_t e = mln_dmalloc((void**)&d_rot, sizeof(unsigned));
  if (e == hipSuccess) e = mln_dmalloc((void**)&d_w, sizeof(double) * (size_t)mp);
  if (e == hipSuccess) e = mln_dmalloc((void**)&d_perm, sizeof(int) * (size_t)mp);
  if (e != hipSuccess) return fail(e, "eigh workspace");

  hipLaunchKernelGGL(k_eigh_init, dim3((unsigned)std::min<int64_t>((mp + 255) / 256, 64), (unsigned)mp), dim3(256), 0,
                     ctx->stream, A, lda, m, Y, ld, mp);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(e, "k_eigh_init");

  // |A|_F and the symmetrised copy As = X_0 (the per-sweep refresh X <- V As needs it)
  double* As = nullptr;
  e = mln_dmalloc((void**)&As, sizeof(double) * (size_t)mp * mp);
  if (e != hipSuccess) return fail(e, "eigh workspace");
  unsigned* h_rot = nullptr;   // pinned: the per-sweep convergence flag comes back without a staging copy
  e = hipHostMalloc((void**)&h_rot, sizeof(unsigned), hipHostMallocDefault);
  if (e != hipSuccess) { (void)mln_dfree(As); return fail(e, "eigh workspace"); }
  auto fail2 = [&](hipError_t err, const char* what) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)mln_dfree(As);
    (void)hipHostFree(h_rot);
    return fail(err, what);
  };
  if (launch_copy_block(ctx, Y, ld, As, mp, mp, mp) != MLN_OK) return fail2(hipGetLastError(), "copy");
  hipLaunchKernelGGL(k_eigh_rowss, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, Y, ld, mp, m, d_w);
  std::vector<double> ss((size_t)m);
  e = hipMemcpyAsync(ss.data(), d_w, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail2(e, "norm");
  double fro2 = 0.0;
  for (double v : ss) fro2 += v;
  if (!(fro2 == fro2) || std::isinf(fro2)) {
    (void)mln_dfree(As);
    (void)hipHostFree(h_rot);
    cleanup();
    mln_set_error(ctx, "eigh: matrix contains NaN or Inf");
    return MLN_ERR_ARG;
  }

  JacArgs a{};
  a.Y = Y; a.ld = ld; a.mp = mp; a.nb = nb; a.nbp = nbp;
  a.tol = 2.0 * std::sqrt((double)m) * 2.220446049250313e-16 * std::sqrt(fro2);
  a.rotations = d_rot;
  a.max_local = 2;   // measured: 1-3 local sweeps per visit give the same number of outer sweeps
  const int max_sweeps = 40;
  int sweep = 0;
  bool converged = (fro2 == 0.0);
  const int rounds = (nbp > 1) ? nbp - 1 : 1;
  for (; sweep < max_sweeps && !converged; ++sweep) {
    e = hipMemsetAsync(d_rot, 0, sizeof(unsigned), ctx->stream);
    if (e != hipSuccess) return fail2(e, "memset");
    if (sweep > 0) {   // X <- V As: removes the drift the rotations accumulate in X
      GemmArgs g{};
      g.A = Y + mp; g.lda = ld; g.ta = 0;
      g.B = As; g.ldb = mp; g.tb = 1;          // As is symmetric: V As = V As^T, the k-contiguous form
      g.C = Y; g.ldc = ld;
      g.M = mp; g.N = mp; g.K = mp; g.alpha = 1.0; g.beta = 0.0; g.split_k = 1;
      if (launch_dgemm(ctx, g) != MLN_OK) return fail2(hipGetLastError(), "refresh");
    }
    for (int r = 0; r < rounds; ++r) {
      a.round = r;
      hipLaunchKernelGGL(k_jacobi_round, dim3((unsigned)(nbp / 2 > 0 ? nbp / 2 : 1)), dim3(JT), 0, ctx->stream, a);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail2(e, "k_jacobi_round");
    *h_rot = 1;
    e = hipMemcpyAsync(h_rot, d_rot, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail2(e, "jacobi sweep");
    converged = (*h_rot == 0);
  }
  if (converged && sweep > 0) {   // final X = V As for the Rayleigh quotients
    GemmArgs g{};
    g.A = Y + mp; g.lda = ld; g.ta = 0; g.B = As; g.ldb = mp; g.tb = 1; g.C = Y; g.ldc = ld;
    g.M = mp; g.N = mp; g.K = mp; g.alpha = 1.0; g.beta = 0.0; g.split_k = 1;
    if (launch_dgemm(ctx, g) != MLN_OK) return fail2(hipGetLastError(), "refresh");
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(As);
  (void)hipHostFree(h_rot);
  if (n_sweeps_out) *n_sweeps_out = sweep;
  if (!converged) {
    cleanup();
    mln_set_error(ctx, "eigh: Jacobi iteration did not converge in 40 sweeps");
    return MLN_ERR_NOCONV;
  }
  hipLaunchKernelGGL(k_eigh_values, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, Y, ld, mp, m, d_w);
  std::vector<double> w((size_t)m);
  e = hipMemcpyAsync(w.data(), d_w, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail(e, "eigenvalues");
  std::vector<int> perm((size_t)m);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return w[(size_t)x] < w[(size_t)y]; });
  for (int64_t j = 0; j < m; ++j) w_host[j] = w[(size_t)perm[(size_t)j]];
  e = hipMemcpyAsync(d_perm, perm.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) return fail(e, "perm upload");
  hipLaunchKernelGGL(k_eigh_gather_rows, dim3((unsigned)std::min<int64_t>((ldv + 255) / 256, 64), (unsigned)m), dim3(256),
                     0, ctx->stream, Y, ld, mp, m, d_perm, Vrows, ldv);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // perm is a host vector: finish before it dies
  if (e != hipSuccess) return fail(e, "gather");
  cleanup();
  return MLN_OK;
}
