// Diagnostics exported through the C ABI: device microbenchmarks used to state rooflines from
// measurement (fp64 MFMA issue rate, HBM read bandwidth) and to time the GEMM kernel in isolation.
#include <cstdlib>
#include <vector>

#include "mln_internal.h"

namespace {
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma_f64_peak(double* out, int iters, double seed) {
  v4d acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (v4d){seed, 0.0, 0.0, 0.0};
  double a = seed * (1.0 + threadIdx.x * 1e-3), b = seed * (1.0 - threadIdx.x * 1e-3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;  // keep the chain live
}

typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k_hbm_read(const d2* __restrict__ p, int64_t n2, double* out) {
  d2 s = (d2){0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    d2 v = __builtin_nontemporal_load(p + i);
    s.x += v.x;
    s.y += v.y;
  }
  if (s.x + s.y == 123.456) out[0] = s.x;
}
__global__ __launch_bounds__(512) void k_hbm_write(d2* __restrict__ p, int64_t n2, double v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = (d2){v, v + (double)i};
}
}  // namespace

// what: 0 = fp64 MFMA peak (TFLOP/s), 1 = HBM streaming read, 4 = HBM streaming write (GB/s over `bytes`)
extern "C" int mln_diag_peak(mln_ctx* ctx, int32_t what, int64_t bytes, double* result) {
  if (!ctx || !result) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  hipEvent_t e0, e1;
  MLN_HIP(ctx, hipEventCreate(&e0));
  MLN_HIP(ctx, hipEventCreate(&e1));
  double* out = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&out, 64));
  float ms = 0.f;
  if (what == 0 || what == 2 || what == 3) {
    // what = 0: non-trivial operands, 2 waves/SIMD; 2: all-zero operands (DVFS probe); 3: 4 waves/SIMD
    const int iters = 4000, grid = ctx->n_cu * (what == 3 ? 4 : 2);
    const double seed = (what == 2) ? 0.0 : 1.0;
    hipLaunchKernelGGL(k_mfma_f64_peak, dim3(grid), dim3(256), 0, ctx->stream, out, 10, seed);
    MLN_HIP(ctx, hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(k_mfma_f64_peak, dim3(grid), dim3(256), 0, ctx->stream, out, iters, seed);
    MLN_HIP(ctx, hipEventRecord(e1, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MLN_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * 4.0 * iters * 8.0 * 2.0 * 16 * 16 * 4;
    *result = flops / (ms * 1e-3) / 1e12;
  } else {
    if (bytes < (1 << 20)) bytes = 1 << 20;
    double* buf = nullptr;
    MLN_HIP(ctx, mln_dmalloc((void**)&buf, (size_t)bytes));
    MLN_HIP(ctx, hipMemsetAsync(buf, 0, (size_t)bytes, ctx->stream));
    const int grid = ctx->n_cu * 4;
    if (what == 4) {
      hipLaunchKernelGGL(k_hbm_write, dim3(grid), dim3(512), 0, ctx->stream, (d2*)buf, bytes / 16, 1.0);
      MLN_HIP(ctx, hipEventRecord(e0, ctx->stream));
      hipLaunchKernelGGL(k_hbm_write, dim3(grid), dim3(512), 0, ctx->stream, (d2*)buf, bytes / 16, 2.0);
      MLN_HIP(ctx, hipEventRecord(e1, ctx->stream));
    } else {
      hipLaunchKernelGGL(k_hbm_read, dim3(grid), dim3(512), 0, ctx->stream, (const d2*)buf, bytes / 16, out);
      MLN_HIP(ctx, hipEventRecord(e0, ctx->stream));
      hipLaunchKernelGGL(k_hbm_read, dim3(grid), dim3(512), 0, ctx->stream, (const d2*)buf, bytes / 16, out);
      MLN_HIP(ctx, hipEventRecord(e1, ctx->stream));
    }
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MLN_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *result = (double)bytes / (ms * 1e-3) / 1e9;
    (void)mln_dfree(buf);
  }
  (void)mln_dfree(out);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return MLN_OK;
}


// Times C = op(A) op(B) on random-free (zero-initialised + diagonal) device data; returns ms per call.
extern "C" int mln_diag_dgemm(mln_ctx* ctx, int32_t ta, int32_t tb, int64_t M, int64_t N, int64_t K,
                              int32_t lower_only, int32_t split_k, int32_t reps, double* ms_out) {
  if (!ctx || !ms_out || M < 1 || N < 1 || K < 1 || reps < 1) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t lda = ((ta ? M : K) + 15) / 16 * 16, ldb = ((tb ? K : N) + 15) / 16 * 16, ldc = (N + 15) / 16 * 16;
  const size_t a_bytes = sizeof(double) * (size_t)(ta ? K : M) * lda, b_bytes = sizeof(double) * (size_t)(tb ? N : K) * ldb;
  const int split = split_k > 1 ? split_k : 1;
  const size_t c_bytes = sizeof(double) * (size_t)M * ldc * split;
  double *A = nullptr, *B = nullptr, *Cm = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&A, a_bytes));
  MLN_HIP(ctx, mln_dmalloc((void**)&B, b_bytes));
  MLN_HIP(ctx, mln_dmalloc((void**)&Cm, c_bytes));
  // non-trivial data: fill with a repeating host pattern (random-like, avoids the zero-data DVFS bonus)
  std::vector<double> pat(1 << 20);
  unsigned long long s = 88172645463325252ULL;
  for (auto& v : pat) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
  for (size_t off = 0; off < a_bytes; off += pat.size() * 8)
    MLN_HIP(ctx, hipMemcpyAsync((char*)A + off, pat.data(), std::min(pat.size() * 8, a_bytes - off), hipMemcpyHostToDevice, ctx->stream));
  for (size_t off = 0; off < b_bytes; off += pat.size() * 8)
    MLN_HIP(ctx, hipMemcpyAsync((char*)B + off, pat.data(), std::min(pat.size() * 8, b_bytes - off), hipMemcpyHostToDevice, ctx->stream));
  GemmArgs g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = Cm; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.alpha = 1.0; g.beta = 0.0; g.ta = ta; g.tb = tb; g.lower_only = lower_only; g.split_k = split;
  g.c_split_stride = (int64_t)M * ldc;
  hipEvent_t e0, e1;
  MLN_HIP(ctx, hipEventCreate(&e0));
  MLN_HIP(ctx, hipEventCreate(&e1));
  MLN_TRY(launch_dgemm(ctx, g));
  MLN_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int r = 0; r < reps; ++r) MLN_TRY(launch_dgemm(ctx, g));
  MLN_HIP(ctx, hipEventRecord(e1, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  MLN_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / reps;
  (void)mln_dfree(A); (void)mln_dfree(B); (void)mln_dfree(Cm);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return MLN_OK;
}

// The mixed-size pipelined kernel against the single-size kernels on the same operands: out[0] = largest |difference| of
// the two results (every element sums its k in the same order under either tiling: 0 is the expected value), out[1] =
// largest |value|.  kmode / lower_only as in GemmArgs; any_size: also shapes below the policy's threshold.
void dgemm_set_mix(int mode);
// (lower_only: only the elements BOTH tilings are asked to produce -- a 128-wide diagonal tile also fills its upper quadrant)
__global__ void k_diag_absdiff(const double* __restrict__ a, const double* __restrict__ b, int64_t count, int64_t ldc, int lower_only,
                               double* __restrict__ out) {
  double d = 0.0, v = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / ldc, col = i % ldc;
    if ((lower_only == 1 && col > row) || (lower_only == 2 && col / 128 >= row / 128) || (lower_only == 3 && col < row)) continue;
    const double x = a[i], y = b[i];
    const double e = (x == y) ? 0.0 : ((x != x || y != y) ? INFINITY : fabs(x - y));
    d = e > d ? e : d;
    v = fabs(x) > v ? fabs(x) : v;
  }
  for (int o = 32; o > 0; o >>= 1) { d = fmax(d, __shfl_down(d, o)); v = fmax(v, __shfl_down(v, o)); }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(d));       // non-negative doubles order as integers
    atomicMax(reinterpret_cast<unsigned long long*>(out + 1), (unsigned long long)__double_as_longlong(v));
  }
}
// zero the strictly upper (which = 0) or strictly lower (1) triangle of a stored rows x cols matrix
__global__ void k_diag_zero_triangle(double* __restrict__ P, int64_t rows, int64_t cols, int64_t ld, int which) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    if ((which == 0 && c > r) || (which == 1 && r > c)) P[r * ld + c] = 0.0;
  }
}
extern "C" int mln_diag_dgemm_compare(mln_ctx* ctx, int32_t ta, int32_t tb, int64_t M, int64_t N, int64_t K, int32_t lower_only,
                                      int32_t kmode, double beta, int32_t any_size, double* out) {
  if (!ctx || !out || M < 1 || N < 1 || K < 1) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t lda = ((ta ? M : K) + 15) / 16 * 16, ldb = ((tb ? K : N) + 15) / 16 * 16, ldc = (N + 15) / 16 * 16;
  const size_t a_bytes = sizeof(double) * (size_t)(ta ? K : M) * lda, b_bytes = sizeof(double) * (size_t)(tb ? N : K) * ldb;
  const size_t c_count = (size_t)M * ldc;
  double *A = nullptr, *B = nullptr, *C0 = nullptr, *C1 = nullptr, *res = nullptr;
  // any_size == 2 (round 5): the BATCH dimension -- two products on different operands, once as two launches, once as one
  // launch with batch = 2; everything twice as large, kmode / lower_only as given
  const bool batch_mode = any_size == 2;
  const size_t nb = batch_mode ? 2 : 1;
  MLN_HIP(ctx, mln_dmalloc((void**)&A, a_bytes * nb));
  MLN_HIP(ctx, mln_dmalloc((void**)&B, b_bytes * nb));
  MLN_HIP(ctx, mln_dmalloc((void**)&C0, c_count * 8 * nb));
  MLN_HIP(ctx, mln_dmalloc((void**)&C1, c_count * 8 * nb));
  MLN_HIP(ctx, mln_dmalloc((void**)&res, 16));
  std::vector<double> pat((1 << 20) + 7);
  unsigned long long s = 88172645463325252ULL;
  for (auto& v : pat) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
  auto fill = [&](double* p, size_t bytes, size_t phase) -> hipError_t {
    for (size_t off = 0; off < bytes; off += (pat.size() - phase) * 8) {
      hipError_t e = hipMemcpyAsync((char*)p + off, pat.data() + phase, std::min((pat.size() - phase) * 8, bytes - off), hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  };
  MLN_HIP(ctx, fill(A, a_bytes * nb, 0));
  MLN_HIP(ctx, fill(B, b_bytes * nb, 3));
  MLN_HIP(ctx, fill(C0, c_count * 8 * nb, 5));     // the untouched part (upper triangle, beta * C) is the same in both
  MLN_HIP(ctx, fill(C1, c_count * 8 * nb, 5));
  MLN_HIP(ctx, hipMemsetAsync(res, 0, 16, ctx->stream));
  // the triangular K ranges are defined on 128-wide blocks for the 128-wide tiles and on 64-wide ones for the 64-wide: the
  // operand has to BE triangular for the two to mean the same product (every caller's is)
  for (size_t b = 0; b < nb; ++b) {
    double* Ab = A + b * (a_bytes / 8); double* Bb = B + b * (b_bytes / 8);
    if (kmode == 3 || kmode == 7)      // op(A)[row, k] = 0 for k > row
      hipLaunchKernelGGL(k_diag_zero_triangle, dim3(1024), dim3(256), 0, ctx->stream, Ab, ta ? K : M, ta ? M : K, lda, ta ? 1 : 0);
    if (kmode == 4)                    // op(B)[k, col] = 0 for k > col
      hipLaunchKernelGGL(k_diag_zero_triangle, dim3(1024), dim3(256), 0, ctx->stream, Bb, tb ? N : K, tb ? K : N, ldb, tb ? 0 : 1);
    if (kmode == 7)                    // op(B)[k, col] = 0 for k < col
      hipLaunchKernelGGL(k_diag_zero_triangle, dim3(1024), dim3(256), 0, ctx->stream, Bb, tb ? N : K, tb ? K : N, ldb, tb ? 1 : 0);
  }
  GemmArgs g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.alpha = 0.75; g.beta = beta; g.ta = ta; g.tb = tb; g.lower_only = lower_only; g.split_k = 1; g.kmode = kmode;
  int rc = MLN_OK;
  if (batch_mode) {
    for (size_t b = 0; b < nb && rc == MLN_OK; ++b) {
      GemmArgs h = g;
      h.A = A + b * (a_bytes / 8); h.B = B + b * (b_bytes / 8); h.C = C0 + b * c_count;
      rc = launch_dgemm(ctx, h);
    }
    g.C = C1; g.batch = 2; g.bsa = (int64_t)(a_bytes / 8); g.bsb = (int64_t)(b_bytes / 8); g.bsc = (int64_t)c_count;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
  } else {
    dgemm_set_mix(0);
    g.C = C0;
    rc = launch_dgemm(ctx, g);
    dgemm_set_mix(any_size ? 1 : -1);
    g.C = C1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
    dgemm_set_mix(-1);
  }
  if (rc == MLN_OK) {
    hipLaunchKernelGGL(k_diag_absdiff, dim3(1024), dim3(256), 0, ctx->stream, C0, C1, (int64_t)(c_count * nb), ldc, batch_mode ? 0 : (int)lower_only, res);
    hipError_t e = hipMemcpyAsync(out, res, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "dgemm compare", __FILE__, __LINE__);
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(A); (void)mln_dfree(B); (void)mln_dfree(C0); (void)mln_dfree(C1); (void)mln_dfree(res);
  return rc;
}

// ---- stream-overlap probe: does a second stream's MFMA-bound GEMM / latency-bound Cholesky run concurrently
// with the VALU-bound kernel-matrix pass?  out[0..5] = ms: K alone, Gram alone, K||Gram, chol alone, K||chol,
// and (K_s rows first) is not modelled here.
#include <chrono>
#include <thread>
static double wall_ms() {
  return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
extern "C" int mln_diag_overlap(mln_ctx* ctx, int64_t n, int64_t m, int32_t d, int64_t gram_rows, double* out) {
  if (!ctx || !out) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  mln_kernel_desc kd{};
  mln_leaf leaf{}; leaf.kind = MLN_K_MATERN52; leaf.ndims = 0; leaf.ls = 20.0; leaf.alpha = 1.0; leaf.dims = nullptr;
  mln_tok tok{}; tok.op = MLN_OP_LEAF; tok.leaf = 0;
  kd.n_leaves = 1; kd.n_toks = 1; kd.leaves = &leaf; kd.toks = &tok;
  DevCov cov;
  MLN_TRY(mln_lower_cov(ctx, &kd, d, &cov));
  const int64_t ld = (m + 15) / 16 * 16;
  double *X = nullptr, *Xu = nullptr, *K = nullptr, *G = nullptr, *A = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&X, sizeof(double) * n * d));
  MLN_HIP(ctx, mln_dmalloc((void**)&Xu, sizeof(double) * m * d));
  MLN_HIP(ctx, mln_dmalloc((void**)&K, sizeof(double) * n * ld));
  MLN_HIP(ctx, mln_dmalloc((void**)&G, sizeof(double) * m * ld));
  MLN_HIP(ctx, mln_dmalloc((void**)&A, sizeof(double) * m * ld));
  std::vector<double> pat((size_t)1 << 20);
  unsigned long long s = 88172645463325252ULL;
  for (auto& v : pat) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = 3.0 * ((double)(s >> 11) / 9007199254740992.0 - 0.5); }
  for (size_t off = 0; off < sizeof(double) * (size_t)n * d; off += pat.size() * 8)
    MLN_HIP(ctx, hipMemcpyAsync((char*)X + off, pat.data(), std::min(pat.size() * 8, sizeof(double) * (size_t)n * d - off), hipMemcpyHostToDevice, ctx->stream));
  MLN_HIP(ctx, hipMemcpyAsync(Xu, pat.data() + 777, sizeof(double) * m * d, hipMemcpyHostToDevice, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  mln_ctx side = *ctx;   // second stream, own status word
  side.scratch = nullptr; side.scratch_bytes = 0; side.err.clear();
  MLN_HIP(ctx, hipStreamCreateWithFlags(&side.stream, hipStreamNonBlocking));
  MLN_HIP(ctx, mln_dmalloc((void**)&side.d_info, 4 * sizeof(int)));
  auto run_k = [&]() { (void)launch_kernel_matrix(ctx, cov, X, n, Xu, m, d, K, ld, 0.0); (void)hipStreamSynchronize(ctx->stream); };
  auto run_gram = [&](mln_ctx* c) {
    GemmArgs g{};
    g.A = K; g.lda = ld * (n / gram_rows); g.B = K; g.ldb = g.lda; g.C = G; g.ldc = ld;
    g.M = m; g.N = m; g.K = gram_rows; g.alpha = 1.0; g.beta = 0.0; g.ta = 1; g.tb = 0; g.lower_only = 1; g.split_k = 1;
    (void)launch_dgemm(c, g);
    (void)hipStreamSynchronize(c->stream);
  };
  auto run_chol = [&](mln_ctx* c) {
    (void)launch_kernel_matrix(c, cov, Xu, m, Xu, m, d, A, ld, 1.0);
    (void)dev_cholesky_lower(c, A, m, ld);
  };
  run_k(); run_gram(ctx); run_chol(ctx);   // warm up
  double t0 = wall_ms(); run_k(); out[0] = wall_ms() - t0;
  t0 = wall_ms(); run_gram(&side); out[1] = wall_ms() - t0;
  t0 = wall_ms(); { std::thread th([&]() { (void)hipSetDevice(ctx->device); run_gram(&side); }); run_k(); th.join(); } out[2] = wall_ms() - t0;
  t0 = wall_ms(); run_chol(&side); out[3] = wall_ms() - t0;
  t0 = wall_ms(); { std::thread th([&]() { (void)hipSetDevice(ctx->device); run_chol(&side); }); run_k(); th.join(); } out[4] = wall_ms() - t0;
  t0 = wall_ms(); { std::thread th([&]() { (void)hipSetDevice(ctx->device); run_chol(&side); run_gram(&side); }); run_k(); th.join(); } out[5] = wall_ms() - t0;
  (void)hipDeviceSynchronize();
  if (side.scratch) (void)mln_dfree(side.scratch);
  (void)mln_dfree(side.d_info);
  (void)hipStreamDestroy(side.stream);
  for (void* p : {(void*)X, (void*)Xu, (void*)K, (void*)G, (void*)A}) (void)mln_dfree(p);
  return MLN_OK;
}
