// C ABI of libmellon_hip.so (see include/mellon_hip.h for the reference citations per entry): context, memory,
// stand-alone operators and prediction.  The fit handle lives in api_fit.hip, api_precond.hip, api_solve.hip; the
// FunctionEstimator's noise models in api_noise.hip; what they share in api_internal.h.
#include "api_internal.h"

// ---- errors -------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void mln_set_error(mln_ctx* ctx, const std::string& msg) {
  g_last_error = msg;
  if (ctx) ctx->err = msg;
}

int mln_hip_fail(mln_ctx* ctx, hipError_t e, const char* what, const char* file, int line) {
  mln_set_error(ctx, std::string("HIP error ") + hipGetErrorString(e) + " in " + what + " (" + file + ":" +
                         std::to_string(line) + ")");
  (void)hipGetLastError();
  return MLN_ERR_HIP;
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- pointer helpers ------------------------------------------------------------------------------
bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int mln_scratch(mln_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->scratch_bytes) {
    if (ctx->scratch) { MLN_HIP(ctx, hipStreamSynchronize(ctx->stream)); MLN_HIP(ctx, mln_dfree(ctx->scratch)); ctx->scratch = nullptr; }
    size_t want = bytes + bytes / 4 + 4096;
    MLN_HIP(ctx, mln_dmalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return MLN_OK;
}

// ---- covariance lowering ---------------------------------------------------------------------------
int mln_lower_cov(mln_ctx* ctx, const mln_kernel_desc* cov, int d, DevCov* out) {
  if (!cov || !cov->leaves || !cov->toks) { mln_set_error(ctx, "null covariance descriptor"); return MLN_ERR_ARG; }
  if (cov->n_leaves < 1 || cov->n_leaves > MLN_MAX_LEAVES || cov->n_toks < 1 || cov->n_toks > MLN_MAX_TOKS) {
    mln_set_error(ctx, "covariance descriptor: too many leaves / tokens for the device program");
    return MLN_ERR_UNSUPPORTED;
  }
  std::memset(out, 0, sizeof(DevCov));
  out->n_leaves = cov->n_leaves;
  out->n_toks = cov->n_toks;
  int off = 0;
  for (int l = 0; l < cov->n_leaves; ++l) {
    const mln_leaf& lf = cov->leaves[l];
    if (lf.kind < MLN_K_MATERN32 || lf.kind > MLN_K_DISTANCE) { mln_set_error(ctx, "unknown kernel kind"); return MLN_ERR_ARG; }
    if (lf.ndims < 0 || off + lf.ndims > MLN_MAX_DIMS || (lf.ndims > 0 && !lf.dims)) {
      mln_set_error(ctx, "covariance descriptor: active dims overflow"); return MLN_ERR_UNSUPPORTED;
    }
    out->leaves[l].kind = lf.kind; out->leaves[l].ndims = lf.ndims; out->leaves[l].dims_off = off;
    out->leaves[l].ls = lf.ls; out->leaves[l].alpha = lf.alpha;
    out->leaves[l].alpha_inv_ls[0] = lf.alpha; out->leaves[l].alpha_inv_ls[1] = 1.0 / lf.ls;
    for (int k = 0; k < lf.ndims; ++k) {
      if (lf.dims[k] < 0 || lf.dims[k] >= d) { mln_set_error(ctx, "active dim out of range"); return MLN_ERR_SHAPE; }
      out->dims[off + k] = (short)lf.dims[k];
    }
    off += lf.ndims;
  }
  int depth = 0;
  for (int t = 0; t < cov->n_toks; ++t) {
    const mln_tok& tk = cov->toks[t];
    out->tok_op[t] = tk.op; out->tok_leaf[t] = tk.leaf; out->tok_val[t] = tk.value;
    if (tk.op == MLN_OP_LEAF) {
      if (tk.leaf < 0 || tk.leaf >= cov->n_leaves) { mln_set_error(ctx, "token references unknown leaf"); return MLN_ERR_ARG; }
      ++depth;
    } else if (tk.op == MLN_OP_CONST) {
      ++depth;
    } else if (tk.op == MLN_OP_ADD || tk.op == MLN_OP_MUL || tk.op == MLN_OP_POW) {
      if (depth < 2) { mln_set_error(ctx, "malformed covariance program"); return MLN_ERR_ARG; }
      --depth;
    } else { mln_set_error(ctx, "unknown token op"); return MLN_ERR_ARG; }
    if (depth > 3) { mln_set_error(ctx, "covariance expression nests deeper than the device stack (3)"); return MLN_ERR_UNSUPPORTED; }
  }
  if (depth != 1) { mln_set_error(ctx, "malformed covariance program"); return MLN_ERR_ARG; }
  if (cov->n_toks == 1 && cov->toks[0].op != MLN_OP_LEAF) { mln_set_error(ctx, "constant covariance"); return MLN_ERR_ARG; }
  return MLN_OK;
}

// ---- context ----------------------------------------------------------------------------------------
extern "C" const char* mln_last_error(mln_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

extern "C" int mln_ctx_create(int device, mln_ctx** out) {
  if (!out) return MLN_ERR_ARG;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    mln_set_error(nullptr, "no HIP device visible: libmellon_hip has no CPU fallback");
    return MLN_ERR_HIP;
  }
  if (device < 0 || device >= count) { mln_set_error(nullptr, "device index out of range"); return MLN_ERR_ARG; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { mln_set_error(nullptr, "hipGetDeviceProperties failed"); return MLN_ERR_HIP; }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    mln_set_error(nullptr, std::string("device is ") + prop.gcnArchName + "; this library is built for gfx950 only");
    return MLN_ERR_UNSUPPORTED;
  }
  mln_ctx* ctx = new mln_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess ||
      mln_dmalloc((void**)&ctx->d_info, 4 * sizeof(int)) != hipSuccess) {
    mln_set_error(nullptr, "failed to initialise the HIP context");
    delete ctx;
    return MLN_ERR_HIP;
  }
  *out = ctx;
  return MLN_OK;
}

void fit_release_copy_lane(mln_ctx* ctx);   // api_fit.hip: the context's copy stream and events

extern "C" void mln_ctx_destroy(mln_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  comm_release(ctx);
  fit_release_copy_lane(ctx);
  if (ctx->scratch) (void)mln_dfree(ctx->scratch);
  if (ctx->d_info) (void)mln_dfree(ctx->d_info);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int mln_device_info(mln_ctx* ctx, char* name, int name_cap, int* n_cu, int64_t* mem_bytes) {
  if (!ctx) return MLN_ERR_ARG;
  hipDeviceProp_t prop;
  MLN_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_cap > 0) { std::strncpy(name, prop.gcnArchName, name_cap - 1); name[name_cap - 1] = 0; }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (mem_bytes) *mem_bytes = (int64_t)prop.totalGlobalMem;
  return MLN_OK;
}

extern "C" int mln_device_count(int* count_out) {
  if (!count_out) return MLN_ERR_ARG;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
  *count_out = count;
  return MLN_OK;
}

extern "C" int mln_synchronize(mln_ctx* ctx) {
  if (!ctx) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MLN_OK;
}

extern "C" int mln_malloc(mln_ctx* ctx, int64_t bytes, void** dev_ptr) {
  if (!ctx || !dev_ptr || bytes < 0) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_HIP(ctx, hipMalloc(dev_ptr, (size_t)(bytes > 0 ? bytes : 8)));
  return MLN_OK;
}

extern "C" int mln_free(mln_ctx* ctx, void* dev_ptr) {
  if (!ctx) return MLN_ERR_ARG;
  if (dev_ptr) { MLN_HIP(ctx, hipStreamSynchronize(ctx->stream)); MLN_HIP(ctx, hipFree(dev_ptr)); }
  return MLN_OK;
}

extern "C" int mln_host_register(mln_ctx* ctx, const void* host_ptr, int64_t bytes) {
  if (!ctx || !host_ptr || bytes < 1) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_HIP(ctx, hipHostRegister(const_cast<void*>(host_ptr), (size_t)bytes, hipHostRegisterDefault));
  return MLN_OK;
}

extern "C" int mln_host_unregister(mln_ctx* ctx, const void* host_ptr) {
  if (!ctx || !host_ptr) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MLN_HIP(ctx, hipHostUnregister(const_cast<void*>(host_ptr)));
  return MLN_OK;
}

extern "C" int mln_release_cached_memory(void) {
  mln_dcache_flush();
  return MLN_OK;
}

extern "C" int mln_memcpy(mln_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  if (!ctx || bytes < 0 || (bytes > 0 && (!dst || !src))) return MLN_ERR_ARG;
  if (bytes == 0) return MLN_OK;
  MLN_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MLN_OK;
}

// ---- communicator (comm.hip) ---------------------------------------------------------------------------
int dev_allreduce(mln_ctx* ctx, double* dev, int64_t count) { return comm_allreduce(ctx, dev, count); }
int dev_bcast0(mln_ctx* ctx, double* dev, int64_t count) { return comm_bcast0(ctx, dev, count); }

extern "C" int mln_comm_allreduce_sum(mln_ctx* ctx, double* buf, int64_t count) {
  if (!ctx || (count > 0 && !buf)) return MLN_ERR_ARG;
  if (!ctx->comm && !ctx->loop && ctx->n_ranks <= 1) return MLN_OK;     // (n_ranks > 1 without either: host-staged)
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevOut o;
  MLN_TRY(o.init(ctx, buf, (size_t)count, true));
  MLN_TRY(dev_allreduce(ctx, o.dev, count));
  return o.commit();
}

// ---- stand-alone operators -------------------------------------------------------------------------
extern "C" int mln_kernel_matrix(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n,
                                 const double* y, int64_t m, int32_t d, double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n < 0 || m < 0 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (n == 0 || m == 0) return MLN_OK;
  if (!x || !y || !out) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  DevIn dx, dy;
  DevOut o;
  MLN_TRY(dx.init(ctx, x, (size_t)n * d));
  MLN_TRY(dy.init(ctx, y, (size_t)m * d));
  MLN_TRY(o.init(ctx, out, (size_t)n * m));
  MLN_TRY(launch_kernel_matrix(ctx, dc, dx.dev, n, dy.dev, m, d, o.dev, m, 0.0));
  return o.commit();
}

int reject_distance_leaf(mln_ctx* ctx, const DevCov& dc) {
  for (int l = 0; l < dc.n_leaves; ++l)
    if (dc.leaves[l].kind == MLN_K_DISTANCE) {
      mln_set_error(ctx, "MLN_K_DISTANCE is a value-only leaf: no derivatives");
      return MLN_ERR_UNSUPPORTED;
    }
  return MLN_OK;
}

extern "C" int mln_kernel_grad(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n,
                               const double* y, int64_t m, int32_t d, double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n < 0 || m < 0 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (n == 0 || m == 0) return MLN_OK;
  if (!x || !y || !out) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  MLN_TRY(reject_distance_leaf(ctx, dc));
  DevIn dx, dy;
  DevOut o;
  MLN_TRY(dx.init(ctx, x, (size_t)n * d));
  MLN_TRY(dy.init(ctx, y, (size_t)m * d));
  MLN_TRY(o.init(ctx, out, (size_t)n * m * d));
  MLN_TRY(launch_kernel_grad(ctx, dc, dx.dev, n, dy.dev, m, d, 0, o.dev));
  return o.commit();
}

extern "C" int mln_predict_gradient(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new,
                                    int32_t d, const double* centers, int64_t m, const double* W, double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n_new < 0 || m < 0 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (n_new == 0) return MLN_OK;
  if (!xnew || !out || (m > 0 && (!centers || !W))) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  MLN_TRY(reject_distance_leaf(ctx, dc));
  DevIn dx, dc_, dw;
  DevOut o;
  MLN_TRY(dx.init(ctx, xnew, (size_t)n_new * d));
  MLN_TRY(dc_.init(ctx, centers, (size_t)m * d));
  MLN_TRY(dw.init(ctx, W, (size_t)m));
  MLN_TRY(o.init(ctx, out, (size_t)n_new * d));
  MLN_TRY(launch_predict_gradient(ctx, dc, dx.dev, n_new, dc_.dev, m, d, dw.dev, o.dev));
  return o.commit();
}

extern "C" int mln_predict_hessian(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new,
                                   int32_t d, const double* centers, int64_t m, const double* W, double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n_new < 0 || m < 1 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (n_new == 0) return MLN_OK;
  if (!xnew || !out || !centers || !W) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  MLN_TRY(reject_distance_leaf(ctx, dc));
  DevIn dx, dc_, dw;
  DevOut o;
  MLN_TRY(dx.init(ctx, xnew, (size_t)n_new * d));
  MLN_TRY(dc_.init(ctx, centers, (size_t)m * d));
  MLN_TRY(dw.init(ctx, W, (size_t)m));
  MLN_TRY(o.init(ctx, out, (size_t)n_new * d * d));
  MLN_TRY(launch_predict_hessian(ctx, dc, dx.dev, n_new, dc_.dev, m, d, dw.dev, o.dev));
  return o.commit();
}


// G (m x m) = cov(x, xu)^T cov(x, xu): the B^T B of the landmark leverage (conditional.py:660-685) without
// the n x m matrix leaving the device; rows in chunks, all-reduced over ranks.
extern "C" int mln_kernel_gram(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n, int32_t d,
                               const double* xu, int64_t m, double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n < 0 || m < 1 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (!xu || !out || (n > 0 && !x)) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  DevIn dx, du;
  DevOut o;
  if (n > 0) MLN_TRY(dx.init(ctx, x, (size_t)n * d));
  MLN_TRY(du.init(ctx, xu, (size_t)m * d));
  MLN_TRY(o.init(ctx, out, (size_t)m * m));
  const int64_t ld = pad16(m);
  const int64_t chunk = (n < 65536) ? (n > 0 ? n : 1) : 65536;
  double *K = nullptr, *G = nullptr, *acc = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&K, sizeof(double) * (size_t)chunk * ld));
  MLN_HIP(ctx, mln_dmalloc((void**)&G, sizeof(double) * (size_t)m * ld));
  MLN_HIP(ctx, mln_dmalloc((void**)&acc, sizeof(double) * (size_t)m * ld));
  int rc = (hipMemsetAsync(acc, 0, sizeof(double) * (size_t)m * ld, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
  bool any = false;
  for (int64_t r0 = 0; (r0 < n || !any) && rc == MLN_OK; r0 += chunk) {   // at least one (possibly empty) round: collectives
    const int64_t rows = (n - r0 < chunk) ? (n - r0 > 0 ? n - r0 : 0) : chunk;
    if (rows > 0) rc = launch_kernel_matrix(ctx, dc, dx.dev + r0 * d, rows, du.dev, m, d, K, ld, 0.0);
    if (rc == MLN_OK) rc = gram_of(ctx, K, ld, rows, m, 1.0, G, ld);
    if (rc == MLN_OK) rc = launch_axpby(ctx, m * ld, 1.0, G, 1.0, acc);
    any = true;
    if (n == 0) break;
  }
  if (rc == MLN_OK) rc = launch_copy_block(ctx, acc, ld, o.dev, m, m, m);
  if (rc == MLN_OK) rc = o.commit();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(K); (void)mln_dfree(G); (void)mln_dfree(acc);
  return rc;
}

extern "C" int mln_nn_distances(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int32_t d,
                                int64_t self_offset, double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n < 0 || m < 0 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (n == 0) return MLN_OK;
  if (!x || !y || !out) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevIn dx, dy;
  DevOut o;
  MLN_TRY(dx.init(ctx, x, (size_t)n * d));
  // (the same buffer only if it also has the same extent: a shard that starts at row 0 of all cells shares their address)
  if (y == x && m == n) dy.dev = dx.dev, dy.ctx = ctx; else MLN_TRY(dy.init(ctx, y, (size_t)m * d));
  MLN_TRY(o.init(ctx, out, (size_t)n));
  MLN_TRY(launch_nn_distances(ctx, dx.dev, n, dy.dev, m, d, self_offset, o.dev));
  return o.commit();
}

int64_t pad16(int64_t m) { return ((m + 15) / 16) * 16; }

extern "C" int mln_chol_lower(mln_ctx* ctx, double* A, int64_t m, double add_diag) {
  if (!ctx || (m > 0 && !A)) return MLN_ERR_ARG;
  if (m < 0 || m > 65535) { mln_set_error(ctx, "cholesky: m out of range"); return MLN_ERR_SHAPE; }
  if (m == 0) return MLN_OK;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevOut o;
  MLN_TRY(o.init(ctx, A, (size_t)m * m, true));
  if (add_diag != 0.0) MLN_TRY(launch_add_diag(ctx, o.dev, m, m, add_diag));
  MLN_TRY(dev_cholesky_lower(ctx, o.dev, m, m));
  return o.commit();
}

extern "C" int mln_trsm_lower(mln_ctx* ctx, const double* Lf, int64_t m, int32_t trans, double* B, int64_t p) {
  if (!ctx || (m > 0 && (!Lf || (p > 0 && !B)))) return MLN_ERR_ARG;
  if (m < 0 || p < 0) return MLN_ERR_SHAPE;
  if (m == 0 || p == 0) return MLN_OK;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevIn dl;
  DevOut b;
  MLN_TRY(dl.init(ctx, Lf, (size_t)m * m));
  MLN_TRY(b.init(ctx, B, (size_t)m * p, true));
  TriInv t;
  MLN_TRY(triinv_build(ctx, dl.dev, m, m, trans == 0, trans != 0, &t));
  int rc = trans == 0 ? triinv_solve_left(ctx, t, b.dev, p, p) : triinv_solve_left_T(ctx, t, b.dev, p, p);
  if (rc == MLN_OK) rc = b.commit();
  (void)hipStreamSynchronize(ctx->stream);
  triinv_free(&t);
  return rc;
}

__global__ void k_ewise(int op, const double* __restrict__ a, const double* __restrict__ b, double scalar,
                        double* __restrict__ out, int64_t count) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const double l = a[i], r = b ? b[i] : scalar;
    out[i] = op == MLN_OP_ADD ? l + r : (op == MLN_OP_MUL ? l * r : pow(l, r));
  }
}

extern "C" int mln_ewise(mln_ctx* ctx, int32_t op, const double* a, const double* b, double scalar, double* out,
                         int64_t count) {
  if (!ctx || count < 0 || (count > 0 && (!a || !out))) return MLN_ERR_ARG;
  if (op != MLN_OP_ADD && op != MLN_OP_MUL && op != MLN_OP_POW) { mln_set_error(ctx, "mln_ewise: unknown operation"); return MLN_ERR_ARG; }
  if (count == 0) return MLN_OK;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevIn da, db;
  DevOut o;
  const bool alias = (out == a);
  MLN_TRY(o.init(ctx, out, (size_t)count, alias));
  if (alias && is_device_ptr(a)) da.dev = a; else if (alias) da.dev = o.dev; else MLN_TRY(da.init(ctx, a, (size_t)count));
  if (b) MLN_TRY(db.init(ctx, b, (size_t)count));
  const int64_t blocks = std::min<int64_t>((count + 255) / 256, 65535);
  hipLaunchKernelGGL(k_ewise, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, op, da.dev, b ? db.dev : nullptr, scalar, o.dev, count);
  MLN_HIP(ctx, hipGetLastError());
  return o.commit();
}

extern "C" int mln_gemm(mln_ctx* ctx, int32_t ta, int32_t tb, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* Cm, int64_t ldc) {
  if (!ctx || M < 0 || N < 0 || K < 0) return MLN_ERR_ARG;
  if (M == 0 || N == 0) return MLN_OK;
  if (!A || !B || !Cm) return MLN_ERR_ARG;
  const int64_t ar = ta ? K : M, ac = ta ? M : K, br = tb ? N : K, bc = tb ? K : N;
  if (lda < ac || ldb < bc || ldc < N) { mln_set_error(ctx, "mln_gemm: leading dimension too small"); return MLN_ERR_SHAPE; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  // host operands are staged compactly (their leading dimension becomes the column count)
  DevIn da, db;
  DevOut oc;
  const bool a_dev = is_device_ptr(A), b_dev = is_device_ptr(B), c_dev = is_device_ptr(Cm);
  std::vector<double> ha, hb, hc;
  const double* Ah = A; const double* Bh = B;
  int64_t lda_d = lda, ldb_d = ldb, ldc_d = ldc;
  if (!a_dev && lda != ac) { ha.resize((size_t)ar * ac); for (int64_t r = 0; r < ar; ++r) std::memcpy(&ha[(size_t)r * ac], A + r * lda, sizeof(double) * ac); Ah = ha.data(); }
  if (!b_dev && ldb != bc) { hb.resize((size_t)br * bc); for (int64_t r = 0; r < br; ++r) std::memcpy(&hb[(size_t)r * bc], B + r * ldb, sizeof(double) * bc); Bh = hb.data(); }
  if (!a_dev) lda_d = ac;
  if (!b_dev) ldb_d = bc;
  MLN_TRY(da.init(ctx, Ah, a_dev ? 1 : (size_t)ar * ac));
  MLN_TRY(db.init(ctx, Bh, b_dev ? 1 : (size_t)br * bc));
  if (a_dev) da.dev = A;
  if (b_dev) db.dev = B;
  double* Cd = Cm;
  double* c_owned = nullptr;
  if (!c_dev) {
    ldc_d = (N + 1) & ~(int64_t)1;
    MLN_HIP(ctx, mln_dmalloc((void**)&c_owned, sizeof(double) * (size_t)M * ldc_d));
    Cd = c_owned;
    if (beta != 0.0)
      MLN_HIP(ctx, hipMemcpy2DAsync(Cd, sizeof(double) * ldc_d, Cm, sizeof(double) * ldc, sizeof(double) * N, (size_t)M, hipMemcpyHostToDevice, ctx->stream));
  }
  GemmArgs g{};
  g.A = da.dev; g.lda = lda_d; g.B = db.dev; g.ldb = ldb_d; g.C = Cd; g.ldc = ldc_d;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.ta = ta ? 1 : 0; g.tb = tb ? 1 : 0;
  int rc = (K > 0) ? launch_dgemm(ctx, g) : MLN_OK;
  if (rc == MLN_OK && K == 0) {
    mln_set_error(ctx, "mln_gemm: K = 0");
    rc = MLN_ERR_SHAPE;
  }
  if (rc == MLN_OK && !c_dev) {
    hipError_t e = hipMemcpy2DAsync(Cm, sizeof(double) * ldc, Cd, sizeof(double) * ldc_d, sizeof(double) * N, (size_t)M, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "mln_gemm download", __FILE__, __LINE__);
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (c_owned) (void)mln_dfree(c_owned);
  return rc;
}

extern "C" int mln_eigh(mln_ctx* ctx, const double* A, int64_t m, double* w, double* V, int32_t* n_sweeps) {
  if (!ctx || (m > 0 && (!A || !w || !V))) return MLN_ERR_ARG;
  if (m < 0 || m > 32768) { mln_set_error(ctx, "eigh: m out of range"); return MLN_ERR_SHAPE; }
  if (n_sweeps) *n_sweeps = 0;
  if (m == 0) return MLN_OK;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevIn a;
  DevOut v;
  MLN_TRY(a.init(ctx, A, (size_t)m * m));
  MLN_TRY(v.init(ctx, V, (size_t)m * m));
  double* rows = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&rows, sizeof(double) * (size_t)m * m));
  std::vector<double> wh((size_t)m);
  int sweeps = 0;
  int rc = dev_eigh(ctx, a.dev, m, m, wh.data(), rows, m, &sweeps);
  if (rc == MLN_OK) rc = launch_transpose(ctx, rows, m, v.dev, m, m);   // eigenvectors as columns
  if (rc == MLN_OK) rc = v.commit();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(rows);
  if (rc != MLN_OK) return rc;
  if (n_sweeps) *n_sweeps = sweeps;
  if (is_device_ptr(w)) MLN_HIP(ctx, hipMemcpy(w, wh.data(), sizeof(double) * (size_t)m, hipMemcpyHostToDevice));
  else std::memcpy(w, wh.data(), sizeof(double) * (size_t)m);
  return MLN_OK;
}

// ---- prediction -----------------------------------------------------------------------------------
extern "C" int mln_predict_mean(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new,
                                int32_t d, const double* centers, int64_t m, const double* W, int64_t p,
                                double mu, double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n_new < 0 || m < 1 || p < 1 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (n_new == 0) return MLN_OK;
  if (!xnew || !centers || !W || !out) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  DevIn dx, dc_, dw;
  DevOut o;
  MLN_TRY(dx.init(ctx, xnew, (size_t)n_new * d));
  MLN_TRY(dc_.init(ctx, centers, (size_t)m * d));
  MLN_TRY(dw.init(ctx, W, (size_t)m * p));
  MLN_TRY(o.init(ctx, out, (size_t)n_new * p));
  if (p == 1) {
    MLN_TRY(launch_predict_mean1(ctx, dc, dx.dev, n_new, dc_.dev, m, d, dw.dev, mu, o.dev));
    return o.commit();
  }
  // p > 1 (FunctionEstimator): materialise row chunks of cov(Xnew, centers) and contract on the
  // matrix cores: out = mu + K W                                   conditional.py:651-658
  int64_t chunk = (int64_t)((1ull << 30) / (sizeof(double) * (size_t)m));
  chunk = (chunk / 128) * 128;
  if (chunk < 128) chunk = 128;
  if (chunk > n_new) chunk = n_new;
  double* Kc = nullptr;
  double* mus = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&Kc, sizeof(double) * (size_t)chunk * m));
  int rc = MLN_OK;
  if (mu != 0.0) {
    std::vector<double> h((size_t)(chunk * p), mu);
    rc = (mln_dmalloc((void**)&mus, sizeof(double) * h.size()) == hipSuccess &&
          hipMemcpy(mus, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
  }
  for (int64_t r0 = 0; r0 < n_new && rc == MLN_OK; r0 += chunk) {
    const int64_t rc_n = (n_new - r0 < chunk) ? (n_new - r0) : chunk;
    rc = launch_kernel_matrix(ctx, dc, dx.dev + r0 * d, rc_n, dc_.dev, m, d, Kc, m, 0.0);
    if (rc != MLN_OK) break;
    double* oc = o.dev + r0 * p;
    double beta = 0.0;
    if (mus) {
      rc = (hipMemcpyAsync(oc, mus, sizeof(double) * rc_n * p, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
      beta = 1.0;
    }
    if (rc != MLN_OK) break;
    GemmArgs g{};
    g.A = Kc; g.lda = m; g.B = dw.dev; g.ldb = p; g.C = oc; g.ldc = p;
    g.M = rc_n; g.N = p; g.K = m; g.alpha = 1.0; g.beta = beta; g.ta = 0; g.tb = 0;
    rc = launch_dgemm(ctx, g);
  }
  if (rc == MLN_OK) rc = o.commit();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(Kc);
  if (mus) (void)mln_dfree(mus);
  return rc;
}

// ---- predictive uncertainty (S8f rank 2) ------------------------------------------------------------
// covariance (conditional.py:409-440, 930-945):  k(x*,x*) - A A^T,  A = cov(x*, centers) Lf^-T
extern "C" int mln_predict_covariance(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew, int64_t n_new,
                                      int32_t d, const double* centers, int64_t m, const double* Lf, int32_t diag,
                                      double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n_new < 0 || m < 1 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (n_new == 0) return MLN_OK;
  if (!xnew || !centers || !Lf || !out) return MLN_ERR_ARG;
  if (!diag && n_new > 32768) { mln_set_error(ctx, "full covariance is limited to 32768 points"); return MLN_ERR_UNSUPPORTED; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  DevIn dx, dcen, dl;
  DevOut o;
  MLN_TRY(dx.init(ctx, xnew, (size_t)n_new * d));
  MLN_TRY(dcen.init(ctx, centers, (size_t)m * d));
  MLN_TRY(dl.init(ctx, Lf, (size_t)m * m));
  MLN_TRY(o.init(ctx, out, diag ? (size_t)n_new : (size_t)n_new * n_new));
  const int64_t ld = pad16(m);
  TriInv t;
  MLN_TRY(triinv_build(ctx, dl.dev, m, m, true, false, &t));
  int64_t chunk = diag ? (int64_t)((1ull << 30) / (sizeof(double) * (size_t)ld)) : n_new;
  if (chunk > n_new) chunk = n_new;
  if (chunk < 1) chunk = 1;
  double *A = nullptr, *kss = nullptr;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "predict_covariance", __FILE__, __LINE__); };
  chk(mln_dmalloc((void**)&A, sizeof(double) * (size_t)chunk * ld));
  chk(mln_dmalloc((void**)&kss, sizeof(double) * (size_t)chunk));
  for (int64_t r0 = 0; r0 < n_new && rc == MLN_OK; r0 += chunk) {
    const int64_t rows = (n_new - r0 < chunk) ? (n_new - r0) : chunk;
    rc = launch_kernel_matrix(ctx, dc, dx.dev + r0 * d, rows, dcen.dev, m, d, A, ld, 0.0);
    if (rc == MLN_OK) rc = triinv_solve_right_T(ctx, t, A, rows, ld);
    if (rc != MLN_OK) break;
    if (diag) {
      rc = launch_cov_diag(ctx, dc, dx.dev + r0 * d, rows, d, kss);
      if (rc == MLN_OK) rc = launch_row_sumsq(ctx, A, ld, rows, m, kss, -1.0, o.dev + r0);
    } else {
      rc = launch_kernel_matrix(ctx, dc, dx.dev, n_new, dx.dev, n_new, d, o.dev, n_new, 0.0);
      GemmArgs g{};
      g.A = A; g.lda = ld; g.B = A; g.ldb = ld; g.C = o.dev; g.ldc = n_new;
      g.M = n_new; g.N = n_new; g.K = m; g.alpha = -1.0; g.beta = 1.0; g.ta = 0; g.tb = 1;
      if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
    }
  }
  if (rc == MLN_OK) rc = o.commit();
  (void)hipStreamSynchronize(ctx->stream);
  triinv_free(&t);
  if (A) (void)mln_dfree(A);
  if (kss) (void)mln_dfree(kss);
  return rc;
}

// mean covariance (conditional.py:423-440, 947-963):  (K W)(K W)^T,  K = cov(x*, centers), W: m x q
extern "C" int mln_predict_mean_covariance(mln_ctx* ctx, const mln_kernel_desc* cov, const double* xnew,
                                           int64_t n_new, int32_t d, const double* centers, int64_t m,
                                           const double* W, int64_t q, int32_t diag, double* out) {
  if (!ctx) return MLN_ERR_ARG;
  if (n_new < 0 || m < 1 || d < 1 || q < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (n_new == 0) return MLN_OK;
  if (!xnew || !centers || !W || !out) return MLN_ERR_ARG;
  if (!diag && n_new > 32768) { mln_set_error(ctx, "full covariance is limited to 32768 points"); return MLN_ERR_UNSUPPORTED; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  DevIn dx, dcen, dw;
  DevOut o;
  MLN_TRY(dx.init(ctx, xnew, (size_t)n_new * d));
  MLN_TRY(dcen.init(ctx, centers, (size_t)m * d));
  MLN_TRY(dw.init(ctx, W, (size_t)m * q));
  MLN_TRY(o.init(ctx, out, diag ? (size_t)n_new : (size_t)n_new * n_new));
  const int64_t ldq = pad16(q);
  int64_t chunk = diag ? (int64_t)((1ull << 30) / (sizeof(double) * (size_t)(m + ldq))) : n_new;
  if (chunk > n_new) chunk = n_new;
  if (chunk < 1) chunk = 1;
  double *Kc = nullptr, *T = nullptr;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "predict_mean_covariance", __FILE__, __LINE__); };
  chk(mln_dmalloc((void**)&Kc, sizeof(double) * (size_t)chunk * m));
  chk(mln_dmalloc((void**)&T, sizeof(double) * (size_t)chunk * ldq));
  for (int64_t r0 = 0; r0 < n_new && rc == MLN_OK; r0 += chunk) {
    const int64_t rows = (n_new - r0 < chunk) ? (n_new - r0) : chunk;
    rc = launch_kernel_matrix(ctx, dc, dx.dev + r0 * d, rows, dcen.dev, m, d, Kc, m, 0.0);
    GemmArgs g{};
    g.A = Kc; g.lda = m; g.B = dw.dev; g.ldb = q; g.C = T; g.ldc = ldq;
    g.M = rows; g.N = q; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
    if (rc != MLN_OK) break;
    if (diag) {
      rc = launch_row_sumsq(ctx, T, ldq, rows, q, nullptr, 1.0, o.dev + r0);
    } else {
      GemmArgs h{};
      h.A = T; h.lda = ldq; h.B = T; h.ldb = ldq; h.C = o.dev; h.ldc = n_new;
      h.M = n_new; h.N = n_new; h.K = q; h.alpha = 1.0; h.beta = 0.0; h.ta = 0; h.tb = 1;
      rc = launch_dgemm(ctx, h);
    }
  }
  if (rc == MLN_OK) rc = o.commit();
  (void)hipStreamSynchronize(ctx->stream);
  if (Kc) (void)mln_dfree(Kc);
  if (T) (void)mln_dfree(T);
  return rc;
}

