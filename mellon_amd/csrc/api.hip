// C ABI of libmellon_hip.so (see include/mellon_hip.h for the reference citations per entry).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "linalg.h"
#include "mln_internal.h"
#include "objective.h"
#include "precond_rebuild.h"
#include "solver.h"

void mln_dfree_defer(std::vector<void*>* sink);   // alloc.hip: frees of the calling thread are collected instead of performed

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

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- pointer helpers ------------------------------------------------------------------------------
static bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// Read-only input that may live on host or device.
struct DevIn {
  mln_ctx* ctx;
  const double* dev = nullptr;
  double* owned = nullptr;
  int init(mln_ctx* c, const double* p, size_t count) {
    ctx = c;
    if (count == 0 || !p) { dev = p; return MLN_OK; }
    if (is_device_ptr(p)) { dev = p; return MLN_OK; }
    MLN_HIP(ctx, mln_dmalloc((void**)&owned, count * sizeof(double)));
    MLN_HIP(ctx, hipMemcpyAsync(owned, p, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    dev = owned;
    return MLN_OK;
  }
  ~DevIn() {
    if (owned) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(owned); }
  }
};

// Output that may live on host or device.
struct DevOut {
  mln_ctx* ctx;
  double* dev = nullptr;
  double* owned = nullptr;
  double* host = nullptr;
  size_t count = 0;
  int init(mln_ctx* c, double* p, size_t n, bool copy_in = false) {
    ctx = c; count = n;
    if (n == 0) { dev = p; return MLN_OK; }
    if (is_device_ptr(p)) { dev = p; return MLN_OK; }
    host = p;
    MLN_HIP(ctx, mln_dmalloc((void**)&owned, n * sizeof(double)));
    if (copy_in) MLN_HIP(ctx, hipMemcpyAsync(owned, p, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    dev = owned;
    return MLN_OK;
  }
  int commit() {
    if (owned && count) {
      MLN_HIP(ctx, hipMemcpyAsync(host, owned, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLN_OK;
  }
  ~DevOut() {
    if (owned) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(owned); }
  }
};

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

extern "C" void mln_ctx_destroy(mln_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  comm_release(ctx);
  masked_streams_release(ctx);
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
static int dev_allreduce(mln_ctx* ctx, double* dev, int64_t count) { return comm_allreduce(ctx, dev, count); }
static int dev_bcast0(mln_ctx* ctx, double* dev, int64_t count) { return comm_bcast0(ctx, dev, count); }

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

static int reject_distance_leaf(mln_ctx* ctx, const DevCov& dc) {
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

static int gram_of(mln_ctx* ctx, const double* A, int64_t lda, int64_t rows, int64_t m, double alpha, double* G,
                   int64_t ldg, bool quantised = false);
static int64_t pad16(int64_t m);

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

static int64_t pad16(int64_t m) { return ((m + 15) / 16) * 16; }

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

struct mln_fit;
// cov(xu, xu) -> Cholesky factor -> block-scaled copies on a second stream, in a helper thread, while the caller runs the
// kernel-matrix pass on `wide` (see fit_prepare_impl).  The helper works on a COPY of the context with its own stream,
// scratch and status word; its temporaries are released by the caller after the join (a free synchronises the device,
// i.e. would sit out the pass).
struct LandmarkChain {
  bool running = false;
  hipStream_t wide = nullptr;
  mln_ctx* ctx = nullptr;
  mln_ctx side;
  std::thread th;
  int rc = MLN_OK;
  double seconds = 0.0;
  double* Lp = nullptr; int64_t ldp = 0;
  TriInv tri;
  std::vector<void*> deferred;
  DevCov cov;
  int start(mln_ctx* c, mln_fit* f, const double* centers, int64_t m, int d, double jitter);
  int finish(mln_fit* f);
  ~LandmarkChain() {
    if (th.joinable()) th.join();
    if (running) {                                       // abandoned on an error path: nothing was handed over
      if (tri.W || tri.W2) triinv_free(&tri);
      if (side.scratch) deferred.push_back(side.scratch);
      if (side.d_info) deferred.push_back(side.d_info);
    }
    for (void* p : deferred) (void)mln_dfree(p);
  }
};

// ---- fit handle --------------------------------------------------------------------------------------
struct mln_fit {
  mln_ctx* ctx = nullptr;
  DevCov cov;
  int d = 0;
  int64_t n = 0, m = 0, ldl = 0, ldp = 0;
  bool full = false;
  double* L = nullptr;   // n x ldl (full GP: aliases Lp)
  double* Lp = nullptr;  // m x ldp
  TriInv tri;            // block-scaled Lp
  double *V = nullptr, *Vdr = nullptr;
  double mu = 0.0;
  // objective workspace
  int n_wg = 0;
  double *part_grad = nullptr, *part_hess = nullptr, *part_loss = nullptr;
  double *d_z = nullptr, *d_out = nullptr;  // m ; 1 + 2m
  double *h_z = nullptr, *h_out = nullptr;  // pinned
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double times[MLN_N_STAGE_TIMES] = {0};
  // preconditioner: C C^T = L^T L + I (the Ridge matrix) and C^-1, both m x ldl lower
  double *C = nullptr, *Cinv = nullptr;
  double *d_u = nullptr, *d_gu = nullptr, *d_tmp = nullptr;  // m ; m ; 1 + m
  int n_wg_cap = 0;
  // implicit ("K-space") mode: the n x m buffer holds K = cov(x, xu) itself and Lp^-T is folded
  // into the m-vectors:  L z = K (Lp^-T z),  L^T v = Lp^-1 (K^T v).  No n x m triangular solve.
  bool kspace = false;
  double* P = nullptr;    // Lp^-T C^-T  (m x ldl), so that  w = Lp^-T z = P u  for z = C^-T u
  double* Linv = nullptr; // Lp^-1 (m x ldp, lower), formed once: the whitening of a Gram and P are then plain GEMMs
  double* d_w = nullptr;  // m
  // last vector pair (z, w = Lp^-T z) produced by the library itself (Ridge init / MAP solve): lets
  // mln_transform / mln_weights_cholesky on that same z skip the triangular solve
  std::vector<double> z_cached;
  double* d_w_cached = nullptr;
  // stacked preconditioner operators, so that one evaluation needs two row-GEMVs and no reductions:
  //   Q1 = [C^-T ; P]  (2m x ldl, implicit) or C^-T (m x ldl):   [z ; w] = Q1 u
  //   Q2 = [C^-1 | P^T] (m x 2 ldl, implicit) or C^-1:            g_u = Q2 [z ; K^T(a-1)]
  double *Q1 = nullptr, *Q2 = nullptr, *d_zw = nullptr, *d_zr = nullptr;
  // eigenvectors of L^T L (rows, ascending eigenvalue), m x ldl: Nystroem rank reduction
  double* eigU = nullptr;
  // fp32 copy of the streamed n x m buffer for the warm-up passes of the MAP solve (mixed precision)
  float* L32 = nullptr;
  double emu_excluded = 0.0;    // MELLON_AMD_EMULATE_RANKS: wall seconds spent on the OTHER ranks' column blocks (tools/emulate_rank.py)
  bool cov_bounded01 = false;   // every covariance value lies in [0, 1] (stationary kernels and their products)
  int l32_fixed = 0;     // format of that copy: 0 = fp32, 1 = 32-bit fixed point (covariances bounded by 1)
  int evals32 = 0;
  double times32 = 0.0;
  // evaluation buffers of the preconditioned objective: d_zr = [z (ld2) | r (ld2)] with the likelihood sum at
  // d_zr[ld2 + m], so that one all-reduce of m + 1 values covers [r ; lik];  ld2 = pad16(m + 1)
  int64_t ld2 = 0;
  int64_t row0 = 0;     // global index of this shard's first cell (subsampling is by global index)
  // device-resident L-BFGS (solver.hip)
  SolverBuffers sv{};
  void* sv_block = nullptr;       // one allocation behind every pointer of sv
  SolverState* h_state = nullptr; // pinned mirror
  int sv_maxcor = 0;
  std::vector<hipEvent_t> evs;    // three per evaluation: before the fp32 pass, between, after the fp64 pass
  // f = L z + mu of every row at the solver's accepted point, kept by the objective passes themselves
  double* f_keep[2] = {nullptr, nullptr};
  int f_final = -1;               // which buffer holds f at z_cached (-1: none; mln_transform then streams the buffer)
  // row subsample shared by the preconditioner's Gram and the solver's first phase: cells whose GLOBAL index is a
  // multiple of precond_stride (0: no preconditioner yet; 1: all cells)
  int64_t precond_stride = 0;
  // handle whose kernel values come from the binding (mln_fit_prepare_from_K): rows received, finished
  bool from_K = false, k_finished = false;
  int64_t k_rows_done = 0;
  double build_seconds = 0.0;     // wall time of the first preconditioner build (Gram + factorisation): the rebuild's price
  double times_sub = 0.0, times_rebuild = 0.0, sub_pass_equiv = 0.0;
  int evals_sub = 0, n_rebuild = 0;
};

int LandmarkChain::start(mln_ctx* c, mln_fit* f, const double* centers, int64_t m, int d, double jitter) {
  ctx = c;
  wide = masked_stream(c, 32);
  hipStream_t second = masked_stream(c, 0);
  hipEvent_t ev = masked_stream_event(c, 2);
  if (!wide || !second || !ev) return MLN_OK;            // no masked streams here: the caller keeps the serial order
  side = *c;
  side.stream = second;
  side.scratch = nullptr; side.scratch_bytes = 0; side.err.clear();
  side.d_info = nullptr;
  MLN_HIP(c, mln_dmalloc((void**)&side.d_info, 4 * sizeof(int)));
  // what has been enqueued so far (the landmarks' upload, the zeroed Lp) precedes both side streams
  MLN_HIP(c, hipEventRecord(ev, c->stream));
  MLN_HIP(c, hipStreamWaitEvent(second, ev, 0));
  MLN_HIP(c, hipStreamWaitEvent(wide, ev, 0));
  Lp = f->Lp; ldp = f->ldp; cov = f->cov;
  const int device = c->device;
  running = true;
  th = std::thread([this, centers, m, d, jitter, device] {
    const double t0 = now_s();
    if (hipSetDevice(device) != hipSuccess) { rc = MLN_ERR_HIP; return; }
    mln_dfree_defer(&deferred);
    set_lookahead_disabled(true);
    rc = launch_kernel_matrix(&side, cov, centers, m, centers, m, d, Lp, ldp, jitter);
    if (rc == MLN_OK) rc = dev_cholesky_lower(&side, Lp, m, ldp);
    if (rc == MLN_OK) rc = triinv_build(&side, Lp, m, ldp, true, true, &tri);
    if (hipStreamSynchronize(side.stream) != hipSuccess && rc == MLN_OK) rc = MLN_ERR_HIP;
    set_lookahead_disabled(false);
    mln_dfree_defer(nullptr);
    seconds = now_s() - t0;
  });
  return MLN_OK;
}

int LandmarkChain::finish(mln_fit* f) {
  if (th.joinable()) th.join();
  running = false;
  if (side.scratch) deferred.push_back(side.scratch);
  if (side.d_info) deferred.push_back(side.d_info);
  side.scratch = nullptr; side.d_info = nullptr;
  for (void* p : deferred) (void)mln_dfree(p);
  deferred.clear();
  f->times[1] += seconds;
  if (rc != MLN_OK) {
    if (tri.W || tri.W2) triinv_free(&tri);
    mln_set_error(ctx, side.err.empty() ? std::string("the landmark chain (cov(xu, xu), Cholesky) failed") : side.err);
    return rc;
  }
  f->tri = tri;
  return MLN_OK;
}

// rows of this shard in the subsample of stride s: first local index and count
static void fit_sample_rows(const mln_fit* f, int64_t s, int64_t* first, int64_t* rows) {
  if (s < 1) s = 1;
  *first = (s - f->row0 % s) % s;
  *rows = (f->n > *first) ? (f->n - *first + s - 1) / s : 0;
}

static void fit_free(mln_fit* f) {
  if (!f) return;
  mln_ctx* ctx = f->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (f->L && f->L != f->Lp) (void)mln_dfree(f->L);
  if (f->Lp) (void)mln_dfree(f->Lp);
  triinv_free(&f->tri);
  void* ptrs[] = {f->V, f->Vdr, f->part_grad, f->part_hess, f->part_loss, f->d_z, f->d_out,
                  f->C, f->Cinv, f->d_u, f->d_gu, f->d_tmp, f->P, f->d_w, f->d_w_cached,
                  f->Q1, f->Q2, f->d_zw, f->d_zr, f->eigU, f->L32, f->sv_block, f->f_keep[0], f->f_keep[1], f->Linv};
  for (void* p : ptrs) if (p) (void)mln_dfree(p);
  if (f->h_state) (void)hipHostFree(f->h_state);
  for (hipEvent_t e : f->evs) (void)hipEventDestroy(e);
  if (f->h_z) (void)hipHostFree(f->h_z);
  if (f->h_out) (void)hipHostFree(f->h_out);
  if (f->ev0) (void)hipEventDestroy(f->ev0);
  if (f->ev1) (void)hipEventDestroy(f->ev1);
  delete f;
}

extern "C" void mln_fit_destroy(mln_fit* fit) { fit_free(fit); }

__global__ void k_round_copy_bits(unsigned* __restrict__ q, int64_t count, int drop) {
  const unsigned half = 1u << (drop - 1), mask = ~((1u << drop) - 1u);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned v = q[i];
    const unsigned r = (v > 0xffffffffu - half) ? (v & mask) : ((v + half) & mask);
    q[i] = r;
  }
}

static int fit_alloc_workspace(mln_fit* f) {
  mln_ctx* ctx = f->ctx;
  int64_t steps = (f->n + 1) / 2;
  int n_wg = ctx->n_cu > 0 ? ctx->n_cu : 256;
  f->n_wg_cap = n_wg;                      // partial buffers are sized for this many workgroups
  if (steps < n_wg) n_wg = (int)(steps > 0 ? steps : 1);
  f->n_wg = n_wg;
  n_wg = f->n_wg_cap;
  const size_t pm = (size_t)f->ldl;
  f->ld2 = pad16(f->m + 1);
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_u, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_gu, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_tmp, sizeof(double) * (1 + pm)));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_w, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_w_cached, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_zw, sizeof(double) * 2 * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_zr, sizeof(double) * 2 * (size_t)f->ld2));
  MLN_HIP(ctx, hipMemsetAsync(f->d_zr, 0, sizeof(double) * 2 * (size_t)f->ld2, ctx->stream));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->part_grad, sizeof(double) * pm * n_wg));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->part_hess, sizeof(double) * pm * n_wg));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->part_loss, sizeof(double) * n_wg));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_z, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_out, sizeof(double) * (1 + 2 * pm)));
  MLN_HIP(ctx, hipHostMalloc((void**)&f->h_z, sizeof(double) * pm, hipHostMallocDefault));
  MLN_HIP(ctx, hipHostMalloc((void**)&f->h_out, sizeof(double) * (1 + 2 * pm), hipHostMallocDefault));
  MLN_HIP(ctx, hipEventCreate(&f->ev0));
  MLN_HIP(ctx, hipEventCreate(&f->ev1));
  return MLN_OK;
}

// Upload of the cells from pageable host memory in row chunks by a helper thread (see fit_prepare_impl).  A copy from
// pageable memory blocks its CALLING thread while the runtime stages it through pinned buffers, but not the device: the
// chunks travel while the main thread's kernels run.  Chunk c is complete on the device when events[c] has fired; the
// main thread makes its stream wait for that event -- after the helper has recorded it (done > c).
struct HostUpload {
  mln_ctx* ctx = nullptr;
  hipStream_t copy = nullptr;
  std::vector<hipEvent_t> events;
  std::atomic<int> done{0};
  std::atomic<int> failed{0};
  std::thread th;
  int n_chunks = 0;
  int64_t n = 0, chunk_rows = 0;
  int d = 0;
  int start(mln_ctx* c, const double* src, double* dst, int64_t n_, int d_) {
    ctx = c; n = n_; d = d_;
    // chunks of whole 128-row workgroup tiles, ~64 MB each, at most 16
    chunk_rows = std::max<int64_t>(128, (((int64_t)64 << 20) / ((int64_t)d * 8) + 127) / 128 * 128);
    if ((n + chunk_rows - 1) / chunk_rows > 16) chunk_rows = ((n + 15) / 16 + 127) / 128 * 128;
    n_chunks = (int)((n + chunk_rows - 1) / chunk_rows);
    MLN_HIP(ctx, hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    events.resize((size_t)n_chunks, nullptr);
    for (auto& e : events) MLN_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int device = ctx->device;
    th = std::thread([this, src, dst, device] {
      if (hipSetDevice(device) != hipSuccess) { failed.store(1); done.store(n_chunks); return; }
      for (int c = 0; c < n_chunks; ++c) {
        const int64_t r0 = (int64_t)c * chunk_rows, rows = std::min(chunk_rows, n - r0);
        hipError_t e = hipMemcpyAsync(dst + r0 * d, src + r0 * d, sizeof(double) * (size_t)(rows * d), hipMemcpyHostToDevice, copy);
        if (e == hipSuccess) e = hipEventRecord(events[(size_t)c], copy);
        if (e != hipSuccess) { failed.store(1); done.store(n_chunks, std::memory_order_release); return; }
        done.store(c + 1, std::memory_order_release);
      }
    });
    return MLN_OK;
  }
  int wait_chunk(int c, int64_t* r0, int64_t* rows) {
    while (done.load(std::memory_order_acquire) <= c) std::this_thread::yield();
    if (failed.load()) { mln_set_error(ctx, "upload of the cells failed (helper thread)"); return MLN_ERR_HIP; }
    MLN_HIP(ctx, hipStreamWaitEvent(ctx->stream, events[(size_t)c], 0));
    *r0 = (int64_t)c * chunk_rows;
    *rows = std::min(chunk_rows, n - *r0);
    return MLN_OK;
  }
  int finish() {
    if (th.joinable()) th.join();
    return failed.load() ? MLN_ERR_HIP : MLN_OK;
  }
  ~HostUpload() {
    if (th.joinable()) th.join();
    if (copy) { (void)hipStreamSynchronize(copy); (void)hipStreamDestroy(copy); }
    for (hipEvent_t e : events) if (e) (void)hipEventDestroy(e);
  }
};

static int fit_prepare_impl(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n, int32_t d,
                            const double* xu, int64_t m, double jitter, const double* Lp_in, int32_t flags,
                            mln_fit* f) {
  f->ctx = ctx;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &f->cov));
  f->d = d; f->n = n; f->full = (xu == nullptr);
  if (f->full) m = n;
  f->m = m;
  if (m < 1 || m > 65535) { mln_set_error(ctx, "number of landmarks out of range"); return MLN_ERR_SHAPE; }
  if (m > objective_max_m()) { mln_set_error(ctx, "m > 8192 landmarks is not supported by this build"); return MLN_ERR_UNSUPPORTED; }
  if (f->full && ctx->n_ranks > 1) { mln_set_error(ctx, "the full (non-sparse) GP cannot be cell-sharded"); return MLN_ERR_UNSUPPORTED; }
  f->ldp = pad16(m);
  f->ldl = pad16(m);
  DevIn dx, du;
  // Cells handed over in HOST memory (the reference's timed region starts there: density_estimator.py:542-581): the upload
  // -- 0.4 GB at C3, ~8 ms over PCIe -- runs in a helper thread on a copy stream, in row chunks, UNDER the work that needs
  // only the landmarks (cov(xu, xu), its Cholesky factor, the block-scaled copies) and under the kernel-matrix pass of the
  // chunks that have already arrived; each chunk's pass waits for that chunk's event only.
  HostUpload up;
  const bool pipelined = !f->full && n > 0 && x && !is_device_ptr(x) && (size_t)n * d * sizeof(double) >= ((size_t)32 << 20) &&
                         !(std::getenv("MELLON_AMD_UPLOAD_PIPELINE") && std::atoi(std::getenv("MELLON_AMD_UPLOAD_PIPELINE")) == 0);
  if (pipelined) {
    dx.ctx = ctx;
    MLN_HIP(ctx, mln_dmalloc((void**)&dx.owned, (size_t)n * d * sizeof(double)));
    dx.dev = dx.owned;
    MLN_TRY(up.start(ctx, x, dx.owned, n, d));
  } else {
    MLN_TRY(dx.init(ctx, x, (size_t)n * d));
  }
  if (!f->full) MLN_TRY(du.init(ctx, xu, (size_t)m * d));
  const double* centers = f->full ? dx.dev : du.dev;

  // Lp = chol(cov(xu, xu) + max(sigma^2, jitter) I), sigma = 0     decomposition.py:111-123
  double t0 = now_s();
  const size_t lp_bytes = sizeof(double) * (size_t)m * f->ldp;
  MLN_HIP(ctx, mln_dmalloc((void**)&f->Lp, lp_bytes));
  MLN_HIP(ctx, hipMemsetAsync(f->Lp, 0, lp_bytes, ctx->stream));
  // Round 4: the landmark-only chain -- cov(xu, xu), its Cholesky factor, the block-scaled copies: ~6 ms at m = 5000, a
  // latency chain that never fills the chip -- runs in a helper thread on a second stream UNDER the kernel-matrix pass,
  // which is launched on a stream whose CU mask leaves 32 compute units to it (linalg.h: masked_stream).  Worth it when
  // the pass is the longer of the two by a margin (the chain is slower with few units): C3 on one GPU, not its 8-rank shard.
  LandmarkChain chain;
  {
    const double km_est = (double)n * (double)m * 3.3e-12, chain_est = 6e-3 * ((double)m / 5000.0) * ((double)m / 5000.0);
    bool want = !f->full && !Lp_in && n > 0 && m >= 1024 && km_est > 2.0 * chain_est;
    if (const char* ev = std::getenv("MELLON_AMD_OVERLAP_LANDMARK_CHAIN")) want = want && std::atoi(ev) != 0;
    if (want) MLN_TRY(chain.start(ctx, f, centers, m, d, jitter));
  }
  if (chain.running) {
    // (its results -- f->Lp, f->tri -- are collected below, after the kernel-matrix pass has been enqueued)
  } else if (Lp_in) {
    DevIn dl;
    MLN_TRY(dl.init(ctx, Lp_in, (size_t)m * m));
    MLN_TRY(launch_copy_block(ctx, dl.dev, m, f->Lp, f->ldp, m, m));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    MLN_TRY(launch_kernel_matrix(ctx, f->cov, centers, m, centers, m, d, f->Lp, f->ldp, jitter));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    f->times[0] += now_s() - t0;
    t0 = now_s();
    MLN_TRY(dev_cholesky_lower(ctx, f->Lp, m, f->ldp));
  }
  if (!chain.running) {
    MLN_TRY(triinv_build(ctx, f->Lp, m, f->ldp, true, true, &f->tri));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    f->times[1] += now_s() - t0;
  }

  if (f->full) {
    f->L = f->Lp;  // parameters.py:847-850
  } else {
    // L = cov(x, xu) Lp^-T                                          decomposition.py:205-210
    t0 = now_s();
    const size_t l_bytes = sizeof(double) * (size_t)(n > 0 ? n : 1) * f->ldl;
    const bool trace = std::getenv("MELLON_AMD_TRACE") != nullptr;
    MLN_HIP(ctx, mln_dmalloc((void**)&f->L, l_bytes));
    if (trace) { (void)hipStreamSynchronize(ctx->stream); fprintf(stderr, "[trace] L alloc %.4f s\n", now_s() - t0); }
    // Mixed precision (default on for large implicit fits, MELLON_AMD_MIXED=0 disables): the kernel-matrix
    // pass also writes an fp32 copy, which the first passes of the MAP solve stream instead of the fp64 one.
    int64_t mixed_min = (int64_t)1 << 27;
    bool mixed = (flags & MLN_FIT_IMPLICIT) != 0;
    if (const char* ev = std::getenv("MELLON_AMD_MIXED")) mixed = mixed && std::atoi(ev) != 0;
    if (const char* ev = std::getenv("MELLON_AMD_MIXED_MIN_ELEMS")) mixed_min = std::atoll(ev);
    if (mixed && n * m >= mixed_min && n > 0 && m <= 8192)     // (beyond 8192 landmarks the pass is segmented: objective.hip)
      MLN_HIP(ctx, mln_dmalloc((void**)&f->L32, sizeof(float) * (size_t)n * f->ldl));
    // Format of the copy.  Covariance values of stationary kernels and of their products lie in [0, 1]: there the
    // fixed-point number round(v 2^32) has an absolute error of 1.2e-10 for EVERY entry, where fp32 carries up to 3e-8
    // on the entries near 1 -- which, with the nearest-neighbour length-scale heuristic, are most of them.  The
    // surrogate objective then sits ~100x closer to the true one, and the solver can stay on the 4-byte stream for
    // more of its iterations.  Sums, scalars, powers, the Linear kernel: fp32.  MELLON_AMD_SURROGATE=float|fixed overrides.
    f->l32_fixed = 0;
    bool bounded = true;
    for (int l = 0; l < f->cov.n_leaves; ++l)
      bounded = bounded && f->cov.leaves[l].kind >= MLN_K_MATERN32 && f->cov.leaves[l].kind <= MLN_K_RATQUAD;
    for (int t = 0; t < f->cov.n_toks; ++t)
      bounded = bounded && (f->cov.tok_op[t] == MLN_OP_LEAF || f->cov.tok_op[t] == MLN_OP_MUL);
    f->cov_bounded01 = bounded;
    if (f->L32) {
      f->l32_fixed = bounded ? 1 : 0;
      if (const char* ev = std::getenv("MELLON_AMD_SURROGATE")) {
        if (std::strcmp(ev, "float") == 0) f->l32_fixed = 0;
        else if (std::strcmp(ev, "fixed") == 0 && bounded) f->l32_fixed = 1;
      }
    }
    hipStream_t const own_stream = ctx->stream;
    struct StreamRestore { mln_ctx* c; hipStream_t s; ~StreamRestore() { c->stream = s; } } restore{ctx, own_stream};   // (early returns included)
    if (chain.running) ctx->stream = chain.wide;         // the pass leaves 32 compute units to the landmark chain
    if (pipelined) {
      for (int c = 0; c < up.n_chunks; ++c) {
        int64_t r0 = 0, rows = 0;
        MLN_TRY(up.wait_chunk(c, &r0, &rows));           // (ctx->stream waits for the chunk's event; the host only for its recording)
        MLN_TRY(launch_kernel_matrix(ctx, f->cov, dx.dev + r0 * d, rows, du.dev, m, d, f->L + r0 * f->ldl, f->ldl, 0.0,
                                     f->L32 ? f->L32 + r0 * f->ldl : nullptr, f->l32_fixed));
      }
      MLN_TRY(up.finish());
    } else {
      MLN_TRY(launch_kernel_matrix(ctx, f->cov, dx.dev, n, du.dev, m, d, f->L, f->ldl, 0.0, f->L32, f->l32_fixed));
    }
    if (f->L32 && f->l32_fixed)
      if (const char* ev = std::getenv("MELLON_AMD_COPY_BITS")) {   // experiment: the copy rounded to fewer bits
        const int bits = std::atoi(ev);
        if (bits >= 8 && bits < 32)
          hipLaunchKernelGGL(k_round_copy_bits, dim3(4096), dim3(256), 0, ctx->stream, reinterpret_cast<unsigned*>(f->L32),
                             (int64_t)n * f->ldl, 32 - bits);
      }
    if (chain.running) {
      const int rc_km = (hipStreamSynchronize(ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
      ctx->stream = own_stream;
      MLN_TRY(chain.finish(f));                          // joins the helper; its error (not positive definite) is the fit's
      MLN_TRY(rc_km);
    }
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (trace) fprintf(stderr, "[trace] L kernel matrix done at %.4f s\n", now_s() - t0);
    f->times[0] += now_s() - t0;
    if (flags & MLN_FIT_IMPLICIT) {
      f->kspace = true;  // keep K; Lp^-T is applied to m-vectors instead of to n rows
    } else {
      t0 = now_s();
      MLN_TRY(triinv_solve_right_T(ctx, f->tri, f->L, n, f->ldl));
      MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
      f->times[2] += now_s() - t0;
    }
  }
  MLN_TRY(fit_alloc_workspace(f));
  return MLN_OK;
}

extern "C" int mln_fit_prepare(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                               int32_t d, const double* xu, int64_t m, double jitter, const double* Lp_in,
                               int32_t flags, mln_fit** out) {
  if (!ctx || !out) return MLN_ERR_ARG;
  *out = nullptr;
  if (n_local < 0 || d < 1 || (n_local > 0 && !x)) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  mln_fit* f = new mln_fit();
  int rc = fit_prepare_impl(ctx, cov, x, n_local, d, xu, m, jitter, Lp_in, flags, f);
  if (rc != MLN_OK) { fit_free(f); return rc; }
  *out = f;
  return MLN_OK;
}

extern "C" int mln_fit_from_L(mln_ctx* ctx, const double* L, int64_t n_local, int64_t m, const double* Lp,
                              mln_fit** out) {
  if (!ctx || !out || !L) return MLN_ERR_ARG;
  *out = nullptr;
  if (n_local < 1 || m < 1 || m > 65535) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (m > objective_max_m()) { mln_set_error(ctx, "m > 8192 columns is not supported by this build"); return MLN_ERR_UNSUPPORTED; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  mln_fit* f = new mln_fit();
  f->ctx = ctx; f->n = n_local; f->m = m; f->d = 0; f->full = false;
  f->ldl = pad16(m); f->ldp = pad16(m);
  auto body = [&]() -> int {
    DevIn dl;
    MLN_TRY(dl.init(ctx, L, (size_t)n_local * m));
    const size_t l_bytes = sizeof(double) * (size_t)n_local * f->ldl;
    MLN_HIP(ctx, mln_dmalloc((void**)&f->L, l_bytes));
    MLN_HIP(ctx, hipMemsetAsync(f->L, 0, l_bytes, ctx->stream));
    MLN_TRY(launch_copy_block(ctx, dl.dev, m, f->L, f->ldl, n_local, m));
    if (Lp) {
      DevIn dp;
      MLN_TRY(dp.init(ctx, Lp, (size_t)m * m));
      const size_t lp_bytes = sizeof(double) * (size_t)m * f->ldp;
      MLN_HIP(ctx, mln_dmalloc((void**)&f->Lp, lp_bytes));
      MLN_HIP(ctx, hipMemsetAsync(f->Lp, 0, lp_bytes, ctx->stream));
      MLN_TRY(launch_copy_block(ctx, dp.dev, m, f->Lp, f->ldp, m, m));
      MLN_TRY(triinv_build(ctx, f->Lp, m, f->ldp, true, true, &f->tri));
    }
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return fit_alloc_workspace(f);
  };
  int rc = body();
  if (rc != MLN_OK) { fit_free(f); return rc; }
  *out = f;
  return MLN_OK;
}

// ---- user-defined kernels / oversized covariance trees: the kernel values arrive from the binding -----------------
extern "C" int mln_fit_prepare_from_K(mln_ctx* ctx, const double* Kuu, int64_t n_local, int64_t m, double jitter,
                                      const double* Lp_in, int32_t flags, mln_fit** out) {
  if (!ctx || !out) return MLN_ERR_ARG;
  *out = nullptr;
  const bool full = (flags & MLN_FIT_FULL) != 0;
  if (n_local < 0 || m < 1 || m > 65535 || (full && n_local != m)) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (!Kuu && !Lp_in) { mln_set_error(ctx, "mln_fit_prepare_from_K needs cov(xu, xu) or its factor"); return MLN_ERR_ARG; }
  if (full && ctx->n_ranks > 1) { mln_set_error(ctx, "the full (non-sparse) GP cannot be cell-sharded"); return MLN_ERR_UNSUPPORTED; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  mln_fit* f = new mln_fit();
  f->ctx = ctx; f->n = n_local; f->m = m; f->d = 0; f->full = full;
  f->ldl = pad16(m); f->ldp = pad16(m);
  f->cov.n_leaves = 0; f->cov.n_toks = 0;          // no device program: values only
  f->from_K = true;
  f->kspace = !full && (flags & MLN_FIT_IMPLICIT) != 0;
  auto body = [&]() -> int {
    double t0 = now_s();
    const size_t lp_bytes = sizeof(double) * (size_t)m * f->ldp;
    MLN_HIP(ctx, mln_dmalloc((void**)&f->Lp, lp_bytes));
    MLN_HIP(ctx, hipMemsetAsync(f->Lp, 0, lp_bytes, ctx->stream));
    DevIn dk;
    MLN_TRY(dk.init(ctx, Lp_in ? Lp_in : Kuu, (size_t)m * m));
    MLN_TRY(launch_copy_block(ctx, dk.dev, m, f->Lp, f->ldp, m, m));
    if (!Lp_in) {
      MLN_TRY(launch_add_diag(ctx, f->Lp, m, f->ldp, jitter));          // decomposition.py:111-114
      MLN_TRY(dev_cholesky_lower(ctx, f->Lp, m, f->ldp));
    }
    MLN_TRY(triinv_build(ctx, f->Lp, m, f->ldp, true, true, &f->tri));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    f->times[1] += now_s() - t0;
    if (full) {
      f->L = f->Lp;
      f->k_rows_done = n_local;
      return fit_alloc_workspace(f);
    }
    const size_t l_bytes = sizeof(double) * (size_t)(n_local > 0 ? n_local : 1) * f->ldl;
    MLN_HIP(ctx, mln_dmalloc((void**)&f->L, l_bytes));
    if (f->ldl != m) MLN_HIP(ctx, hipMemsetAsync(f->L, 0, l_bytes, ctx->stream));      // the pad columns must be zero
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLN_OK;
  };
  int rc = body();
  if (rc != MLN_OK) { fit_free(f); return rc; }
  *out = f;
  return MLN_OK;
}

extern "C" int mln_fit_set_K_rows(mln_fit* f, int64_t row0, int64_t n_rows, const double* K_rows) {
  if (!f || (n_rows > 0 && !K_rows)) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->from_K || f->full || f->k_finished) { mln_set_error(ctx, "mln_fit_set_K_rows: not a handle awaiting kernel rows"); return MLN_ERR_ARG; }
  if (row0 < 0 || n_rows < 0 || row0 + n_rows > f->n) { mln_set_error(ctx, "mln_fit_set_K_rows: rows out of range"); return MLN_ERR_SHAPE; }
  if (n_rows == 0) return MLN_OK;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  double t0 = now_s();
  MLN_HIP(ctx, hipMemcpy2DAsync(f->L + row0 * f->ldl, sizeof(double) * (size_t)f->ldl, K_rows, sizeof(double) * (size_t)f->m,
                                sizeof(double) * (size_t)f->m, (size_t)n_rows, hipMemcpyDefault, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  f->k_rows_done += n_rows;
  f->times[0] += now_s() - t0;
  return MLN_OK;
}

extern "C" int mln_fit_finish_K(mln_fit* f) {
  if (!f) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->from_K) { mln_set_error(ctx, "mln_fit_finish_K: not a handle built from kernel values"); return MLN_ERR_ARG; }
  if (f->k_finished || f->full) { f->k_finished = true; return MLN_OK; }
  if (f->k_rows_done < f->n) { mln_set_error(ctx, "mln_fit_finish_K: kernel rows are missing"); return MLN_ERR_SHAPE; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  if (!f->kspace) {                       // L = K Lp^-T                                   decomposition.py:205-210
    double t0 = now_s();
    MLN_TRY(triinv_solve_right_T(ctx, f->tri, f->L, f->n, f->ldl));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    f->times[2] += now_s() - t0;
  }
  f->k_finished = true;
  return fit_alloc_workspace(f);
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

static int fit_gram(mln_fit* f, double* G, int64_t ldg, int64_t row_stride);

extern "C" int mln_fit_gram_eigh(mln_fit* f, double* w, int32_t* n_sweeps) {
  if (!f || !w) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t m = f->m, ld = f->ldl;
  double* G = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&G, sizeof(double) * (size_t)m * ld));
  // all cells, all ranks; an implicit fit forms Lp^-1 (K^T K) Lp^-T (eigenvalues only are meaningful then:
  // mln_fit_project needs the explicit factor)
  int rc = f->kspace ? fit_gram(f, G, ld, 1) : gram_of(ctx, f->L, f->ldl, f->n, m, 1.0, G, ld);
  if (rc == MLN_OK && !f->eigU) {
    hipError_t e = mln_dmalloc((void**)&f->eigU, sizeof(double) * (size_t)m * ld);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc eigenvectors", __FILE__, __LINE__);
  }
  std::vector<double> wh((size_t)m);
  int sweeps = 0;
  if (rc == MLN_OK) rc = dev_eigh(ctx, G, m, ld, wh.data(), f->eigU, ld, &sweeps);
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(G);
  if (rc != MLN_OK) return rc;
  if (n_sweeps) *n_sweeps = sweeps;
  std::memcpy(w, wh.data(), sizeof(double) * (size_t)m);
  return MLN_OK;
}

// util.test_rank without an eigendecomposition: the number of singular values of L above tol * the largest = the number of
// eigenvalues of L^T L (all cells, all ranks) above tol^2 * lambda_max, counted on the tridiagonalised Gram (tridiag.hip)
extern "C" int mln_fit_gram_rank(mln_fit* f, double tol, int64_t* rank_out, double* sigma_max_out) {
  if (!f || !rank_out || !(tol >= 0.0)) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t m = f->m, ld = f->ldl;
  double* G = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&G, sizeof(double) * (size_t)m * ld));
  // With more than 24 cells per landmark the count is taken from the Gram of ~12 m evenly spaced cells (by global index,
  // scaled by the stride; the integer Gram of the preconditioner where the covariance is bounded): the diagnostic only
  // compares the count with 80 % of m (base_model.py:344-355), and the full fp64 Gram is n m^2 flops -- 0.5 s at C3.
  const int n_ranks = ctx->n_ranks > 1 ? ctx->n_ranks : 1;
  const int64_t n_est = f->n * n_ranks;
  int64_t stride = 1;
  static const bool sampled_ok = !(std::getenv("MELLON_AMD_RANK_SAMPLED") && std::atoi(std::getenv("MELLON_AMD_RANK_SAMPLED")) == 0);
  if (sampled_ok && f->kspace && n_est >= 24 * m) stride = std::max<int64_t>(1, n_est / (12 * m));
  int rc = f->kspace ? fit_gram(f, G, ld, stride) : gram_of(ctx, f->L, f->ldl, f->n, m, 1.0, G, ld);
  double lmax = 0.0;
  if (rc == MLN_OK) rc = dev_sym_rank_above(ctx, G, m, ld, tol * tol, rank_out, &lmax);
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(G);
  if (rc == MLN_OK && sigma_max_out) *sigma_max_out = std::sqrt(std::max(lmax, 0.0));
  return rc;
}

extern "C" int mln_fit_project(mln_fit* f, int64_t p, mln_fit** out) {
  if (!f || !out) return MLN_ERR_ARG;
  *out = nullptr;
  mln_ctx* ctx = f->ctx;
  if (!f->eigU) { mln_set_error(ctx, "project: call mln_fit_gram_eigh first"); return MLN_ERR_ARG; }
  if (f->kspace) { mln_set_error(ctx, "project needs the explicit factor (prepare without MLN_FIT_IMPLICIT)"); return MLN_ERR_UNSUPPORTED; }
  if (p < 1 || p > f->m) { mln_set_error(ctx, "project: rank out of range"); return MLN_ERR_SHAPE; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  mln_fit* g = new mln_fit();
  g->ctx = ctx; g->n = f->n; g->m = p; g->d = 0; g->full = false;
  g->ldl = pad16(p); g->ldp = pad16(p);
  auto body = [&]() -> int {
    const size_t l_bytes = sizeof(double) * (size_t)(g->n > 0 ? g->n : 1) * g->ldl;
    MLN_HIP(ctx, mln_dmalloc((void**)&g->L, l_bytes));
    MLN_HIP(ctx, hipMemsetAsync(g->L, 0, l_bytes, ctx->stream));
    if (g->n > 0) {
      GemmArgs a{};
      a.A = f->L; a.lda = f->ldl; a.ta = 0;                                   // B (n x m)
      a.B = f->eigU + (f->m - p) * f->ldl; a.ldb = f->ldl; a.tb = 1;          // top-p eigenvectors as rows
      a.C = g->L; a.ldc = g->ldl;
      a.M = g->n; a.N = p; a.K = f->m; a.alpha = 1.0; a.beta = 0.0; a.split_k = 1;
      MLN_TRY(launch_dgemm(ctx, a));
    }
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return fit_alloc_workspace(g);
  };
  int rc = body();
  if (rc != MLN_OK) { fit_free(g); return rc; }
  *out = g;
  return MLN_OK;
}

extern "C" int mln_fit_rank(mln_fit* fit, int64_t* m_out) {
  if (!fit || !m_out) return MLN_ERR_ARG;
  *m_out = fit->m;
  return MLN_OK;
}

extern "C" int mln_fit_get_Lp(mln_fit* f, double* out) {
  if (!f || !out) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->Lp) { mln_set_error(ctx, "this fit handle holds no Lp"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevOut o;
  MLN_TRY(o.init(ctx, out, (size_t)f->m * f->m));
  MLN_TRY(launch_copy_block(ctx, f->Lp, f->ldp, o.dev, f->m, f->m, f->m));
  return o.commit();
}

extern "C" int mln_fit_get_L(mln_fit* f, int64_t row0, int64_t n_rows, double* out) {
  if (!f || (n_rows > 0 && !out)) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (row0 < 0 || n_rows < 0 || row0 + n_rows > f->n) { mln_set_error(ctx, "row range out of bounds"); return MLN_ERR_SHAPE; }
  if (n_rows == 0) return MLN_OK;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevOut o;
  MLN_TRY(o.init(ctx, out, (size_t)n_rows * f->m));
  if (f->kspace) {  // materialise the requested rows of L = K Lp^-T on demand
    double* tmp = nullptr;
    MLN_HIP(ctx, mln_dmalloc((void**)&tmp, sizeof(double) * (size_t)n_rows * f->ldl));
    int rc = launch_copy_block(ctx, f->L + row0 * f->ldl, f->ldl, tmp, f->ldl, n_rows, f->ldl);
    if (rc == MLN_OK) rc = triinv_solve_right_T(ctx, f->tri, tmp, n_rows, f->ldl);
    if (rc == MLN_OK) rc = launch_copy_block(ctx, tmp, f->ldl, o.dev, f->m, n_rows, f->m);
    if (rc == MLN_OK) rc = o.commit();
    (void)mln_dfree(tmp);
    return rc;
  }
  MLN_TRY(launch_copy_block(ctx, f->L + row0 * f->ldl, f->ldl, o.dev, f->m, n_rows, f->m));
  return o.commit();
}

extern "C" int mln_fit_set_likelihood(mln_fit* f, const double* V, const double* Vdr, double mu) {
  if (!f || (f->n > 0 && (!V || !Vdr))) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = sizeof(double) * (size_t)(f->n > 0 ? f->n : 1);
  if (!f->V) MLN_HIP(ctx, mln_dmalloc((void**)&f->V, bytes));
  if (!f->Vdr) MLN_HIP(ctx, mln_dmalloc((void**)&f->Vdr, bytes));
  if (f->n > 0) {
    MLN_HIP(ctx, hipMemcpyAsync(f->V, V, sizeof(double) * f->n, hipMemcpyDefault, ctx->stream));
    MLN_HIP(ctx, hipMemcpyAsync(f->Vdr, Vdr, sizeof(double) * f->n, hipMemcpyDefault, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  f->mu = mu;
  return MLN_OK;
}

static ObjArgs obj_args(mln_fit* f) {
  ObjArgs a{};
  a.L = f->L; a.ldl = f->ldl; a.n = f->n; a.m = f->m;
  a.z = f->d_z; a.V = f->V; a.Vdr = f->Vdr; a.mu = f->mu;
  a.part_grad = f->part_grad; a.part_hess = nullptr; a.part_loss = f->part_loss;
  a.weights = nullptr; a.f_out = nullptr;
  a.n_wg = f->n_wg; a.m_pad = f->ldl;
  return a;
}

static int fit_small_gemv(mln_fit* f, const double* M, int trans, const double* w, double* y);

static void obj_account(mln_fit* f, bool f32 = false) {
  float ms = 0.f;
  const bool ok = hipEventElapsedTime(&ms, f->ev0, f->ev1) == hipSuccess;
  if (f32) {   // fp32 warm-up passes are accounted separately: the roofline figure is the fp64 kernel's
    if (ok) f->times32 += 1e-3 * ms;
    f->evals32 += 1;
    return;
  }
  if (ok) f->times[5] += 1e-3 * ms;
  f->times[6] += 1.0;
  f->times[7] = (double)f->n * (double)f->ldl * 8.0;
}

// In implicit mode the streamed matrix is K and the kernel's vector is w = Lp^-T z (device, m).
// `z_host` (may be NULL) is the caller's host copy of z, used to recognise the cached pair.
static int fit_w_from_z(mln_fit* f, const double* z_dev, double* w_dev, const double* z_host = nullptr) {
  mln_ctx* ctx = f->ctx;
  if (z_host && f->z_cached.size() == (size_t)f->m &&
      std::memcmp(z_host, f->z_cached.data(), sizeof(double) * f->m) == 0) {
    MLN_HIP(ctx, hipMemcpyAsync(w_dev, f->d_w_cached, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
    return MLN_OK;
  }
  MLN_HIP(ctx, hipMemcpyAsync(w_dev, z_dev, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
  return triinv_solve_left_T(ctx, f->tri, w_dev, 1, 1);
}

// remember (z, w) computed from the preconditioned variable: z = C^-T u (d_z), w = P u
static int fit_cache_pair_from_u(mln_fit* f, const double* u_dev) {
  mln_ctx* ctx = f->ctx;
  f->z_cached.assign((size_t)f->m, 0.0);
  MLN_HIP(ctx, hipMemcpyAsync(f->z_cached.data(), f->d_z, sizeof(double) * f->m, hipMemcpyDeviceToHost, ctx->stream));
  if (f->kspace) MLN_TRY(fit_small_gemv(f, f->P, 0, u_dev, f->d_w_cached));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MLN_OK;
}

extern "C" int mln_objective(mln_fit* f, const double* z, double* loss, double* grad, double* hess_diag) {
  if (!f || !z || !loss || !grad) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->V) { mln_set_error(ctx, "mln_fit_set_likelihood has not been called"); return MLN_ERR_ARG; }
  if (hess_diag && f->kspace) {
    mln_set_error(ctx, "the Hessian diagonal needs the explicit factor L: prepare the fit without MLN_FIT_IMPLICIT");
    return MLN_ERR_UNSUPPORTED;
  }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t m = f->m;
  MLN_HIP(ctx, hipMemcpyAsync(f->d_z, z, sizeof(double) * m, hipMemcpyDefault, ctx->stream));
  MLN_HIP(ctx, hipMemcpyAsync(f->h_z, f->d_z, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  ObjArgs a = obj_args(f);
  if (f->kspace) {
    MLN_TRY(fit_w_from_z(f, f->d_z, f->d_w));
    a.z = f->d_w;
  }
  if (hess_diag) a.part_hess = f->part_hess;
  const int64_t nout = 1 + m + (hess_diag ? m : 0);
  MLN_HIP(ctx, hipEventRecord(f->ev0, ctx->stream));
  MLN_TRY(launch_objective(ctx, a));
  MLN_HIP(ctx, hipEventRecord(f->ev1, ctx->stream));
  MLN_TRY(launch_reduce_obj(ctx, a, f->d_out));
  MLN_TRY(dev_allreduce(ctx, f->d_out, nout));
  if (f->kspace) MLN_TRY(triinv_solve_left(ctx, f->tri, f->d_out + 1, 1, 1));   // L^T v = Lp^-1 (K^T v)
  MLN_HIP(ctx, hipMemcpyAsync(f->h_out, f->d_out, sizeof(double) * nout, hipMemcpyDeviceToHost, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  obj_account(f);
  // prior terms, added once (inference.py:45-46): 1/2 |z|^2 + (k/2) log 2 pi ; d/dz = z ; d2/dz2 = 1
  double zz = 0.0;
  for (int64_t j = 0; j < m; ++j) zz += f->h_z[j] * f->h_z[j];
  *loss = f->h_out[0] + 0.5 * zz + 0.5 * (double)m * std::log(2.0 * M_PI);
  std::vector<double> tmp;
  double* gh = grad;
  if (is_device_ptr(grad)) { tmp.resize(m); gh = tmp.data(); }
  for (int64_t j = 0; j < m; ++j) gh[j] = f->h_out[1 + j] + f->h_z[j];
  if (gh != grad) MLN_HIP(ctx, hipMemcpy(grad, gh, sizeof(double) * m, hipMemcpyHostToDevice));
  if (hess_diag) {
    std::vector<double> th;
    double* hh = hess_diag;
    if (is_device_ptr(hess_diag)) { th.resize(m); hh = th.data(); }
    for (int64_t j = 0; j < m; ++j) hh[j] = f->h_out[1 + m + j] + 1.0;
    if (hh != hess_diag) MLN_HIP(ctx, hipMemcpy(hess_diag, hh, sizeof(double) * m, hipMemcpyHostToDevice));
  }
  return MLN_OK;
}

extern "C" int mln_transform(mln_fit* f, const double* z, double mu, double* f_out) {
  if (!f || !z || (f->n > 0 && !f_out)) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  if (f->n == 0) return MLN_OK;
  MLN_HIP(ctx, hipMemcpyAsync(f->d_z, z, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  if (f->f_final >= 0 && mu == f->mu && !is_device_ptr(z) && f->z_cached.size() == (size_t)f->m &&
      std::memcmp(z, f->z_cached.data(), sizeof(double) * f->m) == 0) {
    // the last accepted pass of the MAP solve stored exactly this vector (same kernel, same operands)
    MLN_HIP(ctx, hipMemcpyAsync(f_out, f->f_keep[f->f_final], sizeof(double) * (size_t)f->n, hipMemcpyDefault, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLN_OK;
  }
  DevOut o;
  MLN_TRY(o.init(ctx, f_out, (size_t)f->n));
  ObjArgs a = obj_args(f);
  if (f->kspace) {
    MLN_TRY(fit_w_from_z(f, f->d_z, f->d_w, is_device_ptr(z) ? nullptr : z));
    a.z = f->d_w;
  }
  a.f_out = o.dev;
  a.mu = mu;
  MLN_TRY(launch_objective(ctx, a));
  return o.commit();
}

// G (m x ldg, full symmetric) = alpha * A^T A for the row-major A (rows x m, leading dim lda), all-reduced.
// `quantised`: A holds covariance values in [0, 1] and the result only feeds a preconditioner -- the Gram of A rounded
// to 23 fractional bits, exact in integers on the int8 matrix cores (gram_i8.hip).
static int gram_of(mln_ctx* ctx, const double* A, int64_t lda, int64_t rows, int64_t m, double alpha, double* G,
                   int64_t ldg, bool quantised) {
  int split = quantised ? gram_i8_splits(rows, m) : (int)(rows / 8192);
  if (split < 1) split = 1;
  if (split > 16 && !quantised) split = 16;
  const size_t stride = (size_t)m * ldg;
  double* parts = nullptr;
  if (split > 1) MLN_HIP(ctx, mln_dmalloc((void**)&parts, sizeof(double) * stride * split));
  GemmArgs g{};
  g.A = A; g.lda = lda; g.B = A; g.ldb = lda;
  g.C = (split > 1) ? parts : G; g.ldc = ldg;
  g.M = m; g.N = m; g.K = rows; g.alpha = alpha; g.beta = 0.0; g.ta = 1; g.tb = 0; g.lower_only = 1;
  g.split_k = split; g.c_split_stride = (int64_t)stride;
  int rc = MLN_OK;
  if (split > 1) rc = (hipMemsetAsync(parts, 0, sizeof(double) * stride * split, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
  else rc = (hipMemsetAsync(G, 0, sizeof(double) * stride, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
  if (rc == MLN_OK && rows > 0)
    rc = quantised ? launch_gram_i8(ctx, A, lda, rows, m, alpha, g.C, ldg, (int64_t)stride, split) : launch_dgemm(ctx, g);
  if (rc == MLN_OK && split > 1) rc = launch_sum_partials(ctx, parts, split, (int64_t)stride, G, (int64_t)stride, 0.0);
  if (rc == MLN_OK) rc = launch_symmetrize_from_lower(ctx, G, m, ldg);
  if (rc == MLN_OK) rc = dev_allreduce(ctx, G, (int64_t)stride);
  (void)hipStreamSynchronize(ctx->stream);
  if (parts) (void)mln_dfree(parts);
  return rc;
}

// Test / measurement hook for gram_i8.hip: out (m x m) = the Gram of round(A 8355711) / 8355711^2, A rows x m with values
// in [0, 1]; ms_out (may be NULL) = milliseconds per call of digit extraction + integer GEMM + sum of the k-chunks.
extern "C" int mln_diag_gram_i8(mln_ctx* ctx, const double* A, int64_t rows, int64_t m, double* out, int32_t reps,
                                double* ms_out) {
  if (!ctx || !A || !out || rows < 1 || m < 1) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevIn a;
  DevOut o;
  MLN_TRY(a.init(ctx, A, (size_t)rows * m));
  MLN_TRY(o.init(ctx, out, (size_t)m * m));
  const int split = gram_i8_splits(rows, m);
  const size_t stride = (size_t)m * m;
  double *parts = nullptr, *G = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&parts, sizeof(double) * stride * split));
  MLN_HIP(ctx, mln_dmalloc((void**)&G, sizeof(double) * stride));
  hipEvent_t e0, e1;
  MLN_HIP(ctx, hipEventCreate(&e0));
  MLN_HIP(ctx, hipEventCreate(&e1));
  int rc = MLN_OK;
  if (reps < 1) reps = 1;
  for (int r = 0; r <= reps && rc == MLN_OK; ++r) {   // round 0 warms up
    if (r == 1) (void)hipEventRecord(e0, ctx->stream);
    rc = (hipMemsetAsync(parts, 0, sizeof(double) * stride * split, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
    if (rc == MLN_OK) rc = launch_gram_i8(ctx, a.dev, m, rows, m, 1.0, parts, m, (int64_t)stride, split);
    if (rc == MLN_OK) rc = launch_sum_partials(ctx, parts, split, (int64_t)stride, G, (int64_t)stride, 0.0);
  }
  (void)hipEventRecord(e1, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_out) *ms_out = ms / reps;
  if (rc == MLN_OK) rc = launch_symmetrize_from_lower(ctx, G, m, m);
  if (rc == MLN_OK) rc = launch_copy_block(ctx, G, m, o.dev, m, m, m);
  if (rc == MLN_OK) rc = o.commit();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(parts); (void)mln_dfree(G);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

// G ~ L^T L from every `row_stride`-th cell of this rank (scaled by row_stride), all-reduced.
// device scratch that frees itself (after draining the stream) on every exit path
struct DevScratch {
  mln_ctx* ctx;
  double* p = nullptr;
  explicit DevScratch(mln_ctx* c) : ctx(c) {}
  hipError_t alloc(size_t bytes) { return mln_dmalloc((void**)&p, bytes); }
  ~DevScratch() {
    if (p) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(p); }
  }
  DevScratch(const DevScratch&) = delete;
  DevScratch& operator=(const DevScratch&) = delete;
};

// ---- column-split m x m work (strong scaling) -----------------------------------------------------------
// The whitening of the Gram and the inverses behind the per-evaluation products are "m right-hand sides through a
// triangular solve": replicated, they cost every rank ~3.7 m^3 flops.  From 3 ranks on, rank r solves only its block of
// columns [r b, (r + 1) b), writes it into a zeroed full matrix, and ONE all-reduce (a sum with zeros: exact, the same
// bits on every rank) assembles the result -- 4 m^3 / N flops per rank for the whitening, 2 m^3 / N for the inverses.
// MELLON_AMD_EMULATE_RANKS=N (tools/emulate_rank.py, one process): rank 0's block is timed, the other blocks are
// computed too (the fit must go on) with their wall time recorded in emu_excluded.
static int split_ranks(const mln_ctx* ctx, int* my_rank, bool* emulate) {
  *my_rank = ctx->rank; *emulate = false;
  int n = ctx->n_ranks;
  if (n <= 1)
    if (const char* ev = std::getenv("MELLON_AMD_EMULATE_RANKS")) { n = std::atoi(ev); *my_rank = 0; *emulate = n > 1; }
  static const int from = std::getenv("MELLON_AMD_COLSPLIT_RANKS") ? std::atoi(std::getenv("MELLON_AMD_COLSPLIT_RANKS")) : 3;
  return (from > 0 && n >= from) ? n : 1;
}

template <typename Body>
static int for_my_column_blocks(mln_fit* f, int n_split, int my_rank, bool emulate, int64_t b, Body body) {
  mln_ctx* ctx = f->ctx;
  for (int r = 0; r < n_split; ++r) {
    if (!emulate && r != my_rank) continue;
    const int64_t c0 = (int64_t)r * b, nb = std::min<int64_t>(b, f->m - c0);
    if (nb <= 0) continue;
    const bool excluded = emulate && r != my_rank;
    double t0 = 0.0;
    if (excluded) { MLN_HIP(ctx, hipStreamSynchronize(ctx->stream)); t0 = now_s(); }
    MLN_TRY(body(c0, nb));
    if (excluded) { MLN_HIP(ctx, hipStreamSynchronize(ctx->stream)); f->emu_excluded += now_s() - t0; }
  }
  return MLN_OK;
}

// G (S = K_s^T K_s, all-reduced, symmetric) <- Lp^-1 S Lp^-T, columns split over the ranks
static int fit_whiten_split(mln_fit* f, double* G, int64_t ldg, int n_split, int my_rank, bool emulate) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, b = pad16((m + n_split - 1) / n_split);
  DevScratch zb(ctx), tb(ctx), ob(ctx);
  const size_t blk = sizeof(double) * (size_t)m * b, full = sizeof(double) * (size_t)m * ldg;
  MLN_HIP(ctx, zb.alloc(blk));
  MLN_HIP(ctx, tb.alloc(blk));
  MLN_HIP(ctx, ob.alloc(full));
  double *Z = zb.p, *T = tb.p, *Out = ob.p;
  MLN_HIP(ctx, hipMemsetAsync(Out, 0, full, ctx->stream));
  MLN_TRY(for_my_column_blocks(f, n_split, my_rank, emulate, b, [&](int64_t c0, int64_t nb) -> int {
    MLN_HIP(ctx, hipMemsetAsync(Z, 0, blk, ctx->stream));
    MLN_TRY(launch_add_diag(ctx, Z + c0 * b, nb, b, 1.0));             // unit columns c0 .. c0 + nb
    MLN_TRY(triinv_solve_left_T(ctx, f->tri, Z, nb, b));                // (Lp^-T)[:, block]
    GemmArgs g{};                                                       // T = S (Lp^-T)[:, block]
    g.A = G; g.lda = ldg; g.B = Z; g.ldb = b; g.C = T; g.ldc = b;
    g.M = m; g.N = nb; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0;
    MLN_TRY(launch_dgemm(ctx, g));
    MLN_TRY(triinv_solve_left(ctx, f->tri, T, nb, b));                  // Lp^-1 S Lp^-T [:, block]
    return launch_copy_block(ctx, T, b, Out + c0, ldg, m, nb);
  }));
  MLN_HIP(ctx, hipMemcpyAsync(G, Out, full, hipMemcpyDeviceToDevice, ctx->stream));
  MLN_TRY(dev_allreduce(ctx, G, (int64_t)m * ldg));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MLN_OK;
}

// inv (m x ld) <- C^-1 and P (m x ld) <- Lp^-T C^-T, column blocks of [C^-T ; P] split over the ranks (both zeroed
// by the caller); tc: the block-scaled copies of C
static int fit_inverses_split(mln_fit* f, const TriInv& tc, double* inv, double* P, int64_t ld, int n_split, int my_rank,
                              bool emulate) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, b = pad16((m + n_split - 1) / n_split);
  DevScratch zb(ctx), qb(ctx);
  const size_t blk = sizeof(double) * (size_t)m * b, full = sizeof(double) * (size_t)m * ld;
  MLN_HIP(ctx, zb.alloc(blk));
  MLN_HIP(ctx, qb.alloc(2 * full));                                     // [C^-T ; P], this rank's columns only
  double *Z = zb.p, *Q = qb.p;
  MLN_HIP(ctx, hipMemsetAsync(Q, 0, 2 * full, ctx->stream));
  MLN_TRY(for_my_column_blocks(f, n_split, my_rank, emulate, b, [&](int64_t c0, int64_t nb) -> int {
    MLN_HIP(ctx, hipMemsetAsync(Z, 0, blk, ctx->stream));
    MLN_TRY(launch_add_diag(ctx, Z + c0 * b, nb, b, 1.0));
    MLN_TRY(triinv_solve_left_T(ctx, tc, Z, nb, b));                    // (C^-T)[:, block]
    MLN_TRY(launch_copy_block(ctx, Z, b, Q + c0, ld, m, nb));
    MLN_TRY(triinv_solve_left_T(ctx, f->tri, Z, nb, b));                // P[:, block] = Lp^-T (C^-T)[:, block]
    return launch_copy_block(ctx, Z, b, Q + (size_t)m * ld + c0, ld, m, nb);
  }));
  MLN_TRY(dev_allreduce(ctx, Q, 2 * (int64_t)m * ld));
  MLN_TRY(launch_transpose(ctx, Q, ld, inv, ld, m));                                  // C^-1
  MLN_TRY(launch_copy_block(ctx, Q + (size_t)m * ld, ld, P, ld, m, ld));              // P
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MLN_OK;
}

__global__ void k_round_bits(double* __restrict__ A, int64_t count, double scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    A[i] = rint(A[i] * scale) / scale;
}

// Lp^-1 as an explicit lower-triangular matrix (once per fit).  With it the whitening of a Gram, Lp^-1 S Lp^-T, and
// P = Lp^-T C^-T are GEMMs over the non-zero K ranges (dgemm kmodes 3 / 4 / 7) instead of chains of 40 dependent block
// solves: 9.3 -> ~5 ms per whitening, 3.9 -> ~1 ms for P at m = 5000.  The explicit inverse multiplies rounding by
// cond(Lp) ~ 1e3-1e4 where the block solves are backward stable -- immaterial for a preconditioner built from a Gram
// quantised to 23 bits, and 1e-12 relative on w = P u.  MELLON_AMD_EXPLICIT_LINV=0 restores the solves.
static bool use_explicit_linv() {
  static const bool on = !(std::getenv("MELLON_AMD_EXPLICIT_LINV") && std::atoi(std::getenv("MELLON_AMD_EXPLICIT_LINV")) == 0);
  return on;
}

static int fit_ensure_linv(mln_fit* f) {
  if (f->Linv) return MLN_OK;
  mln_ctx* ctx = f->ctx;
  const size_t bytes = sizeof(double) * (size_t)f->m * f->ldp;
  MLN_HIP(ctx, mln_dmalloc((void**)&f->Linv, bytes));
  MLN_HIP(ctx, hipMemsetAsync(f->Linv, 0, bytes, ctx->stream));
  MLN_TRY(launch_add_diag(ctx, f->Linv, f->m, f->ldp, 1.0));
  return triinv_solve_left(ctx, f->tri, f->Linv, f->m, f->ldp, true);      // Lp^-1 I, lower triangular right-hand side
}

// G (symmetric, full storage) <- Lp^-1 G Lp^-T through the explicit inverse: two GEMMs
static int fit_whiten_gemm(mln_fit* f, double* G, int64_t ldg) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m;
  MLN_TRY(fit_ensure_linv(f));
  double* T = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&T, sizeof(double) * (size_t)m * ldg));
  GemmArgs g{};
  g.A = f->Linv; g.lda = f->ldp; g.B = G; g.ldb = ldg; g.C = T; g.ldc = ldg;           // T = Lp^-1 G   (rows of Lp^-1 end at the diagonal)
  g.M = m; g.N = m; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.kmode = 3;
  int rc = launch_dgemm(ctx, g);
  GemmArgs h{};
  h.A = T; h.lda = ldg; h.B = f->Linv; h.ldb = f->ldp; h.C = G; h.ldc = ldg;           // G = T Lp^-T, lower tiles (symmetric)
  h.M = m; h.N = m; h.K = m; h.alpha = 1.0; h.beta = 0.0; h.ta = 0; h.tb = 1; h.kmode = 4; h.lower_only = 1;
  if (rc == MLN_OK) rc = launch_dgemm(ctx, h);
  if (rc == MLN_OK) rc = launch_symmetrize_from_lower(ctx, G, m, ldg);
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(T);
  return rc;
}

static int fit_gram(mln_fit* f, double* G, int64_t ldg, int64_t row_stride) {
  mln_ctx* ctx = f->ctx;
  if (row_stride < 1) row_stride = 1;
  // cells whose GLOBAL index is a multiple of row_stride: the sample -- and with it the preconditioner and the
  // iteration path -- does not depend on how the cells are sharded (up to the order of the all-reduce sum)
  const int64_t first = (row_stride - f->row0 % row_stride) % row_stride;
  const int64_t rows = (f->n > first) ? (f->n - first + row_stride - 1) / row_stride : 0;
  const double* Ls = f->L + first * f->ldl;
  if (!f->kspace) return gram_of(ctx, Ls, f->ldl * row_stride, rows, f->m, (double)row_stride, G, ldg);
  // implicit mode: G = Lp^-1 (K_s^T K_s) Lp^-T -- the Gram of the sampled rows of K itself (strided
  // rows read in place) followed by two m x m block solves.  Rounding in K_s^T K_s is amplified by
  // |Lp^-1|^2, which would matter for a quantity that enters the result; as a preconditioner the
  // outcome is spectrally equivalent to the row-solved Gram within 1e-3 (measured), at none of the
  // n_s m^2 triangular-solve flops.
  // With many ranks the sampled rows are few per rank (~12 m / N) while the two m x m block solves are replicated:
  // from N = 7 on it is cheaper for every rank to whiten ITS rows first, L_s = K_s Lp^-T (rows x m^2 flops, < 2 m^3),
  // and to all-reduce the Gram of those -- the explicit route's arithmetic, no replicated solve, same single
  // collective.  (The choice depends on the rank count only, so every rank takes the same branch.)
  static const int row_solve_from = std::getenv("MELLON_AMD_GRAM_ROWSOLVE_RANKS") ? std::atoi(std::getenv("MELLON_AMD_GRAM_ROWSOLVE_RANKS")) : 0;   // superseded by the column split (fit_whiten_split)
  if (ctx->n_ranks >= row_solve_from && row_solve_from > 0) {
    double* R = nullptr;
    const int64_t rr = rows > 0 ? rows : 1;
    MLN_HIP(ctx, mln_dmalloc((void**)&R, sizeof(double) * (size_t)rr * f->ldl));
    int rc = MLN_OK;
    if (rows > 0) rc = launch_copy_block(ctx, Ls, f->ldl * row_stride, R, f->ldl, rows, f->ldl);
    if (rc == MLN_OK && rows > 0) rc = triinv_solve_right_T(ctx, f->tri, R, rows, f->ldl);
    if (rc == MLN_OK) rc = gram_of(ctx, R, f->ldl, rows, f->m, (double)row_stride, G, ldg);   // all-reduced
    (void)hipStreamSynchronize(ctx->stream);
    (void)mln_dfree(R);
    return rc;
  }
  int rc = MLN_OK;
  const int qbits = std::getenv("MELLON_AMD_GRAM_QBITS") ? std::atoi(std::getenv("MELLON_AMD_GRAM_QBITS")) : 0;
  if (qbits > 0) {   // experiment: the Gram of the sampled rows rounded to `qbits` fractional bits
    double* R = nullptr;
    const int64_t rr = rows > 0 ? rows : 1;
    MLN_HIP(ctx, mln_dmalloc((void**)&R, sizeof(double) * (size_t)rr * f->ldl));
    if (rows > 0) rc = launch_copy_block(ctx, Ls, f->ldl * row_stride, R, f->ldl, rows, f->ldl);
    if (rc == MLN_OK && rows > 0)
      hipLaunchKernelGGL(k_round_bits, dim3(2048), dim3(256), 0, ctx->stream, R, rows * f->ldl, std::ldexp(1.0, qbits));
    if (rc == MLN_OK) rc = gram_of(ctx, R, f->ldl, rows, f->m, (double)row_stride, G, ldg);
    (void)hipStreamSynchronize(ctx->stream);
    (void)mln_dfree(R);
  } else {
    // bounded covariances: 23-bit integer Gram on the int8 matrix cores (the preconditioner needs ~20 bits: gram_i8.hip)
    // ... and only where the caller asked for a SAMPLED Gram (row_stride > 1: a preconditioner by construction);
    // row_stride == 1 is the reference's exact Ridge matrix / the Gram whose eigenvalues are results
    bool quant = f->cov_bounded01 && f->m >= 256 && row_stride > 1;
    if (const char* ev = std::getenv("MELLON_AMD_GRAM_I8")) quant = quant && std::atoi(ev) != 0;
    rc = gram_of(ctx, Ls, f->ldl * row_stride, rows, f->m, (double)row_stride, G, ldg, quant);   // all-reduced
  }
  {
    int my_rank = 0; bool emulate = false;
    const int n_split = split_ranks(ctx, &my_rank, &emulate);
    if (rc == MLN_OK && n_split > 1) return fit_whiten_split(f, G, ldg, n_split, my_rank, emulate);
  }
  if (rc == MLN_OK && use_explicit_linv()) return fit_whiten_gemm(f, G, ldg);
  double* T = nullptr;
  if (rc == MLN_OK) {
    hipError_t e = mln_dmalloc((void**)&T, sizeof(double) * (size_t)f->m * ldg);
    if (e == hipSuccess) e = hipMemsetAsync(T, 0, sizeof(double) * (size_t)f->m * ldg, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc Gram temp", __FILE__, __LINE__);
  }
  if (rc == MLN_OK) rc = triinv_solve_left(ctx, f->tri, G, f->m, ldg);          // Lp^-1 S
  if (rc == MLN_OK) rc = launch_transpose(ctx, G, ldg, T, ldg, f->m);           // (Lp^-1 S)^T
  if (rc == MLN_OK) rc = triinv_solve_left(ctx, f->tri, T, f->m, ldg);          // Lp^-1 S Lp^-T (symmetric)
  if (rc == MLN_OK) rc = (hipMemcpyAsync(G, T, sizeof(double) * (size_t)f->m * ldg, hipMemcpyDeviceToDevice,
                                         ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
  (void)hipStreamSynchronize(ctx->stream);
  if (T) (void)mln_dfree(T);
  return rc;
}

// rhs (m) = L^T t over this rank's rows, all-reduced; t is a device vector of length n
static int fit_gemvT(mln_fit* f, const double* t_dev, double* rhs_dev) {
  mln_ctx* ctx = f->ctx;
  ObjArgs a = obj_args(f);
  a.weights = t_dev;
  a.part_loss = nullptr;
  MLN_TRY(launch_objective(ctx, a));
  MLN_TRY(launch_reduce_obj(ctx, a, f->d_out));
  MLN_TRY(dev_allreduce(ctx, f->d_out, 1 + f->m));
  if (f->kspace) MLN_TRY(triinv_solve_left(ctx, f->tri, f->d_out + 1, 1, 1));   // Lp^-1 (K^T t)
  MLN_HIP(ctx, hipMemcpyAsync(rhs_dev, f->d_out + 1, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
  return MLN_OK;
}

// C C^T = L^T L + I and C^-1 (explicit, lower): the Ridge matrix of parameters.py:895-896 doubles as
// the preconditioner of the MAP solve, because the MAP Hessian I + L^T diag(e^{f+V}) L equals it
// wherever e^{f+V} = 1 (i.e. where f matches the nearest-neighbour estimate the Ridge regresses on).
// With row_stride > 1 the Gram is estimated from every row_stride-th cell: any SPD matrix is a valid
// preconditioner / initial guess for a strictly convex problem, and ~8 m rows already give the same
// iteration count as all n (measured), at 1/row_stride of the n m^2 flops.
static void fit_drop_precond_operators(mln_fit* f) {
  (void)hipStreamSynchronize(f->ctx->stream);
  void* ptrs[] = {f->Cinv, f->P, f->Q1, f->Q2};
  for (void* p : ptrs) if (p) (void)mln_dfree(p);
  f->Cinv = nullptr; f->P = nullptr; f->Q1 = nullptr; f->Q2 = nullptr;
}

// f->C holds the (whitened) Gram: add the prior's identity, factor C C^T, and build C^-1, P = Lp^-T C^-T and the stacked
// per-evaluation operators Q1, Q2
static int fit_factor_precond(mln_fit* f) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, ldg = f->ldl;
  const size_t bytes = sizeof(double) * (size_t)m * ldg;
  int my_rank = 0; bool emulate = false;
  const int n_split = f->kspace ? split_ranks(ctx, &my_rank, &emulate) : 1;
  int rc = launch_add_diag(ctx, f->C, m, ldg, 1.0);  // Ridge alpha = 1 / the prior's Hessian
  if (rc == MLN_OK) rc = dev_cholesky_lower(ctx, f->C, m, ldg);
  TriInv t;
  if (rc == MLN_OK) rc = triinv_build(ctx, f->C, m, ldg, true, false, &t);
  double* inv = nullptr;
  if (rc == MLN_OK) {
    hipError_t e = mln_dmalloc((void**)&inv, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(inv, 0, bytes, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc C^-1", __FILE__, __LINE__);
  }
  if (rc == MLN_OK && n_split > 1) {                                // column blocks over the ranks, one all-reduce
    hipError_t e = mln_dmalloc((void**)&f->P, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(f->P, 0, bytes, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc P", __FILE__, __LINE__);
    if (rc == MLN_OK) rc = fit_inverses_split(f, t, inv, f->P, ldg, n_split, my_rank, emulate);
  } else {
  if (rc == MLN_OK) rc = launch_add_diag(ctx, inv, m, ldg, 1.0);
  if (rc == MLN_OK) rc = triinv_solve_left(ctx, t, inv, m, ldg, true);   // C^-1 = C^-1 I (lower triangular B)
  if (rc == MLN_OK && f->kspace) {                                  // P = Lp^-T C^-T
    hipError_t e = mln_dmalloc((void**)&f->P, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(f->P, 0, bytes, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc P", __FILE__, __LINE__);
    if (rc == MLN_OK && use_explicit_linv()) {
      // P^T = C^-1 Lp^-1: two lower triangular factors, lower triangular product (K range column .. row)
      rc = fit_ensure_linv(f);
      double* X = nullptr;
      if (rc == MLN_OK) {
        e = mln_dmalloc((void**)&X, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(X, 0, bytes, ctx->stream);
        if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc P^T", __FILE__, __LINE__);
      }
      GemmArgs g{};
      g.A = inv; g.lda = ldg; g.B = f->Linv; g.ldb = f->ldp; g.C = X; g.ldc = ldg;
      g.M = m; g.N = m; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.kmode = 7; g.lower_only = 1;
      if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
      if (rc == MLN_OK) rc = launch_transpose(ctx, X, ldg, f->P, ldg, m);
      (void)hipStreamSynchronize(ctx->stream);
      if (X) (void)mln_dfree(X);
    } else {
    if (rc == MLN_OK) rc = launch_transpose(ctx, inv, ldg, f->P, ldg, m);
    if (rc == MLN_OK) rc = triinv_solve_left_T(ctx, f->tri, f->P, m, ldg, true);   // C^-T is upper triangular
    }
  }
  }
  if (rc == MLN_OK) {   // stacked operators for the per-evaluation row-GEMVs
    const int64_t ld = ldg;
    const size_t blk = (size_t)m * ld;
    const int nq1 = f->kspace ? 2 : 1;
    hipError_t e = mln_dmalloc((void**)&f->Q1, sizeof(double) * blk * nq1);
    if (e == hipSuccess) e = mln_dmalloc((void**)&f->Q2, sizeof(double) * blk * 2);
    if (e == hipSuccess) e = hipMemsetAsync(f->Q1, 0, sizeof(double) * blk * nq1, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(f->Q2, 0, sizeof(double) * blk * 2, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc stacked operators", __FILE__, __LINE__);
    if (rc == MLN_OK) rc = launch_transpose(ctx, inv, ld, f->Q1, ld, m);                      // C^-T
    if (rc == MLN_OK && f->kspace) rc = launch_copy_block(ctx, f->P, ld, f->Q1 + blk, ld, m, ld);   // P below it
    if (rc == MLN_OK) rc = launch_copy_block(ctx, inv, ld, f->Q2, ld * 2, m, ld);             // C^-1
    // explicit factor: g_u = C^-1 (z + L^T(a-1)) = [C^-1 | C^-1] [z ; r] -- the same two-segment product
    if (rc == MLN_OK && !f->kspace) rc = launch_copy_block(ctx, inv, ld, f->Q2 + ld, ld * 2, m, ld);
    if (rc == MLN_OK && f->kspace) {                                                           // P^T beside it
      double* Pt = nullptr;
      e = mln_dmalloc((void**)&Pt, sizeof(double) * blk);
      if (e == hipSuccess) e = hipMemsetAsync(Pt, 0, sizeof(double) * blk, ctx->stream);
      if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc P^T", __FILE__, __LINE__);
      if (rc == MLN_OK) rc = launch_transpose(ctx, f->P, ld, Pt, ld, m);
      if (rc == MLN_OK) rc = launch_copy_block(ctx, Pt, ld, f->Q2 + ld, ld * 2, m, ld);
      (void)hipStreamSynchronize(ctx->stream);
      if (Pt) (void)mln_dfree(Pt);
    }
  }
  (void)hipStreamSynchronize(ctx->stream);
  triinv_free(&t);
  if (rc == MLN_OK) f->Cinv = inv; else if (inv) (void)mln_dfree(inv);
  return rc;
}

static int fit_build_precond(mln_fit* f, int64_t row_stride) {
  if (f->Cinv) return MLN_OK;
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, ldg = f->ldl;
  const size_t bytes = sizeof(double) * (size_t)m * ldg;
  double t0 = now_s(), ex0 = f->emu_excluded;
  const double ex_start = f->emu_excluded;
  MLN_HIP(ctx, mln_dmalloc((void**)&f->C, bytes));
  int rc = fit_gram(f, f->C, ldg, row_stride);
  f->times[3] += now_s() - t0 - (f->emu_excluded - ex0);
  double t1 = now_s(); ex0 = f->emu_excluded;
  if (rc == MLN_OK) rc = fit_factor_precond(f);
  f->times[4] += now_s() - t1 - (f->emu_excluded - ex0);
  if (rc == MLN_OK) { f->precond_stride = row_stride < 1 ? 1 : row_stride; f->build_seconds = now_s() - t0 - (f->emu_excluded - ex_start); }
  return rc;
}

// The solver's SECOND preconditioner (precond_rebuild.hip): C C^T = I + sum_i a_i L_i L_i^T estimated from an importance
// sample of ~rows_per_m * m cells at the point whose rows' f = L z + mu is `f_dev`; replaces C, C^-1, P, Q1, Q2.
static int fit_rebuild_precond(mln_fit* f, const double* f_dev, double rows_per_m) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, ldg = f->ldl;
  RebuildSelection sel{};
  const double target = rows_per_m * (double)m;
  const bool tr_on = std::getenv("MELLON_AMD_TRACE") != nullptr;
  double tt[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int i, double& t0) { if (tr_on) { (void)hipStreamSynchronize(ctx->stream); const double t1 = now_s(); tt[i] += t1 - t0; t0 = t1; } };
  double tl = now_s();
  MLN_TRY(rebuild_select_rows(ctx, f_dev, f->V, f->n, f->row0, target, 0x6d656c6c6f6eull, &sel));
  lap(0, tl);
  double* R = nullptr;
  int rc = MLN_OK;
  const int64_t rr = sel.rows > 0 ? sel.rows : 1;
  if (mln_dmalloc((void**)&R, sizeof(double) * (size_t)rr * f->ldl) != hipSuccess) rc = MLN_ERR_HIP;
  if (rc == MLN_OK) rc = launch_gather_scale_rows(ctx, f->L, f->ldl, sel.idx, sel.scale, sel.rows, R);
  fit_drop_precond_operators(f);
  lap(1, tl);
  if (rc == MLN_OK) {
    // scaled covariances stay in [0, 1]: the integer Gram applies where it did for the first preconditioner
    bool quant = f->kspace && f->cov_bounded01 && m >= 256;
    if (const char* ev = std::getenv("MELLON_AMD_GRAM_I8")) quant = quant && std::atoi(ev) != 0;
    rc = gram_of(ctx, R, f->ldl, sel.rows, m, sel.w_max, f->C, ldg, quant);                   // all-reduced
    // (The integer Gram is that of the rows ROUNDED to 1 / 8355711; the rounding's own Gram, rows * var * I times w_max
    //  and the whitening's |Lp^-1|^2, is an O(0.1) multiple of K_uu^-1.  Subtracting its expectation was tried: no change
    //  in the pass count at w_max ~ 5e3, and at w_max ~ 1e5 the subtraction itself made the matrix indefinite.)
    if (std::getenv("MELLON_AMD_TRACE"))
      fprintf(stderr, "[trace] rebuild: %lld of %lld local rows kept (target %.0f global), c = %.4g, 1/c = %.4g, w_max = %.4g, sum a = %.6g\n",
              (long long)sel.rows, (long long)f->n, target, sel.c, 1.0 / sel.c, sel.w_max, sel.sum_a);
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (R) (void)mln_dfree(R);
  rebuild_selection_free(ctx, &sel);
  lap(2, tl);
  if (rc == MLN_OK && f->kspace) {                                                              // Lp^-1 G Lp^-T
    int my_rank = 0; bool emulate = false;
    const int n_split = split_ranks(ctx, &my_rank, &emulate);
    if (n_split > 1) rc = fit_whiten_split(f, f->C, ldg, n_split, my_rank, emulate);
    else if (use_explicit_linv()) rc = fit_whiten_gemm(f, f->C, ldg);
    else {
      double* T = nullptr;
      hipError_t e = mln_dmalloc((void**)&T, sizeof(double) * (size_t)m * ldg);
      if (e == hipSuccess) e = hipMemsetAsync(T, 0, sizeof(double) * (size_t)m * ldg, ctx->stream);
      if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc Gram temp", __FILE__, __LINE__);
      if (rc == MLN_OK) rc = triinv_solve_left(ctx, f->tri, f->C, m, ldg);
      if (rc == MLN_OK) rc = launch_transpose(ctx, f->C, ldg, T, ldg, m);
      if (rc == MLN_OK) rc = triinv_solve_left(ctx, f->tri, T, m, ldg);
      if (rc == MLN_OK) rc = (hipMemcpyAsync(f->C, T, sizeof(double) * (size_t)m * ldg, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
      (void)hipStreamSynchronize(ctx->stream);
      if (T) (void)mln_dfree(T);
    }
  }
  lap(3, tl);
  if (rc == MLN_OK) rc = fit_factor_precond(f);
  lap(4, tl);
  if (tr_on) fprintf(stderr, "[trace] rebuild ms: select %.2f, gather+drop %.2f, gram %.2f, whiten %.2f, factor+inverses+stacks %.2f\n",
                     1e3 * tt[0], 1e3 * tt[1], 1e3 * tt[2], 1e3 * tt[3], 1e3 * tt[4]);
  return rc;
}

// y (m) = M^T w  (trans = 1)  or  M w  (trans = 0) for an m x ldl matrix M, via the streaming kernels
// of objective.hip (GEMV-T mode / f-only mode); all pointers on the device.
static int fit_small_gemv(mln_fit* f, const double* M, int trans, const double* w, double* y) {
  mln_ctx* ctx = f->ctx;
  ObjArgs a{};
  a.L = M; a.ldl = f->ldl; a.n = f->m; a.m = f->m; a.mu = 0.0;
  a.part_grad = f->part_grad; a.part_hess = nullptr; a.part_loss = nullptr;
  a.m_pad = f->ldl;
  int64_t steps = (f->m + 1) / 2;
  a.n_wg = (int)((steps < f->n_wg_cap) ? (steps > 0 ? steps : 1) : f->n_wg_cap);
  if (trans) {
    a.weights = w;
    MLN_TRY(launch_objective(ctx, a));
    MLN_TRY(launch_reduce_obj(ctx, a, f->d_tmp));
    MLN_HIP(ctx, hipMemcpyAsync(y, f->d_tmp + 1, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    a.z = w;
    a.f_out = y;
    MLN_TRY(launch_objective(ctx, a));
  }
  return MLN_OK;
}

extern "C" int mln_fit_set_row_offset(mln_fit* f, int64_t global_row0) {
  if (!f || global_row0 < 0) return MLN_ERR_ARG;
  f->row0 = global_row0;
  return MLN_OK;
}

extern "C" int mln_precond_build(mln_fit* f, int64_t row_stride) {
  if (!f) return MLN_ERR_ARG;
  MLN_HIP(f->ctx, hipSetDevice(f->ctx->device));
  if (row_stride < 1) row_stride = 1;
  if (f->Cinv && f->precond_stride != row_stride) {
    // an explicit request for a DIFFERENT sample (e.g. the reference's exact Ridge, stride 1, after a sampled
    // preconditioner had been built): drop the factor and build the one asked for
    fit_drop_precond_operators(f);
    if (f->C) { (void)mln_dfree(f->C); f->C = nullptr; }
    f->precond_stride = 0;
  }
  return fit_build_precond(f, row_stride);
}

extern "C" int mln_ridge_init(mln_fit* f, const double* target, double* z0) {
  if (!f || !z0 || (f->n > 0 && !target)) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_TRY(fit_build_precond(f, 1));   // exact Ridge unless a (subsampled) factor was built before
  double t0 = now_s();
  DevIn dt;
  MLN_TRY(dt.init(ctx, target, (size_t)f->n));
  // z0 = (L^T L + I)^-1 L^T t = C^-T C^-1 (L^T t);  implicit mode: C^-1 L^T t = P^T (K^T t)
  // With a sampled Gram (stride s >= 11) the right-hand side is taken over the SAME cells, s L_s^T t_s: z0 is then the
  // exact Ridge solution of the subsample -- the problem the solver's first phase works on -- and costs 1/s of a pass.
  // (Not beyond 8192 landmarks: the segmented pass, launch_objective_wide, has no row map; there the right-hand side
  //  runs over all cells against the sampled Gram -- a valid start for a solve that the host's L-BFGS-B drives anyway.)
  int64_t rs = (f->precond_stride >= 11 && f->m <= objective_max_m_one_pass()) ? f->precond_stride : 1;
  if (const char* ev = std::getenv("MELLON_AMD_SUBSAMPLE")) { if (std::atoi(ev) == 0) rs = 1; }
  ObjArgs a = obj_args(f);
  a.weights = dt.dev;
  a.part_loss = nullptr;
  if (rs > 1) {
    int64_t first = 0, rows = 0;
    fit_sample_rows(f, rs, &first, &rows);
    a.n = rows; a.row_first = first; a.row_stride = rs; a.out_scale = (double)rs;
  }
  if (f->kspace) {
    a.L32 = f->L32;   // the Ridge solution only seeds the solve: its right-hand side may come from the 32-bit copy
    a.l32_fixed = f->l32_fixed;
    MLN_TRY(launch_objective(ctx, a));
    MLN_TRY(launch_reduce_obj(ctx, a, f->d_out));
    MLN_TRY(dev_allreduce(ctx, f->d_out, 1 + f->m));
    MLN_TRY(fit_small_gemv(f, f->P, 1, f->d_out + 1, f->d_gu));
  } else {
    MLN_TRY(launch_objective(ctx, a));
    MLN_TRY(launch_reduce_obj(ctx, a, f->d_out));
    MLN_TRY(dev_allreduce(ctx, f->d_out, 1 + f->m));
    MLN_TRY(fit_small_gemv(f, f->Cinv, 0, f->d_out + 1, f->d_gu));
  }
  MLN_TRY(fit_small_gemv(f, f->Cinv, 1, f->d_gu, f->d_z));       // z0 = C^-T (.)   [d_gu plays the role of u0]
  MLN_TRY(fit_cache_pair_from_u(f, f->d_gu));
  MLN_HIP(ctx, hipMemcpyAsync(z0, f->d_z, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  f->times[4] += now_s() - t0;
  return MLN_OK;
}

// u <-> z of the preconditioned variable  z = C^-T u
extern "C" int mln_precond_apply(mln_fit* f, int32_t mode, const double* in, double* out) {
  if (!f || !in || !out) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_TRY(fit_build_precond(f, 1));
  MLN_HIP(ctx, hipMemcpyAsync(f->d_u, in, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  if (mode == 0) MLN_TRY(fit_small_gemv(f, f->C, 1, f->d_u, f->d_gu));          // u = C^T z
  else if (mode == 1) MLN_TRY(fit_small_gemv(f, f->Cinv, 1, f->d_u, f->d_gu));  // z = C^-T u
  else if (mode == 2) MLN_TRY(fit_small_gemv(f, f->Cinv, 0, f->d_u, f->d_gu));  // g_u = C^-1 g_z
  else { mln_set_error(ctx, "mln_precond_apply: unknown mode"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipMemcpyAsync(out, f->d_gu, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MLN_OK;
}

// One evaluation of the preconditioned objective at the device vector `u`, enqueued without any host wait:
//   [z ; w] = Q1 u  ->  one pass over the n x m buffer  ->  fixed-order reduction  ->  all-reduce of [r ; lik]
//   ->  g_u = Q2 [z ; r]                                     (z -> d_zr, r -> d_zr + ld2, lik -> d_zr[ld2 + m], g_u -> gn)
//   explicit mode: f = L z + mu,            g_u = C^-1 (z + L^T (a - 1))             Q2 = [C^-1 | C^-1]
//   implicit mode: f = K (P u) + mu,        g_u = C^-1 z + P^T (K^T (a - 1)),        Q2 = [C^-1 | P^T],  P = Lp^-T C^-T
// gate == nullptr: `use32` picks the streamed copy.  gate != nullptr (device-resident solver): both objective kernels
// are launched and the one the solver's state does not select returns at once; everything is a no-op after DONE.
// ev (optional): three events -- before the fp32 pass, between the two, after the fp64 pass.
static int fit_enqueue_eval(mln_fit* f, const double* u_dev, double* gn_dev, bool use32, const int* gate,
                            hipEvent_t* ev, const std::vector<int64_t>* sub_strides = nullptr) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, ld = f->ldl, ld2 = f->ld2;
  GemvTri g1{f->Q1, ld, f->kspace ? 2 * m : m, u_dev, f->d_zr, f->kspace ? f->d_w : nullptr, 1, m, m, 0, 0, gate};
  MLN_TRY(launch_gemv_tri(ctx, g1));                                   // C^-T, P: upper triangular blocks
  ObjArgs a = obj_args(f);
  a.z = f->kspace ? f->d_w : f->d_zr;
  a.gate = gate;
  static const bool no_fkeep = std::getenv("MELLON_AMD_NO_FKEEP") != nullptr;
  if (gate && !no_fkeep && f->f_keep[0] && objective_can_keep_f(f->n, f->n_wg)) { a.f_keep[0] = f->f_keep[0]; a.f_keep[1] = f->f_keep[1]; a.f_slot = &f->sv.st->f_slot; }
  if (ev) MLN_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
  if (f->L32 && (gate || use32)) {
    ObjArgs a32 = a;
    a32.L32 = f->L32;
    a32.l32_fixed = f->l32_fixed;
    a32.gate_want = MLN_GATE_F32;
    if (gate) a32.cap = &f->sv.st->cap;
    MLN_TRY(launch_objective(ctx, a32));
  }
  if (ev) MLN_HIP(ctx, hipEventRecord(ev[1], ctx->stream));
  if (gate || !use32 || !f->L32) {
    a.gate_want = MLN_GATE_F64;
    MLN_TRY(launch_objective(ctx, a));
  }
  if (gate && sub_strides) {
    // the subsample objectives of the solver's first phase: the same fp64 kernel over every s-th row, one launch per
    // level (the solver's state says which one works; same grid: workgroups past the shorter row range write zero
    // partials), partial sums scaled by s
    for (size_t lv = 0; lv < sub_strides->size(); ++lv) {
      const int64_t sub_stride = (*sub_strides)[lv];
      ObjArgs as = a;
      int64_t first = 0, rows = 0;
      fit_sample_rows(f, sub_stride, &first, &rows);
      as.n = rows; as.row_first = first; as.row_stride = sub_stride; as.out_scale = (double)sub_stride;
      as.f_keep[0] = as.f_keep[1] = nullptr; as.f_slot = nullptr;
      as.gate_want = MLN_GATE_SUB;
      as.gate2 = &f->sv.st->sub_level; as.gate2_want = (int)lv;
      MLN_TRY(launch_objective(ctx, as));
    }
  }
  if (ev) MLN_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
  MLN_TRY(launch_reduce_obj2(ctx, a, f->d_zr + ld2 + m, f->d_zr + ld2));
  MLN_TRY(dev_allreduce(ctx, f->d_zr + ld2, m + 1));
  GemvTri g2{f->Q2, 2 * ld, m, f->d_zr, gn_dev, nullptr, 0, m, m, ld, ld2, gate};
  MLN_TRY(launch_gemv_tri(ctx, g2));                                   // C^-1 | P^T: lower triangular blocks
  return MLN_OK;
}

// host-synchronous form (SciPy-driven route, mln_objective_precond)
static int fit_objective_u(mln_fit* f, const double* u, double* loss, double* grad_u, double* z_out,
                           bool use32 = false) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m;
  MLN_HIP(ctx, hipMemcpyAsync(f->d_u, u, sizeof(double) * m, hipMemcpyDefault, ctx->stream));
  hipEvent_t ev[3] = {f->ev0, use32 ? f->ev1 : f->ev0, f->ev1};
  MLN_TRY(fit_enqueue_eval(f, f->d_u, f->d_gu, use32, nullptr, ev));
  MLN_HIP(ctx, hipMemcpyAsync(f->h_out, f->d_zr + f->ld2 + m, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MLN_HIP(ctx, hipMemcpyAsync(f->h_out + 1, f->d_gu, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  MLN_HIP(ctx, hipMemcpyAsync(f->h_z, f->d_zr, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  obj_account(f, use32);
  double zz = 0.0;
  for (int64_t j = 0; j < m; ++j) zz += f->h_z[j] * f->h_z[j];
  *loss = f->h_out[0] + 0.5 * zz + 0.5 * (double)m * std::log(2.0 * M_PI);
  std::memcpy(grad_u, f->h_out + 1, sizeof(double) * m);
  if (z_out) std::memcpy(z_out, f->h_z, sizeof(double) * m);
  return MLN_OK;
}

extern "C" int mln_objective_precond(mln_fit* f, const double* u, double* loss, double* grad_u, double* z_out) {
  if (!f || !u || !loss || !grad_u) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->V) { mln_set_error(ctx, "mln_fit_set_likelihood has not been called"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_TRY(fit_build_precond(f, 1));
  return fit_objective_u(f, u, loss, grad_u, z_out);
}

// ---- a-8: the MAP solve: device-resident L-BFGS (solver.hip) --------------------------------------------------
// Reference: inference.minimize_lbfgsb (inference.py:272-288) = SciPy L-BFGS-B without bounds.  Same method
// (limited-memory BFGS two-loop recursion, H0 = s.y / y.y, sufficient-decrease backtracking from step 1 -- Armijo
// only: SciPy's dcsrch also enforces the curvature condition, so iteration counts are not comparable one to one) and
// SciPy's stopping tests (relative decrease <= ftol, max|g| <= gtol, maxiter) on the preconditioned variable u.
// The optimiser's vectors and decisions live on the device; the host only enqueues evaluation chains in batches
// and looks at the solver's state once per batch.
static int fit_solver_alloc(mln_fit* f, int maxcor) {
  mln_ctx* ctx = f->ctx;
  if (f->sv_block && f->sv_maxcor >= maxcor) return MLN_OK;
  if (f->sv_block) { MLN_HIP(ctx, hipStreamSynchronize(ctx->stream)); MLN_HIP(ctx, mln_dfree(f->sv_block)); f->sv_block = nullptr; }
  const size_t ld = (size_t)f->ldl;
  const size_t n_dbl = 6 * ld + 2 * (size_t)maxcor * ld + 2 * 64 + 4 * 512 + (sizeof(SolverState) + 63) / 64 * 8;
  MLN_HIP(ctx, mln_dmalloc(&f->sv_block, sizeof(double) * n_dbl));
  MLN_HIP(ctx, hipMemsetAsync(f->sv_block, 0, sizeof(double) * n_dbl, ctx->stream));
  double* p = (double*)f->sv_block;
  SolverBuffers& b = f->sv;
  b.u = p; p += ld; b.g = p; p += ld; b.un = p; p += ld; b.gn = p; p += ld; b.d = p; p += ld;
  b.S = p; p += (size_t)maxcor * ld; b.Y = p; p += (size_t)maxcor * ld;
  b.rho = p; p += 64; b.yy = p; p += 64;
  b.c = p; p += ld;
  b.trace = p; p += 4 * 512;
  b.st = (SolverState*)p;
  b.ld = (int64_t)ld;
  b.z = f->d_zr;
  b.lik = f->d_zr + f->ld2 + f->m;
  f->sv_maxcor = maxcor;
  if (!f->h_state) MLN_HIP(ctx, hipHostMalloc((void**)&f->h_state, sizeof(SolverState), hipHostMallocDefault));
  return MLN_OK;
}

extern "C" int mln_map_solve(mln_fit* f, const double* z0, const mln_solver_opts* opts_in, double* z_out,
                             double* loss_out, int32_t* n_eval_out, int32_t* n_iter_out, int32_t* status_out) {
  if (!f || !z0 || !z_out) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->V) { mln_set_error(ctx, "mln_fit_set_likelihood has not been called"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  if (f->m > 8192) {
    mln_set_error(ctx, "mln_map_solve: the device-resident solver holds at most 8192 landmarks; drive mln_objective_precond "
                       "from the host instead (the Python binding does: inference.minimize_lbfgsb)");
    return MLN_ERR_UNSUPPORTED;
  }
  MLN_TRY(fit_build_precond(f, 1));
  mln_solver_opts o = {5000, 10, 30, 1e-13, 1e-7};
  if (opts_in) o = *opts_in;
  if (o.maxcor < 1) o.maxcor = 1;
  if (o.maxcor > 64) o.maxcor = 64;
  if (o.maxls < 1) o.maxls = 1;
  const int64_t m = f->m;
  MLN_TRY(fit_solver_alloc(f, o.maxcor));
  for (int b = 0; b < 2; ++b)
    if (!f->f_keep[b]) MLN_HIP(ctx, mln_dmalloc((void**)&f->f_keep[b], sizeof(double) * (size_t)(f->n > 0 ? f->n : 1)));
  f->f_final = -1;
  // u0 = C^T z0, identical on every rank
  MLN_HIP(ctx, hipMemcpyAsync(f->d_u, z0, sizeof(double) * m, hipMemcpyDefault, ctx->stream));
  MLN_TRY(fit_small_gemv(f, f->C, 1, f->d_u, f->d_gu));
  MLN_TRY(dev_bcast0(ctx, f->d_gu, m));
  // Mixed precision: while an fp32 copy of the n x m buffer exists, the first passes stream it (half the bytes);
  // the solver switches to the fp64 buffer by itself (see k_solver_step) and finishes at the same tolerances as
  // a pure fp64 run.
  const int trace_lvl = std::getenv("MELLON_AMD_TRACE") ? std::atoi(std::getenv("MELLON_AMD_TRACE")) : 0;
  const bool phase32 = f->kspace && f->L32 != nullptr;
  SolverState init{};
  init.gate = phase32 ? MLN_GATE_F32 : MLN_GATE_F64;
  init.mode = MLN_SOLVE_FIRST;
  init.status = 1;
  init.maxiter = o.maxiter; init.maxcor = o.maxcor; init.maxls = o.maxls;
  init.m = (int)m;
  init.ftol = o.ftol; init.gtol = o.gtol;
  // progress per iteration below which the 32-bit surrogate is left for the fp64 buffer (relative to the loss):
  // the fp32 copy's optimum sits ~1e-5 (relative loss) from the true one, the fixed-point copy's ~1e-9
  init.ftol32 = f->l32_fixed ? 1e-9 : 3e-6;
  if (const char* ev = std::getenv("MELLON_AMD_MIXED_FTOL")) init.ftol32 = std::atof(ev);
  // ... and after that first fp64 evaluation the solve continues on the 32-bit copy WITH its first-order correction
  // (solver.hip), the fp64 objective verifying the final point (MELLON_AMD_CORRECTED=0: finish on the fp64 buffer)
  init.use_corr = (phase32 && f->l32_fixed) ? 1 : 0;
  if (const char* ev = std::getenv("MELLON_AMD_CORRECTED")) init.use_corr = init.use_corr && std::atoi(ev) != 0;
  init.prior_const = 0.5 * (double)m * std::log(2.0 * M_PI);
  init.t0 = 1.0;
  init.boost = 0.15;    // solver.hip "step-length memory"; MELLON_AMD_LS_BOOST=0 keeps every first trial at 1
  if (const char* ev = std::getenv("MELLON_AMD_LS_BOOST")) init.boost = std::atof(ev);
  // capped start (solver.hip): on the 32-bit copy the likelihood's e^t is continued linearly beyond t = 7 while the loss
  // still falls steeply; MELLON_AMD_EXP_CAP=<t> moves the cap, MELLON_AMD_EXP_CAP=off removes it
  init.cap = phase32 ? 7.0 : __builtin_inf();
  if (const char* ev = std::getenv("MELLON_AMD_EXP_CAP"))
    if (phase32) init.cap = (std::strcmp(ev, "off") == 0 || std::atof(ev) <= 0.0) ? __builtin_inf() : std::atof(ev);
  init.cap_fall = 0.15;
  if (const char* ev = std::getenv("MELLON_AMD_EXP_CAP_FALL")) init.cap_fall = std::atof(ev);
  init.boost_fall = 0.15;
  if (const char* ev = std::getenv("MELLON_AMD_LS_BOOST_FALL")) init.boost_fall = std::atof(ev);
  // Subsample start (solver.h): when the preconditioner's Gram came from every s-th cell (s >= 4), the solve starts
  // on the MAP problem of exactly those cells -- the Ridge matrix is ITS Hessian at a = 1 -- at 1/s of the bytes per
  // pass, and moves to all cells once that problem's progress per iteration is below sub_tol.  The walk down from the
  // Ridge start (a dozen passes) then costs about two.  MELLON_AMD_SUBSAMPLE=0 disables, MELLON_AMD_SUB_TOL moves it.
  // Which cells: ~32 m of them (every (3 s / 16)-th cell for a Gram stride s = n / 6 m; nested levels are possible,
  // MELLON_AMD_SUB_LEVELS="16:8", but did not pay).  tools/solver_sweep.py, five data seeds at C3, mean step in ms:
  // no subsample 302 | stride 16: 241 | 12: 204 | 8: 203 | 6: 193 | 4: 203 | 16 then 8: 213 | 16 then 4: 215.
  // The smaller the sample, the cheaper its passes but the more its optimum overfits (at stride 16 the first full
  // evaluation finds the loss 60 % above the optimum's and e^{f+V} of unseen cells up to 1e5).
  std::vector<int64_t> sub_strides;
  if (f->precond_stride >= 11) sub_strides.push_back(std::max<int64_t>(2, 3 * f->precond_stride / 16));
  if (const char* ev = std::getenv("MELLON_AMD_SUB_LEVELS")) {
    if (!sub_strides.empty()) {
      sub_strides.clear();
      for (const char* p = ev; *p;) {
        char* end = nullptr;
        const long long v = std::strtoll(p, &end, 10);
        if (end == p) break;
        if (v >= 2) sub_strides.push_back((int64_t)v);
        p = (*end != 0) ? end + 1 : end;        // any one separator character ("16,4", "16:4")
      }
    }
  }
  if (const char* ev = std::getenv("MELLON_AMD_SUBSAMPLE")) { if (std::atoi(ev) == 0) sub_strides.clear(); }
  const std::vector<int64_t>* subs = sub_strides.empty() ? nullptr : &sub_strides;
  init.gate_full = init.gate;
  init.sub_tol = 1e-3;
  if (const char* ev = std::getenv("MELLON_AMD_SUB_TOL")) init.sub_tol = std::atof(ev);
  init.n_sub_levels = (int)sub_strides.size();
  init.sub_level = 0;
  if (subs) { init.gate = MLN_GATE_SUB; init.cap = __builtin_inf(); }
  // Preconditioner rebuild (solver.h, precond_rebuild.hip): pays when the evaluations it saves (measured: 33-40 full
  // passes without it, 15-26 with it) cost more than the m^3 work of a second factorisation -- decided from rank 0's
  // measurement of the first build, the same on every rank.  An evaluation = one pass of this rank's rows + ~0.14 ms of
  // small launches.  MELLON_AMD_REBUILD=0 / 1 forces the decision.
  const double pass_s = (double)f->n * (double)f->ldl * 8.0 / 6.5e12 + 1.4e-4;
  // (emulated ranks of C3, tools/emulate_rank.py: the rebuild gains 9 ms per step at 4 ranks -- first build = 8.9 evaluations
  //  -- and loses 2.5 ms at 8 -- 12.7 evaluations: the threshold sits between)
  double want_rebuild = (f->build_seconds > 0.0 && 11.0 * pass_s > f->build_seconds) ? 1.0 : 0.0;
  if (const char* ev = std::getenv("MELLON_AMD_REBUILD")) want_rebuild = std::atoi(ev) != 0 ? 1.0 : 0.0;
  if (phase32 && !(f->l32_fixed)) want_rebuild = 0.0;   // (mixed solves pause at their fp64 anchor, which only the corrected fixed-point surrogate has)
  // The rebuild reads the rows' f of the last accepted pass (f_keep), which a rank only has while its shard fits the
  // kernel's f staging: with uneven or very large shards that is a per-rank fact, and the branch at the pause issues
  // collectives (Gram all-reduce, the sample's global sum) -- so the decision is made ONCE, here, for all ranks: rank 0's
  // cost rule AND every rank able to keep f (one all-reduce of two numbers: rank 0's vote, the count of ranks that cannot).
  static const bool no_fkeep_env = std::getenv("MELLON_AMD_NO_FKEEP") != nullptr;
  const bool keeps_f = !no_fkeep_env && f->f_keep[0] && f->f_keep[1] && objective_can_keep_f(f->n, f->n_wg);
  {
    double vote[2] = {ctx->rank == 0 ? want_rebuild : 0.0, keeps_f ? 0.0 : 1.0};
    MLN_HIP(ctx, hipMemcpyAsync(f->d_tmp, vote, sizeof(vote), hipMemcpyHostToDevice, ctx->stream));
    MLN_TRY(dev_allreduce(ctx, f->d_tmp, 2));
    MLN_HIP(ctx, hipMemcpyAsync(vote, f->d_tmp, sizeof(vote), hipMemcpyDeviceToHost, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    want_rebuild = (vote[0] != 0.0 && vote[1] == 0.0) ? 1.0 : 0.0;
  }
  init.rebuild_armed = want_rebuild != 0.0 ? 1 : 0;
  init.rebuild_at_switch = 0;     // (measured: at the switch the unseen cells' weights are still too wild -- 37-96 full passes)
  if (const char* ev = std::getenv("MELLON_AMD_REBUILD_AT_SWITCH")) init.rebuild_at_switch = std::atoi(ev) != 0 ? 1 : 0;
  init.switch_t0 = 0.35;
  if (const char* ev = std::getenv("MELLON_AMD_SWITCH_T0")) init.switch_t0 = std::atof(ev);
  init.gap_tol = 0.2 * o.ftol;     // (tools/solver_sweep.py, seven data seeds at C3: 15.9 -> 14.7 full passes with both rules, log-density
                                   //  within 4e-8 of the old stop -- the spread between two runs of the old rule; 0.5 ftol: 14.3 passes, 1.8e-7)
  if (const char* ev = std::getenv("MELLON_AMD_GAP_TOL")) init.gap_tol = std::atof(ev);
  init.dec_prev = 0.0; init.dec_prev2 = 0.0;
  init.rebuild_tol = 1e-3;       // (tools/solver_sweep.py at C3, two seeds: 1e-2 -> 23-28 full passes, 1e-3 -> 20-22, 2e-4 -> 22-26)
  if (const char* ev = std::getenv("MELLON_AMD_REBUILD_TOL")) init.rebuild_tol = std::atof(ev);
  double rebuild_rows_per_m = 6.0;   // (6 m, 12 m, 24 m importance-sampled rows: the same pass counts; 6 m is the cheapest Gram)
  if (const char* ev = std::getenv("MELLON_AMD_REBUILD_ROWS_PER_M")) rebuild_rows_per_m = std::atof(ev);
  MLN_TRY(launch_solver_init(ctx, f->sv, init, f->d_gu));
  const int* gate = &f->sv.st->gate;
  static const bool timing = !(std::getenv("MELLON_AMD_TIMING") && std::atoi(std::getenv("MELLON_AMD_TIMING")) == 0);
  int n_enq = 0;
  auto events_for = [&](int i) -> hipEvent_t* {
    if (!timing || i >= 512) return nullptr;
    while ((int)f->evs.size() < 3 * (i + 1)) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      f->evs.push_back(e);
    }
    return &f->evs[3 * i];
  };
  MLN_TRY(fit_enqueue_eval(f, f->sv.un, f->sv.gn, false, gate, events_for(n_enq), subs));
  ++n_enq;
  // evaluation t of the solver's trace ran in enqueue slot t + shift: the chains left in a batch after a pause are
  // no-ops that use up slots (their events time nothing)
  std::vector<std::pair<int, int>> slot_shift;      // (first trace index, shift)
  int batch = 8;
  if (const char* ev = std::getenv("MELLON_AMD_SOLVER_BATCH")) batch = std::max(1, std::atoi(ev));
  const int64_t hard_cap = (int64_t)o.maxiter * o.maxls + 16;
  for (;;) {
    for (int b = 0; b < batch; ++b) {
      MLN_TRY(launch_solver_step(ctx, f->sv, (int)m));
      MLN_TRY(fit_enqueue_eval(f, f->sv.un, f->sv.gn, false, gate, events_for(n_enq), subs));
      ++n_enq;
    }
    // rank 0's state decides for everyone (it is the same state on every rank by construction: identical inputs,
    // identical all-reduced sums, deterministic kernels -- this only rules out a hang should that ever fail)
    MLN_TRY(dev_bcast0(ctx, (double*)f->sv.st, (int64_t)(sizeof(SolverState) / sizeof(double))));
    MLN_HIP(ctx, hipMemcpyAsync(f->h_state, f->sv.st, sizeof(SolverState), hipMemcpyDeviceToHost, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (f->h_state->gate == MLN_GATE_DONE) break;
    if (f->h_state->gate == MLN_GATE_PAUSE) {
      // ---- second preconditioner at the accepted point (whose rows' f the last accepted fp64 pass left in f_keep) ----
      const double tr0 = now_s(), ex_r0 = f->emu_excluded;
      const SolverState ps = *f->h_state;
      if (!ps.f_valid) {      // (a function of the solver's state, identical on every rank; keeping f was settled collectively above)
        // no per-row f to weight the cells with: resume with the preconditioner we have
        MLN_TRY(launch_solver_resume(ctx, f->sv, ps.gate_after_pause, 0));
      } else {
        double *zt = nullptr, *gz = nullptr, *cz = nullptr;
        MLN_HIP(ctx, mln_dmalloc((void**)&zt, sizeof(double) * 3 * (size_t)f->ldl));
        gz = zt + f->ldl; cz = gz + f->ldl;
        MLN_HIP(ctx, hipMemsetAsync(zt, 0, sizeof(double) * 3 * (size_t)f->ldl, ctx->stream));
        // old variable -> z-space:  z = C^-T u,  g_z = C g_u  (and the surrogate's correction c, a gradient in u, likewise)
        int rc = fit_small_gemv(f, f->Cinv, 1, f->sv.u, zt);
        if (rc == MLN_OK) rc = fit_small_gemv(f, f->C, 0, f->sv.g, gz);
        if (rc == MLN_OK && ps.corr) rc = fit_small_gemv(f, f->C, 0, f->sv.c, cz);
        // The curvature pairs survive the change of variable u' = T u, T = C'^T C^-T:  s' = T s,  y' = T^-T y  (s'.y' = s.y).
        // First half here (into z-space, in place), second half once the new factor exists.  Measured (five data seeds at C3):
        // carrying them over costs 1-3 full passes MORE than starting the history afresh -- the new factor already holds
        // the curvature the old pairs describe, relative to a metric that is gone -- so they are dropped by default
        // (MELLON_AMD_REBUILD_KEEP_PAIRS=1 keeps them).
        const bool keep_pairs = std::getenv("MELLON_AMD_REBUILD_KEEP_PAIRS") && std::atoi(std::getenv("MELLON_AMD_REBUILD_KEEP_PAIRS")) != 0;
        const int n_pairs = keep_pairs ? ps.k : 0;
        double* ptmp = zt;     // (reuses zt after z has been consumed below: see order)
        std::vector<int> slots;
        for (int j = 0; j < n_pairs; ++j) slots.push_back((ps.head + j) % ps.maxcor);
        double* pbuf = nullptr;
        if (n_pairs > 0 && rc == MLN_OK) {
          if (mln_dmalloc((void**)&pbuf, sizeof(double) * (size_t)f->ldl) != hipSuccess) rc = MLN_ERR_HIP;
          if (rc == MLN_OK && hipMemsetAsync(pbuf, 0, sizeof(double) * (size_t)f->ldl, ctx->stream) != hipSuccess) rc = MLN_ERR_HIP;
          for (int sl : slots) {
            double* S = f->sv.S + (size_t)sl * f->sv.ld;
            double* Y = f->sv.Y + (size_t)sl * f->sv.ld;
            if (rc == MLN_OK) rc = fit_small_gemv(f, f->Cinv, 1, S, pbuf);          // s_z = C^-T s
            if (rc == MLN_OK && hipMemcpyAsync(S, pbuf, sizeof(double) * m, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = MLN_ERR_HIP;
            if (rc == MLN_OK) rc = fit_small_gemv(f, f->C, 0, Y, pbuf);             // y_z = C y
            if (rc == MLN_OK && hipMemcpyAsync(Y, pbuf, sizeof(double) * m, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = MLN_ERR_HIP;
          }
        }
        (void)ptmp;
        if (rc == MLN_OK) rc = fit_rebuild_precond(f, f->f_keep[ps.f_slot], rebuild_rows_per_m);
        // z-space -> new variable:  u = C^T z,  g_u = C^-1 g_z
        if (rc == MLN_OK) rc = fit_small_gemv(f, f->C, 1, zt, f->sv.u);
        if (rc == MLN_OK) rc = fit_small_gemv(f, f->Cinv, 0, gz, f->sv.g);
        if (rc == MLN_OK && ps.corr) rc = fit_small_gemv(f, f->Cinv, 0, cz, f->sv.c);
        for (int sl : slots) {
          double* S = f->sv.S + (size_t)sl * f->sv.ld;
          double* Y = f->sv.Y + (size_t)sl * f->sv.ld;
          if (rc == MLN_OK) rc = fit_small_gemv(f, f->C, 1, S, pbuf);               // s' = C'^T s_z
          if (rc == MLN_OK && hipMemcpyAsync(S, pbuf, sizeof(double) * m, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = MLN_ERR_HIP;
          if (rc == MLN_OK) rc = fit_small_gemv(f, f->Cinv, 0, Y, pbuf);            // y' = C'^-1 y_z
          if (rc == MLN_OK && hipMemcpyAsync(Y, pbuf, sizeof(double) * m, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = MLN_ERR_HIP;
        }
        if (rc == MLN_OK && n_pairs > 0) rc = launch_solver_refresh_pairs(ctx, f->sv, ps.maxcor);
        (void)hipStreamSynchronize(ctx->stream);
        (void)mln_dfree(zt);
        if (pbuf) (void)mln_dfree(pbuf);
        MLN_TRY(rc);
        MLN_TRY(launch_solver_resume(ctx, f->sv, ps.gate_after_pause, n_pairs > 0 ? 0 : 1));
        if (const char* ev = std::getenv("MELLON_AMD_RESUME_T0")) {     // experiment: first trial step under the new preconditioner
          const double t0v = std::atof(ev);
          MLN_HIP(ctx, hipMemcpyAsync(&f->sv.st->t0, &t0v, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        }
        f->n_rebuild += 1;
      }
      MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
      f->times_rebuild += now_s() - tr0 - (f->emu_excluded - ex_r0);
      slot_shift.push_back({ps.n_eval, n_enq - ps.n_eval});
      continue;
    }
    if (n_enq > hard_cap) { mln_set_error(ctx, "map_solve: the device solver did not terminate"); return MLN_ERR_NOCONV; }
    if (batch < 16 && f->h_state->gate == MLN_GATE_F64) batch = std::min(batch, 6);
  }
  const SolverState st = *f->h_state;
  // kernel-time accounting from the per-evaluation events (the pass the solver did not select is a ~2 us no-op)
  std::vector<double> tr;
  const int n_done = st.n_eval < 512 ? st.n_eval : 512;
  if (n_done > 0) {
    tr.resize((size_t)4 * n_done);
    MLN_HIP(ctx, hipMemcpy(tr.data(), f->sv.trace, sizeof(double) * 4 * n_done, hipMemcpyDeviceToHost));
    for (int i = 0; i < n_done && timing; ++i) {
      const int gcode = (int)tr[4 * i + 3] & 15, lvl = (int)tr[4 * i + 3] >> 4;
      const bool was32 = (gcode & 3) == MLN_GATE_F32, was_sub = gcode == MLN_GATE_SUB;
      int slot = i;
      for (const auto& sh : slot_shift) if (i >= sh.first) slot = i + sh.second;
      if (3 * (slot + 1) > (int)f->evs.size()) continue;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, f->evs[3 * slot + (was32 ? 0 : 1)], f->evs[3 * slot + (was32 ? 1 : 2)]) != hipSuccess) continue;
      if (was_sub) {
        f->times_sub += 1e-3 * ms; f->evals_sub += 1;
        f->sub_pass_equiv += 1.0 / (double)((lvl >= 0 && lvl < (int)sub_strides.size()) ? sub_strides[lvl] : 1);
      }
      else if (was32) { f->times32 += 1e-3 * ms; f->evals32 += 1; }
      else { f->times[5] += 1e-3 * ms; f->times[6] += 1.0; f->times[7] = (double)f->n * (double)f->ldl * 8.0; }
    }
    if (trace_lvl >= 2)
      for (int i = 0; i < n_done; ++i)
        fprintf(stderr, "[eval %d] %s mode=%d t=%.3g f=%.15g\n", i, ((int)tr[4 * i + 3] & 15) == MLN_GATE_F32 ? "f32" : (((int)tr[4 * i + 3] & 15) == MLN_GATE_F32C ? "f32c" : (((int)tr[4 * i + 3] & 15) == MLN_GATE_SUB ? (((int)tr[4 * i + 3] >> 4) ? "sub1" : "sub0") : "f64")),
                (int)tr[4 * i + 2], tr[4 * i + 1], tr[4 * i]);
  }
  // z = C^-T u and w = P u at the accepted point (one stacked product), remembered for transform / predictor weights
  {
    GemvTri g1{f->Q1, f->ldl, f->kspace ? 2 * m : m, f->sv.u, f->d_z, f->kspace ? f->d_w_cached : nullptr, 1, m, m, 0, 0, nullptr};
    MLN_TRY(launch_gemv_tri(ctx, g1));
    f->z_cached.assign((size_t)m, 0.0);
    MLN_HIP(ctx, hipMemcpyAsync(f->z_cached.data(), f->d_z, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    MLN_HIP(ctx, hipMemcpyAsync(z_out, f->d_z, sizeof(double) * m, hipMemcpyDefault, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (st.f_valid && objective_can_keep_f(f->n, f->n_wg) && !std::getenv("MELLON_AMD_NO_FKEEP")) f->f_final = st.f_slot;   // f = L z + mu at this z is already there (mln_transform)
  if (trace_lvl)
    fprintf(stderr, "[trace] map_solve: %d evaluations (%d on the 32-bit copy, %d on the row subsample of stride %lld), %d iterations, "
            "%d rebuild(s), %d enqueued, status %d\n", st.n_eval, st.n_eval32, st.n_eval_sub, (long long)(subs ? sub_strides[0] : 0), st.it,
            f->n_rebuild, n_enq, st.status);
  if (loss_out) *loss_out = st.fx;
  if (n_eval_out) *n_eval_out = st.n_eval;
  if (n_iter_out) *n_iter_out = st.it;
  if (status_out) *status_out = st.status;
  return MLN_OK;
}

extern "C" int mln_weights_cholesky(mln_fit* f, const double* z, double* w) {
  if (!f || !z || !w) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->tri.W2) { mln_set_error(ctx, "this fit handle holds no Lp"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevOut o;
  MLN_TRY(o.init(ctx, w, (size_t)f->m));
  if (f->kspace && !is_device_ptr(z) && f->z_cached.size() == (size_t)f->m &&
      std::memcmp(z, f->z_cached.data(), sizeof(double) * f->m) == 0) {
    MLN_HIP(ctx, hipMemcpyAsync(o.dev, f->d_w_cached, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
    return o.commit();
  }
  MLN_HIP(ctx, hipMemcpyAsync(o.dev, z, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  MLN_TRY(triinv_solve_left_T(ctx, f->tri, o.dev, 1, 1));  // conditional.py:818
  return o.commit();
}

extern "C" int mln_weights_full(mln_fit* f, const double* y, int64_t p, double mu, double* w) {
  if (!f || !y || !w || p < 1) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->tri.W2) { mln_set_error(ctx, "this fit handle holds no Lp"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t cnt = f->m * p;
  DevOut o;
  MLN_TRY(o.init(ctx, w, (size_t)cnt));
  MLN_HIP(ctx, hipMemcpyAsync(o.dev, y, sizeof(double) * cnt, hipMemcpyDefault, ctx->stream));
  // r = y - mu ; w = Lp^-T Lp^-1 r                                conditional.py:263-264
  if (mu != 0.0) {
    double* ones = nullptr;
    MLN_HIP(ctx, mln_dmalloc((void**)&ones, sizeof(double) * cnt));
    std::vector<double> h((size_t)cnt, 1.0);
    MLN_HIP(ctx, hipMemcpyAsync(ones, h.data(), sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_axpby(ctx, cnt, -mu, ones, 1.0, o.dev);
    (void)hipStreamSynchronize(ctx->stream);
    (void)mln_dfree(ones);
    if (rc != MLN_OK) return rc;
  }
  MLN_TRY(triinv_solve_left(ctx, f->tri, o.dev, p, p));
  MLN_TRY(triinv_solve_left_T(ctx, f->tri, o.dev, p, p));
  return o.commit();
}

extern "C" int mln_stage_times(mln_fit* f, double* out) {
  if (!f || !out) return MLN_ERR_ARG;
  for (int i = 0; i < 8; ++i) out[i] = f->times[i];
  out[8] = f->times32;                                  // 32-bit warm-up passes: kernel seconds (HIP events)
  out[9] = (double)f->evals32;                          //                        launches
  out[10] = f->L32 ? (f->l32_fixed ? 2.0 : 1.0) : 0.0;  //                        format of the copy
  out[11] = f->emu_excluded;                            // MELLON_AMD_EMULATE_RANKS: seconds spent on other ranks' blocks
  out[12] = f->times_sub;                               // subsample passes of the solver's first phase: kernel seconds
  out[13] = (double)f->evals_sub;                       //                                                launches
  out[14] = (double)(f->precond_stride > 0 ? f->precond_stride : 1);   // their row stride (= the Gram sample's)
  out[15] = f->times_rebuild;                           // second preconditioner: wall seconds (selection, Gram, factorisation)
  out[16] = (double)f->n_rebuild;
  // passes over the n x m buffer in full-fp64-pass equivalents (bytes streamed / bytes of one fp64 pass)
  out[17] = f->times[6] + 0.5 * (double)f->evals32 + f->sub_pass_equiv;
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

// ---- FunctionEstimator sparse solve ----------------------------------------------------------------
// C[i][j] *= (row ? row[i] : 1) * (col ? col[j] : 1)
__global__ void k_scale_rows_cols(double* __restrict__ A, int64_t ld, int64_t rows, int64_t cols,
                                  const double* __restrict__ row, const double* __restrict__ col) {
  const int64_t i = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (i >= rows) return;
  const double ri = row ? row[i] : 1.0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cols; j += (int64_t)gridDim.x * blockDim.x)
    A[i * ld + j] *= ri * (col ? col[j] : 1.0);
}

// T[k][j] /= lam[k] * inv_s2[j] + 1  -- the resolvent (G / s_j^2 + I)^-1 in the eigenbasis of G
__global__ void k_resolvent_scale(double* __restrict__ T, int64_t ld, int64_t rows, int64_t cols,
                                  const double* __restrict__ lam, const double* __restrict__ inv_s2) {
  const int64_t k = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (k >= rows) return;
  const double l = lam[k] > 0.0 ? lam[k] : 0.0;       // A A^T is positive semi-definite; rounding may say -1e-13
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cols; j += (int64_t)gridDim.x * blockDim.x)
    T[k * ld + j] /= l * inv_s2[j] + 1.0;
}

static int launch_scale_rows_cols(mln_ctx* ctx, double* A, int64_t ld, int64_t rows, int64_t cols, const double* row,
                                  const double* col) {
  if (rows <= 0 || cols <= 0) return MLN_OK;
  int64_t bx = (cols + 255) / 256;
  if (bx > 64) bx = 64;
  const int64_t by = rows < 65535 ? rows : 65535, bz = (rows + 65534) / 65535;
  hipLaunchKernelGGL(k_scale_rows_cols, dim3((unsigned)bx, (unsigned)by, (unsigned)bz), dim3(256), 0, ctx->stream, A, ld,
                     rows, cols, row, col);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

// Noise models of the landmark conditional (conditional.py:140-159, 529-545):
//   MLN_SIGMA_SCALAR      sigma[1]        r / s^2, A / s^2
//   MLN_SIGMA_PER_OUTPUT  sigma[p]        one scalar solve per output column ("per-gene", the vmap of :529-545);
//                                         A A^T and A r are formed once, columns with equal sigma share L_B
//   MLN_SIGMA_PER_CELL    sigma[n_local]  element-wise std of the cells: A diag(1/s^2) A^T and A (r / s^2)
static constexpr int SPECTRAL_MIN_LEVELS = 32;   // runs of equal sigma above which the per-output solve goes spectral

static int sparse_solve_impl(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                             int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                             const double* sigmas, int32_t kind, double jitter, double* W, double* Lp_out,
                             double* Cs_out) {
  if (!ctx || !xu || !W || !sigmas || (n_local > 0 && (!x || !y))) return MLN_ERR_ARG;
  if (p < 1 || m < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (kind < MLN_SIGMA_SCALAR || kind > MLN_SIGMA_PER_CELL) { mln_set_error(ctx, "unknown sigma kind"); return MLN_ERR_ARG; }
  if (kind != MLN_SIGMA_SCALAR && (Lp_out || Cs_out)) {
    mln_set_error(ctx, "the L_B factor is only defined for one scalar sigma (conditional.py:574-577)");
    return MLN_ERR_ARG;
  }
  const int64_t n_sig = (kind == MLN_SIGMA_SCALAR) ? 1 : (kind == MLN_SIGMA_PER_OUTPUT ? p : n_local);
  for (int64_t i = 0; i < n_sig; ++i)
    if (!(sigmas[i] > 0.0)) {
      mln_set_error(ctx, "sigma must be > 0 for the sparse solve (conditional.py:157-159 divides by sigma^2)");
      return MLN_ERR_ARG;
    }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  // A^T = cov(x, xu) Lp^-T is exactly the factor L of the density path   conditional.py:516-522
  mln_fit* f = nullptr;
  MLN_TRY(mln_fit_prepare(ctx, cov, x, n_local, d, xu, m, jitter, nullptr, 0, &f));
  const int64_t ldg = pad16(m), n = n_local;
  // groups of adjacent output columns with one noise level
  std::vector<int64_t> g_begin;
  std::vector<double> g_s2;
  if (kind == MLN_SIGMA_PER_OUTPUT) {
    for (int64_t j = 0; j < p; ++j)
      if (j == 0 || sigmas[j] != sigmas[j - 1]) { g_begin.push_back(j); g_s2.push_back(sigmas[j] * sigmas[j]); }
  } else {
    g_begin.push_back(0);
    g_s2.push_back(kind == MLN_SIGMA_SCALAR ? sigmas[0] * sigmas[0] : 1.0);
  }
  g_begin.push_back(p);
  const size_t n_groups = g_s2.size();
  double *G = nullptr, *G0 = nullptr, *R = nullptr, *C = nullptr, *parts = nullptr, *d_scale = nullptr, *d_col = nullptr;
  TriInv tb;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "sparse_solve", __FILE__, __LINE__); };
  chk(mln_dmalloc((void**)&G, sizeof(double) * (size_t)m * ldg));
  if (n_groups > 1) chk(mln_dmalloc((void**)&G0, sizeof(double) * (size_t)m * ldg));
  chk(mln_dmalloc((void**)&C, sizeof(double) * (size_t)m * p));
  DevIn dy;
  if (rc == MLN_OK) rc = dy.init(ctx, y, (size_t)n * p);
  // r = y - mu
  if (rc == MLN_OK && n > 0) {
    chk(mln_dmalloc((void**)&R, sizeof(double) * (size_t)n * p));
    chk(hipMemcpyAsync(R, dy.dev, sizeof(double) * (size_t)n * p, hipMemcpyDeviceToDevice, ctx->stream));
    if (rc == MLN_OK && mu != 0.0) {
      std::vector<double> ones((size_t)n * p, 1.0);
      double* d1 = nullptr;
      chk(mln_dmalloc((void**)&d1, sizeof(double) * ones.size()));
      chk(hipMemcpyAsync(d1, ones.data(), sizeof(double) * ones.size(), hipMemcpyHostToDevice, ctx->stream));
      if (rc == MLN_OK) rc = launch_axpby(ctx, (int64_t)ones.size(), -mu, d1, 1.0, R);
      (void)hipStreamSynchronize(ctx->stream);
      if (d1) (void)mln_dfree(d1);
    }
  }
  // per-cell noise: rows of A^T and of r divided by sigma_i, after which the solve is the sigma = 1 one
  if (rc == MLN_OK && kind == MLN_SIGMA_PER_CELL && n > 0) {
    std::vector<double> inv((size_t)n);
    for (int64_t i = 0; i < n; ++i) inv[(size_t)i] = 1.0 / sigmas[i];
    chk(mln_dmalloc((void**)&d_scale, sizeof(double) * (size_t)n));
    chk(hipMemcpyAsync(d_scale, inv.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (rc == MLN_OK) rc = launch_scale_rows_cols(ctx, f->L, f->ldl, n, m, d_scale, nullptr);
    if (rc == MLN_OK) rc = launch_scale_rows_cols(ctx, R, p, n, p, d_scale, nullptr);
    (void)hipStreamSynchronize(ctx->stream);
  }
  // A A^T (all-reduced), kept when several noise levels need it
  if (rc == MLN_OK) rc = fit_gram(f, n_groups > 1 ? G0 : G, ldg, 1);
  // C = A r = L^T r   (m x p), split over cells; the 1 / sigma^2 is applied per group below
  if (rc == MLN_OK) {
    int split = (int)(n / 8192);
    if (split < 1) split = 1;
    if (split > 16) split = 16;
    const size_t stride = (size_t)m * p;
    if (split > 1) chk(mln_dmalloc((void**)&parts, sizeof(double) * stride * split));
    if (rc == MLN_OK) chk(hipMemsetAsync(split > 1 ? parts : C, 0, sizeof(double) * stride * (split > 1 ? split : 1), ctx->stream));
    GemmArgs g{};
    g.A = f->L; g.lda = f->ldl; g.B = R; g.ldb = p; g.C = (split > 1) ? parts : C; g.ldc = p;
    g.M = m; g.N = p; g.K = n; g.alpha = (n_groups == 1) ? 1.0 / g_s2[0] : 1.0; g.beta = 0.0; g.ta = 1; g.tb = 0;
    g.split_k = split; g.c_split_stride = (int64_t)stride;
    if (rc == MLN_OK && n > 0) rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK && split > 1) rc = launch_sum_partials(ctx, parts, split, (int64_t)stride, C, (int64_t)stride, 0.0);
    if (rc == MLN_OK) rc = dev_allreduce(ctx, C, (int64_t)stride);
    if (rc == MLN_OK && n_groups > 1) {
      std::vector<double> inv((size_t)p);
      for (int64_t j = 0; j < p; ++j) inv[(size_t)j] = 1.0 / (sigmas[j] * sigmas[j]);
      chk(mln_dmalloc((void**)&d_col, sizeof(double) * (size_t)p));
      chk(hipMemcpyAsync(d_col, inv.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
      if (rc == MLN_OK) rc = launch_scale_rows_cols(ctx, C, p, m, p, nullptr, d_col);
      (void)hipStreamSynchronize(ctx->stream);
    }
  }
  if (rc == MLN_OK && n_groups > (size_t)SPECTRAL_MIN_LEVELS) {
    // Many noise levels: one eigendecomposition A A^T = U diag(lam) U^T serves them all,
    //   (A A^T / s^2 + I)^-1 c = U diag(1 / (lam / s^2 + 1)) U^T c,
    // O(m^3 + m^2 p) instead of one m^3/3 Cholesky per level.  The matrix inverted has eigenvalues >= 1, so the
    // spectral form is as well conditioned as the factorisation it replaces.
    double *V = nullptr, *T = nullptr, *d_lam = nullptr;
    std::vector<double> lam((size_t)m);
    int sweeps = 0;
    chk(mln_dmalloc((void**)&V, sizeof(double) * (size_t)m * ldg));
    chk(mln_dmalloc((void**)&T, sizeof(double) * (size_t)m * p));
    chk(mln_dmalloc((void**)&d_lam, sizeof(double) * (size_t)m));
    if (rc == MLN_OK) rc = dev_eigh(ctx, G0, m, ldg, lam.data(), V, ldg, &sweeps);     // row k of V = eigenvector k
    if (rc == MLN_OK) chk(hipMemcpyAsync(d_lam, lam.data(), sizeof(double) * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    GemmArgs g{};
    g.A = V; g.lda = ldg; g.B = C; g.ldb = p; g.C = T; g.ldc = p;
    g.M = m; g.N = p; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);                                      // U^T c
    if (rc == MLN_OK) {
      int64_t bx = (p + 255) / 256;
      if (bx > 64) bx = 64;
      hipLaunchKernelGGL(k_resolvent_scale, dim3((unsigned)bx, (unsigned)(m < 65535 ? m : 65535), (unsigned)((m + 65534) / 65535)),
                         dim3(256), 0, ctx->stream, T, p, m, p, d_lam, d_col);
      chk(hipGetLastError());
    }
    g.A = V; g.B = T; g.C = C; g.ta = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);                                      // U (.)
    (void)hipStreamSynchronize(ctx->stream);
    void* tmp[] = {V, T, d_lam};
    for (void* q : tmp) if (q) (void)mln_dfree(q);
  } else {
    for (size_t gi = 0; gi < n_groups && rc == MLN_OK; ++gi) {
      const int64_t c0 = g_begin[gi], nc = g_begin[gi + 1] - c0;
      // LBB = A A^T / sigma^2 + I                                      conditional.py:62 (stabilize(.., 1))
      rc = launch_axpby(ctx, m * ldg, 1.0 / g_s2[gi], n_groups > 1 ? G0 : G, 0.0, G);
      if (rc == MLN_OK) rc = launch_add_diag(ctx, G, m, ldg, 1.0);
      if (rc == MLN_OK) rc = dev_cholesky_lower(ctx, G, m, ldg);
      // L_B^-T L_B^-1 C                                                conditional.py:64-65
      if (rc == MLN_OK) rc = triinv_build(ctx, G, m, ldg, true, true, &tb);
      if (rc == MLN_OK) rc = triinv_solve_left(ctx, tb, C + c0, nc, p);
      if (rc == MLN_OK) rc = triinv_solve_left_T(ctx, tb, C + c0, nc, p);
      if (gi + 1 < n_groups) { (void)hipStreamSynchronize(ctx->stream); triinv_free(&tb); }
    }
  }
  // weights = Lp^-T (.)
  if (rc == MLN_OK) rc = triinv_solve_left_T(ctx, f->tri, C, p, p);
  if (rc == MLN_OK) chk(hipMemcpyAsync(W, C, sizeof(double) * (size_t)m * p, hipMemcpyDefault, ctx->stream));
  // with_uncertainty state of the noisy landmark conditional: L = Lp and Cs = Lp L_B   conditional.py:571-577
  if (rc == MLN_OK && Lp_out) {
    DevOut o;
    rc = o.init(ctx, Lp_out, (size_t)m * m);
    if (rc == MLN_OK) rc = launch_copy_block(ctx, f->Lp, f->ldp, o.dev, m, m, m);
    if (rc == MLN_OK) rc = o.commit();
  }
  if (rc == MLN_OK && Cs_out) {
    double* cs = nullptr;
    chk(mln_dmalloc((void**)&cs, sizeof(double) * (size_t)m * ldg));
    if (rc == MLN_OK) chk(hipMemsetAsync(cs, 0, sizeof(double) * (size_t)m * ldg, ctx->stream));
    GemmArgs g{};
    g.A = f->Lp; g.lda = f->ldp; g.B = G; g.ldb = ldg; g.C = cs; g.ldc = ldg;
    g.M = m; g.N = m; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
    DevOut o;
    if (rc == MLN_OK) rc = o.init(ctx, Cs_out, (size_t)m * m);
    if (rc == MLN_OK) rc = launch_copy_block(ctx, cs, ldg, o.dev, m, m, m);
    if (rc == MLN_OK) rc = o.commit();
    (void)hipStreamSynchronize(ctx->stream);
    if (cs) (void)mln_dfree(cs);
  }
  (void)hipStreamSynchronize(ctx->stream);
  triinv_free(&tb);
  void* ptrs[] = {G, G0, R, C, parts, d_scale, d_col};
  for (void* q : ptrs) if (q) (void)mln_dfree(q);
  fit_free(f);
  return rc;
}

extern "C" int mln_sparse_solve(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                                int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                                double sigma, double jitter, double* W) {
  return sparse_solve_impl(ctx, cov, x, n_local, d, xu, m, y, p, mu, &sigma, MLN_SIGMA_SCALAR, jitter, W, nullptr, nullptr);
}

extern "C" int mln_sparse_solve_factors(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                                        int32_t d, const double* xu, int64_t m, const double* y, int64_t p,
                                        double mu, double sigma, double jitter, double* W, double* Lp_out,
                                        double* Cs_out) {
  return sparse_solve_impl(ctx, cov, x, n_local, d, xu, m, y, p, mu, &sigma, MLN_SIGMA_SCALAR, jitter, W, Lp_out, Cs_out);
}

// Z <- Z o Z
__global__ void k_square_inplace(double* __restrict__ Z, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    Z[i] *= Z[i];
}

// D[k][j] = 1 / (s2[j] + theta[k])
__global__ void k_resolvent_table(double* __restrict__ D, int64_t ld, int64_t rows, int64_t cols,
                                  const double* __restrict__ theta, const double* __restrict__ s2) {
  const int64_t k = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (k >= rows) return;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cols; j += (int64_t)gridDim.x * blockDim.x)
    D[k * ld + j] = 1.0 / (s2[j] + theta[k]);
}

// Leverage of the landmark conditional for p noise levels from one eigendecomposition.
//   h_ij = b_i^T M_j^-1 b_i,  M_j = s_j^2 K_uu + B^T B + jitter I,  B = cov(x, xu)      conditional.py:660-685
// With K_uu = Lk Lk^T and l_i = Lk^-1 b_i (the rows of the low-rank factor L = B Lk^-T):
//   M_j = Lk (s_j^2 I + N) Lk^T,  N = L^T L + jitter Lk^-1 Lk^-T = V diag(theta) V^T
//   h_ij = sum_k (V^T l_i)_k^2 / (s_j^2 + theta_k)
// i.e. one TRSM, one Gram, one m x m eigensolve and two GEMMs for all levels, instead of one m x m Cholesky and one
// n m^2 solve per level.  L is formed by the triangular solve (not from B^T B), so N is accurate to eps |N|.
extern "C" int mln_landmark_leverage(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                                     int32_t d, const double* xu, int64_t m, const double* Lk, const double* sigma,
                                     int64_t p, double jitter, double* out) {
  if (!ctx || !xu || !Lk || !sigma || (n_local > 0 && (!x || !out))) return MLN_ERR_ARG;
  if (p < 1 || m < 1 || n_local < 0) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  mln_fit* f = nullptr;
  MLN_TRY(mln_fit_prepare(ctx, cov, x, n_local, d, xu, m, jitter, Lk, 0, &f));
  const int64_t ldg = pad16(m), n = n_local;
  double *N = nullptr, *Li = nullptr, *J = nullptr, *V = nullptr, *D = nullptr, *Z = nullptr, *d_theta = nullptr, *d_s2 = nullptr;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "landmark_leverage", __FILE__, __LINE__); };
  const size_t mm = sizeof(double) * (size_t)m * ldg;
  chk(mln_dmalloc((void**)&N, mm));
  chk(mln_dmalloc((void**)&Li, mm));
  chk(mln_dmalloc((void**)&J, mm));
  chk(mln_dmalloc((void**)&V, mm));
  chk(mln_dmalloc((void**)&D, sizeof(double) * (size_t)m * p));
  chk(mln_dmalloc((void**)&d_theta, sizeof(double) * (size_t)m));
  chk(mln_dmalloc((void**)&d_s2, sizeof(double) * (size_t)p));
  DevOut o;
  if (rc == MLN_OK && n > 0) rc = o.init(ctx, out, (size_t)n * p);
  // N = L^T L (all cells, all ranks) + jitter Lk^-1 Lk^-T
  if (rc == MLN_OK) rc = fit_gram(f, N, ldg, 1);
  if (rc == MLN_OK) chk(hipMemsetAsync(Li, 0, mm, ctx->stream));
  if (rc == MLN_OK) rc = launch_add_diag(ctx, Li, m, ldg, 1.0);
  if (rc == MLN_OK) rc = triinv_solve_left(ctx, f->tri, Li, m, ldg);                  // Lk^-1
  if (rc == MLN_OK) chk(hipMemsetAsync(J, 0, mm, ctx->stream));
  {
    GemmArgs g{};
    g.A = Li; g.lda = ldg; g.B = Li; g.ldb = ldg; g.C = J; g.ldc = ldg;
    g.M = m; g.N = m; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1; g.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
  }
  if (rc == MLN_OK) rc = launch_axpby(ctx, m * ldg, jitter, J, 1.0, N);
  std::vector<double> theta((size_t)m), s2((size_t)p);
  int sweeps = 0;
  if (rc == MLN_OK) rc = dev_eigh(ctx, N, m, ldg, theta.data(), V, ldg, &sweeps);     // row k of V = eigenvector k
  if (rc == MLN_OK) {
    double smin = sigma[0] * sigma[0];
    for (int64_t j = 0; j < p; ++j) { s2[(size_t)j] = sigma[j] * sigma[j]; if (s2[(size_t)j] < smin) smin = s2[(size_t)j]; }
    if (!(smin + theta[0] > 0.0)) {
      mln_set_error(ctx, "sigma^2 K_uu + B^T B + jitter I is not positive definite");
      rc = MLN_ERR_NOT_PD;
    }
  }
  if (rc == MLN_OK) {
    chk(hipMemcpyAsync(d_theta, theta.data(), sizeof(double) * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    chk(hipMemcpyAsync(d_s2, s2.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
    int64_t bx = (p + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_resolvent_table, dim3((unsigned)bx, (unsigned)(m < 65535 ? m : 65535), (unsigned)((m + 65534) / 65535)),
                       dim3(256), 0, ctx->stream, D, p, m, p, d_theta, d_s2);
    chk(hipGetLastError());
  }
  // per chunk of cells: Z = L V^T (coordinates of l_i in the eigenbasis), squared, times the resolvent table
  const int64_t chunk = (n < 32768) ? (n > 0 ? n : 1) : 32768;
  if (rc == MLN_OK) chk(mln_dmalloc((void**)&Z, sizeof(double) * (size_t)chunk * ldg));
  for (int64_t r0 = 0; r0 < n && rc == MLN_OK; r0 += chunk) {
    const int64_t rows = (n - r0 < chunk) ? n - r0 : chunk;
    GemmArgs g{};
    g.A = f->L + r0 * f->ldl; g.lda = f->ldl; g.B = V; g.ldb = ldg; g.C = Z; g.ldc = ldg;
    g.M = rows; g.N = m; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1; g.split_k = 1;
    rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK) {
      const int64_t count = rows * ldg;
      int64_t nb = (count + 255) / 256;
      if (nb > 16384) nb = 16384;
      hipLaunchKernelGGL(k_square_inplace, dim3((unsigned)nb), dim3(256), 0, ctx->stream, Z, count);
      chk(hipGetLastError());
    }
    GemmArgs h{};
    h.A = Z; h.lda = ldg; h.B = D; h.ldb = p; h.C = o.dev + r0 * p; h.ldc = p;
    h.M = rows; h.N = p; h.K = m; h.alpha = 1.0; h.beta = 0.0; h.ta = 0; h.tb = 0; h.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, h);
  }
  if (rc == MLN_OK && n > 0) rc = o.commit();
  (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {N, Li, J, V, D, Z, d_theta, d_s2};
  for (void* q : ptrs) if (q) (void)mln_dfree(q);
  fit_free(f);
  return rc;
}

__global__ void k_mul_inplace(double* __restrict__ T, const double* __restrict__ D, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    T[i] *= D[i];
}

__global__ void k_shift(double* __restrict__ R, const double* __restrict__ y, double mu, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    R[i] = y[i] - mu;
}

// H[i][j] <- 1 - sig2[j] * H[i][j]
__global__ void k_leverage_finish(double* __restrict__ H, int64_t count, int64_t p, const double* __restrict__ sig2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    H[i] = 1.0 - sig2[i % p] * H[i];
}

// out = (y - mu - KW)^2 / (1 - h)^2                                 conditional.py:330-333
__global__ void k_hc3(double* __restrict__ out, const double* __restrict__ y, const double* __restrict__ kw, double mu,
                      const double* __restrict__ h, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const double r = y[i] - (mu + kw[i]), q = 1.0 - h[i];
    out[i] = (r * r) / (q * q);
  }
}

static unsigned grid_1d(int64_t count) {
  int64_t nb = (count + 255) / 256;
  return (unsigned)(nb > 16384 ? 16384 : (nb < 1 ? 1 : nb));
}

// Full GP conditioned on p outputs, each with its own noise level (conditional.py:239-251), from one
// eigendecomposition K = U diag(lam) U^T:  (K + s_j I)^-1 = U diag(1 / (lam + s_j)) U^T,  s_j = sigma_j^2 + jitter.
//   W[:, j]   = (K + s_j I)^-1 (y_j - mu)
//   lev[:, j] = 1 - sigma_j^2 diag((K + s_j I)^-1) = 1 - sigma_j^2 sum_k U_ik^2 / (lam_k + s_j)    :313-323,385-403
//   cr2       = (y - mu - K W)^2 / (1 - lev)^2                                                     :330-333
//   VW[:, j]  = (K + s_j I)^-1 cr2[:, j]                                                           :338-350
extern "C" int mln_full_conditional_noise(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n,
                                          int32_t d, const double* y, int64_t p, double mu, const double* sigma,
                                          double jitter, double* W, double* leverage, double* corrected_r2,
                                          double* variance_W) {
  if (!ctx || !x || !y || !sigma || !W) return MLN_ERR_ARG;
  if (n < 1 || n > 32768 || p < 1 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if ((corrected_r2 || variance_W) && !leverage) { mln_set_error(ctx, "corrected_r2 / variance_W need the leverage output"); return MLN_ERR_ARG; }
  if (variance_W && !corrected_r2) { mln_set_error(ctx, "variance_W needs the corrected_r2 output"); return MLN_ERR_ARG; }
  if (ctx->n_ranks > 1) { mln_set_error(ctx, "the full (non-sparse) GP cannot be cell-sharded"); return MLN_ERR_UNSUPPORTED; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  DevIn dx, dy;
  MLN_TRY(dx.init(ctx, x, (size_t)n * d));
  MLN_TRY(dy.init(ctx, y, (size_t)n * p));
  DevOut oW, oH, oC, oV;
  MLN_TRY(oW.init(ctx, W, (size_t)n * p));
  if (leverage) MLN_TRY(oH.init(ctx, leverage, (size_t)n * p));
  if (corrected_r2) MLN_TRY(oC.init(ctx, corrected_r2, (size_t)n * p));
  if (variance_W) MLN_TRY(oV.init(ctx, variance_W, (size_t)n * p));
  const int64_t ld = pad16(n), np_ = n * p;
  double *K = nullptr, *V = nullptr, *D = nullptr, *T = nullptr, *d_lam = nullptr, *d_s = nullptr, *d_sig2 = nullptr;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "full_conditional_noise", __FILE__, __LINE__); };
  const size_t nn = sizeof(double) * (size_t)n * ld;
  chk(mln_dmalloc((void**)&K, nn));
  chk(mln_dmalloc((void**)&V, nn));
  chk(mln_dmalloc((void**)&D, sizeof(double) * (size_t)np_));
  chk(mln_dmalloc((void**)&T, sizeof(double) * (size_t)np_));
  chk(mln_dmalloc((void**)&d_lam, sizeof(double) * (size_t)n));
  chk(mln_dmalloc((void**)&d_s, sizeof(double) * (size_t)p));
  chk(mln_dmalloc((void**)&d_sig2, sizeof(double) * (size_t)p));
  if (rc == MLN_OK) chk(hipMemsetAsync(K, 0, nn, ctx->stream));
  if (rc == MLN_OK) rc = launch_kernel_matrix(ctx, dc, dx.dev, n, dx.dev, n, d, K, ld, 0.0);
  std::vector<double> lam((size_t)n), s((size_t)p), sig2((size_t)p);
  int sweeps = 0;
  if (rc == MLN_OK) rc = dev_eigh(ctx, K, n, ld, lam.data(), V, ld, &sweeps);        // row k of V = eigenvector k
  if (rc == MLN_OK) {
    double smin = 0.0;
    for (int64_t j = 0; j < p; ++j) {
      sig2[(size_t)j] = sigma[j] * sigma[j];
      s[(size_t)j] = sig2[(size_t)j] + jitter;
      if (j == 0 || s[(size_t)j] < smin) smin = s[(size_t)j];
    }
    if (!(lam[0] + smin > 0.0)) {
      mln_set_error(ctx, "Covariance not positively definite with the given sigma and jitter");
      rc = MLN_ERR_NOT_PD;
    }
  }
  auto resolvent = [&](const double* rhs, double* out) -> int {      // out = U diag(1/(lam + s_j)) U^T rhs, per column j
    GemmArgs g{};
    g.A = V; g.lda = ld; g.B = rhs; g.ldb = p; g.C = T; g.ldc = p;
    g.M = n; g.N = p; g.K = n; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.split_k = 1;
    int r = launch_dgemm(ctx, g);
    if (r != MLN_OK) return r;
    hipLaunchKernelGGL(k_mul_inplace, dim3(grid_1d(np_)), dim3(256), 0, ctx->stream, T, D, np_);
    g.B = T; g.C = out; g.ta = 1;
    return launch_dgemm(ctx, g);
  };
  if (rc == MLN_OK) {
    chk(hipMemcpyAsync(d_lam, lam.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    chk(hipMemcpyAsync(d_s, s.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
    chk(hipMemcpyAsync(d_sig2, sig2.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
    int64_t bx = (p + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_resolvent_table, dim3((unsigned)bx, (unsigned)n, 1u), dim3(256), 0, ctx->stream, D, p, n, p, d_lam, d_s);
    chk(hipGetLastError());
  }
  // weights: R = y - mu in oW, then the resolvent
  double* R = nullptr;
  chk(mln_dmalloc((void**)&R, sizeof(double) * (size_t)np_));
  if (rc == MLN_OK) hipLaunchKernelGGL(k_shift, dim3(grid_1d(np_)), dim3(256), 0, ctx->stream, R, dy.dev, mu, np_);
  if (rc == MLN_OK) rc = resolvent(R, oW.dev);
  if (rc == MLN_OK && leverage) {
    // Q = V o V (in place: V is not needed unsquared again until the variance solve, which re-reads K's eigenvectors
    // from a copy), H = Q^T D, then 1 - sigma^2 H
    double* Q = nullptr;
    chk(mln_dmalloc((void**)&Q, nn));
    if (rc == MLN_OK) chk(hipMemcpyAsync(Q, V, nn, hipMemcpyDeviceToDevice, ctx->stream));
    if (rc == MLN_OK) hipLaunchKernelGGL(k_square_inplace, dim3(grid_1d(n * ld)), dim3(256), 0, ctx->stream, Q, n * ld);
    GemmArgs g{};
    g.A = Q; g.lda = ld; g.B = D; g.ldb = p; g.C = oH.dev; g.ldc = p;
    g.M = n; g.N = p; g.K = n; g.alpha = 1.0; g.beta = 0.0; g.ta = 1; g.tb = 0; g.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK) hipLaunchKernelGGL(k_leverage_finish, dim3(grid_1d(np_)), dim3(256), 0, ctx->stream, oH.dev, np_, p, d_sig2);
    (void)hipStreamSynchronize(ctx->stream);
    if (Q) (void)mln_dfree(Q);
  }
  if (rc == MLN_OK && corrected_r2) {
    GemmArgs g{};
    g.A = K; g.lda = ld; g.B = oW.dev; g.ldb = p; g.C = R; g.ldc = p;                 // K W (K is symmetric, kept by dev_eigh)
    g.M = n; g.N = p; g.K = n; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.split_k = 1;
    rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK) hipLaunchKernelGGL(k_hc3, dim3(grid_1d(np_)), dim3(256), 0, ctx->stream, oC.dev, dy.dev, R, mu, oH.dev, np_);
    if (rc == MLN_OK && variance_W) rc = resolvent(oC.dev, oV.dev);                    // variance_mu = 0
  }
  chk(hipGetLastError());
  if (rc == MLN_OK) rc = oW.commit();
  if (rc == MLN_OK && leverage) rc = oH.commit();
  if (rc == MLN_OK && corrected_r2) rc = oC.commit();
  if (rc == MLN_OK && variance_W) rc = oV.commit();
  (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {K, V, D, T, R, d_lam, d_s, d_sig2};
  for (void* q : ptrs) if (q) (void)mln_dfree(q);
  return rc;
}

extern "C" int mln_sparse_solve_noise(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                                      int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                                      const double* sigma, int32_t sigma_kind, double jitter, double* W) {
  return sparse_solve_impl(ctx, cov, x, n_local, d, xu, m, y, p, mu, sigma, sigma_kind, jitter, W, nullptr, nullptr);
}
