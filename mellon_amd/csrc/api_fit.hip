// C ABI, part 2: the fit handle (see api_internal.h for the map of the api*.hip files).
#include "api_internal.h"
#include "mln_options.h"
#include <mutex>
#include <unordered_map>

// rows of this shard in the subsample of stride s: first local index and count
void fit_sample_rows(const mln_fit* f, int64_t s, int64_t* first, int64_t* rows) {
  if (s < 1) s = 1;
  *first = (s - f->row0 % s) % s;
  *rows = (f->n > *first) ? (f->n - *first + s - 1) / s : 0;
}

void fit_free(mln_fit* f) {
  if (!f) return;
  mln_ctx* ctx = f->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (f->L && f->L != f->Lp) (void)mln_dfree(f->L);
  if (f->Lp) (void)mln_dfree(f->Lp);
  triinv_free(&f->tri);
  void* ptrs[] = {f->V, f->Vdr, f->part_grad, f->part_hess, f->part_loss, f->d_z, f->d_out,
                  f->C, f->Cinv, f->d_u, f->d_gu, f->d_tmp, f->P, f->d_w, f->d_w_cached,
                  f->Q1, f->Q2, f->d_zw, f->d_zr, f->eigU, f->L32, f->sv_block, f->f_keep[0], f->f_keep[1], f->Kj, f->d_over};
  for (void* p : ptrs) if (p) (void)mln_dfree(p);
  for (double* p : f->saved_precond) if (p) (void)mln_dfree(p);
  if (f->h_state) (void)mln_hfree(f->h_state);
  fit_events_return(ctx, &f->evs);
  if (f->h_z) (void)mln_hfree(f->h_z);
  if (f->h_out) (void)mln_hfree(f->h_out);
  if (f->ev0) (void)hipEventDestroy(f->ev0);
  if (f->ev1) (void)hipEventDestroy(f->ev1);
  delete f;
}

extern "C" void mln_fit_destroy(mln_fit* fit) { fit_free(fit); }

int fit_alloc_workspace(mln_fit* f) {
  mln_ctx* ctx = f->ctx;
  int64_t steps = (f->n + 1) / 2;
  int n_wg = ctx->n_cu > 0 ? ctx->n_cu : 256;
  f->n_wg_cap = n_wg;                      // partial buffers are sized for this many workgroups
  if (steps < n_wg) n_wg = (int)(steps > 0 ? steps : 1);
  f->n_wg = n_wg;
  n_wg = f->n_wg_cap;
  const size_t pm = (size_t)f->ldl;
  f->ld2 = pad16(f->m + 2);
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_u, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_gu, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_tmp, sizeof(double) * (1 + pm)));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_w, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_w_cached, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_zw, sizeof(double) * 2 * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_zr, sizeof(double) * 2 * (size_t)f->ld2));
  MLN_HIP(ctx, hipMemsetAsync(f->d_zr, 0, sizeof(double) * 2 * (size_t)f->ld2, ctx->stream));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_over, 64));
  MLN_HIP(ctx, hipMemsetAsync(f->d_over, 0, 64, ctx->stream));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->part_grad, sizeof(double) * pm * n_wg));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->part_hess, sizeof(double) * pm * n_wg));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->part_loss, sizeof(double) * n_wg));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_z, sizeof(double) * pm));
  MLN_HIP(ctx, mln_dmalloc((void**)&f->d_out, sizeof(double) * (1 + 2 * pm)));
  MLN_HIP(ctx, mln_hmalloc((void**)&f->h_z, sizeof(double) * pm));
  MLN_HIP(ctx, mln_hmalloc((void**)&f->h_out, sizeof(double) * (1 + 2 * pm)));
  MLN_HIP(ctx, hipEventCreate(&f->ev0));
  MLN_HIP(ctx, hipEventCreate(&f->ev1));
  return MLN_OK;
}

// Upload of the cells from host memory in row chunks UNDER the kernel-matrix pass (see fit_prepare_impl).
// History (C3, host-to-host step minus the step with resident cells): one plain copy before everything else +13.6 ms; sixteen
// equal chunks copied by a helper thread, each chunk's pass waiting for its event +10.9 ms (round 4); a pool of pinned staging
// buffers filled by eight memcpy threads +20 ms; page-locked source (mln_host_register), copies enqueued by the calling thread
// +6.8 ms (round 5); growing chunks (below) take the pass's own share of that from +2.0 to +0.8 ms.
// A copy from pageable memory blocks its CALLING thread while the runtime stages it, but not the device: a helper thread
// issues those.  Chunk c is complete on the device when events[c] has fired; the main thread makes its stream wait for that
// event -- after it has been recorded (done > c).
// The copy stream and its events live as long as the context (created by the first upload, released by mln_ctx_destroy):
// hipStreamCreate alone costs ~3 ms, which every host-to-host fit paid before the first chunk was even enqueued
// (MELLON_AMD_TRACE: "upload started at 0.0031 s").
namespace {
struct CopyLane { hipStream_t stream = nullptr; std::vector<hipEvent_t> events; };
std::mutex g_copy_mu;
std::unordered_map<mln_ctx*, CopyLane> g_copy_lanes;
}  // namespace

static int copy_lane_get(mln_ctx* ctx, int n_events, hipStream_t* stream, std::vector<hipEvent_t>* events) {
  std::lock_guard<std::mutex> lk(g_copy_mu);
  CopyLane& lane = g_copy_lanes[ctx];
  if (!lane.stream) MLN_HIP(ctx, hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking));
  while ((int)lane.events.size() < n_events) {
    hipEvent_t e = nullptr;
    MLN_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    lane.events.push_back(e);
  }
  *stream = lane.stream;
  events->assign(lane.events.begin(), lane.events.begin() + n_events);
  return MLN_OK;
}

// The per-evaluation timing events of the device-resident solver (three per evaluation, ~100 per fit) are the context's too:
// a solve borrows them into its handle and hands them back (fit_free does so for a solve that ended early).
namespace {
std::unordered_map<mln_ctx*, std::vector<hipEvent_t>> g_event_pools;
}  // namespace
void fit_events_borrow(mln_ctx* ctx, std::vector<hipEvent_t>* evs) {
  std::lock_guard<std::mutex> lk(g_copy_mu);
  auto& pool = g_event_pools[ctx];
  if (evs->empty()) evs->swap(pool);
}
void fit_events_return(mln_ctx* ctx, std::vector<hipEvent_t>* evs) {
  std::lock_guard<std::mutex> lk(g_copy_mu);
  auto& pool = g_event_pools[ctx];
  if (pool.empty()) pool.swap(*evs);
  else { for (hipEvent_t e : *evs) (void)hipEventDestroy(e); evs->clear(); }
}

void fit_release_copy_lane(mln_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_copy_mu);
  auto pe = g_event_pools.find(ctx);
  if (pe != g_event_pools.end()) { for (hipEvent_t e : pe->second) (void)hipEventDestroy(e); g_event_pools.erase(pe); }
  auto it = g_copy_lanes.find(ctx);
  if (it == g_copy_lanes.end()) return;
  if (it->second.stream) { (void)hipStreamSynchronize(it->second.stream); (void)hipStreamDestroy(it->second.stream); }
  for (hipEvent_t e : it->second.events) if (e) (void)hipEventDestroy(e);
  g_copy_lanes.erase(it);
}

struct HostUpload {
  mln_ctx* ctx = nullptr;
  hipStream_t copy = nullptr;
  std::vector<hipEvent_t> events;
  std::atomic<int> done{0};
  std::atomic<int> failed{0};
  std::thread th;
  int n_chunks = 0;
  int64_t n = 0;
  std::vector<int64_t> row0;     // chunk c = rows [row0[c], row0[c + 1])
  int d = 0;
  int start(mln_ctx* c, const double* src, double* dst, int64_t n_, int d_) {
    ctx = c; n = n_; d = d_;
    // Chunks of whole 128-row workgroup tiles, GROWING: 1/32, 1/32, 1/16, 1/8, 1/4 of the cells and the rest.  The copy
    // (56 GB/s measured, pageable or page-locked: tools/h2d_probe.py) delivers a cell 2.3x faster than the kernel-matrix
    // pass consumes it, so after the first chunk (0.2 ms at C3) every chunk is there before the pass of the one before it
    // ends -- and six launches have six tails where round 4's sixteen equal chunks had sixteen (each launch ends with CUs
    // idling behind its slowest workgroups: +2.0 ms on the pass at C3).
    {
      const int64_t tiles = (n + 127) / 128;
      int64_t size = std::max<int64_t>(1, tiles / 32), at = 0;
      row0.assign(1, 0);
      for (int k = 0; at < tiles; ++k) {
        const int64_t take = (tiles - at <= 2 * size || k >= 14) ? tiles - at : size;   // the rest once it is within two steps
        at += take;
        row0.push_back(std::min(n, at * 128));
        if (k >= 1) size *= 2;
      }
      n_chunks = (int)row0.size() - 1;
    }
    MLN_TRY(copy_lane_get(ctx, n_chunks, &copy, &events));     // (the context's own: not created or destroyed per fit)
    // a page-locked source (mln_host_register): the copies are DMA transfers the runtime queues without blocking -- all of
    // them are enqueued here, by the calling thread, and run under whatever the main stream does meanwhile
    {
      hipPointerAttribute_t attr;
      const bool pinned_src = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
      if (!pinned_src) (void)hipGetLastError();
      if (pinned_src) {
        for (int c = 0; c < n_chunks; ++c) {
          const int64_t r0 = row0[(size_t)c], rows = row0[(size_t)c + 1] - r0;
          MLN_HIP(ctx, hipMemcpyAsync(dst + r0 * d, src + r0 * d, sizeof(double) * (size_t)(rows * d), hipMemcpyHostToDevice, copy));
          MLN_HIP(ctx, hipEventRecord(events[(size_t)c], copy));
        }
        done.store(n_chunks, std::memory_order_release);
        return MLN_OK;
      }
    }
    const int device = ctx->device;
    th = std::thread([this, src, dst, device] {
      if (hipSetDevice(device) != hipSuccess) { failed.store(1); done.store(n_chunks); return; }
      for (int c = 0; c < n_chunks; ++c) {
        const int64_t r0 = row0[(size_t)c], rows = row0[(size_t)c + 1] - r0;
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
    *r0 = row0[(size_t)c];
    *rows = row0[(size_t)c + 1] - *r0;
    return MLN_OK;
  }
  int finish() {
    if (th.joinable()) th.join();
    return failed.load() ? MLN_ERR_HIP : MLN_OK;
  }
  ~HostUpload() {
    if (th.joinable()) th.join();
    if (copy) (void)hipStreamSynchronize(copy);
  }
};

// ---- deferred Lp (MLN_FIT_DEFER_LP) --------------------------------------------------------------------------------------
// rc_chol: what the factorisation of f->Lp (alone or batched with the preconditioner's matrix) returned
int fit_lp_finish(mln_fit* f, int rc_chol, double t0) {
  mln_ctx* ctx = f->ctx;
  if (rc_chol == MLN_ERR_NOT_PD) {
    f->lp_failed = true;
    char buf[160];
    snprintf(buf, sizeof buf, "cov(xu, xu) + jitter I is not positive definite (jitter = %g): %s", f->jitter, ctx->err.c_str());
    mln_set_error(ctx, buf);
    return rc_chol;
  }
  MLN_TRY(rc_chol);
  f->lp_pending = false;
  f->tri_pending = true;       // the block-scaled copies for triangular SOLVES with Lp: built by the first call that solves
  f->times[1] += now_s() - t0;
  return MLN_OK;
}

int fit_ensure_lp(mln_fit* f, bool need_tri) {
  if (f->lp_failed) { mln_set_error(f->ctx, "cov(xu, xu) + jitter I is not positive definite"); return MLN_ERR_NOT_PD; }
  if (f->lp_pending) {
    const double t0 = now_s();
    MLN_TRY(fit_lp_finish(f, dev_cholesky_lower(f->ctx, f->Lp, f->m, f->ldp), t0));
  }
  if (need_tri && f->tri_pending) {
    const double t0 = now_s();
    MLN_TRY(triinv_build(f->ctx, f->Lp, f->m, f->ldp, true, true, &f->tri));
    MLN_HIP(f->ctx, hipStreamSynchronize(f->ctx->stream));
    f->tri_pending = false;
    f->times[1] += now_s() - t0;
  }
  return MLN_OK;
}

int fit_prepare_impl(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n, int32_t d,
                            const double* xu, int64_t m, double jitter, const double* Lp_in, int32_t flags,
                            mln_fit* f) {
  f->ctx = ctx;
  const double t_enter = now_s();
  const bool trace_all = std::getenv("MELLON_AMD_TRACE") != nullptr;
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
  bool pipelined = !f->full && n > 0 && x && !is_device_ptr(x) && (size_t)n * d * sizeof(double) >= ((size_t)32 << 20);
  if (pipelined) {
    dx.ctx = ctx;
    MLN_HIP(ctx, mln_dmalloc((void**)&dx.owned, (size_t)n * d * sizeof(double)));
    dx.dev = dx.owned;
    MLN_TRY(up.start(ctx, x, dx.owned, n, d));
    if (trace_all) fprintf(stderr, "[trace] fit_prepare: upload started at %.4f s (%d chunks)\n", now_s() - t_enter, up.n_chunks);
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
  // (Round 4 built the landmark-only chain -- cov(xu, xu), its Cholesky factor, the block-scaled copies -- in a helper thread
  //  on a second stream UNDER the kernel-matrix pass on a CU-masked stream: 1.9 ms less kernel time, 2.3 ms more wall time;
  //  taken out in round 5, profiles/HISTORY.md.)
  if (Lp_in) {
    DevIn dl;
    MLN_TRY(dl.init(ctx, Lp_in, (size_t)m * m));
    MLN_TRY(launch_copy_block(ctx, dl.dev, m, f->Lp, f->ldp, m, m));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    MLN_TRY(launch_kernel_matrix(ctx, f->cov, centers, m, centers, m, d, f->Lp, f->ldp, jitter));
    const bool implicit = !f->full && (flags & MLN_FIT_IMPLICIT) != 0;
    if (implicit) {
      // Kj = cov(xu, xu) + jitter I survives the factorisation: the prior's Hessian in w-space (fit_build_precond)
      MLN_HIP(ctx, mln_dmalloc((void**)&f->Kj, lp_bytes));
      MLN_HIP(ctx, hipMemcpyAsync(f->Kj, f->Lp, lp_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    f->times[0] += now_s() - t0;
    t0 = now_s();
    // MLN_FIT_DEFER_LP: nothing between here and the preconditioner needs the factor (the n x m buffer keeps K itself), and
    // its chain of 40 dependent block steps rides along with the preconditioner's (api_precond.hip fit_factor_precond)
    if (implicit && (flags & MLN_FIT_DEFER_LP)) f->lp_pending = true;
    else MLN_TRY(dev_cholesky_lower(ctx, f->Lp, m, f->ldp));
  }
  f->jitter = jitter;
  if (!f->lp_pending) {
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
    // Mixed precision (OPT-IN since round 5: MELLON_AMD_MIXED=1, large implicit fits): the kernel-matrix pass also writes
    // a 32-bit copy, which the first passes of the MAP solve stream instead of the fp64 one.  The product default is the
    // pure-fp64 solve: with the subsample phase and the rebuilt preconditioner the mixed solve measured no faster
    // (174.3 against 176.9 ms at C3) for +20 GB of HBM and a surrogate-correction state machine on the path.
    int64_t mixed_min = (int64_t)1 << 27;
    bool mixed = false;
    if (const char* ev = std::getenv("MELLON_AMD_MIXED")) mixed = (flags & MLN_FIT_IMPLICIT) != 0 && std::atoi(ev) != 0;
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
    // the evaluation workspace (device vectors, pinned mirrors, events: ~0.5 ms of host calls) while the pass runs
    MLN_TRY(fit_alloc_workspace(f));
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
  if (!f->d_u) MLN_TRY(fit_alloc_workspace(f));
  if (trace_all) fprintf(stderr, "[trace] fit_prepare: body done at %.4f s\n", now_s() - t_enter);
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
  const double t_call = now_s();
  int rc = fit_prepare_impl(ctx, cov, x, n_local, d, xu, m, jitter, Lp_in, flags, f);
  if (std::getenv("MELLON_AMD_TRACE")) fprintf(stderr, "[trace] mln_fit_prepare: %.4f s with the upload's teardown\n", now_s() - t_call);
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
  if (f->kspace && n_est >= 24 * m) stride = std::max<int64_t>(1, n_est / (12 * m));
  auto gram = [&]() { return f->kspace ? fit_gram(f, G, ld, stride) : gram_of(ctx, f->L, f->ldl, f->n, m, 1.0, G, ld); };
  int rc = gram();
  double lmax = 0.0;
  // The count by inertia (ldl_inertia.hip: Lanczos for lambda_max, then the signs of the pivots of G - x I; ~15 ms at
  // m = 5000) first; where it cannot certify its pivots, the tridiagonal path (0.25 s) on a fresh Gram.
  const bool ldl_ok = !(mln_experiment("MELLON_AMD_RANK_LDL") && std::atoi(mln_experiment("MELLON_AMD_RANK_LDL")) == 0);
  bool done = false;
  if (rc == MLN_OK && ldl_ok) {
    rc = dev_sym_rank_above_ldl(ctx, G, m, ld, tol * tol, rank_out, &lmax, &done);
    if (rc == MLN_OK && !done) rc = gram();
  }
  if (rc == MLN_OK && !done) rc = dev_sym_rank_above(ctx, G, m, ld, tol * tol, rank_out, &lmax);
  f->rank_path = done ? 1 : 2;
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
  MLN_TRY(fit_ensure_lp(f, false));
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
    MLN_TRY(fit_ensure_lp(f));
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

ObjArgs obj_args(mln_fit* f) {
  ObjArgs a{};
  a.L = f->L; a.ldl = f->ldl; a.n = f->n; a.m = f->m;
  a.z = f->d_z; a.V = f->V; a.Vdr = f->Vdr; a.mu = f->mu;
  a.part_grad = f->part_grad; a.part_hess = nullptr; a.part_loss = f->part_loss;
  a.weights = nullptr; a.f_out = nullptr;
  a.n_wg = f->n_wg; a.m_pad = f->ldl;
  return a;
}

