// The stable core every translation unit of libmellon_hip.so shares: the context, error plumbing, the caching
// allocator and the device-side covariance program.  (Kept apart from mln_internal.h so that the slow-to-compile
// persistent-row kernels only rebuild when one of THESE declarations changes.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/mellon_hip.h"

struct mln_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_cu = 0;
  void* comm = nullptr;  // ncclComm_t when multi-GPU (one process per GPU)
  struct mln_loopback* loop = nullptr;  // in-process thread-rank communicator (comm.hip)
  int n_ranks = 1;
  int rank = 0;
  std::string err;
  // grow-only device scratch
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  int* d_info = nullptr;  // device int[4] for factorisation status
};

void mln_set_error(mln_ctx* ctx, const std::string& msg);
int mln_hip_fail(mln_ctx* ctx, hipError_t e, const char* what, const char* file, int line);

#define MLN_HIP(ctx, call)                                                   \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess) return mln_hip_fail((ctx), e__, #call, __FILE__, __LINE__); \
  } while (0)

#define MLN_TRY(call)            \
  do {                           \
    int s__ = (call);            \
    if (s__ != MLN_OK) return s__; \
  } while (0)

// alloc.hip: caching device allocator (every internal device buffer goes through it)
hipError_t mln_dmalloc(void** out, size_t bytes);
hipError_t mln_dfree(void* p);
void mln_dcache_flush();

// ---- device-side covariance program (by-value kernel argument) ------------------------------
struct DevLeaf {
  int kind;
  int ndims;
  int dims_off;  // offset into DevCov::dims
  int pad;
  double ls;
  double alpha;
  double alpha_inv_ls[2];  // [1] = 1 / ls
};
struct DevCov {
  int n_leaves;
  int n_toks;
  DevLeaf leaves[MLN_MAX_LEAVES];
  int tok_op[MLN_MAX_TOKS];
  int tok_leaf[MLN_MAX_TOKS];
  double tok_val[MLN_MAX_TOKS];
  short dims[MLN_MAX_DIMS];
};
int mln_lower_cov(mln_ctx* ctx, const mln_kernel_desc* cov, int d, DevCov* out);

// helpers (api.hip)
int mln_scratch(mln_ctx* ctx, size_t bytes, void** out);
