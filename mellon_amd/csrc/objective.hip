// The iterative hot loop: one pass over the cell-sharded factor L (n x m, fp64, row-major) per
// objective evaluation.  Reference: mellon/inference.py:35-92,167-192 (loss), whose autodiff
// gradient is z + L^T (exp(L z + mu + V) - 1); inference.py:291-338 (diagonal Hessian) and
// inference.py:341-354 (f = L z + mu).  The reference reads L twice per evaluation (L z, then
// L^T via reverse mode); here each row block is read ONCE:
//
//   512-thread workgroup; thread t owns the double2 column pairs {t, t+512, ...} (CPT of them,
//   m <= 1024*CPT), keeps z, the gradient partial (and the Hessian-diagonal partial) for those
//   columns in registers, and streams R rows at a time through registers with 16-byte coalesced
//   loads: partial dots -> wave64 shuffle reduction -> 8 wave partials through LDS (fixed order)
//   -> a_i = exp(f_i + V_i) -> g += (a_i - 1) * row, from the SAME registers.  The next R rows
//   are requested before the current ones are consumed (two register sets), so ~2 x R x m x 8 B
//   per CU are in flight against HBM latency with one workgroup per CU.
//
// HBM-bound: algorithmic bytes per launch = n_local * ldl * 8 (+ O(n + n_wg * m)).
// Determinism: rows are split into contiguous per-workgroup ranges; per-workgroup partials are
// summed in workgroup order by k_reduce_obj.
#include "mln_internal.h"

namespace {

constexpr int WG = 512;
enum { MODE_OBJ = 0, MODE_OBJ_HESS = 1, MODE_GEMVT = 2, MODE_FONLY = 3 };

typedef double d2 __attribute__((ext_vector_type(2)));

template <int CPT, int R>
__device__ __forceinline__ void load_rows(const d2* __restrict__ L2, int64_t ld2, int64_t row, int64_t row_end,
                                          unsigned tid, d2 (&v)[R][CPT]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool rok = (row + r) < row_end;
    // wave-uniform row base (scalar registers) + 32-bit per-lane offset -> saddr addressing
    const d2* rowp = L2 + (rok ? (row + r) : row) * ld2;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const unsigned off = (unsigned)c * WG + tid;
      const bool ok = rok && off < (unsigned)ld2;
      v[r][c] = ok ? __builtin_nontemporal_load(rowp + off) : (d2){0.0, 0.0};
    }
  }
}

template <int CPT, int R, int MODE>
__device__ __forceinline__ void process_rows(const ObjArgs& a, int64_t row, int64_t row_end, int tid, int par,
                                             const d2 (&v)[R][CPT], const d2 (&z)[CPT], d2 (&g)[CPT],
                                             d2 (&h)[CPT], double& loss, double (*red)[8][R]) {
  double coef[R], aexp[R];
  if (MODE == MODE_GEMVT) {
#pragma unroll
    for (int r = 0; r < R; ++r) coef[r] = (row + r < row_end) ? a.weights[row + r] : 0.0;
  } else {
    double dot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        s = fma(v[r][c].x, z[c].x, s);
        s = fma(v[r][c].y, z[c].y, s);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      dot[r] = s;
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) red[par][wave][r] = dot[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[par][w][r];
      const bool rok = (row + r) < row_end;
      const double f = s + a.mu;
      if (MODE == MODE_FONLY) {
        if (rok && tid == 0) a.f_out[row + r] = f;
        coef[r] = 0.0;
        aexp[r] = 0.0;
      } else {
        const double Vi = rok ? a.V[row + r] : 0.0;
        const double e = rok ? exp(f + Vi) : 0.0;
        aexp[r] = e;
        coef[r] = rok ? (e - 1.0) : 0.0;
        if (tid == 0 && rok) loss -= (f + a.Vdr[row + r]) - e;   // inference.py:89-91
      }
    }
  }
  if (MODE != MODE_FONLY) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        g[c].x = fma(coef[r], v[r][c].x, g[c].x);
        g[c].y = fma(coef[r], v[r][c].y, g[c].y);
        if (MODE == MODE_OBJ_HESS) {
          h[c].x = fma(aexp[r] * v[r][c].x, v[r][c].x, h[c].x);
          h[c].y = fma(aexp[r] * v[r][c].y, v[r][c].y, h[c].y);
        }
      }
    }
  }
}

template <int CPT, int R, int MODE>
__global__ __launch_bounds__(WG) void k_objective(ObjArgs a) {
  __shared__ double red[2][8][R];
  const int tid = threadIdx.x;
  const int64_t ld2 = a.ldl / 2;
  const d2* __restrict__ L2 = reinterpret_cast<const d2*>(a.L);
  // contiguous row range of this workgroup, in units of R rows
  const int64_t nsteps = (a.n + R - 1) / R;
  const int64_t per = (nsteps + a.n_wg - 1) / a.n_wg;
  const int64_t s_beg = (int64_t)blockIdx.x * per;
  int64_t s_end = s_beg + per;
  if (s_end > nsteps) s_end = nsteps;

  d2 z[CPT], g[CPT], h[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int64_t col = 2 * ((int64_t)c * WG + tid);
    z[c] = (d2){0.0, 0.0};
    if (MODE != MODE_GEMVT) {
      if (col < a.m) z[c].x = a.z[col];
      if (col + 1 < a.m) z[c].y = a.z[col + 1];
    }
    g[c] = (d2){0.0, 0.0};
    h[c] = (d2){0.0, 0.0};
  }
  double loss = 0.0;
  d2 va[R][CPT], vb[R][CPT];
  if (s_beg < s_end) load_rows<CPT, R>(L2, ld2, s_beg * R, a.n, tid, va);
  for (int64_t s = s_beg; s < s_end; s += 2) {
    if (s + 1 < s_end) load_rows<CPT, R>(L2, ld2, (s + 1) * R, a.n, tid, vb);
    process_rows<CPT, R, MODE>(a, s * R, a.n, tid, 0, va, z, g, h, loss, red);
    if (s + 2 < s_end) load_rows<CPT, R>(L2, ld2, (s + 2) * R, a.n, tid, va);
    if (s + 1 < s_end) process_rows<CPT, R, MODE>(a, (s + 1) * R, a.n, tid, 1, vb, z, g, h, loss, red);
  }
  if (MODE != MODE_FONLY) {
    double* pg = a.part_grad + (int64_t)blockIdx.x * a.m_pad;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int64_t col = 2 * ((int64_t)c * WG + tid);
      if (col < a.m_pad) *reinterpret_cast<d2*>(pg + col) = g[c];
    }
    if (MODE == MODE_OBJ_HESS) {
      double* ph = a.part_hess + (int64_t)blockIdx.x * a.m_pad;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int64_t col = 2 * ((int64_t)c * WG + tid);
        if (col < a.m_pad) *reinterpret_cast<d2*>(ph + col) = h[c];
      }
    }
    if (tid == 0 && a.part_loss) a.part_loss[blockIdx.x] = loss;
  }
}

// out[0] = sum_wg loss ; out[1 + j] = sum_wg grad[wg][j] ; out[1 + m + j] = sum_wg hess[wg][j]
__global__ void k_reduce_obj(ObjArgs a, double* __restrict__ out, int with_hess) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < a.m) {
    double s = 0.0;
    for (int w = 0; w < a.n_wg; ++w) s += a.part_grad[(int64_t)w * a.m_pad + j];
    out[1 + j] = s;
    if (with_hess) {
      double t = 0.0;
      for (int w = 0; w < a.n_wg; ++w) t += a.part_hess[(int64_t)w * a.m_pad + j];
      out[1 + a.m + j] = t;
    }
  }
  if (j == 0) {
    double s = 0.0;
    if (a.part_loss)
      for (int w = 0; w < a.n_wg; ++w) s += a.part_loss[w];
    out[0] = s;
  }
}

template <int CPT, int R>
int launch_mode(mln_ctx* ctx, const ObjArgs& a, int mode) {
  dim3 grid((unsigned)a.n_wg), block(WG);
  switch (mode) {
    case MODE_OBJ: hipLaunchKernelGGL((k_objective<CPT, R, MODE_OBJ>), grid, block, 0, ctx->stream, a); break;
    case MODE_OBJ_HESS: hipLaunchKernelGGL((k_objective<CPT, R, MODE_OBJ_HESS>), grid, block, 0, ctx->stream, a); break;
    case MODE_GEMVT: hipLaunchKernelGGL((k_objective<CPT, R, MODE_GEMVT>), grid, block, 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((k_objective<CPT, R, MODE_FONLY>), grid, block, 0, ctx->stream, a); break;
  }
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

}  // namespace

int objective_max_m() { return 1024 * 8; }

int launch_objective(mln_ctx* ctx, const ObjArgs& a) {
  if (a.ldl % 2 != 0) { mln_set_error(ctx, "objective: leading dimension of L must be even"); return MLN_ERR_ARG; }
  int mode = MODE_OBJ;
  if (a.weights) mode = MODE_GEMVT;
  else if (a.f_out) mode = MODE_FONLY;
  else if (a.part_hess) mode = MODE_OBJ_HESS;
  const int64_t pairs = a.ldl / 2;
  const int cpt = (int)((pairs + WG - 1) / WG);
  // rows per step chosen so that one register set holds <= 12 double2 per thread
  switch (cpt) {
    case 1: return launch_mode<1, 8>(ctx, a, mode);
    case 2: return launch_mode<2, 5>(ctx, a, mode);
    case 3: return launch_mode<3, 3>(ctx, a, mode);
    case 4: return launch_mode<4, 2>(ctx, a, mode);
    case 5: return launch_mode<5, 2>(ctx, a, mode);
    case 6: return launch_mode<6, 2>(ctx, a, mode);
    case 7:
    case 8: return launch_mode<8, 1>(ctx, a, mode);
    default:
      mln_set_error(ctx, "objective: m > 8192 landmarks is not supported by this build");
      return MLN_ERR_UNSUPPORTED;
  }
}

int launch_reduce_obj(mln_ctx* ctx, const ObjArgs& a, double* out) {
  hipLaunchKernelGGL(k_reduce_obj, dim3((unsigned)((a.m + 255) / 256)), dim3(256), 0, ctx->stream, a, out,
                     a.part_hess ? 1 : 0);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
