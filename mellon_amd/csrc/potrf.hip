// Diagonal-block kernel of the blocked Cholesky (reference arithmetic: jnp.linalg.cholesky behind
// decomposition.py:115 `_full_rank` and the Ridge factor of parameters.py:895-896, i.e. LAPACK potrf).
//
// One 256-thread workgroup factors a 128 x 128 diagonal block entirely in LDS and also returns the inverse of the
// factor, so that the rest of the block column becomes ONE GEMM (P <- P T^-T) and the trailing update ONE SYRK-shaped
// GEMM with K = 128: 40 steps for m = 5000 instead of 79 steps of a single-wave 64 x 64 kernel.
//
//   T (128 x 130 doubles, row stride = 2 mod 32 so that "16 rows x 2 k" MFMA operand reads touch every bank once)
//   eight micro-steps over 16-column panels:
//     (a) wave 0: Cholesky of the 16 x 16 diagonal tile in registers (lane i = row i, v_readlane broadcasts, no
//         barrier) and its inverse X_pp -- for panel p + 1 this runs UNDER step (c) of panel p (look-ahead: wave 0
//         updates the next diagonal tile first, then factors it while waves 1-3 update the other tiles)
//     (b) all waves: L_ip = A_ip X_pp^T for the tiles below                     (4 x v_mfma_f64_16x16x4 per tile)
//     (c) all waves: A_ij -= L_ip L_jp^T for the lower tiles right of the panel (4 MFMAs per tile)
//   then X = T^-1 by block columns: wave w owns block columns w and 7 - w and keeps X_kj in registers -- the D layout
//   of one MFMA is exactly the B-operand layout of the next (row = (lane >> 4) + 4 reg), so the chain
//   X_ij = -X_ii sum_k T_ik X_kj never leaves the register file.
// A non-positive or NaN pivot sets *info = global pivot index + 1 (first failure wins) and leaves the block as is.
#include "mln_internal.h"
#include "linalg.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NB = 128, LDT = 130, MB = 16, NMB = NB / MB, LDX = 18;

__device__ __forceinline__ double bcast_lane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// MFMA operand fetches from a row-major LDS matrix M (row stride ldm):
//   A operand of slice s: A[row = lane & 15][k = 4 s + (lane >> 4)] = M[r0 + (lane & 15)][c0 + 4 s + (lane >> 4)]
//   B operand of slice s holding M^T: B[k][col = lane & 15] = M[r0 + (lane & 15)][c0 + 4 s + (lane >> 4)]  -- same fetch
__device__ __forceinline__ double frag(const double* M, int ldm, int r0, int c0, int s, int lane) {
  return M[(r0 + (lane & 15)) * ldm + c0 + 4 * s + (lane >> 4)];
}

__global__ __launch_bounds__(256) void k_potrf128(double* A, int64_t lda, int nb, double* __restrict__ Dinv,
                                                  int* info, int64_t j0) {
  extern __shared__ double smem[];
  double* T = smem;                        // NB x LDT
  double* Xd = smem + NB * LDT;            // NMB x MB x LDX: inverses of the 16 x 16 diagonal tiles
  __shared__ int bad_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) bad_s = 0;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 7, j = e & 127;
    T[i * LDT + j] = (i < nb && j <= i) ? A[(int64_t)i * lda + j] : ((i == j) ? 1.0 : 0.0);
  }
  __syncthreads();

  // (a) diagonal tile p: Cholesky in registers + inverse, by wave 0 alone (every 16-lane group computes the same thing)
  auto diag_tile = [&](int p) {
    const int c0 = p * MB;
    const int li = lane & 15;
    double a[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) a[j] = (j <= li) ? T[(c0 + li) * LDT + c0 + j] : 0.0;
    double dinv[MB];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      const double dk = bcast_lane(a[k], k);
      if (!(dk > 0.0)) {   // wave-uniform
        if (!bad && lane == 0) { atomicCAS(info, 0, (int)(j0 + c0 + k + 1)); bad_s = 1; }
        bad = true;
      }
      double inv = __builtin_amdgcn_rsq(dk);
      inv = inv * fma(-0.5 * dk, inv * inv, 1.5);
      inv = inv * fma(-0.5 * dk, inv * inv, 1.5);
      double sq = dk * inv;
      sq = fma(0.5 * inv, fma(-sq, sq, dk), sq);
      dinv[k] = inv;
      const double lik = (li == k) ? sq : ((li > k) ? a[k] * inv : 0.0);
      a[k] = lik;
#pragma unroll
      for (int j = k + 1; j < MB; ++j) {
        const double ljk = bcast_lane(lik, j);
        a[j] = fma(-lik, ljk, a[j]);
      }
    }
    // X = L^-1, lane c owns column c:  x_i = (delta_ic - sum_{k<i} L_ik x_k) / L_ii   (L_ik = a[k] of lane i)
    double x[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      double s = (i == li) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s = fma(-bcast_lane(a[k], i), x[k], s);
      x[i] = s * dinv[i];
    }
    if (lane < MB) {
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        T[(c0 + li) * LDT + c0 + j] = (j <= li) ? a[j] : 0.0;
        Xd[(p * MB + j) * LDX + li] = x[j];    // X[j][li]
      }
    }
  };
  // (c) one trailing tile: A_ij -= L_ip L_jp^T
  auto trailing_tile = [&](int p, int t) {
    const int c0 = p * MB;
    int ii = 0, rem = t;           // t -> (ii, jj) with 0 <= jj <= ii, row-major over the lower triangle
    while (rem > ii) { rem -= ii + 1; ++ii; }
    const int jj = rem;
    const int r0 = (p + 1 + ii) * MB, q0 = (p + 1 + jj) * MB;
    v4d acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = T[(r0 + (lane >> 4) + 4 * r) * LDT + q0 + (lane & 15)];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag(T, LDT, r0, c0, s, lane), frag(T, LDT, q0, c0, s, lane), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) T[(r0 + (lane >> 4) + 4 * r) * LDT + q0 + (lane & 15)] = acc[r];
  };

  if (wave == 0) diag_tile(0);
  __syncthreads();
  if (bad_s) return;   // uniform: leave the block unfactorised, *info says where
  for (int p = 0; p < NMB; ++p) {
    const int c0 = p * MB;
    // ---- (b) panel tiles below the diagonal: L_ip = A_ip X_pp^T ---------------------------------------------
    for (int i = p + 1 + wave; i < NMB; i += 4) {
      const int r0 = i * MB;
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(frag(T, LDT, r0, c0, s, lane), frag(Xd + p * MB * LDX, LDX, 0, 0, s, lane),
                                                   acc, 0, 0, 0);
      // the whole tile has been read into A operands before any lane stores (MFMA results depend on all loads)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(r0 + (lane >> 4) + 4 * r) * LDT + c0 + (lane & 15)] = acc[r];
    }
    __syncthreads();
    if (p + 1 == NMB) break;
    // ---- (c) trailing tiles, with look-ahead: wave 0 updates the NEXT diagonal tile first and factors it right away
    //      (the serial 16-step recurrence of (a): ~3 us) while waves 1-3 update the other tiles ----------------------
    {
      const int nt = NMB - 1 - p;               // tiles per side
      const int ntri = nt * (nt + 1) / 2;
      if (wave == 0) {
        trailing_tile(p, 0);                    // tile (p + 1, p + 1); LDS operations of one wave complete in order
        diag_tile(p + 1);
      } else {
        for (int t = wave; t < ntri; t += 3) trailing_tile(p, t);
      }
    }
    __syncthreads();
    if (bad_s) return;
  }

  // ---- factor back to global memory (strict upper part of the block zeroed) ------------------------------------
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 7, j = e & 127;
    if (i < nb && j < nb) A[(int64_t)i * lda + j] = (j <= i) ? T[i * LDT + j] : 0.0;
  }
  // ---- X = T^-1 by block columns ----------------------------------------------------------------------------------
  for (int pass = 0; pass < 2; ++pass) {
    const int j = pass == 0 ? wave : (NMB - 1 - wave);
    v4d X[NMB];                                 // X_kj for k = j .. 7 (B-operand / D layout)
#pragma unroll
    for (int k = 0; k < NMB; ++k) X[k] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NMB; ++k) {
      if (k == j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) X[k][r] = Xd[(j * MB + (lane >> 4) + 4 * r) * LDX + (lane & 15)];
      }
    }
#pragma unroll
    for (int i = 1; i < NMB; ++i) {
      if (i > j) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NMB; ++k) {
          if (k >= j && k < i) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(frag(T, LDT, i * MB, k * MB, s, lane), X[k][s], acc, 0, 0, 0);
          }
        }
        v4d out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s)
          out = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag(Xd + i * MB * LDX, LDX, 0, 0, s, lane), acc[s], out, 0, 0, 0);
        X[i] = out;
      }
    }
#pragma unroll
    for (int i = 0; i < NMB; ++i) {
      if (i >= j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Dinv[(i * MB + (lane >> 4) + 4 * r) * NB + j * MB + (lane & 15)] = X[i][r];
      }
    }
  }
}

}  // namespace

// Factorises the nb x nb (nb <= 128) lower block at A in place and writes T^-1 to Dinv (128 x 128, row-major, leading
// dimension 128; entries above the diagonal are never written: the caller zeroes the buffer once).
int launch_potrf128(mln_ctx* ctx, double* A, int64_t lda, int nb, double* Dinv, int* info, int64_t j0) {
  const size_t lds = sizeof(double) * (size_t)(NB * LDT + NMB * MB * LDX);
  static bool configured = false;
  if (!configured) {
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
  hipLaunchKernelGGL(k_potrf128, dim3(1), dim3(256), lds, ctx->stream, A, lda, nb, Dinv, info, j0);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
