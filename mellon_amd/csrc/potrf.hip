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

// Phase stamps for tools/potrf_probe.py: build with -DMLN_POTRF_TIMING (100 MHz wall clock in [0, 32), shader clock in [32, 64))
#ifdef MLN_POTRF_TIMING
__device__ long long g_potrf_ts[64];
#define TS(i) do { if (threadIdx.x == 0) { g_potrf_ts[i] = wall_clock64(); g_potrf_ts[32 + i] = clock64(); } } while (0)
#else
#define TS(i) do { } while (0)
#endif
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NB = 128, LDT = 130, MB = 16, NMB = NB / MB, LDX = 18;
constexpr bool NEWTON2 = true;     // (one step: no faster -- the chain is hidden behind the LDS round trip -- and 3e-15 instead of 5e-16 backward error)

__device__ __forceinline__ double bcast_lane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Broadcast inside each 16-lane row: v_mov_b64_dpp row_newbcast:K (gfx90a+) -- lane K of every row to all 16 lanes of the
// row, one instruction, no scalar registers, no LDS round trip.  The diagonal-tile recurrence keeps row i of the
// tile on lane i of EVERY row (the four rows of the wave compute the same thing), so "the value lane K holds" is exactly
// what a pivot step needs from the others.  k is a loop counter of fully unrolled loops: the switch folds away.
template <int K> __device__ __forceinline__ double row_bcast_c(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, false);      // v_mov_b64_dpp (row_newbcast is the one 64-bit DPP control)
}
__device__ __forceinline__ double row_bcast(double v, int k) {
  switch (k) {
    case 0: return row_bcast_c<0>(v);   case 1: return row_bcast_c<1>(v);   case 2: return row_bcast_c<2>(v);
    case 3: return row_bcast_c<3>(v);   case 4: return row_bcast_c<4>(v);   case 5: return row_bcast_c<5>(v);
    case 6: return row_bcast_c<6>(v);   case 7: return row_bcast_c<7>(v);   case 8: return row_bcast_c<8>(v);
    case 9: return row_bcast_c<9>(v);   case 10: return row_bcast_c<10>(v); case 11: return row_bcast_c<11>(v);
    case 12: return row_bcast_c<12>(v); case 13: return row_bcast_c<13>(v); case 14: return row_bcast_c<14>(v);
    default: return row_bcast_c<15>(v);
  }
}
#ifdef MLN_POTRF_LDS_BCAST
constexpr bool DPP_BCAST = false;      // (the round-3 form: broadcasts through LDS, 9 100 cycles per tile)
#else
constexpr bool DPP_BCAST = true;
#endif

// MFMA operand fetches from a row-major LDS matrix M (row stride ldm):
//   A operand of slice s: A[row = lane & 15][k = 4 s + (lane >> 4)] = M[r0 + (lane & 15)][c0 + 4 s + (lane >> 4)]
//   B operand of slice s holding M^T: B[k][col = lane & 15] = M[r0 + (lane & 15)][c0 + 4 s + (lane >> 4)]  -- same fetch
__device__ __forceinline__ double frag(const double* M, int ldm, int r0, int c0, int s, int lane) {
  return M[(r0 + (lane & 15)) * ldm + c0 + 4 * s + (lane >> 4)];
}

// SIGNED (the inertia count of ldl_inertia.hip): the block is factored as T S T^T with S = diag(+-1) -- a pivot may be
// negative, T_kk = sqrt|d_k|, s_k = sign d_k; a zero or NaN pivot is the failure.  Also written then: DinvS = S T^-1 (so that
// the caller's panel GEMM against it yields L = A_panel T^-T S), the number of negative pivots and the smallest |pivot|.
template <bool SIGNED>
__global__ __launch_bounds__(256) void k_potrf128(double* A, int64_t lda, int nb, double* __restrict__ Dinv,
                                                  int* info, int64_t j0, double* __restrict__ DinvS, int* n_neg,
                                                  unsigned long long* min_piv, int64_t a_bs) {
  extern __shared__ double smem[];
  // batched launch (dev_cholesky_lower2): workgroup b factors the block of matrix b -- its own Dinv slab and info word
  A += (int64_t)blockIdx.x * a_bs; Dinv += (int64_t)blockIdx.x * NB * NB; info += blockIdx.x;
  double* T = smem;                        // NB x LDT
  double* Xd = smem + NB * LDT;            // NMB x MB x LDX: inverses of the 16 x 16 diagonal tiles
  double* Sg = Xd + NMB * MB * LDX;        // NB signs (SIGNED only)
  __shared__ int bad_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TS(0);
  if (tid == 0) bad_s = 0;
  // 128 x 128 doubles as 8192 pairs, 32 per thread in two batches of sixteen UNCONDITIONAL loads (clamped row, value
  // selected afterwards): the branchy one-element-at-a-time form waited for every load on its own -- 11.7 us of a 66 us
  // kernel (s_memtime stamps, tools/potrf_probe.py).  lda is even and A 16-byte aligned wherever this is called from.
  {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const bool pairs_ok = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      d2_t v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * (16 * b + q), i = e >> 6, j = (e & 63) * 2;
        const int ic = (i < nb) ? i : (nb - 1), jc = (j + 1 < nb) ? j : 0;
        if (pairs_ok) v[q] = *reinterpret_cast<const d2_t*>(A + (int64_t)ic * lda + jc);
        else { v[q].x = A[(int64_t)ic * lda + jc]; v[q].y = A[(int64_t)ic * lda + jc + 1]; }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * (16 * b + q), i = e >> 6, j = (e & 63) * 2;
        const bool in0 = i < nb && j <= i && j + 1 < nb, in1 = i < nb && j + 1 <= i && j + 1 < nb;
        double x0 = in0 ? v[q].x : ((i == j) ? 1.0 : 0.0);
        double x1 = in1 ? v[q].y : ((i == j + 1) ? 1.0 : 0.0);
        if (i < nb && j <= i && j + 1 >= nb) x0 = A[(int64_t)i * lda + j];      // (odd nb: the last column, alone)
        T[i * LDT + j] = x0;
        T[i * LDT + j + 1] = x1;
      }
    }
  }
  __syncthreads();
  TS(1);

  // (a) diagonal tile p: Cholesky in registers + inverse, by wave 0 alone (every 16-lane group computes the same thing)
  auto diag_tile = [&](int p) {
    if (DPP_BCAST) {
      // Round 4b: broadcasts by DPP row_newbcast instead of LDS.  Lane i of each 16-lane row holds row i of the tile in
      // registers from start to end; what a pivot step needs from other lanes -- the pivot itself (lane k), column k of L
      // (lane j's L_jk for the update of a_j) and row k of L (lane k's L_kj for the substitution of X = L^-1) -- are all
      // "the value lane K holds": one v_mov_b64_dpp, where the LDS round trip cost ~130 cycles.  The tile
      // and its inverse are written to LDS once, at the end.  Critical chain per pivot: L_{k+1,k} -> next pivot ->
      // v_rsq_f64 + two Newton steps; everything else fills its shadow.
      const int c0 = p * MB;
      const int li = lane & 15;
      double* Tt = T + c0 * LDT + c0;
      double a[MB], x[MB];
#pragma unroll
      for (int j = 0; j < MB; ++j) a[j] = (j <= li) ? Tt[li * LDT + j] : 0.0;
      int badk = -1;
      double minabs = INFINITY;
      double dk = row_bcast(a[0], 0);
      double sg = 1.0;
      if (SIGNED) { sg = (dk < 0.0) ? -1.0 : 1.0; dk = fabs(dk); }
      double inv = __builtin_amdgcn_rsq(dk);
      inv = inv * fma(-0.5 * dk, inv * inv, 1.5);
      if (NEWTON2) inv = inv * fma(-0.5 * dk, inv * inv, 1.5);
#pragma unroll
      for (int k = 0; k < MB; ++k) {
        badk = (!(dk > 0.0) && badk < 0) ? k : badk;
        if (SIGNED) {
          minabs = fmin(minabs, (c0 + k < nb) ? dk : INFINITY);
          if (lane == 0) Sg[c0 + k] = sg;
        }
        const double sq = dk * inv;
        const double lsg = SIGNED ? inv * sg : inv;
        const double lik = (li == k) ? sq : ((li > k) ? a[k] * lsg : 0.0);
        a[k] = lik;
        const double nlik = SIGNED ? -lik * sg : -lik;
        double dk1 = 1.0, inv1 = 1.0, sg1 = 1.0;
        if (k + 1 < MB) {                                   // the next pivot first: it heads the longest chain
          a[k + 1] = fma(nlik, row_bcast(lik, k + 1), a[k + 1]);
          dk1 = row_bcast(a[k + 1], k + 1);
          if (SIGNED) { sg1 = (dk1 < 0.0) ? -1.0 : 1.0; dk1 = fabs(dk1); }
          inv1 = __builtin_amdgcn_rsq(dk1);
          inv1 = inv1 * fma(-0.5 * dk1, inv1 * inv1, 1.5);
          if (NEWTON2) inv1 = inv1 * fma(-0.5 * dk1, inv1 * inv1, 1.5);
        }
#pragma unroll
        for (int j = k + 2; j < MB; ++j) a[j] = fma(nlik, row_bcast(lik, j), a[j]);
        // column li of X = L^-1: x_k = (delta_{k,li} - sum_{j<k} L_kj x_j) / L_kk, row k of L from lane k
        double s0 = (k == li) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j) {
          const double lkj = row_bcast(a[j], k);
          if ((j & 3) == 0) s0 = fma(-lkj, x[j], s0);
          else if ((j & 3) == 1) s1 = fma(-lkj, x[j], s1);
          else if ((j & 3) == 2) s2 = fma(-lkj, x[j], s2);
          else s3 = fma(-lkj, x[j], s3);
        }
        x[k] = ((s0 + s1) + (s2 + s3)) * inv;
        dk = dk1; inv = inv1; sg = sg1;
      }
      if (badk >= 0 && lane == 0) { atomicCAS(info, 0, (int)(j0 + c0 + badk + 1)); bad_s = 1; }
      if (SIGNED && lane == 0 && c0 < nb) {
        int real_negs = 0;
        for (int k = 0; k < MB; ++k) real_negs += (c0 + k < nb && Sg[c0 + k] < 0.0) ? 1 : 0;
        if (real_negs) atomicAdd(n_neg, real_negs);
        atomicMin(min_piv, (unsigned long long)__double_as_longlong(minabs));
      }
      if (lane < MB) {
#pragma unroll
        for (int j = 0; j < MB; ++j) {
          Tt[li * LDT + j] = (j <= li) ? a[j] : 0.0;
          Xd[(p * MB + j) * LDX + li] = x[j];    // X[j][li]
        }
      }
      return;
    }
    // Broadcasts go through LDS, not through v_readlane: with ~250 lane broadcasts per tile the scalar registers they
    // land in ran out (the compiler spilled 253 of them into VGPR lanes, 822 readlanes in all) and the tile took 11 000
    // cycles.  Here lane i writes L_ik into the tile the moment it is final, and everybody reads column k / row k back
    // with wave-uniform addresses (an LDS broadcast).  One wave: its LDS operations complete in order, no barrier.
    const int c0 = p * MB;
    const int li = lane & 15;
    double* Tt = T + c0 * LDT + c0;                    // Tt[i * LDT + j]
    double a[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) a[j] = (j <= li) ? Tt[li * LDT + j] : 0.0;
    double x[MB];      // column li of X = L^-1: x_k = (delta_{k,li} - sum_{j<k} L_kj x_j) / L_kk
    int badk = -1;
    // Software pipeline over the pivots: the next pivot's diagonal is a[k+1] - L_{k+1,k}^2 on lane k + 1 -- its OWN
    // values, no broadcast -- so its reciprocal square root (the longest dependent chain of a pivot) is computed while
    // column k travels through LDS to the other lanes.
    // (tried: X by row operations on [L | I] -- k + 1 independent FMAs per pivot instead of the substitution's dependent
    //  chain, but divergent over lanes and 270 more LDS operations: 14 500 instead of 9 100 cycles per tile)
    double dk = bcast_lane(a[0], 0);
    double sg = 1.0;                                     // sign of the pivot (SIGNED; wave-uniform)
    double minabs = INFINITY;
    if (SIGNED) { sg = (dk < 0.0) ? -1.0 : 1.0; dk = fabs(dk); }
    double inv = __builtin_amdgcn_rsq(dk);
    inv = inv * fma(-0.5 * dk, inv * inv, 1.5);
    if (NEWTON2) inv = inv * fma(-0.5 * dk, inv * inv, 1.5);
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      badk = (!(dk > 0.0) && badk < 0) ? k : badk;        // (reported after the loop: no branch inside the recurrence)
      if (SIGNED) {
        minabs = fmin(minabs, (c0 + k < nb) ? dk : INFINITY);      // (identity padding past nb does not count)
        if (lane == 0) Sg[c0 + k] = sg;
      }
      const double sq = dk * inv;                        // sqrt(dk) to ~1 ulp
      const double lsg = SIGNED ? inv * sg : inv;
      const double lik = (li == k) ? sq : ((li > k) ? a[k] * lsg : 0.0);
      a[k] = lik;
      if (lane < MB && li >= k) Tt[li * LDT + k] = lik;  // column k of L, final (same wave: later reads see it)
      double lcol[MB];
#pragma unroll
      for (int j = k + 1; j < MB; ++j) lcol[j] = Tt[j * LDT + k];
      // row k of L (final) for x_k; four partial sums keep the dependent chain short
      double s0 = (k == li) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int j = 0; j < k; ++j) {
        const double lkj = Tt[k * LDT + j];
        if ((j & 3) == 0) s0 = fma(-lkj, x[j], s0);
        else if ((j & 3) == 1) s1 = fma(-lkj, x[j], s1);
        else if ((j & 3) == 2) s2 = fma(-lkj, x[j], s2);
        else s3 = fma(-lkj, x[j], s3);
      }
      x[k] = ((s0 + s1) + (s2 + s3)) * inv;
      // (pinning x[k] here with an empty asm, so that the substitution runs under column k's LDS round trip instead of
      //  after the loop, made every pivot slower by more than the tail it removed: 10 300 instead of 9 250 cycles per tile)
      const double nlik = SIGNED ? -lik * sg : -lik;     // update coefficient: a_j -= L_ik s_k L_jk
      double dk1 = 1.0, inv1 = 1.0, sg1 = 1.0;
      if (k + 1 < MB) {
        dk1 = bcast_lane(fma(nlik, lik, a[k + 1]), k + 1);
        if (SIGNED) { sg1 = (dk1 < 0.0) ? -1.0 : 1.0; dk1 = fabs(dk1); }
        inv1 = __builtin_amdgcn_rsq(dk1);
        inv1 = inv1 * fma(-0.5 * dk1, inv1 * inv1, 1.5);
        if (NEWTON2) inv1 = inv1 * fma(-0.5 * dk1, inv1 * inv1, 1.5);
      }
#pragma unroll
      for (int j = k + 1; j < MB; ++j) a[j] = fma(nlik, lcol[j], a[j]);
      dk = dk1; inv = inv1; sg = sg1;
    }
    if (badk >= 0 && lane == 0) { atomicCAS(info, 0, (int)(j0 + c0 + badk + 1)); bad_s = 1; }   // first non-positive / NaN pivot
    if (SIGNED && lane == 0 && c0 < nb) {
      // (tiles past nb are identity padding: their pivots are 1 and do not count; a partly padded tile counts its real rows)
      int real_negs = 0;
      for (int k = 0; k < MB; ++k) real_negs += (c0 + k < nb && Sg[c0 + k] < 0.0) ? 1 : 0;
      if (real_negs) atomicAdd(n_neg, real_negs);
      atomicMin(min_piv, (unsigned long long)__double_as_longlong(minabs));
    }
    if (lane < MB) {
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        if (j > li) Tt[li * LDT + j] = 0.0;
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
    {
      const double sk = SIGNED ? -Sg[c0 + 4 * s + (lane >> 4)] : -1.0;          // A_ij -= L_ik s_k L_jk
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sk * frag(T, LDT, r0, c0, s, lane), frag(T, LDT, q0, c0, s, lane), acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) T[(r0 + (lane >> 4) + 4 * r) * LDT + q0 + (lane & 15)] = acc[r];
  };

  if (wave == 0) diag_tile(0);
  __syncthreads();
  TS(2);
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
      const double sc = SIGNED ? Sg[c0 + (lane & 15)] : 1.0;                    // L_ip = A_ip X_pp^T S_p
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(r0 + (lane >> 4) + 4 * r) * LDT + c0 + (lane & 15)] = SIGNED ? sc * acc[r] : acc[r];
    }
    __syncthreads();
    TS(3 + 2 * p);
    if (p + 1 == NMB) break;
    // ---- (c) trailing tiles, with look-ahead: wave 0 updates the NEXT diagonal tile first and factors it right away
    //      (the serial 16-step recurrence of (a): ~3 us) while waves 1-3 update the other tiles ----------------------
    {
      const int nt = NMB - 1 - p;               // tiles per side
      const int ntri = nt * (nt + 1) / 2;
      if (wave == 0) {
        if (p == 3) TS(23);
        trailing_tile(p, 0);                    // tile (p + 1, p + 1); LDS operations of one wave complete in order
        diag_tile(p + 1);
        if (p == 3) TS(26);
      } else {
        for (int t = wave; t < ntri; t += 3) trailing_tile(p, t);
      }
    }
    __syncthreads();
    TS(4 + 2 * p);
    if (bad_s) return;
  }
  TS(20);

  // ---- factor back to global memory (strict upper part of the block zeroed) ------------------------------------
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 7, j = e & 127;
    if (i < nb && j < nb) A[(int64_t)i * lda + j] = (j <= i) ? T[i * LDT + j] : 0.0;
  }
  __syncthreads(); TS(21);
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
        for (int r = 0; r < 4; ++r) {
          Dinv[(i * MB + (lane >> 4) + 4 * r) * NB + j * MB + (lane & 15)] = X[i][r];
          if (SIGNED) DinvS[(i * MB + (lane >> 4) + 4 * r) * NB + j * MB + (lane & 15)] = Sg[i * MB + (lane >> 4) + 4 * r] * X[i][r];
        }
      }
    }
  }
  __syncthreads(); TS(22);
}

}  // namespace

// Factorises the nb x nb (nb <= 128) lower block at A in place and writes T^-1 to Dinv (128 x 128, row-major, leading
// dimension 128; entries above the diagonal are never written: the caller zeroes the buffer once).
int launch_potrf128(mln_ctx* ctx, double* A, int64_t lda, int nb, double* Dinv, int* info, int64_t j0, int nbatch, int64_t a_bs) {
  const size_t lds = sizeof(double) * (size_t)(NB * LDT + NMB * MB * LDX + NB);
  static bool configured = false;
  if (!configured) {
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
  hipLaunchKernelGGL(k_potrf128<false>, dim3((unsigned)(nbatch > 1 ? nbatch : 1)), dim3(256), lds, ctx->stream, A, lda, nb, Dinv, info, j0,
                     (double*)nullptr, (int*)nullptr, (unsigned long long*)nullptr, a_bs);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

// The signed variant (see the kernel): A = T S T^T on the nb x nb block; DinvS = S T^-1; *n_neg += negative pivots;
// *min_piv = min(*min_piv, bits of the smallest |pivot|).
int launch_potrf128_signed(mln_ctx* ctx, double* A, int64_t lda, int nb, double* Dinv, double* DinvS, int* info, int64_t j0,
                           int* n_neg, unsigned long long* min_piv) {
  const size_t lds = sizeof(double) * (size_t)(NB * LDT + NMB * MB * LDX + NB);
  static bool configured = false;
  if (!configured) {
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
  hipLaunchKernelGGL(k_potrf128<true>, dim3(1), dim3(256), lds, ctx->stream, A, lda, nb, Dinv, info, j0, DinvS, n_neg, min_piv, (int64_t)0);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

#ifdef MLN_POTRF_TIMING
extern "C" int mln_diag_potrf_times(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_potrf_ts), sizeof(long long) * 64) == hipSuccess ? 0 : 1; }
#endif
