// The folded fp16-split row-minimum sweep with ONE wave per SIMD (round 6): 4 waves x 64 query rows per workgroup.
// Same contract, operands and products as k_rowmin_f16x3<TOP2, true> (rowmin_f16.hip: the pre-filter of the exact 1-NN search,
// parameters.py:352-433, and the assignment step of k-means, parameters.py:243-291) -- per accumulator the same MFMAs in the
// same order, hence the same approximate values, certified rows, runner-up values and labels.
//
// Why another shape.  The counters of the 8 waves x 32 rows kernel (profiles/r05_rowmin_pmc.txt) had the fp16 matrix cores busy
// 0.476 of the sweep: the two waves of a SIMD pass the same barrier per stage, so their MFMA and epilogue phases coincide
// instead of interleaving, and at 256 registers per lane a wave has no room to overlap the two phases itself.  Here
//   * a wave owns 64 rows (two 32-row halves) and the whole 512-register file of its SIMD lanes;
//   * each B fragment (32 candidates x 64 k x {hi, lo}) is read from LDS once for BOTH row halves: a quarter of the LDS reads
//     per matrix instruction;
//   * the unit of work is a SUB-TILE (64 rows x 32 candidates = 24 MFMAs, the two row halves' accumulator chains alternating):
//     while the matrix pipe works on sub-tile t, the vector pipe runs the min / median epilogue of sub-tile t - 1 on the other
//     accumulator pair, the LDS reads of the next B fragment and the LDS stores of the next stage -- placed BY HAND between the
//     matrix instructions (scheduling fences
//     after every slot): left to the scheduler the compares piled up in scalar registers (200 v_writelane / v_readlane
//     spills per stage) and the accumulators went to AGPRs (one v_accvgpr_read per element);
//   * the stage body has no branch: a stage that holds a padded candidate or meets the wave's diagonal band (the excluded
//     pair) takes the masked body instead -- at most three stages per wave.
// Measured and taken out (round 6): the same interleave with 32 rows per wave and TWO independent workgroups per CU (the
// micro-benchmark tools/ubench/f16_overlap.hip: one wave per SIMD reaches 0.86 of the fp16 MFMA rate alone and 0.72 with four
// vector instructions per MFMA in its stream, two waves reach the rate) -- matrix cores busy 0.60 instead of 0.56, but at
// 1.77 GHz instead of 2.06: the sweep is POWER-bound, and the shape with twice the LDS reads and 27 % more vector
// instructions lost what the second wave won (1-NN at C3: 0.423 s against 0.405 on the same box; round-5 kernel 0.441).
// Built with -mllvm -amdgpu-mfma-vgpr-form (mellon_amd/_build.py): the accumulators are architectural VGPRs, which the
// epilogue's vector instructions read directly.
#include <cmath>
#include <cstdlib>

#include "mln_internal.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int KP = 64;               // padded feature count (d <= 61 for folded operands)
constexpr int ROWH = 2 * KP;         // halves per split row: hi[0..63] | lo[0..63]
constexpr int RT = 256;              // candidates per stage (eight sub-tiles of 32)
constexpr int NSUB = RT / 32;
constexpr int PITCH = 272;           // LDS bytes per candidate row (256 + 16): conflict-free 16-byte reads

// one v_min_f32 (fminf() quiets signalling NaNs first -- a v_max x, x per operand; an MFMA only produces quiet ones)
__device__ __forceinline__ float vmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

#define MLN_FENCE() __builtin_amdgcn_sched_barrier(0)

// one step of the SMIN reduction: the lane keeps HALF of its 2 HALF values (which half: its bit HALF of lr) and takes the
// minimum with the partner's copies of the same elements
template <int HALF>
__device__ __forceinline__ void smin_exchange(float (&v)[32], int lr) {
  const bool up = (lr & HALF) != 0;
#pragma unroll
  for (int e = 0; e < HALF; ++e) {
    const float keep = up ? v[e + HALF] : v[e], send = up ? v[e] : v[e + HALF];
    v[e] = vmin_raw(keep, __shfl_xor(send, HALF, 64));
  }
}

// TOP2: the runner-up per row is tracked (1-NN certification) and the winner's column only per stage and lane (out_arg = the
// half-stage's first candidate of that lane; the winner is one of out_arg + {0, 32, 64, 96}); else (labels) the
// exact column.
// SMIN (k-means with group bounds, kmeans.hip): per row and STAGE the smallest value of the stage's 256 candidates,
// smin[stage * smin_stride + row] (a stage-local minimum, reduced over the wave's 32 column lanes and reset at the end of
// every stage) -- on its own (no winner at all) or together with TOP2.
template <bool TOP2, bool SMIN = false>
__global__ __launch_bounds__(256) void k_rowmin_w64(const _Float16* __restrict__ Xs, int64_t n,
                                                    const _Float16* __restrict__ Ys, int64_t m,
                                                    int64_t self_offset, int exclude_self,
                                                    float* __restrict__ out_m1, float* __restrict__ out_m2,
                                                    int* __restrict__ out_arg, const int* __restrict__ row_idx,
                                                    const uint32_t* __restrict__ stage_mask, int mask_words,
                                                    const int* __restrict__ wg_order, const int* __restrict__ n_dev,
                                                    float* __restrict__ smin, int64_t smin_stride) {
  extern __shared__ unsigned char lds[];                      // 2 x (RT x PITCH) candidate rows
  // n_dev (k-means sweeps queued ahead of the host): the number of query rows lives on the device, n only sized the grid
  if (n_dev) {
    n = *n_dev;
    if ((int64_t)blockIdx.x * 256 >= n) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lg = lane >> 5;
  // wg_order (with a stage mask): the row block this workgroup takes -- the blocks with the most candidate stages first, so
  // that a block that must sweep everything does not start in the last round
  const int64_t blk = wg_order ? (int64_t)wg_order[blockIdx.x] : (int64_t)blockIdx.x;
  const int64_t row0w = blk * 256 + wave * 64;
  const int64_t d0 = row0w + self_offset;            // the excluded candidates of this wave's rows are [d0, d0 + 64)
  h8 ahi[2][4], alo[2][4];
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) {
    const int64_t r = row0w + rh * 32 + lr;
    const int64_t ar0 = (r < n) ? r : n - 1;
    const int64_t ar = row_idx ? (int64_t)row_idx[ar0] : ar0;          // (k-means with bounds: the rows still open)
    const _Float16* src = Xs + ar * ROWH + 8 * lg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ahi[rh][ks] = *reinterpret_cast<const h8*>(src + 16 * ks);
      alo[rh][ks] = *reinterpret_cast<const h8*>(src + KP + 16 * ks);
    }
  }
  float m1[2][16], m2[2][16];
  int a1[2][16];
#pragma unroll
  for (int rh = 0; rh < 2; ++rh)
#pragma unroll
    for (int r = 0; r < 16; ++r) { m1[rh][r] = INFINITY; m2[rh][r] = INFINITY; a1[rh][r] = 0; }
  float ms[2][16];                     // TOP2 && SMIN: the minima of the stage in flight (SMIN alone: m1 is that)
#pragma unroll
  for (int rh = 0; rh < 2; ++rh)
#pragma unroll
    for (int r = 0; r < 16; ++r) ms[rh][r] = INFINITY;

  // staging: a stage is 256 rows x 256 B = two halves of 2048 16-byte pieces, 8 per thread and half (the second half is
  // requested when the first has gone to LDS: 32 staging registers instead of 64)
  v4i st[8];
  auto g_load = [&](int64_t col0, int half) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int piece = tid + 256 * q, r = piece >> 4, seg = piece & 15;
      int64_t c = col0 + 128 * half + r;
      c = (c < m) ? c : m - 1;                      // (unconditional: a padded row re-reads the last candidate and is masked)
      st[q] = *reinterpret_cast<const v4i*>(Ys + c * ROWH + seg * 8);
    }
  };
  auto l_store1 = [&](int buf, int half, int q) {
    const int piece = tid + 256 * q, r = 128 * half + (piece >> 4), seg = piece & 15;
    *reinterpret_cast<v4i*>(lds + buf * (RT * PITCH) + r * PITCH + seg * 16) = st[q];
  };
  struct BFrag { h8 hi[4], lo[4]; };
  // piece i of 8 of the B fragment of sub-tile `sub`
  auto read_b1 = [&](BFrag& b, const unsigned char* base, int sub, int i) {
    const unsigned char* brow = base + (sub * 32 + lr) * PITCH + 16 * lg;
    if (i < 4) b.hi[i] = *reinterpret_cast<const h8*>(brow + 32 * i);
    else b.lo[i - 4] = *reinterpret_cast<const h8*>(brow + 2 * KP + 32 * (i - 4));
  };
  // MFMA i of 12 of one half-tile: the small cross terms first (hi.lo, lo.hi per k-step), the hi.hi terms on top -- the order of
  // k_rowmin_f16x3: the fp32 roundings that matter are those of the last four
  auto mfma1 = [&](f16v& acc, int rh, const BFrag& b, int i) {
    if (i == 0) {                                      // the chain starts from the inline constant 0: no accumulator clear
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[rh][0], b.lo[0], f16v{}, 0, 0, 0);
    } else if (i < 8) {
      const int ks = i >> 1;
      if ((i & 1) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[rh][ks], b.lo[ks], acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[rh][ks], b.hi[ks], acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[rh][i - 8], b.hi[i - 8], acc, 0, 0, 0);
    }
  };
  // epilogue of element r of a finished sub-tile (row half rh, candidate column c0 + 32 sub of this lane).  masked: the
  // candidate may be padding (32 sub >= mrem) or the row's excluded one (dlt[rh] + 32 sub == (r & 3) + 8 (r >> 2)): 32-bit
  // quantities prepared once per stage (mask_prepare)
  int mrem = 0, dlt[2] = {0, 0};
  auto mask_prepare = [&](int64_t col0) {
    const int64_t left = m - col0 - lr;                                       // candidates from this lane's first column on
    mrem = left > (1 << 20) ? (1 << 20) : (left < 0 ? 0 : (int)left);
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      // the excluded candidate of row (r & 3) + 8 (r >> 2) + 4 lg of this half is d0 + 32 rh + that
      int64_t t = col0 + lr - d0 - 32 * rh - 4 * lg;
      t = t > (1 << 20) ? (1 << 20) : (t < -(1 << 20) ? -(1 << 20) : t);
      dlt[rh] = exclude_self ? (int)t : -(1 << 20);
    }
  };
  auto epi1 = [&](const f16v& acc, int rh, int r, int sub, int c0, bool masked) {
    float sv = acc[r];
    if (masked) sv = (32 * sub < mrem && dlt[rh] + 32 * sub != (r & 3) + 8 * (r >> 2)) ? sv : INFINITY;
    if (TOP2) {
      m2[rh][r] = __builtin_amdgcn_fmed3f(m1[rh][r], m2[rh][r], sv);
    } else if (!SMIN) {
      a1[rh][r] = (sv < m1[rh][r]) ? c0 + 32 * sub : a1[rh][r];
    }
    m1[rh][r] = vmin_raw(m1[rh][r], sv);
    if (TOP2 && SMIN) ms[rh][r] = vmin_raw(ms[rh][r], sv);
  };

  // Which stages (blocks of RT candidates) this workgroup multiplies: all of them, or -- stage_mask -- the set bits of this
  // workgroup's row of a bit matrix (the pruned 1-NN search: candidate blocks that a triangle-inequality bound cannot
  // exclude, rowmin_f16.hip nn_distances_pruned).  Wave-uniform scalar work.
  // (the workgroup's mask row is copied to LDS first: a global load per stage on the critical path cost a quarter of the sweep)
  uint32_t* mrow = stage_mask ? reinterpret_cast<uint32_t*>(lds + 2 * RT * PITCH) : nullptr;
  if (mrow) {
    for (int w = tid; w < mask_words; w += 256) mrow[w] = stage_mask[blk * mask_words + w];
    __syncthreads();
  }
  auto next_stage = [&](int64_t from_stage) -> int64_t {      // first selected stage >= from_stage, as a column; m when none
    if (!mrow) return from_stage * RT;
    int w = (int)(from_stage >> 5);
    if (w >= mask_words) return m;
    uint32_t bits = __builtin_amdgcn_readfirstlane(mrow[w]) & (0xffffffffu << (from_stage & 31));
    while (bits == 0 && ++w < mask_words) bits = __builtin_amdgcn_readfirstlane(mrow[w]);
    if (bits == 0) return m;
    const int64_t c = ((int64_t)w * 32 + __builtin_ctz(bits)) * RT;
    return c < m ? c : m;
  };
  int64_t col0 = next_stage(0);
  if (col0 < m) {
    g_load(col0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) l_store1(0, 0, q);
    g_load(col0, 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) l_store1(0, 1, q);
  }
  __syncthreads();
  int buf = 0;
  for (; col0 < m; buf ^= 1) {
    const int64_t following = next_stage(col0 / RT + 1);
    const int64_t ncol0 = following < m ? following : col0;   // (no branch around the requests; past the end: a re-read nobody multiplies)
    g_load(ncol0, 0);
    // The loop-invariant A operands and the staging registers belong in AGPRs (MFMA sources, global loads and LDS stores take
    // them directly); the 256 architectural VGPRs are for what the vector epilogue touches (accumulators, minima, columns).
    // Left alone the allocator shuffles ~200 values per stage between the two files (v_accvgpr_read / _write / _mov).
#pragma unroll
    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { asm volatile("" : "+a"(ahi[rh][ks])); asm volatile("" : "+a"(alo[rh][ks])); }
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+a"(st[q]));
    const unsigned char* base = lds + buf * (RT * PITCH);
    const int nbuf = buf ^ 1;
    const bool plain = (col0 + RT <= m) && !(exclude_self && col0 < d0 + 64 && col0 + RT > d0);
    const int c0 = (int)col0 + lr;
    BFrag b0, b1;
    f16v accA[2], accB[2];                           // the two row halves of the sub-tile in flight / of the one before
    float m1_in[2][16];                              // TOP2: the minima the stage began with (which lane improved, and in which stage)
    if (TOP2) {
#pragma unroll
      for (int rh = 0; rh < 2; ++rh)
#pragma unroll
        for (int r = 0; r < 16; ++r) m1_in[rh][r] = m1[rh][r];
    }
    // One step = one sub-tile (32 candidates x 64 rows): 24 slots of [one matrix instruction | a few vector / LDS
    // instructions].  The two row halves' accumulator chains ALTERNATE: a chain of dependent 32 x 32 MFMAs runs at their
    // 64-cycle latency, two independent chains fill the 32-cycle issue rate.
    //   epilogue: the 32 elements of the finished sub-tile subE (accE[2]) over slots 0-23 (none if subE < 0)
    //   rd: the 8 LDS reads of the B fragment `bR` of sub-tile rd in slots 0-7 (none if rd < 0)
    //   WRH: the 8 LDS stores of half WRH of the NEXT stage in slots 12-19 (none if WRH < 0)
#define MLN_STEP(accW, bW, accE, subE, bR, rd, WRH, MASKED)                                                        \
    _Pragma("unroll")                                                                                              \
    for (int i = 0; i < 24; ++i) {                                                                                 \
      mfma1(accW[i & 1], i & 1, bW, i >> 1);                                                                       \
      if (rd >= 0 && i < 8) read_b1(bR, base, rd, i);                                                              \
      if (subE >= 0) {                                                                                             \
        _Pragma("unroll")                                                                                          \
        for (int e = (i * 4) / 3; e < ((i + 1) * 4) / 3; ++e)                                                      \
          epi1(accE[e >> 4], e >> 4, e & 15, subE < 0 ? 0 : subE, c0, MASKED); \
      }                                                                                                            \
      if (WRH >= 0 && i >= 12 && i < 20) l_store1(nbuf, WRH < 0 ? 0 : WRH, i - 12);                                \
      MLN_FENCE();                                                                                                 \
    }
    // TOP2: the winner's column is recorded per HALF stage (128 candidates) and lane: after the epilogue of sub-tile 3 (in the
    // step of sub-tile 4) and at the end of the stage -- the winner is one of arg + {0, 32, 64, 96}, four exact evaluations per
    // row in k_nn_certify / k_km_resolve.  Measured at C3 (1-NN sweep + certification | k-means sweeps + resolve, ms): per whole
    // stage (eight candidates) 41 + 14 | 202 + 90; per half stage, between the steps as here 50 + 8 | 227 + 52; per half stage with
    // the updates placed inside the steps' slots 55 + 8 | 244 + 52 (five vector instructions per MFMA: past the issue knee).
#define MLN_HALF_STAGE_ARG(REC)                                                                                    \
    if (TOP2) {                                                                                                    \
      _Pragma("unroll") for (int rh = 0; rh < 2; ++rh)                                                             \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                             \
        a1[rh][r] = (m1[rh][r] < m1_in[rh][r]) ? (REC) : a1[rh][r];                                                \
        m1_in[rh][r] = m1[rh][r];                                                                                  \
      }                                                                                                            \
    }
#define MLN_STAGE(MASKED)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) read_b1(b0, base, 0, i);                                         \
    MLN_FENCE();                                                                                                   \
    MLN_STEP(accA, b0, accB, -1, b1, 1, -1, MASKED)                                                                \
    MLN_STEP(accB, b1, accA, 0, b0, 2, -1, MASKED)                                                                 \
    MLN_STEP(accA, b0, accB, 1, b1, 3, -1, MASKED)                                                                 \
    MLN_STEP(accB, b1, accA, 2, b0, 4, 0, MASKED)                                                                  \
    g_load(ncol0, 1);                                                                                              \
    MLN_FENCE();                                                                                                   \
    MLN_STEP(accA, b0, accB, 3, b1, 5, -1, MASKED)                                                                 \
    MLN_HALF_STAGE_ARG(c0)                                                                                         \
    MLN_STEP(accB, b1, accA, 4, b0, 6, -1, MASKED)                                                                 \
    MLN_STEP(accA, b0, accB, 5, b1, 7, -1, MASKED)                                                                 \
    MLN_STEP(accB, b1, accA, 6, b0, -1, 1, MASKED)                                                                 \
    _Pragma("unroll") for (int e = 0; e < 32; ++e) epi1(accB[e >> 4], e >> 4, e & 15, 7, c0, MASKED);
    if (plain) { MLN_STAGE(false) } else { mask_prepare(col0); MLN_STAGE(true) }
    MLN_HALF_STAGE_ARG(c0 + 128)
#undef MLN_STAGE
#undef MLN_STEP
#undef MLN_HALF_STAGE_ARG
    if (SMIN) {
      // 32 values per lane (one per row of its half) x 32 column lanes -> lane lr keeps the minimum of element lr: five exchange
      // steps in which a lane hands over the half of its values the partner keeps (31 shuffles instead of 160)
      float v[32];
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        if (TOP2) { v[e] = ms[e >> 4][e & 15]; ms[e >> 4][e & 15] = INFINITY; }
        else { v[e] = m1[e >> 4][e & 15]; m1[e >> 4][e & 15] = INFINITY; }
      }
      smin_exchange<16>(v, lr); smin_exchange<8>(v, lr); smin_exchange<4>(v, lr); smin_exchange<2>(v, lr); smin_exchange<1>(v, lr);
      const int64_t row = row0w + (lr >> 4) * 32 + (lr & 3) + 8 * ((lr & 15) >> 2) + 4 * lg;
      if (row < n) smin[(col0 / RT) * smin_stride + row] = v[0];
    }
    __syncthreads();
    col0 = following;
  }
  if (SMIN && !TOP2) return;
  // merge the 32 column-lanes of each row (lanes with the same lg hold the same rows)
#pragma unroll
  for (int rh = 0; rh < 2; ++rh)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) {
        const float o1 = __shfl_xor(m1[rh][r], off, 64), o2 = __shfl_xor(m2[rh][r], off, 64);
        const int oa = __shfl_xor(a1[rh][r], off, 64);
        if (TOP2) m2[rh][r] = fminf(fmaxf(m1[rh][r], o1), fminf(m2[rh][r], o2));
        // ties go to the smaller candidate index: the result does not depend on the lane order of the merge
        const bool take = (o1 < m1[rh][r]) || (o1 == m1[rh][r] && oa < a1[rh][r]);
        a1[rh][r] = take ? oa : a1[rh][r];
        m1[rh][r] = fminf(m1[rh][r], o1);
      }
      const int64_t row = row0w + rh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
      if (lr == 0 && row < n) {
        out_m1[row] = m1[rh][r];
        if (TOP2) out_m2[row] = m2[rh][r];
        out_arg[row] = a1[rh][r];
      }
    }
}

}  // namespace

// stage_mask (optional, device): a bit matrix, one row of mask_words 32-bit words per 256-row workgroup; bit s of a row selects
// the candidate block [256 s, 256 s + 256)
int launch_rowmin_w64(mln_ctx* ctx, const _Float16* X, int64_t n, const _Float16* Y, int64_t m, int64_t self_offset, int exclude_self,
                      float* m1, float* m2, int* arg, const int* row_idx, const uint32_t* stage_mask, int mask_words, const int* wg_order,
                      const int* n_dev, float* smin, int64_t smin_stride) {
  if (stage_mask && mask_words > 4096) { mln_set_error(ctx, "rowmin: stage mask too wide for LDS"); return MLN_ERR_UNSUPPORTED; }
  const size_t lds_max = (size_t)2 * RT * PITCH + 4096 * sizeof(uint32_t);
  const size_t lds_bytes = (size_t)2 * RT * PITCH + (stage_mask ? (size_t)mask_words * sizeof(uint32_t) : 0);
  static bool attr = false;
  if (!attr) {
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowmin_w64<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowmin_w64<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowmin_w64<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowmin_w64<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    attr = true;
  }
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (smin && m2)
    hipLaunchKernelGGL((k_rowmin_w64<true, true>), grid, block, lds_bytes, ctx->stream, X, n, Y, m, self_offset, exclude_self, m1, m2, arg, row_idx,
                       stage_mask, mask_words, wg_order, n_dev, smin, smin_stride);
  else if (smin)
    hipLaunchKernelGGL((k_rowmin_w64<false, true>), grid, block, lds_bytes, ctx->stream, X, n, Y, m, self_offset, exclude_self, (float*)nullptr,
                       (float*)nullptr, (int*)nullptr, row_idx, stage_mask, mask_words, wg_order, n_dev, smin, smin_stride);
  else if (m2)
    hipLaunchKernelGGL((k_rowmin_w64<true>), grid, block, lds_bytes, ctx->stream, X, n, Y, m, self_offset, exclude_self, m1, m2, arg, row_idx,
                       stage_mask, mask_words, wg_order, n_dev, (float*)nullptr, (int64_t)0);
  else
    hipLaunchKernelGGL((k_rowmin_w64<false>), grid, block, lds_bytes, ctx->stream, X, n, Y, m, self_offset, exclude_self, m1, (float*)nullptr, arg,
                       row_idx, stage_mask, mask_words, wg_order, n_dev, (float*)nullptr, (int64_t)0);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
