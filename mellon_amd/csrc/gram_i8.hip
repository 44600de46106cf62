// The preconditioner's Gram S = K_s^T K_s on the int8 matrix cores (v_mfma_i32_32x32x32_i8), exact in integers.
//
// Reference: parameters.compute_initial_value / compute_initial_ridge_value (parameters.py:895-896) form L^T L of ALL
// cells; here the Gram only has to make a good preconditioner, and for that 20 fractional bits of K are
// enough (tools/gram_bits_sweep.py: 46-48 passes at 20...40 bits and unquantised, 2x the passes at 16).  Every
// covariance that takes this route is bounded by [0, 1] (same eligibility test as the fixed-point copy of K), so
//     q = round(K * 8355711) = d0 + 256 d1 + 65536 d2,   d_a in [-128, 127]   (8355711 = 127 (65536 + 256 + 1))
// holds 23 bits in three signed bytes, and
//     Q^T Q = sum_{a,b} 256^(a+b) D_a^T D_b
// is nine int8 GEMMs with int32 accumulation, one accumulator per weight a + b -- exact as long as
// 3 * 2^14 * k < 2^31, i.e. per k-chunk of at most 32768 sampled cells.  ALL nine products are kept, although those of
// weight <= 1 are below the quantisation step: the result must be the Gram of SOMETHING (here: of Q), because
// Lp^-1 . Lp^-T amplifies any symmetric perturbation that is not itself a Gram by up to 1e6 and the Ridge matrix
// Lp^-1 S Lp^-T + I then stops being positive definite (seen at C3 with the low products dropped).
// At ~4 Pop/s against 78 Tflop/s of fp64 MFMA the 1.5 Tflop fp64 GEMM of C3 (35.7 ms) becomes 13.5 Top (a few ms).
//
// Layout: the digit planes are stored TRANSPOSED (and tiled, see k_gram_digits), plane a = [landmark i][sampled cell k] with k contiguous, so that
// both operands of the Gram are "k-contiguous rows": a lane's MFMA operand (16 consecutive k of one landmark) is one
// 16-byte LDS read, and A and B use the same lane -> k map, so the k order inside the instruction does not matter.
#include "mln_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

constexpr int GQ_SCALE = 8355711;
constexpr int GT = 128;          // output tile (landmarks x landmarks) per workgroup
constexpr int GBK = 64;          // sampled cells per stage
constexpr int GROW = GBK + 16;   // LDS row pitch in bytes: 16-byte aligned, 20 banks -> conflict-free b128 reads
constexpr int GSTAGE = 2 * 3 * GT * GROW;   // bytes per stage: two operands x three planes

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// K_s (rows x m doubles, row pitch ldk) -> planes[a][i][k] (int8), zero padded to Mp x Kp
__global__ __launch_bounds__(256) void k_gram_digits(const double* __restrict__ K, int64_t ldk, int64_t rows, int64_t m,
                                                     int8_t* __restrict__ planes, int64_t Mp, int64_t Kp) {
  __shared__ uint32_t tile[3][64][17];
  const int t = threadIdx.x, li = t & 63, kq = t >> 6;
  const int64_t k0 = (int64_t)blockIdx.x * 64, i0 = (int64_t)blockIdx.y * 64;
  const int64_t gi = i0 + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int g = kq * 4 + r;
    uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t k = k0 + g * 4 + e;
      double v = 0.0;
      if (k < rows && gi < m) v = K[k * ldk + gi];
      v = fmin(fmax(v, 0.0), 1.0);
      const int q = (int)rint(v * (double)GQ_SCALE);
      const int d0 = ((q + 128) & 255) - 128;
      const int q1 = (q - d0) >> 8;
      const int d1 = ((q1 + 128) & 255) - 128;
      const int d2 = (q1 - d1) >> 8;
      w0 |= (uint32_t)(d0 & 255) << (8 * e);
      w1 |= (uint32_t)(d1 & 255) << (8 * e);
      w2 |= (uint32_t)(d2 & 255) << (8 * e);
    }
    tile[0][li][g] = w0; tile[1][li][g] = w1; tile[2][li][g] = w2;
  }
  __syncthreads();
  const int oi = t >> 2, seg = t & 3;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    v4i_t o;
    o.x = (int)tile[a][oi][seg * 4 + 0]; o.y = (int)tile[a][oi][seg * 4 + 1];
    o.z = (int)tile[a][oi][seg * 4 + 2]; o.w = (int)tile[a][oi][seg * 4 + 3];
    // tiled layout: [plane][128-landmark tile][64-cell chunk][landmark in tile][64 bytes] -- what one stage of k_gram_i8
    // loads per operand and plane is ONE contiguous 8 KB block (full cache lines; 64-byte pieces of Kp-strided rows
    // made every line travel twice)
    const int64_t gi2 = i0 + oi;
    *reinterpret_cast<v4i_t*>(planes + (int64_t)a * Mp * Kp + (((gi2 >> 7) * (Kp >> 6) + (k0 >> 6)) * 128 + (gi2 & 127)) * 64 +
                              seg * 16) = o;
  }
}

struct GramTile { int ti, tj; };

// One sweep over the k-chunk for one 128 x 128 tile.  PASS 0: the six digit products of weight >= 2 (three int32
// accumulators per element); PASS 1: the three of weight <= 1 (two accumulators, planes 0 and 1 only).  Five
// accumulators at once would be 320 registers per lane for them alone -- the two sweeps cost the same MFMAs and keep
// every accumulator in place.  res (fp64) += sum_w 256^w acc_w.
// NW waves per workgroup: 4 (wave tile 64 x 64, one wave per SIMD, 2 MFMAs per operand read) or 8 (wave tile 64 x 32,
// two waves per SIMD hide each other's LDS latency and barrier waits, 1.3 MFMAs per operand read).
template <int PASS, int NW>
__device__ __forceinline__ void gram_sweep(unsigned char* lds, const int8_t* gA, const int8_t* gB, int64_t plane_sz, int64_t half,
                                           int n_steps, int ldst0, int fa, int fb, double (&res)[2][8 / NW][16]) {
  constexpr int NP = PASS == 0 ? 3 : 2;       // planes staged
  constexpr int NA = PASS == 0 ? 3 : 2;       // accumulators (weights 4, 3, 2 | 1, 0)
  constexpr int NBJ = 8 / NW;                 // 32-column blocks per wave
  constexpr int NRH = 512 / (64 * NW);        // row halves a thread stages per plane (64 NW / 4 rows per sweep of the threads)
  // global -> registers -> LDS, both operands of a stage at once, ONE FULL STAGE ahead: the loads of stage st + 2 are
  // issued at the top of stage st, right after the registers holding stage st + 1 went to LDS, and have the whole stage
  // (24-48 MFMAs per SIMD) to land.  (Half a stage ahead, the matrix pipe was busy 34 % of the time.)
  v4i_t stage[2][NRH * NP];
  auto g_load = [&](int st) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int j = 0; j < NRH * NP; ++j) {
        const int8_t* src = (o == 0 ? gA : gB) + (j / NRH) * plane_sz + (j % NRH) * half + (int64_t)st * (GT * GBK);
        stage[o][j] = *reinterpret_cast<const v4i_t*>(src);
      }
  };
  auto l_store = [&](int buf) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int j = 0; j < NRH * NP; ++j)
        *reinterpret_cast<v4i_t*>(lds + buf * GSTAGE + ldst0 + ((o * 3 + (j / NRH)) * GT + 64 * (j % NRH)) * GROW) = stage[o][j];
  };
  v16i_t acc[NA][2][NBJ];
#pragma unroll
  for (int w = 0; w < NA; ++w)
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < NBJ; ++bj) acc[w][bi][bj] = v16i_t{};

  // Operand fetches run ONE HALF-STAGE ahead of the matrix instructions that consume them (round 6): the 32 cells of
  // half-stage h + 1 are requested from LDS before the MFMAs of half-stage h are issued, so a wave's LDS latency is covered
  // by its own matrix work.  Before, every stage began with all its ds_reads and `s_waitcnt lgkmcnt(0)`, and the two waves
  // of a SIMD -- kept in step by the stage barrier -- sat out their LDS phases TOGETHER: the matrix cores were busy 0.53 of
  // the kernel with no bank conflict left (profiles/r06_gram_i8_pmc.txt).  The barrier now sits between the two
  // half-stages: it is reached with half a stage of MFMAs still in the pipe.
  struct Ops { v4i_t A[NP][2], B[NP][NBJ]; };
  auto read_ops = [&](Ops& o, int buf, int ks) {
    const unsigned char* base = lds + buf * GSTAGE;
#pragma unroll
    for (int a = 0; a < NP; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b) o.A[a][b] = *reinterpret_cast<const v4i_t*>(base + fa + (a * GT + b * 32) * GROW + ks * 32);
#pragma unroll
      for (int b = 0; b < NBJ; ++b) o.B[a][b] = *reinterpret_cast<const v4i_t*>(base + fb + (a * GT + b * 32) * GROW + ks * 32);
    }
  };
  // One row block bi: its digit products, ordered so that no two CONSECUTIVE matrix instructions write the same accumulator
  // when the two row blocks are issued alternately (mfmas2): a dependent 32 x 32 MFMA waits for its predecessor's full latency.
  auto mfma1 = [&](const Ops& o, int bi, int bj, int k) {
    if constexpr (PASS == 0) {
      if (k == 0) acc[0][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[2][bi], o.B[2][bj], acc[0][bi][bj], 0, 0, 0);
      if (k == 1) acc[1][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[1][bi], o.B[2][bj], acc[1][bi][bj], 0, 0, 0);
      if (k == 2) acc[2][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[0][bi], o.B[2][bj], acc[2][bi][bj], 0, 0, 0);
      if (k == 3) acc[1][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[2][bi], o.B[1][bj], acc[1][bi][bj], 0, 0, 0);
      if (k == 4) acc[2][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[1][bi], o.B[1][bj], acc[2][bi][bj], 0, 0, 0);
      if (k == 5) acc[2][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[2][bi], o.B[0][bj], acc[2][bi][bj], 0, 0, 0);
    } else {
      if (k == 0) acc[0][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[0][bi], o.B[1][bj], acc[0][bi][bj], 0, 0, 0);
      if (k == 1) acc[1][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[0][bi], o.B[0][bj], acc[1][bi][bj], 0, 0, 0);
      if (k == 2) acc[0][bi][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.A[1][bi], o.B[0][bj], acc[0][bi][bj], 0, 0, 0);
    }
  };
  // both row blocks, alternating: products k of bi = 0 and bi = 1 back to back
  constexpr int NK = PASS == 0 ? 6 : 3;
  auto mfmas2 = [&](const Ops& o, int k0, int k1) {
#pragma unroll
    for (int k = k0; k < k1; ++k)
#pragma unroll
      for (int bj = 0; bj < NBJ; ++bj) { mfma1(o, 0, bj, k); mfma1(o, 1, bj, k); }
  };
  // Stage st lives in LDS buffer st & 1; the registers `stage` hold the data of stage st + 2 while stage st is multiplied:
  //   request the second half-stage's operands | MFMAs of the first | BARRIER (buffer of stage st + 1 complete; every read
  //   of stage st's buffer has landed, so it is free) | request stage st + 1's first operands | MFMAs of the second
  //   half-stage with, in their shadow, stage st + 2 going from registers to the freed buffer and the request for st + 3.
  // The scheduling fences keep the compiler from sinking the first half-stage's MFMAs below the barrier (it did: matrix
  // instructions touch no memory), which would leave the barrier wait with an empty matrix pipe.
  g_load(0); l_store(0);
  if (n_steps > 1) g_load(1);
  lds_barrier();
  Ops o0, o1;
  read_ops(o0, 0, 0);
  if (n_steps > 1) l_store(1);
  if (n_steps > 2) g_load(2);
  for (int st = 0; st < n_steps; ++st) {
    const int buf = st & 1;
    read_ops(o1, buf, 1);
    mfmas2(o0, 0, NK);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // (no branch around the requests: at a join the compiler's wait-count bookkeeping gives up and waits for EVERYTHING
    //  before the next matrix instruction.  Past the last stage the reads fetch a buffer nobody multiplies, the stores put
    //  stale registers into a buffer nobody reads, and the global request re-reads the last stage.)
    read_ops(o0, buf ^ 1, 0);
    mfmas2(o1, 0, NK / 2);
    l_store(buf);
    g_load(st + 3 < n_steps ? st + 3 : n_steps - 1);
    mfmas2(o1, NK / 2, NK);
  }
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < NBJ; ++bj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (PASS == 0)
          res[bi][bj][r] = ((double)acc[2][bi][bj][r] * 65536.0 + (double)acc[1][bi][bj][r] * 16777216.0) +
                           (double)acc[0][bi][bj][r] * 4294967296.0;
        else
          res[bi][bj][r] += (double)acc[1][bi][bj][r] + (double)acc[0][bi][bj][r] * 256.0;
      }
}

// One 128 x 128 tile (ti >= tj) of one k-chunk: parts[split][i][j] = scale sum_w 256^w acc_w
template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void k_gram_i8(const int8_t* __restrict__ planes, int64_t Mp, int64_t Kp, int64_t kchunk,
                                                        const GramTile* __restrict__ tiles, int n_tiles, int n_wg, double scale,
                                                        double* __restrict__ parts, int64_t ldg, int64_t part_stride, int64_t m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs, so XCD x gets the contiguous range
  // [x n_wg / 8, (x + 1) n_wg / 8) of the logical order -- neighbouring tiles (shared landmark blocks), same k-chunk
  const int id = blockIdx.x;
  const int per = (n_wg + 7) / 8;
  const int logical = (id & 7) * per + (id >> 3);
  if (logical >= n_wg) return;
  const int split = logical / n_tiles;
  const GramTile tl = tiles[logical - split * n_tiles];
  constexpr int NBJ = 8 / NW, WJ = 64 / (32 * NBJ) * 1;   // column blocks per wave; waves per 64 columns
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wi = (NW == 4) ? (w >> 1) : (w >> 2), wj = (NW == 4) ? (w & 1) : (w & 3);
  const int64_t kbeg = (int64_t)split * kchunk;
  const int n_steps = (int)(kchunk / GBK);
  const int64_t plane_sz = Mp * Kp;
  // global -> register -> LDS staging: 16-byte chunks, one 64-bit address per operand, the rest wave-uniform
  // Row of the staged tile a thread moves: rows 0, 4 | 1, 5 | 2, 6 | 3, 7 of every 8-row block go to CONSECUTIVE groups of
  // four lanes.  A ds_write_b128 is serviced in groups of 8 contiguous lanes against 32 banks (128 B): with rows r, r + 1 in
  // a group (pitch 80 B) the second row's [80, 144) wraps onto the first row's first 16 bytes -- a 2-way conflict on every
  // store, 29 % of all LDS cycles (profiles/r05_gram_i8_pmc.txt); rows r, r + 4 sit 320 = 64 (mod 128) bytes apart and
  // tile the 128-byte bank row exactly.  The 32 lanes of a half-wave still cover 8 rows x 64 B = 512 contiguous bytes of
  // the tiled global layout (full cache lines), and the b128 operand READS (pitch 80 B: 16 distinct 16-byte slots per lane
  // group) are unchanged.
  const int seg = t & 3, q = t >> 2, r0 = (q & ~7) | ((q & 1) << 2) | ((q >> 1) & 3);
  const int8_t* gA = planes + (((int64_t)tl.ti * (Kp >> 6) + (kbeg >> 6)) * GT + r0) * GBK + seg * 16;   // tiled layout, see k_gram_digits
  const int8_t* gB = planes + (((int64_t)tl.tj * (Kp >> 6) + (kbeg >> 6)) * GT + r0) * GBK + seg * 16;
  const int64_t half = 64 * GBK;
  const int ldst0 = r0 * GROW + seg * 16;
  const int fa = ((wi * 64 + (lane & 31)) * GROW) + (lane >> 5) * 16;                             // operand A: rows of tile ti
  const int fb = (3 * GT * GROW) + ((wj * 32 * NBJ + (lane & 31)) * GROW) + (lane >> 5) * 16;     // operand B: rows of tile tj
  (void)WJ;

  double res[2][NBJ][16];
  gram_sweep<0, NW>(lds, gA, gB, plane_sz, half, n_steps, ldst0, fa, fb, res);
  gram_sweep<1, NW>(lds, gA, gB, plane_sz, half, n_steps, ldst0, fa, fb, res);

  double* out = parts + (int64_t)split * part_stride;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < NBJ; ++bj) {
      const int64_t gj = (int64_t)tl.tj * GT + wj * 32 * NBJ + bj * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gi = (int64_t)tl.ti * GT + wi * 64 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gi < m && gj < m) out[gi * ldg + gj] = scale * res[bi][bj][r];
      }
    }
}

}  // namespace

int gram_i8_splits(int64_t rows, int64_t m) {
  const int64_t T = (m + GT - 1) / GT, tiles = T * (T + 1) / 2;
  int smin = (int)((rows + 32767) / 32768);
  if (smin < 1) smin = 1;
  int best = smin;
  double best_eff = -1.0;
  for (int s = smin; s <= smin + 7; ++s) {
    if ((rows + s - 1) / s < 2048 && s > smin) break;   // keep the chunks long: the tile's output costs m^2 bytes per split
    const double wg = (double)tiles * s;
    const double eff = wg / (std::ceil(wg / 256.0) * 256.0);
    if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
  }
  return best;
}

// parts[s] (m x ldg each, lower 128-tiles written, the rest untouched) for s < n_splits: the caller sums them in order
int launch_gram_i8(mln_ctx* ctx, const double* K, int64_t ldk, int64_t rows, int64_t m, double alpha, double* parts,
                   int64_t ldg, int64_t part_stride, int n_splits) {
  if (rows <= 0 || m <= 0) return MLN_OK;
  const int64_t T = (m + GT - 1) / GT, Mp = T * GT;
  int64_t kchunk = (rows + n_splits - 1) / n_splits;
  kchunk = (kchunk + GBK - 1) / GBK * GBK;
  if (kchunk > 32768) { mln_set_error(ctx, "gram_i8: k-chunk exceeds the exact int32 range"); return MLN_ERR_ARG; }
  const int64_t Kp = kchunk * n_splits;
  int8_t* planes = nullptr;
  GramTile* d_tiles = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&planes, (size_t)3 * Mp * Kp));
  // tile order: strips of 8 tile rows, column by column inside a strip -> 32 consecutive tiles share ~12 landmark blocks
  std::vector<GramTile> order;
  order.reserve((size_t)(T * (T + 1) / 2));
  for (int64_t s0 = 0; s0 < T; s0 += 8)
    for (int64_t j = 0; j < std::min(T, s0 + 8); ++j)
      for (int64_t i = std::max(s0, j); i < std::min(T, s0 + 8); ++i) order.push_back(GramTile{(int)i, (int)j});
  const int n_tiles = (int)order.size();
  hipError_t e = mln_dmalloc((void**)&d_tiles, sizeof(GramTile) * order.size());
  if (e != hipSuccess) { (void)mln_dfree(planes); return mln_hip_fail(ctx, e, "alloc Gram tile list", __FILE__, __LINE__); }
  int rc = MLN_OK;
  e = hipMemcpyAsync(d_tiles, order.data(), sizeof(GramTile) * order.size(), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "upload Gram tile list", __FILE__, __LINE__);
  if (rc == MLN_OK) {
    hipLaunchKernelGGL(k_gram_digits, dim3((unsigned)(Kp / 64), (unsigned)(Mp / 64)), dim3(256), 0, ctx->stream, K, ldk, rows, m,
                       planes, Mp, Kp);
    static bool attr_set = false;
    const int lds_bytes = 2 * GSTAGE;
    if (!attr_set) {
      // (eight waves: wave tile 64 x 32, two waves per SIMD; the four-wave 64 x 64 variant was 12 % slower and is gone)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram_i8<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "gram_i8 LDS size", __FILE__, __LINE__);
      attr_set = true;
    }
    if (rc == MLN_OK) {
      const int n_wg = n_tiles * n_splits;
      const int grid = (n_wg + 7) / 8 * 8;
      const double scale = alpha / ((double)GQ_SCALE * (double)GQ_SCALE);
      hipLaunchKernelGGL(k_gram_i8<8>, dim3((unsigned)grid), dim3(512), lds_bytes, ctx->stream, planes, Mp, Kp, kchunk, d_tiles,
                         n_tiles, n_wg, scale, parts, ldg, part_stride, m);
      e = hipGetLastError();
      if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "k_gram_i8", __FILE__, __LINE__);
    }
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(planes);
  (void)mln_dfree(d_tiles);
  return rc;
}
