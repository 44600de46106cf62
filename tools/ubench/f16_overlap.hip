// Micro-benchmark (gfx950): do fp16 MFMAs (v_mfma_f32_32x32x16_f16) and fp32 vector instructions (v_min_f32 / v_med3_f32,
// the epilogue of the row-minimum sweep) overlap on one SIMD?
//   mode 0: one wave per SIMD issues N MFMAs (4 independent accumulators)
//   mode 1: one wave per SIMD issues N * R vector instructions (8 independent chains)
//   mode 2: one wave per SIMD issues both, interleaved in one instruction stream (1 MFMA : R vector instructions)
//   mode 3: two waves per SIMD, one issues the MFMAs of mode 0, the other the vector instructions of mode 1
// 256 workgroups (one per CU).  Prints ms per launch.      hipcc --offload-arch=gfx950 -O3 f16_overlap.hip -o f16_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#ifndef RR
#define RR 5
#endif
constexpr int R = RR;   // vector instructions per MFMA

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int n, float s) {
  const int wave = threadIdx.x >> 6;
  f16v acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = s * (threadIdx.x + i);
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(s * threadIdx.x); b[i] = (_Float16)(s + 1.0f); }
  const float c = s + 2.0f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_f = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  if ((MODE == 0 || MODE == 1 || MODE == 2) && wave >= 4) return;
  if (MODE == 2) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int q = (t * R + r) & 7;
          f[q] = __builtin_amdgcn_fmed3f(f[q], c, f[(q + 1) & 7]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (do_m) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
    }
  } else if (do_f) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int t = 0; t < 4 * R; ++t) { const int q = t & 7; f[q] = __builtin_amdgcn_fmed3f(f[q], c, f[(q + 1) & 7]); }
    }
  }
  float r = 0;
  for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) r += acc[t][e];
  for (int i = 0; i < 8; ++i) r += f[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE>
float run(float* out, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, n, 1e-3f);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, n, 1e-3f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

template <int NACC, bool TWO>
__global__ __launch_bounds__(512) void kc(float* out, int n, float s) {
  const int wave = threadIdx.x >> 6;
  if (!TWO && wave >= 4) return;
  f16v acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(s * threadIdx.x); b[i] = (_Float16)(s + 1.0f); }
  for (int i = 0; i < n * 4; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i % NACC], 0, 0, 0);
  float r = 0;
  for (int t = 0; t < NACC; ++t) for (int e = 0; e < 16; ++e) r += acc[t][e];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int NACC> float runc(float* out, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kc<NACC, false>), dim3(256), dim3(512), 0, 0, out, n, 1e-3f);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((kc<NACC, false>), dim3(256), dim3(512), 0, 0, out, n, 1e-3f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
template <int NACC> float runw(float* out, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kc<NACC, true>), dim3(256), dim3(512), 0, 0, out, n, 1e-3f);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((kc<NACC, true>), dim3(256), dim3(512), 0, 0, out, n, 1e-3f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 512);
  const int n = 20000;     // 80000 MFMAs per wave
  const float t0 = run<0>(out, n), t1 = run<1>(out, n), t2 = run<2>(out, n), t3 = run<3>(out, n);
  const double mf = 4.0 * n, ff = 4.0 * R * n;
  printf("R = %d vector instructions per MFMA\n", R);
  printf("mode0 one wave per SIMD, MFMA only      %8.3f ms  (%.1f cycles per MFMA at 2.4 GHz)\n", t0, t0 * 2.4e6 / mf);
  printf("mode1 one wave per SIMD, vector only    %8.3f ms  (%.2f cycles per instruction)\n", t1, t1 * 2.4e6 / ff);
  printf("mode2 one wave per SIMD, interleaved    %8.3f ms  (sum of 0 and 1: %.3f, max: %.3f)\n", t2, t0 + t1, t0 > t1 ? t0 : t1);
  printf("mode3 MFMA wave + vector wave per SIMD  %8.3f ms  (sum: %.3f, max: %.3f)\n", t3, t0 + t1, t0 > t1 ? t0 : t1);
  // dependent chains: the same number of MFMAs per wave over 1, 2, 3 accumulators (one wave per SIMD)
  const float c1 = runc<1>(out, n), c2 = runc<2>(out, n), c3 = runc<3>(out, n);
  printf("chains 1 / 2 / 3 accumulators           %8.3f / %.3f / %.3f ms  (4 accumulators: %.3f)\n", c1, c2, c3, t0);
  const float w1 = runw<1>(out, n), w2 = runw<2>(out, n);
  printf("two waves per SIMD, 1 / 2 accumulators each  %8.3f / %.3f ms for twice the MFMAs (one wave, 4 accumulators: %.3f)\n", w1, w2, t0);
  return 0;
}
