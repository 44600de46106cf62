// Micro-benchmark (gfx950): do fp64 MFMAs and fp64 vector FMAs overlap on one SIMD?
//   mode 0: every wave issues N v_mfma_f64_16x16x4_f64 (4 independent accumulators)
//   mode 1: every wave issues N * R v_fma_f64 (8 independent chains)
//   mode 2: every wave issues both, interleaved in one instruction stream (1 MFMA : R FMAs)
//   mode 3: waves 0-3 of the workgroup (one per SIMD) issue the MFMAs of mode 0, waves 4-7 the FMAs of mode 1
//   mode 4 / 5: like 0 / 1 but only waves 0-3 / 4-7 work (the others exit): the single-wave-per-SIMD rates
// One 512-thread workgroup per CU (2 waves per SIMD), 256 workgroups.  Prints ms per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int R = 9;   // FMAs per MFMA (the kernel-matrix pass: 472 VALU / 52 MFMA)

template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, int n, double s) {
  const int wave = threadIdx.x >> 6;
  v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double f[8];
  for (int i = 0; i < 8; ++i) f[i] = s * (threadIdx.x + i);
  const double a = s * threadIdx.x, b = s + 1.0;
  const bool do_m = MODE == 0 || MODE == 2 || ((MODE == 3 || MODE == 4) && wave < 4);
  const bool do_f = MODE == 1 || MODE == 2 || ((MODE == 3 || MODE == 5) && wave >= 4);
  if (MODE == 4 && wave >= 4) return;
  if (MODE == 5 && wave < 4) return;
  if (MODE == 2) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) f[(t * R + r) & 7] = __builtin_fma(f[(t * R + r) & 7], b, a);
      }
    }
  } else if (do_m) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  } else if (do_f) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int t = 0; t < 4 * R; ++t) f[t & 7] = __builtin_fma(f[t & 7], b, a);
    }
  }
  double r = 0;
  for (int t = 0; t < 4; ++t) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int i = 0; i < 8; ++i) r += f[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE>
float run(double* out, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, n, 1e-9);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, n, 1e-9);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  double* out; hipMalloc(&out, sizeof(double) * 256 * 512);
  const int n = 20000;     // 80000 MFMAs per wave
  const float t0 = run<0>(out, n), t1 = run<1>(out, n), t2 = run<2>(out, n), t3 = run<3>(out, n), t4 = run<4>(out, n), t5 = run<5>(out, n);
  const double mf = 4.0 * n, ff = 4.0 * R * n;
  printf("mode0 all waves MFMA           %8.3f ms  (%.1f cycles per MFMA per SIMD at 2.4 GHz, 2 waves)\n", t0, t0 * 2.4e6 / (2 * mf));
  printf("mode1 all waves FMA            %8.3f ms  (%.2f cycles per FMA per SIMD, 2 waves)\n", t1, t1 * 2.4e6 / (2 * ff));
  printf("mode2 both, one stream         %8.3f ms  (sum of 0 and 1: %.3f, max: %.3f)\n", t2, t0 + t1, t0 > t1 ? t0 : t1);
  printf("mode4 one wave per SIMD MFMA   %8.3f ms  (%.1f cycles per MFMA)\n", t4, t4 * 2.4e6 / mf);
  printf("mode5 one wave per SIMD FMA    %8.3f ms  (%.2f cycles per FMA)\n", t5, t5 * 2.4e6 / ff);
  printf("mode3 MFMA wave + FMA wave     %8.3f ms  (sum of 4 and 5: %.3f, max: %.3f)\n", t3, t4 + t5, t4 > t5 ? t4 : t5);
  return 0;
}
