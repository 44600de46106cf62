// The element epilogue of the persistent-row covariance kernels (cov_rows_impl.h, kernel_rows_prod_impl.h,
// predict_rows.hip, predict_rows_prod.hip): distance -> stationary kernel value, as few fp64 issue slots as the
// arithmetic allows.  Reference: util.py:351-366 `distance` (x.x - 2 x.y + y.y + 1e-12, clamped at 0, square root),
// cov.py:62-66 Matern32, :157-161 Matern52, :255-259 ExpQuad, :352-356 Exponential, :453-457 RatQuad.
//
// Why it is written this way (profiles/r03_pmc_sq.txt: 54 VALU instructions per element, fp64 matrix and vector pipe
// shared -- every instruction here is a slot the MFMAs of the next tile wait for):
//   * every kind is a function of  s = c^2 (|x|^2 - 2 x.y + |y|^2 + 1e-12) / ls^2  with the kind's constant c folded
//     in: the norms arrive pre-scaled (c^2 / ls^2 (|x|^2 + 1e-12) per row, c^2 / ls^2 |y|^2 per column) and the -2 c^2 / ls^2
//     rides in the FMA that consumes the MFMA's dot product -- two instructions, and r^2 = s comes for free
//     (Matern52's r^2 / 3; ExpQuad and RatQuad need no square root at all);
//   * sqrt(s) from v_rsq_f64's seed by ONE correction of second order in the seed's residual (see sqrt_newton): one
//     transcendental + five multiply-adds instead of thirteen;
//   * e^{-r}: k = rint(-r log2 e), t = -r - k ln2 in ONE FMA (the rounding error of the constant, k 2.3e-17, only matters
//     where e^{-r} is long negligible), a degree-11 minimax polynomial (Remez on [-ln2/2, ln2/2], relative error 3.2e-18;
//     with Horner's roundings 1.7e-16 -- tools/exp_poly_remez.py), v_ldexp_f64 with the saturating v_cvt_i32_f64 as its
//     exponent: no clamp, no special cases, underflows to 0 like exp();
//   * a PRODUCT of two leaves needs one exponential: p0 e^{-r0} p1 e^{-r1} = p0 p1 e^{-(r0 + r1)}.
// 28 fp64 instructions per Matern52 element (round 3: 45 + address arithmetic); covariance values within a few 1e-16
// (relative) of a long-double evaluation (tools/epilogue_accuracy.py).
#pragma once
#include "mln_core.h"

namespace covepi {

// c^2 / ls^2 of the kind: s = sq_scale * (squared distance + 1e-12) is what leaf_terms() consumes
template <int KIND>
__device__ __forceinline__ double sq_scale(const DevLeaf& lf) {
  const double il2 = lf.alpha_inv_ls[1] * lf.alpha_inv_ls[1];
  if (KIND == MLN_K_MATERN32) return 3.0 * il2;                 // r = sqrt(3) dist / ls
  if (KIND == MLN_K_MATERN52) return 5.0 * il2;                 // r = sqrt(5) dist / ls
  if (KIND == MLN_K_EXPQUAD) return 0.5 * il2;                  // exponent (dist / ls)^2 / 2
  if (KIND == MLN_K_EXPONENTIAL) return 0.25 * il2;             // exponent dist / (2 ls)
  return 0.5 * il2 / lf.alpha;                                  // RatQuad: (1 + s)^-alpha
}

// sqrt(s), s > 0 and finite: with y = v_rsq_f64(s) = (1 + delta) / sqrt(s), t = s y and e = 1 - t y = 1 - (1 + delta)^2,
// sqrt(s) = t (1 - e)^(-1/2) = t (1 + e / 2 + 3 e^2 / 8 + ...).  The seed is only good to ~2^-23 (measured: the first-order
// form left 1.4e-14 on Matern52 values, tools/epilogue_accuracy.py), so the series is taken to second order: error
// 5 e^3 / 16 ~ 1e-21, one transcendental + five multiply-adds.
__device__ __forceinline__ double sqrt_newton(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  const double t = s * y;
  const double e = fma(-t, y, 1.0);
  const double q = fma(e, 0.375, 0.5) * e;
  return fma(t, q, t);
}

// e^{-r}, r >= 0 (any finite r: large r underflows to 0)
__device__ __forceinline__ double exp_neg(double r) {
  const double k = __builtin_rint(r * -1.4426950408889634);
  const double t = fma(k, -0.6931471805599453, -r);
  double p = 0x1.ad6ffb5024030p-26;
  p = fma(p, t, 0x1.28b376bdee3a6p-22);
  p = fma(p, t, 0x1.71df40058fb45p-19);
  p = fma(p, t, 0x1.a019926153312p-16);
  p = fma(p, t, 0x1.a01a0111fd1f5p-13);
  p = fma(p, t, 0x1.6c16c187aa032p-10);
  p = fma(p, t, 0x1.1111111130b4cp-7);
  p = fma(p, t, 0x1.555555554f2bfp-5);
  p = fma(p, t, 0x1.55555555554a2p-3);
  p = fma(p, t, 0x1.0000000000010p-1);
  p = fma(p, t, 1.0);
  p = fma(p, t, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)k);     // v_cvt_i32_f64 saturates: k below int range still means "underflow"
}

// value = poly * e^{-arg}: the two factors of a stationary kind at scaled squared distance s (s > 0 for the kinds that
// take a square root -- callers clamp with fmax(s, 1e-300), which also turns the rounding-negative s of coincident
// points into distance ~0, the reference's max(., 0))
template <int KIND>
__device__ __forceinline__ void leaf_terms(double s, double& poly, double& arg) {
  if (KIND == MLN_K_MATERN32) { const double r = sqrt_newton(s); poly = r + 1.0; arg = r; }
  else if (KIND == MLN_K_MATERN52) { const double r = sqrt_newton(s); poly = fma(s, 0.3333333333333333, 1.0) + r; arg = r; }
  else if (KIND == MLN_K_EXPQUAD) { poly = 1.0; arg = s; }
  else { poly = 1.0; arg = sqrt_newton(s); }                   // MLN_K_EXPONENTIAL
}

template <int KIND>
__device__ __forceinline__ double leaf_value_s(double s) {
  double poly, arg;
  leaf_terms<KIND>(s, poly, arg);
  if (KIND == MLN_K_EXPQUAD || KIND == MLN_K_EXPONENTIAL) return exp_neg(arg);
  return poly * exp_neg(arg);
}

// two leaves of the same kind multiplied (the time-sensitive product kernel, parameters.py:641-644): one exponential
template <int KIND>
__device__ __forceinline__ double leaf_product_s(double s0, double s1) {
  double p0, a0, p1, a1;
  leaf_terms<KIND>(s0, p0, a0);
  leaf_terms<KIND>(s1, p1, a1);
  if (KIND == MLN_K_EXPQUAD || KIND == MLN_K_EXPONENTIAL) return exp_neg(a0 + a1);
  return (p0 * p1) * exp_neg(a0 + a1);
}

}  // namespace covepi
