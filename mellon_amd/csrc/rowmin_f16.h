// fp16-split row minimum (rowmin_f16.hip): pre-filter of the exact 1-NN search, assignment step of k-means.
#pragma once
#include "mln_core.h"

size_t rowmin_split_bytes(int64_t rows);   // device bytes of the split (hi | lo halves) copy of `rows` rows
// x (n x d doubles, d <= 64) -> split rows, squared norms in fp64 (xx, may be null) and rounded to fp32 (xxf, may be null)
// role 0: plain rows; 1: query rows of the folded product (coordinates x -2, ones in three spare slots); 2: candidate rows of
// it (|y|^2 as three halves in those slots) -- roles 1 / 2 need d <= 61
// prep (device, ROWMIN_PREP_DOUBLES doubles, from rowmin_prepare): the split holds (x - centre) * scale and xx / xxf are the
// squared norms of THAT
#define ROWMIN_PREP_DOUBLES 66
int rowmin_prepare(mln_ctx* ctx, const double* y, int64_t m, const double* x, int64_t n, int d, double* prep);
int launch_split_f16(mln_ctx* ctx, const double* x, int64_t n, int d, void* split, double* xx, float* xxf, int role,
                     const double* prep);
// per row i of xs: m1 = min_j (yyf_j - 2 x_i.y_j), arg = its j, m2 = the second smallest (null: not tracked);
// exclude_self: the pair (i, i + self_offset) does not count
// fold: operands split with roles 1 / 2; arg is then the first of four candidates arg + {0, 32, 64, 96} (launch_resolve_labels
// picks the closest in fp64)
int launch_rowmin_f16x3(mln_ctx* ctx, const void* xs, int64_t n, const void* ys, int64_t m, const float* yyf,
                        int64_t self_offset, int exclude_self, float* m1, float* m2, int* arg, int fold,
                        const int* row_idx = nullptr);   // row_idx (optional, device): query row r is row row_idx[r] of xs
int launch_resolve_labels(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d, const double* yy, int* arg);
// k-means with group bounds (kmeans.hip): per row i and stage s of the permuted centres, lbg[i * nstage + s] - cum[s] is a lower
// bound of the distance to every centre of the stage other than the row's label
struct KmGroups {
  float* lbg;               // n x nstage
  const double* cum;        // nstage: the stages' accumulated movement
  const float* smin;        // the stage-minimum sweep's output, smin[s * smin_stride + r]
  int64_t smin_stride;
  const int* cpos;          // centre id -> position in the sweep order (stage = position / 256)
  int nstage;
};
// k-means with bounds: label, upper and lower bound of the searched rows from a (TOP2, fold) sweep's m2 / arg (see the kernel)
int launch_km_resolve(mln_ctx* ctx, const double* x, int64_t cnt, const int* idx, const double* c, int64_t m, int d,
                      const double* xxs, const double* yy_max, const double* prep, const float* m2, const int* arg,
                      int* label, double* ub, double* lb, double* sums, double* counts, const double* colscale,
                      const int* cnt_dev = nullptr, const int* cperm = nullptr, const KmGroups* grp = nullptr,
                      const uint32_t* stage_mask = nullptr, int mask_words = 0);
// cnt_dev (optional, device): the row count when the host does not know it (cnt bounds the grid); cperm: candidate position ->
// centre id when the sweep saw permuted centres; grp / stage_mask: the group bounds to renew and the stages swept, see the kernel
int launch_km_init_groups(mln_ctx* ctx, int64_t n, const int* label, const double* lb, const double* xxs, const double* yy_max,
                          const double* prep, const KmGroups* grp);
// the folded TOP2 sweep over the rows row_idx[0 .. *n_dev) (n_dev null: n_max rows), each 256-row block restricted to the
// candidate blocks its row of stage_mask selects (null: all)
// smin (optional): instead of the winner, per row and stage the smallest value of the stage, smin[stage * smin_stride + row]
int launch_rowmin_masked(mln_ctx* ctx, const void* xs, int64_t n_max, const int* n_dev, const void* ys, int64_t m, float* m1,
                         float* m2, int* arg, const int* row_idx, const uint32_t* stage_mask, int mask_words,
                         float* smin = nullptr, int64_t smin_stride = 0);
int launch_max_norm(mln_ctx* ctx, const double* xx, int64_t n, double* out);   // out[0] = max xx (one workgroup)
// exact nearest-neighbour distances via the pre-filter + fp64 certification (+ exact re-search of uncertified rows)
int nn_distances_prefiltered(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d,
                             int64_t self_offset, double* out, double* stats);
// the exact fp64 search (cov_kernels.hip); excl (optional, device): the excluded candidate of each row
int launch_nn_distances_exact(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d,
                              int64_t self_offset, const int64_t* excl, double* out);
