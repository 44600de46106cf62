// k_kernel_matrix_rows for MLN_K_EXPQUAD (see cov_rows_impl.h)
#include "cov_rows_impl.h"

MLN_DEFINE_ROWS_KIND(launch_kernel_matrix_rows_expquad, MLN_K_EXPQUAD)
