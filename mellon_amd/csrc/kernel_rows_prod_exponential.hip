// k_kernel_matrix_rows_prod for MLN_K_EXPONENTIAL (see kernel_rows_prod_impl.h)
#include "kernel_rows_prod_impl.h"

MLN_DEFINE_ROWS_PROD_KIND(launch_kernel_matrix_rows_prod_exponential, MLN_K_EXPONENTIAL)
