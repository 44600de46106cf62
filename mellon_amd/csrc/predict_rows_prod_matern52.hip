// k_predict_mean_rows_prod for MLN_K_MATERN52 (see predict_rows_prod_impl.h)
#include "predict_rows_prod_impl.h"

MLN_DEFINE_PREDICT_ROWS_PROD_KIND(launch_predict_mean_rows_prod_matern52, MLN_K_MATERN52)
