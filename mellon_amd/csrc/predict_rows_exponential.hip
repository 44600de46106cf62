// k_predict_mean_rows for MLN_K_EXPONENTIAL (see predict_rows_impl.h)
#include "predict_rows_impl.h"

MLN_DEFINE_PREDICT_ROWS_KIND(launch_predict_mean_rows_exponential, MLN_K_EXPONENTIAL)
