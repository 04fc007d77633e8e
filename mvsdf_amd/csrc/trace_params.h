// trace_params.h -- internal aliases of the public C structs.
#pragma once
#include "../../include/mvsdf_hip.h"
typedef MvsdfTraceParams MvTraceParams;
#define MV_CNT_ROWS_SPHERE MVSDF_CNT_ROWS_SPHERE
#define MV_CNT_ROWS_SAMPLER MVSDF_CNT_ROWS_SAMPLER
#define MV_CNT_ROWS_SECANT MVSDF_CNT_ROWS_SECANT
#define MV_CNT_ROWS_MINSDF MVSDF_CNT_ROWS_MINSDF
#define MV_CNT_N_SECANT MVSDF_CNT_N_SECANT
#define MV_CNT_N_SAMPLER MVSDF_CNT_N_SAMPLER
#define MV_CNT_N_MINSDF MVSDF_CNT_N_MINSDF
#define MV_ITEM_SAMPLER 1
#define MV_ITEM_MINSDF 2
#define MV_ITEM_OM 4
