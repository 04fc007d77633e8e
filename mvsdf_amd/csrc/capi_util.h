// capi_util.h -- helpers shared by the extern "C" entry points.
#pragma once
#include <hip/hip_runtime.h>
#include "mlp_common.h"
#include "../../include/mvsdf_hip.h"

int mv_fail(int code, const char* msg);          // records msg, returns code
int mv_check(hipError_t e, const char* where);   // 0 on success
int mv_make_net(const MvsdfNetDesc* d, MvNet* net);

static inline bool mv_wide(const MvNet& net) {
    int maxnt = 0;
    for (int l = 0; l < net.n_layers - 1; ++l) maxnt = net.L[l].NT > maxnt ? net.L[l].NT : maxnt;
    return maxnt > 16;
}
