#include <stdio.h>
#include <string.h>
#include "capi_util.h"

static thread_local char g_err[512] = "";

int mv_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
int mv_check(hipError_t e, const char* where) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}
int mv_make_net_mode(const MvsdfNetDesc* d, MvNet* net, int mode) {
    if (!d || d->n_layers < 1 || d->n_layers > MV_MAXL) return mv_fail(-2, "net descriptor: n_layers out of range");
    memset(net, 0, sizeof(*net));
    int maxk = 0;
    for (int l = 0; l < d->n_layers; ++l) {
        if (!d->wp[l] || !d->bias[l] || d->K[l] <= 0 || d->N[l] <= 0) return mv_fail(-2, "net descriptor: null pointer or bad dims");
        if (mode == 0) {
            if (l > 0) {
                const int expect = (l == d->skip_layer) ? d->N[l - 1] + 3 + 6 * d->multires : d->N[l - 1];
                if (d->K[l] != expect) return mv_fail(-2, "net descriptor: layer dims do not chain");
            } else if (d->K[0] != 3 + 6 * d->multires) {
                return mv_fail(-2, "net descriptor: first layer K != 3 + 6*multires");
            }
        } else if (mode == 1 && l > 0 && d->K[l] != d->N[l - 1]) {
            return mv_fail(-2, "net descriptor: layer dims do not chain");
        }
        MvLayer& L = net->L[l];
        L.wp = (const float4*)d->wp[l];
        L.bias = d->bias[l];
        L.K = d->K[l]; L.N = d->N[l];
        L.KB = mv_kpad(d->K[l]) / 16; L.NT = mv_ceil16(d->N[l]) / 16;
        if (L.KB * 16 > maxk) maxk = L.KB * 16;
        if (mode == 0 && l < d->n_layers - 1 && L.NT > 32) return mv_fail(-2, "net descriptor: hidden width > 512 not supported");
    }
    net->n_layers = d->n_layers;
    net->skip_layer = mode == 0 ? d->skip_layer : -1;
    net->multires = mode == 0 ? d->multires : 0;
    net->S = ((maxk + 63) & ~63) + 8;
    return 0;
}

extern "C" const char* mvsdf_last_error(void) { return g_err; }
