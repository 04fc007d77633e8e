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
    const unsigned skm = mode == 0 ? mv_desc_skip_mask(d) : 0u;
    // (idr.py:46-49,86: any layer but the first may take cat([x, PE]) / sqrt(2) -- the last Linear included)
    if (skm & 1u || skm >> d->n_layers) return mv_fail(-2, "net descriptor: layer 0 cannot be a skip layer / skip mask beyond the last layer");
    for (int l = 0; l < d->n_layers; ++l) {
        if (!d->wp[l] || !d->bias[l] || d->K[l] <= 0 || d->N[l] <= 0) return mv_fail(-2, "net descriptor: null pointer or bad dims");
        if (mode == 0) {
            if (l > 0) {
                const int expect = mv_skip_at(skm, l) ? d->N[l - 1] + 3 + 6 * d->multires : d->N[l - 1];
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
    net->skip_mask = skm;
    net->multires = mode == 0 ? d->multires : 0;
    net->S = ((maxk + 63) & ~63) + 8;
    return 0;
}

int mv_make_net_trace(const MvsdfNetDesc* d, MvNet* net) {
    if (!d || d->trace_dtype != 2) return mv_make_net(d, net);
    MvsdfNetDesc r = *d;
    for (int l = 0; l < d->n_layers && l < MVSDF_MAX_LAYERS; ++l) {
        if (!d->wp16[l]) return mv_fail(-2, "net descriptor: trace_dtype = 2 without the rounded packs (mvsdf_pack_bf16w_net)");
        r.wp[l] = (const float*)d->wp16[l];
    }
    return mv_make_net(&r, net);
}

int mv_make_net_x3(const MvsdfNetDesc* d, MvNetBf* net) {
    if (!d || d->n_layers < 2 || d->n_layers > MV_MAXL) return 1;
    memset(net, 0, sizeof(*net));
    int maxkb = 0;
    for (int l = 0; l < d->n_layers; ++l) {
        if (!d->wx3[l] || !d->bias[l]) return 1;
        MvLayerBf& L = net->L[l];
        L.wp = (const uint4*)d->wx3[l];
        L.bias = d->bias[l];
        L.K = d->K[l]; L.N = d->N[l];
        L.nsplit = 0;
        L.KB = mv_bf_kb(L.K, 0); L.NT = mv_ceil16(L.N) / 16;
        if (L.KB > maxkb) maxkb = L.KB;
    }
    net->n_layers = d->n_layers; net->skip_mask = mv_desc_skip_mask(d); net->multires = d->multires;
    net->S = 32 * maxkb + 8;                                       // bf16 elements per LDS row of one term tile (64 KB + 16 bytes: conflict-free b128 reads)
    return 0;
}

int mv_make_net_bs(const MvsdfNetDesc* d, MvNetBf* net, int ns) {
    MvNet chk;
    int rc = mv_make_net_mode(d, &chk, 0);
    if (rc) return rc;
    if (ns < 2 || ns > 3) return mv_fail(-2, "net descriptor: 2 or 3 activation terms");
    memset(net, 0, sizeof(*net));
    int maxk = 0;
    for (int l = 0; l < d->n_layers; ++l) {
        if (!d->wp16[l]) return mv_fail(-2, "net descriptor: trace_dtype = 3 / 4 / 5 without bf16 packs (mvsdf_pack_bf16s_net / mvsdf_pack_bf16x3_net)");
        MvLayerBf& L = net->L[l];
        L.wp = (const uint4*)d->wp16[l];
        L.bias = d->bias[l];
        L.K = d->K[l]; L.N = d->N[l];
        L.nsplit = 0;
        L.KB = mv_bf_kb(L.K, 0); L.NT = mv_ceil16(d->N[l]) / 16;
        if (L.KB * 32 > maxk) maxk = L.KB * 32;
    }
    net->n_layers = d->n_layers; net->skip_mask = chk.skip_mask; net->multires = d->multires;
    net->S = ns * ((maxk + 8) / 2);                               // ns term tiles of bf16 rows of 32*KB + 8 elements (64*KB + 16 bytes: conflict-free b128 reads)
    return 0;
}

extern "C" const char* mvsdf_last_error(void) { return g_err; }
