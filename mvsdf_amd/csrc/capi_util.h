// capi_util.h -- helpers shared by the extern "C" entry points.
#pragma once
#include <hip/hip_runtime.h>
#include "mlp_common.h"
#include "tile_engine_bf16.h"
#include "tile_engine_bf16s.h"
#include "../../include/mvsdf_hip.h"

#include <stdlib.h>
int mv_fail(int code, const char* msg);          // records msg, returns code
// Development / A-B switches (alternative launch paths that are independent implementations of the same passes: MVSDF_FUSE, MVSDF_SPLIT_CHAINS, MVSDF_CHAIN_W8,
// MVSDF_CHAIN_MT, MVSDF_DELTA_CHAIN, MVSDF_LAYER_MT, MVSDF_WG_XCD, MVSDF_BF_CARRY, MVSDF_TAIL_STOP, MVSDF_NFIRST, MVSDF_MT_FIRST) exist only in a library built with
// -DMVSDF_DEV_SWITCHES (mvsdf_amd/build.py: build(tag='dev'); tests/test_gpu_alt_paths.py and the sweep tools load it through MVSDF_LIB).  The product library never
// reads them: its behaviour does not depend on stray environment variables.  (Product switches, read with getenv directly: MVSDF_TAIL, MVSDF_SPLIT_ROWS.)
static inline const char* mv_dev_env(const char* name) {
#ifdef MVSDF_DEV_SWITCHES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
int mv_check(hipError_t e, const char* where);   // 0 on success
// mode 0: SDF net (PE input + skip chaining checked); 1: plain chain; 2: transposed packs (no chaining check)
int mv_make_net_mode(const MvsdfNetDesc* d, MvNet* net, int mode);
static inline int mv_make_net(const MvsdfNetDesc* d, MvNet* net) { return mv_make_net_mode(d, net, 0); }
int mv_make_net_bs(const MvsdfNetDesc* d, MvNetBf* net, int ns); // SDF net on the bf16 packs without duplicated columns, activations as ns bf16 terms (trace_dtype 3 / 4; net = an MvNetBs<ns>)
int mv_make_net_x3(const MvsdfNetDesc* d, MvNetBf* net);        // the three-term packs d->wx3 of the differentiable chains (chain_x3.h); 1: a layer has none (-> the fp32 chains)
int mv_make_net_trace(const MvsdfNetDesc* d, MvNet* net);     // the fp32-engine tracing net: the fp32 packs, or (trace_dtype == 2) the fp32 packs of the bf16-rounded weights
// skip layers of a descriptor as a bit mask (skip_mask wins; else the single skip_layer)
static inline unsigned mv_desc_skip_mask(const MvsdfNetDesc* d) { return d->skip_mask ? d->skip_mask : (d->skip_layer >= 0 ? 1u << d->skip_layer : 0u); }
static inline int mv_bf_nsplit(const MvsdfNetDesc* d, int l) { return (l == 0 || mv_skip_at(mv_desc_skip_mask(d), l)) ? 3 + 6 * d->multires : 0; }

// column tiles per wave the fused chain kernels need for this network (8 waves per workgroup): 2 up to width 256, 4 up to 512,
// 0 = too wide for them (per-layer kernels take over).  Width = the widest tile row ANY phase of a chain produces: the hidden layers' outputs N_l (value /
// E.1 chains over W_l) AND their inputs K_l (normal / E.2 chains over W_l^T produce K_l columns) -- a skip layer fed by a wider-than-hidden layer
// (dims[skip] > the other hidden widths) has K_skip > every N_l.
static inline int mv_chain_ntw(const MvNet& net) {
    int maxnt = 0;
    for (int l = 0; l < net.n_layers - 1; ++l) maxnt = net.L[l].NT > maxnt ? net.L[l].NT : maxnt;
    for (int l = 1; l < net.n_layers; ++l) { const int kt = (net.L[l].K + 15) / 16; maxnt = kt > maxnt ? kt : maxnt; }
    return maxnt <= 16 ? 2 : (maxnt <= 32 ? 4 : 0);
}

static inline bool mv_wide(const MvNet& net) {
    int maxnt = 0;
    for (int l = 0; l < net.n_layers - 1; ++l) maxnt = net.L[l].NT > maxnt ? net.L[l].NT : maxnt;
    return maxnt > 16;
}
