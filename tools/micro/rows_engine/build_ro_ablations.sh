#!/bin/bash
# dev tool: side builds of the library with rows_engine_bf16.h's MV_RO_ABLATE switches (1 activation -> conversion, 2 no matrix instructions, 4 no weight
# loads, 8 no LDS reads of the weight fragments, 16 no chunk barriers); only basic.hip is rebuilt: the probe is `tools/bench_mlp.py --mt 64`
# NOTE (round 5): the row-owner engine is parked in this directory and no longer compiled into basic.hip -- to re-run the ablation, re-include
# rows_engine_bf16.h from basic.hip (see basic_hip_kernel.inc here) first; without that the -DMV_RO_ABLATE builds are identical to the product.
# usage: tools/micro/rows_engine/build_ro_ablations.sh 1 2 4 ...   ->  mvsdf_amd/libmvsdf_hip_ro<N>.so   (run with MVSDF_LIB=...)
set -e
cd "$(dirname "$0")/../../../mvsdf_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-pass-failed"
OTHERS="capi_util.o trace.o diff_mlp.o loss_kernels.o optim_kernels.o step_kernels.o sample_kernels.o step_driver.o"
for a in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DMV_RO_ABLATE=$a -c basic.hip -o basic_ro$a.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmvsdf_hip_ro$a.so basic_ro$a.o $OTHERS ) &
done
wait
