#!/bin/bash
# dev tool: side builds of the library with tile_engine.h's MV_ABLATE switches (only basic.hip is rebuilt: the probe is tools/bench_mlp.py)
# usage: tools/build_ablations.sh 1 2 4 8 12 13   ->  mvsdf_amd/libmvsdf_hip_abl<N>.so   (run with MVSDF_LIB=...)
set -e
cd "$(dirname "$0")/../mvsdf_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-pass-failed"
OTHERS="capi_util.o trace.o diff_mlp.o loss_kernels.o optim_kernels.o step_kernels.o sample_kernels.o"
for a in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DMV_ABLATE=$a -c basic.hip -o basic_abl$a.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmvsdf_hip_abl$a.so basic_abl$a.o $OTHERS ) &
done
wait
