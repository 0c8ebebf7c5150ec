#!/bin/bash
# A/B of the exec-restore repair (pyneuralempc_amd/_isa.py) on the GPU box: the two round-3 repro scripts under the library
# assembled from the compiler's text as it is (tools/build_variant.py norepair --no-repair --only
# kernels_mfma_f32.hip,kernels_mfma_f64_relu.hip) and under the shipped, repaired library.
set -e
for lib in pyneuralempc_amd/build_norepair/libnempc_norepair.so pyneuralempc_amd/libnempc.so; do
  echo "=== $lib"
  NEMPC_LIB=$PWD/$lib python tools/repro_tile_rk4_fp32.py 2>&1 | grep -v amdgpu.ids
  NEMPC_LIB=$PWD/$lib python tools/repro_tile_relu_fp64.py 2>&1 | grep -v amdgpu.ids
done
