#!/bin/bash
# Where the layered GEMM's time goes, by leaving parts out (timing experiments; the variants compute garbage):
#   for v in NOLOAD NOLDS NOEPI NOBARRIER; do python tools/build_variant.py lg_$v -DNEMPC_LG_EXP_$v --only kernels_layered.hip; done
#   python tools/build_variant.py lg_NOLDS_NOLOAD -DNEMPC_LG_EXP_NOLDS -DNEMPC_LG_EXP_NOLOAD -DNEMPC_LG_EXP_NOBARRIER --only kernels_layered.hip
# then on the GPU box: bash tools/lg_limiter_exp.sh   (rocprofv3 kernel table of tools/layered_bench.py wide256_c2/float64)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" lg_NOLOAD lg_NOLDS lg_NOEPI lg_NOBARRIER lg_NOLDS_NOLOAD; do
  if [ -z "$v" ]; then lib=pyneuralempc_amd/libnempc.so; else lib=pyneuralempc_amd/build_$v/libnempc_$v.so; fi
  rm -rf gpurun_out/lgx
  NEMPC_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lgx -- python3 tools/layered_bench.py wide256_c2/float64 > /dev/null 2>&1
  echo "== ${v:-shipped}"
  python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("gpurun_out/lgx/*/*kernel_stats.csv")[0])):
    if "gemm" in r["Name"]: print("  gemm avg %.1f min %s max %s" % (float(r["AverageNs"])/1e3, r["MinNs"], r["MaxNs"]))
PY
done
