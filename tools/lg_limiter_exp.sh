#!/bin/bash
# Where the layered GEMM's time goes, by leaving parts out (timing experiments; the variants compute garbage):
#   for v in NOLOAD NOLDS NOEPI NOBARRIER; do python tools/build_variant.py lg_$v -DNEMPC_LG_EXP_$v --only kernels_layered.hip; done
# then on the GPU box: bash tools/lg_limiter_exp.sh   (rocprofv3 kernel trace of tools/layered_bench.py wide256_c2/float64 with
# NEMPC_LAYERED_FUSE=0, so that the plain products stand alone; launches told apart by their grid size)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" lg_NOLOAD lg_NOLDS lg_NOEPI lg_NOBARRIER; do
  if [ -z "$v" ]; then lib=pyneuralempc_amd/libnempc.so; else lib=pyneuralempc_amd/build_$v/libnempc_$v.so; fi
  rm -rf gpurun_out/lgx
  NEMPC_LAYERED_FUSE=0 NEMPC_LIB=$PWD/$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lgx -- python3 tools/layered_bench.py wide256_c2/float64 > /dev/null 2>&1
  echo "== ${v:-shipped}"
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("gpurun_out/lgx/*/*kernel_trace.csv")[0])):
    if "gemm" in r["Kernel_Name"]: acc[(int(r["Grid_Size_X"])//256, r["Kernel_Name"].split("<")[1].split(">")[0])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(acc.items()): print("  gemm workgroups %5d <%s>: avg %.1f us over %d launches" % (k[0], k[1], sum(v)/len(v)/1e3, len(v)))
PY
done
