#!/bin/bash
# A/B of the 32-row blocks for FORWARD products (NEMPC_LG_RM = 4: never, 0: the launcher's rule): per-launch times by kernel and grid
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for case in ${1:-wide256_c2/float64}; do
for rm in 4 0; do
  rm -rf gpurun_out/lgx
  NEMPC_LG_RM=$rm rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lgx -- python3 tools/layered_bench.py $case > gpurun_out/lgx.log 2>&1
  echo "== $case NEMPC_LG_RM=$rm"; grep "^wide\|^deep\|^mixed" gpurun_out/lgx.log | cut -c1-160
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("gpurun_out/lgx/*/*kernel_trace.csv")[0])):
    if "gemm" in r["Kernel_Name"]: acc[(r["Kernel_Name"].split("<")[1].split(">")[0], int(r["Grid_Size_X"])//256)].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(acc.items()): print("  gemm <%s> workgroups %5d: avg %.1f us over %d launches" % (k[0], k[1], sum(v)/len(v)/1e3, len(v)))
PY
done
done
