#!/bin/bash
# Refresh the rocprofv3 evidence for the current build (run on the GPU box through gpurun; see profiles/README.md).
# usage: tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*  (then tools/summarize_profiles.py per config)
set -e -o pipefail
TAG=${1:-r01c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for cfg in c2 c2_b256 c3 c5; do
  python3 bench.py --config $cfg --hessian --steps 200 --warmup 20 > $O/${TAG}_bench_${cfg}.json 2> $O/${TAG}_bench_${cfg}.err
  echo "bench $cfg done"
done
for cfg in c2 c3 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${cfg}_trace -- python3 bench.py --config $cfg --hessian --steps 100 --warmup 10 --no-cpu > /dev/null 2> $O/${TAG}_${cfg}_trace.err
  echo "trace $cfg done"
done
for cfg in c2 c5; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_${cfg}_fetch -- python3 bench.py --config $cfg --steps 50 --warmup 5 --no-cpu > /dev/null 2> $O/${TAG}_${cfg}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_${cfg}_write -- python3 bench.py --config $cfg --steps 50 --warmup 5 --no-cpu > /dev/null 2> $O/${TAG}_${cfg}_write.err
  echo "pmc $cfg done"
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_c2_mfma -- python3 bench.py --config c2 --steps 50 --warmup 5 --no-cpu > /dev/null 2> $O/${TAG}_c2_mfma.err
echo "all done"
