#!/bin/bash
# Round-2 rocprofv3 evidence (run on the GPU box through gpurun; profiles/README.md).  The traces run bench.py in its
# profiling mode (--only-eval: the timed loop of the headline launch, the row-kernel-only timing loop and the accuracy
# check of the timed launch -- no solver / Hessian / two-stream / other-config / CPU legs), so every average in the
# kernel-stats CSV is an average over launches of ONE configuration.
# usage: tools/profile_round2.sh <tag>     -> gpurun_out/<tag>_*  (then tools/summarize_profiles.py per config)
set -e -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 bench.py --hessian --steps 200 --warmup 20 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
echo "bench default done"
for cfg in c3 c5; do
  python3 bench.py --config $cfg --hessian --steps 100 --warmup 10 --no-cpu > $O/${TAG}_bench_${cfg}.json 2> $O/${TAG}_bench_${cfg}.err
  echo "bench $cfg done"
done
# enough timed steps that the clock-priming launches of bench.py (40 ms per timing loop, in synchronised batches of 16)
# are a small share of the per-kernel averages
for cfg in c2 c3 c5; do
  case $cfg in c2) ST=4000;; c5) ST=1500;; *) ST=200;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${cfg}_trace -- python3 bench.py --config $cfg --only-eval --steps $ST --warmup 20 > $O/${TAG}_${cfg}_trace.json 2> $O/${TAG}_${cfg}_trace.err
  echo "trace $cfg done"
done
for cfg in c2 c5; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_${cfg}_fetch -- python3 bench.py --config $cfg --only-eval --steps 50 --warmup 5 > /dev/null 2> $O/${TAG}_${cfg}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_${cfg}_write -- python3 bench.py --config $cfg --only-eval --steps 50 --warmup 5 > /dev/null 2> $O/${TAG}_${cfg}_write.err
  echo "pmc $cfg done"
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_c2_mfma -- python3 bench.py --config c2 --only-eval --steps 50 --warmup 5 > /dev/null 2> $O/${TAG}_c2_mfma.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_c2_pipe1 -- python3 bench.py --config c2 --only-eval --steps 50 --warmup 5 > /dev/null 2> $O/${TAG}_c2_pipe1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/${TAG}_c2_pipe2 -- python3 bench.py --config c2 --only-eval --steps 50 --warmup 5 > /dev/null 2> $O/${TAG}_c2_pipe2.err
echo "all done"
