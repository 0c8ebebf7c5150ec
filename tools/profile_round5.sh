#!/bin/bash
# Round-5 rocprofv3 evidence (run on the GPU box through gpurun; profiles/README.md).  Traces run bench.py in its profiling
# modes (--only-eval: the timed loop of the headline launch, the row-kernel-only timing loop and the accuracy check of
# the timed launch; --only-hessian: the Hessian-callback legs alone), so every average in a kernel-stats CSV is an average
# over launches of ONE configuration.  Counters in passes of their own (no trace domain combined with --pmc).
# usage: tools/profile_round5.sh <tag> [part]   part: bench | trace | layered | pmc | all   -> gpurun_out/<tag>_*
set -e -o pipefail
TAG=${1:-r05}; PART=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
if [ "$PART" = bench ] || [ "$PART" = all ]; then
  python3 bench.py --steps 200 --warmup 20 --details-file ${TAG}_bench_default_details.json > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
  echo "bench default done"
  python3 bench.py --steps 20 --warmup 5 --details-file ${TAG}_bench_driver_form_details.json > $O/${TAG}_bench_driver_form.json 2> $O/${TAG}_bench_driver_form.err
  echo "bench driver form done"
  for cfg in c3 c5; do
    python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu --details-file ${TAG}_bench_${cfg}_details.json > $O/${TAG}_bench_${cfg}.json 2> $O/${TAG}_bench_${cfg}.err
    echo "bench $cfg done"
  done
fi
if [ "$PART" = trace ] || [ "$PART" = all ]; then
  for cfg in c2 c3 c5; do
    case $cfg in c2) ST=4000;; c5) ST=1500;; *) ST=200;; esac
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${cfg}_trace -- python3 bench.py --config $cfg --only-eval --steps $ST --warmup 20 --details-file ${TAG}_${cfg}_trace_details.json > $O/${TAG}_${cfg}_trace.json 2> $O/${TAG}_${cfg}_trace.err
    echo "trace $cfg done"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${cfg}_hess_trace -- python3 bench.py --config $cfg --only-hessian --steps 800 --details-file ${TAG}_${cfg}_hess_trace_details.json > $O/${TAG}_${cfg}_hess_trace.json 2> $O/${TAG}_${cfg}_hess_trace.err
    echo "hessian trace $cfg done"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${cfg}_sparse_trace -- python3 bench.py --config $cfg --only-sparse --steps 800 --details-file ${TAG}_${cfg}_sparse_trace_details.json > $O/${TAG}_${cfg}_sparse_trace.json 2> $O/${TAG}_${cfg}_sparse_trace.err
    echo "sparse trace $cfg done"
  done
fi
if [ "$PART" = trace ] || [ "$PART" = layered ] || [ "$PART" = all ]; then
  # the layer-at-a-time GEMM path (networks outside the register-resident kernels)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_layered_wide256_trace -- python3 tools/layered_bench.py wide256_c2/float64 > $O/${TAG}_layered_wide256_trace.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_layered_wide512_trace -- python3 tools/layered_bench.py wide512x4_c3/float64 > $O/${TAG}_layered_wide512_trace.log 2>&1
  python3 tools/layered_bench.py > $O/${TAG}_layered_bench.log 2>&1
  echo "layered done"
fi
if [ "$PART" = pmc ] || [ "$PART" = all ]; then
  for cfg in c2 c3 c5; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_${cfg}_fetch -- python3 bench.py --config $cfg --only-eval --steps 50 --warmup 5 --details-file scratch_details.json > /dev/null 2> $O/${TAG}_${cfg}_fetch.err
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_${cfg}_write -- python3 bench.py --config $cfg --only-eval --steps 50 --warmup 5 --details-file scratch_details.json > /dev/null 2> $O/${TAG}_${cfg}_write.err
    echo "pmc $cfg done"
  done
  for cfg in c2 c3; do
    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_${cfg}_pipe1 -- python3 bench.py --config $cfg --only-eval --steps 50 --warmup 5 --details-file scratch_details.json > /dev/null 2> $O/${TAG}_${cfg}_pipe1.err
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/${TAG}_${cfg}_pipe2 -- python3 bench.py --config $cfg --only-eval --steps 50 --warmup 5 --details-file scratch_details.json > /dev/null 2> $O/${TAG}_${cfg}_pipe2.err
    echo "pipe $cfg done"
  done
fi
echo "all done"
