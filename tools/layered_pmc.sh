#!/bin/bash
# Pipe counters of the layered GEMM launches (gpurun; counters in passes of their own, no trace domain)
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
CASE=${1:-wide256_c2/float64}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/lg_pmc1 -- python3 tools/layered_bench.py $CASE > $O/lg_pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/lg_pmc2 -- python3 tools/layered_bench.py $CASE > $O/lg_pmc2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_DEP_WAIT SQ_IFETCH SQ_WAIT_IFETCH --output-format csv -d $O/lg_pmc3 -- python3 tools/layered_bench.py $CASE > $O/lg_pmc3.log 2>&1 || true
echo done
# (round 5) memory side of the same launches
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $O/lg_pmc4 -- python3 tools/layered_bench.py $CASE > $O/lg_pmc4.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACCUM_PREV_HIRES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --output-format csv -d $O/lg_pmc5 -- python3 tools/layered_bench.py $CASE > $O/lg_pmc5.log 2>&1 || true
echo done2
