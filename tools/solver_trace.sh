#!/bin/bash
# per-kernel time of batched solves under rocprofv3: tools/solver_trace.sh [c2|c3] [max_iter] [linesearch]
set -e -o pipefail
CFG=${1:-c2}; MI=${2:-40}; LS=${3:-auto}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/solver_trace_$CFG -- python3 tools/solver_one.py $CFG $MI $LS > gpurun_out/solver_trace_$CFG.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/solver_trace_$CFG/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:86]:86s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s} %")
PY
