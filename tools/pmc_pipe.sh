#!/bin/bash
# instruction mix of the headline launch's kernels: tools/pmc_pipe.sh <tag> [config]  -> gpurun_out/<tag>_pipe2/
set -e -o pipefail
TAG=${1:-x}; CFG=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/${TAG}_pipe2 -- python3 bench.py --config $CFG --only-eval --steps 20 --warmup 3 > /dev/null 2> gpurun_out/${TAG}_pipe2.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/${TAG}_pipe2/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k, d in acc.items():
    print(k, "launches", cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in d.items()})
PY
