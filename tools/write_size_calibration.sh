#!/bin/bash
# WRITE_SIZE calibration for isolated 8 / 16-byte stores (tools/ubench_write_size.hip) -> gpurun_out/r04_write_size_calibration.txt
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p tools/_bin
hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_write_size tools/ubench_write_size.hip
rm -rf gpurun_out/r04_wsz
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r04_wsz -- tools/_bin/ubench_write_size > gpurun_out/r04_wsz.log 2>&1
python3 - <<'PY' | tee gpurun_out/r04_write_size_calibration.txt
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/r04_wsz/*/*_counter_collection.csv"))[-1]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "WRITE_SIZE" and "store_pieces" in r["Kernel_Name"]:
        rows[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
count = 1 << 20
print("WRITE_SIZE (KiB -> bytes) per piece, by store shape and stride; 1,048,576 pieces per launch; mean of 3 repetitions")
for name, vals in sorted(rows.items()):
    piece = 8 if "<8" in name else 16
    strides = [piece, 32, 64, 512]
    out = []
    for k, st in enumerate(strides):
        v = [vals[i] for i in range(k, len(vals), 4)]
        out.append(f"stride {st:3d}: {sum(v) / len(v) * 1024 / count:6.2f} B")
    print(f"{name:32s} " + "   ".join(out))
PY
