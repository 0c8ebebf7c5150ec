#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (gpurun_out/prof*_{trace,fetch,write,mfma}) into small tracked files
under profiles/: the --stats kernel table as is, and a JSON of per-kernel mean counter values per launch.
HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB, FETCH_SIZE is doubled on
gfx950 (128-B requests tallied at 64 B); WRITE_SIZE is exact for 8- / 16-B-per-lane streams and charges an isolated
narrower store its whole 32-B sector (round-4 calibration: tools/ubench_write_size.hip)."""
import collections, csv, glob, json, os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]          # e.g. gpurun_out/prof2  r01_c2_b1024
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(REPO, "profiles")
os.makedirs(out, exist_ok=True)
def newest(pattern):
    """gpurun merges every call's output into the local gpurun_out/, so a directory can hold several runs: take the
    latest file only"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


for f in newest(f"{src}_trace/*/*_kernel_stats.csv"):
    shutil.copy(f, os.path.join(out, f"{tag}_kernel_stats.csv"))
for f in newest(f"{src}_hess_trace/*/*_kernel_stats.csv"):        # bench.py --only-hessian (the Hessian-callback legs alone)
    shutil.copy(f, os.path.join(out, f"{tag}_hess_kernel_stats.csv"))
for f in newest(f"{src}_sparse_trace/*/*_kernel_stats.csv"):      # bench.py --only-sparse (the sparse-contract leg alone)
    shutil.copy(f, os.path.join(out, f"{tag}_sparse_kernel_stats.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write", "mfma"):
    if not glob.glob(f"{src}_{kind}"):
        continue
    for f in newest(f"{src}_{kind}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("nempc::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in agg.items():
    if "copyBuffer" in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    m["launches_sampled"] = max(len(x) for x in v.values())
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_read_bytes_corrected"] = m["FETCH_SIZE"] * 1024 * 2
        m["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        m["hbm_traffic_bytes"] = m["hbm_read_bytes_corrected"] + m["hbm_write_bytes"]
        # WRITE_SIZE is exact for 8- and 16-byte-per-lane STREAMS and charges an isolated 8- or 16-byte store one 32-byte
        # sector, write-through or not (tools/ubench_write_size.hip, profiles/r04_write_size_calibration.txt): that is what
        # the memory system moves for a partial-sector write, so the figure is traffic, not a counter artefact
        m["write_size_calibrated"] = True
    res[k] = m
json.dump(res, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
# pipe-utilisation / instruction-mix passes (<src>_pipe1, <src>_pipe2) -> <tag>_pipe.json
pagg = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("pipe1", "pipe2"):
    for f in newest(f"{src}_{kind}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("nempc::", "")
            if name.startswith(("rows_", "post_", "rowhess", "assemble")):
                pagg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
if pagg:
    pres = {k: dict({c: sum(x) / len(x) for c, x in v.items()}, launches_sampled=max(len(x) for x in v.values()))
            for k, v in pagg.items()}
    json.dump(pres, open(os.path.join(out, f"{tag}_pipe.json"), "w"), indent=1)
