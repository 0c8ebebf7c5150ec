#!/usr/bin/env python3
"""Print VGPR/AGPR/spill/occupancy per kernel of one csrc/*.hip file (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "pyneuralempc_amd", "csrc")
src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(REPO, "include"),
       "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s*(.*?) \[-Rpass", line)
    if not m: continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.rsplit(":", 1); rows[cur][k.strip()] = v.strip()
for name, d in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "")).replace("nempc::", "").replace("void ", "")
    print(f"{dem:60s} VGPR {d.get('VGPRs','?'):>4} AGPR {d.get('AGPRs','?'):>4} spillV {d.get('VGPRs Spill','?'):>5} "
          f"scratch {d.get('ScratchSize [bytes/lane]','?'):>6} occ {d.get('Occupancy [waves/SIMD]','?'):>2} LDS {d.get('LDS Size [bytes/block]','?')}")
