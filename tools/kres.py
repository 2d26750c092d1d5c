#!/usr/bin/env python
"""Register / scratch / LDS / occupancy of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kres.py theanet_amd/csrc/conv_c8.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-result",
       "-Wno-unused-value", "-Wno-pass-failed", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if flt and flt not in name:
        continue
    print("%-52s vgpr %4s agpr %4s scratch %5s occ %2s lds %6s sgpr %3s" % (
        name[:52], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
        r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"), r.get("SGPRs", "?")))
