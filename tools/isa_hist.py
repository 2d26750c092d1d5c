#!/usr/bin/env python
"""Instruction histogram of one kernel of a .hip file (device assembly):  python tools/isa_hist.py file.hip mangled-substring [top]"""
import re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
out = "/tmp/isa_hist.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-result",
                "-Wno-unused-value", "-Wno-pass-failed", "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL, check=True)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ops = {}
for l in lines[start:end + 1]:
    m = re.match(r"\s+([a-z][a-z0-9_]+)\s", l)
    if m and not m.group(1).startswith("."):
        ops[m.group(1)] = ops.get(m.group(1), 0) + 1
print(lines[start], "instructions:", sum(ops.values()))
for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:top]:
    print("  %-30s %d" % (k, v))
open("/tmp/isa_kernel.s", "w").write("\n".join(lines[start:end + 1]))
