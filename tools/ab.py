#!/usr/bin/env python
"""A/B of bench.py under different environments on ONE box, interleaved (A B A B ...) so that clock / thermal drift hits
both arms:   python tools/ab.py [--reps 3] [--args "--prms wide6.prms --dtype f16"] "" "TN_PIPE_MID=1" "TN_X=2 TN_Y=3"
Prints ms_per_step (timed region) and the sustained figure of every run and the per-arm medians."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reps, extra = 3, ""
while args and args[0].startswith("--"):
    if args[0] == "--reps":
        reps = int(args[1])
    elif args[0] == "--args":
        extra = args[1]
    args = args[2:]
arms = args or [""]
res = {a: [] for a in arms}
for r in range(reps):
    for a in arms:
        env = dict(os.environ)
        for kv in a.split():
            k, v = kv.split("=", 1)
            env[k] = v
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-other-configs", "--no-roofline",
               "--steps", "400", "--warmup", "50"] + extra.split()
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().split("\n")[-1])
            ms = j["ms_per_step"]
            sus = (j.get("sustained") or {}).get("ms_per_step", float("nan"))
        except Exception:
            print("FAILED [%s]: %s" % (a, (out.stdout + out.stderr)[-800:]))
            continue
        res[a].append((ms, sus))
        print("rep %d  [%-40s]  ms_per_step %.4f   sustained %.4f" % (r, a, ms, sus), flush=True)
for a in arms:
    if res[a]:
        print("MEDIAN [%-40s]  ms_per_step %.4f   sustained %.4f" % (a, statistics.median(x[0] for x in res[a]),
                                                                   statistics.median(x[1] for x in res[a])))
