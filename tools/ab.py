#!/usr/bin/env python
"""A/B of bench.py under different environments on ONE box, interleaved so that clock / thermal drift hits every arm,
WITH AN A/A CONTROL: the first arm runs twice per repetition (as "A" and as "A'", identical in every respect); the spread
between those two is what this harness cannot resolve on this box today, and a change is reported KEPT only when its
median lies outside it (round 5 credited -2.3 % to a code path no configuration took: two identical builds).

    python tools/ab.py [--reps 4] [--args "--prms wide6.prms --dtype f16"] "" "TN_X=1" "TN_X=2 TN_Y=3"
    python tools/ab.py --libs old.so new.so            (two builds of the library, through TN_HIP_LIB)

Order inside a repetition rotates (A B A', B A' A, ...) so that no arm always runs first.  Prints every run, per-arm
median / min / max of ms_per_step (the timed region) and of the sustained figure, the A/A spread, and a verdict per arm."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reps, extra, libs = 4, "", False
while args and args[0].startswith("--"):
    if args[0] == "--reps":
        reps = int(args[1]); args = args[2:]
    elif args[0] == "--args":
        extra = args[1]; args = args[2:]
    elif args[0] == "--libs":
        libs = True; args = args[1:]
    else:
        sys.exit("unknown option " + args[0])
arms = args or [""]
if libs:
    arms = ["TN_HIP_LIB=" + os.path.abspath(a) for a in arms]
CONTROL = "A/A control (arm 0 again)"
labels = list(arms) + [CONTROL]
envs = {a: a for a in arms}
envs[CONTROL] = arms[0]
res = {a: [] for a in labels}


def run(label):
    env = dict(os.environ)
    for kv in envs[label].split():
        k, v = kv.split("=", 1)
        env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-other-configs", "--no-roofline",
           "--steps", "400", "--warmup", "50"] + extra.split()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().split("\n")[-1])
        return j["ms_per_step"], (j.get("sustained") or {}).get("ms_per_step", float("nan"))
    except Exception:
        print("FAILED [%s]: %s" % (label, (out.stdout + out.stderr)[-800:]))
        return None


for r in range(reps):
    order = labels[r % len(labels):] + labels[:r % len(labels)]
    for a in order:
        got = run(a)
        if got:
            res[a].append(got)
            print("rep %d  [%-44s]  ms_per_step %.4f   sustained %.4f" % (r, a[-44:], got[0], got[1]), flush=True)

med = {a: statistics.median(x[0] for x in v) for a, v in res.items() if v}
for a in labels:
    if res[a]:
        ms = [x[0] for x in res[a]]
        print("MEDIAN [%-44s]  ms_per_step %.4f  (min %.4f max %.4f, n=%d)   sustained %.4f"
              % (a[-44:], med[a], min(ms), max(ms), len(ms), statistics.median(x[1] for x in res[a])))
if arms[0] in med and CONTROL in med:
    base = med[arms[0]]
    aa = abs(med[CONTROL] - base) / base
    # the run-to-run scatter inside the two identical arms counts too: half the larger min-max range
    rng = max((max(x[0] for x in res[a]) - min(x[0] for x in res[a])) / 2 / base for a in (arms[0], CONTROL))
    floor = max(aa, rng)
    print("A/A: identical arms differ by %.2f %% (medians), half-range %.2f %% -> resolution %.2f %%" % (100 * aa, 100 * rng, 100 * floor))
    for a in arms[1:]:
        if a in med:
            d = (med[a] - base) / base
            verdict = "KEPT" if d < -floor else ("WORSE" if d > floor else "WITHIN NOISE")
            print("ARM [%-44s]  %+.2f %% vs arm 0 -> %s" % (a[-44:], 100 * d, verdict))
