#!/usr/bin/env python
"""Turns the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/<tag>/{stats,fetch,write})
into the committed evidence: profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.md and
profiles/r01_traffic.json (HBM bytes per launch, gfx950 FETCH_SIZE correction).

    python tools/make_traffic_json.py r01_c
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def pmc(sub, counter):
    out = {}
    for path in glob.glob(os.path.join(src, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot, cnt = out.get(k, (0.0, 0))
            out[k] = (tot + float(r["Counter_Value"]), cnt + 1)
    return {k: t / c for k, (t, c) in out.items()}


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    kernels[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_corrected": int((2 * f + w) * 1024)}
note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 5 --warmup 2, "
        "mnist.prms B=4096, average per launch. gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE "
        "reports half of the bytes of a coalesced streaming read, so hbm_bytes_corrected = "
        "(2*FETCH + WRITE)*1024. Collected by tools/collect_profiles.sh, tabulated by "
        "tools/make_traffic_json.py (%s)." % tag)
json.dump({"note": note, "kernels": kernels}, open(os.path.join(ROOT, "profiles", "r01_traffic.json"), "w"),
          indent=1)

stats = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
dst = os.path.join(ROOT, "profiles", "%s_mnist_bs4096_kernel_stats.csv" % tag)
with open(stats) as fh, open(dst, "w") as out:
    out.write(fh.read())
line = [l for l in open(os.path.join(src, "stats.log")).read().splitlines() if l.startswith("{")][-1]
with open(os.path.join(ROOT, "profiles", "%s_mnist_bs4096_summary.md" % tag), "w") as md:
    md.write("# %s: rocprofv3 --kernel-trace --stats, `python bench.py --sequential --steps 50 --warmup 5`\n\n" % tag)
    md.write("mnist.prms, 4096 images/step, 1 MI355X.  Kernel durations are averages over all launches of\n"
             "the run (timed steps, warm-up and the roofline leg).  HBM bytes: separate `--pmc FETCH_SIZE` /\n"
             "`--pmc WRITE_SIZE` passes, corrected as in MI355X_MICROARCH.md (see r01_traffic.json).\n\n")
    md.write("| kernel | calls | avg us | % of GPU time | HBM bytes / launch (PMC) |\n|---|---:|---:|---:|---:|\n")
    for r in rows:
        k = short(r["Name"])
        hb = kernels.get(k, {}).get("hbm_bytes_corrected")
        md.write("| `%s` | %s | %.1f | %s | %s |\n" % (k[:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                     r["Percentage"], "%.2f MB" % (hb / 1e6) if hb else "-"))
    md.write("\nbench line of the same run:\n\n```\n%s\n```\n" % line[:1500])
print("wrote", dst)
pipe = glob.glob(os.path.join(src, "stats_pipe", "*kernel_stats.csv"))
if pipe:
    dstp = os.path.join(ROOT, "profiles", "%s_mnist_bs4096_pipelined_kernel_stats.csv" % tag)
    with open(pipe[0]) as fh, open(dstp, "w") as out:
        out.write(fh.read())
    lines = [l for l in open(os.path.join(src, "stats_pipe.log")).read().splitlines() if l.startswith("{")]
    if lines:
        with open(os.path.join(ROOT, "profiles", "%s_mnist_bs4096_pipelined_bench.json" % tag), "w") as out:
            out.write(lines[-1] + "\n")
    print("wrote", dstp)
for cfg in ("cifar_like", "wide6"):
    found = glob.glob(os.path.join(src, "stats_%s" % cfg, "*kernel_stats.csv"))
    if not found:
        continue
    dst2 = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, cfg))
    with open(found[0]) as fh, open(dst2, "w") as out:
        out.write(fh.read())
    lines = [l for l in open(os.path.join(src, "stats_%s.log" % cfg)).read().splitlines() if l.startswith("{")]
    if lines:
        with open(os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (tag, cfg)), "w") as out:
            out.write(lines[-1] + "\n")
    print("wrote", dst2)
