#!/usr/bin/env python
"""Runs ON THE GPU BOX at the end of tools/collect_profiles.sh: shrinks gpurun_out/<tag>/ to what is
copied back (<= 64 MiB): per directory the rocprofv3 kernel_stats.csv (untouched) and, for PMC passes,
counters.csv = per (kernel, counter) mean value and launch count aggregated from
*_counter_collection.csv; the raw traces are deleted.  The bench JSON lines are kept in <name>.json."""
import csv
import glob
import os
import re
import shutil
import sys

out = sys.argv[1]


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


for d in sorted(glob.glob(os.path.join(out, "*/"))):
    name = os.path.basename(d.rstrip("/"))
    keep = {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        keep["kernel_stats.csv"] = open(path).read()
    agg = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            tot, cnt = agg.get(k, (0.0, 0))
            agg[k] = (tot + float(r["Counter_Value"]), cnt + 1)
    if agg:
        lines = ["kernel,counter,mean_per_launch,launches"]
        for (k, c), (t, n) in sorted(agg.items()):
            lines.append('"%s",%s,%.6g,%d' % (k, c, t / n, n))
        keep["counters.csv"] = "\n".join(lines) + "\n"
    shutil.rmtree(d)
    os.makedirs(d)
    for fn, text in keep.items():
        with open(os.path.join(d, fn), "w") as fh:
            fh.write(text)
    log = os.path.join(out, name + ".log")
    if os.path.exists(log):
        js = [l for l in open(log, errors="replace").read().splitlines() if l.startswith("{")]
        if js:
            with open(os.path.join(out, name + ".json"), "w") as fh:
                fh.write(js[-1] + "\n")
        tail = open(log, errors="replace").read()[-2000:]
        with open(log, "w") as fh:
            fh.write(tail)
