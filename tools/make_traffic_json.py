#!/usr/bin/env python
"""Turns the condensed rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/<tag>/...) into the
committed evidence under profiles/:

  <tag>_<config>_kernel_stats.csv   rocprofv3 --kernel-trace --stats, untouched
  <tag>_<config>_summary.md         per kernel: calls, average us, share of GPU time, HBM bytes per launch
                                    (PMC), matrix-core utilisation (PMC)
  <tag>_<config>_bench.json         the bench.py line of the profiled run
  <round>_traffic.json              {config: {kernel: {FETCH_SIZE_KB, WRITE_SIZE_KB, hbm_bytes_corrected,
                                    mfma_busy_frac, mfma_flop_issued}}} -- what bench.py's `roofline.traffic` cites

    python tools/make_traffic_json.py r02_a

HBM bytes: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; gfx950 correction from
MI355X_MICROARCH.md (FETCH_SIZE reports half of the bytes of a coalesced streaming read):
hbm_bytes_corrected = (2*FETCH + WRITE) * 1024.  Matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES
(summed over the 1024 SIMDs) / (1024 * GRBM_GUI_ACTIVE per XCD), i.e. rocprof's MfmaUtil; issued MFMA
FLOP = 512 * (SQ_INSTS_VALU_MFMA_MOPS_F32 + _F16).  PMC passes serialise kernels and run at a lower
clock: use them for bytes and ratios, the kernel-trace pass for durations."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_a"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
N_SIMD, N_XCD = 1024, 8

CONFIGS = {  # name -> (stats dir, pmc prefix, description)
    "mnist_bs4096": ("mnist_seq", "mnist", "mnist.prms, 4096 images/step, one step at a time (bench.py --sequential)"),
    "mnist_bs4096_pipelined": ("mnist_pipe", None, "mnist.prms, 4096 images/step, two steps in flight (default schedule)"),
    "mnist_bs512": ("mnist512_seq", None, "mnist.prms, 512 images/step (one rank of the 8-GPU strong-scaling run), one step at a time"),
    "mnist_bs512_pipelined": ("mnist512_pipe", None, "mnist.prms, 512 images/step, two steps in flight"),
    "cifar_like_f32": ("cifar_like_f32", "cifar_like_f32", "cifar_like.prms, 2048 images/step, fp32, one step at a time (bench.py --sequential)"),
    "cifar_like_f16": ("cifar_like_f16", "cifar_like_f16", "cifar_like.prms, 2048 images/step, DTYPE float16 (fp16-resident tensors), one step at a time"),
    "wide6_f32": ("wide6_f32", "wide6_f32", "wide6.prms 64x64x3, 128 images/step, fp32, one step at a time"),
    "wide6_f16": ("wide6_f16", "wide6_f16", "wide6.prms 64x64x3, 128 images/step, DTYPE float16 (fp16-resident tensors), one step at a time"),
    "cifar_like_f32_pipelined": ("cifar_like_f32_pipe", None, "cifar_like.prms, 2048 images/step, fp32, two steps in flight (default schedule: kernel durations include the other stream's share of the GPU)"),
    "cifar_like_f16_pipelined": ("cifar_like_f16_pipe", None, "cifar_like.prms, 2048 images/step, DTYPE float16, two steps in flight"),
    "wide6_f32_pipelined": ("wide6_f32_pipe", None, "wide6.prms 64x64x3, 128 images/step, fp32, two steps in flight"),
    "wide6_f16_pipelined": ("wide6_f16_pipe", None, "wide6.prms 64x64x3, 128 images/step, DTYPE float16, two steps in flight"),
}


def counters(sub):
    path = os.path.join(src, sub, "counters.csv")
    out = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            out.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_launch"])
    return out


def short(name):
    import re
    return re.sub(r"^void ", "", name).split("(")[0]


traffic = {"note": __doc__.split("HBM bytes:")[1].strip().replace("\n", " "), "tag": tag, "configs": {}}
for cfg, (sdir, pmc, desc) in CONFIGS.items():
    stats_path = os.path.join(src, sdir, "kernel_stats.csv")
    if not os.path.exists(stats_path):
        continue
    with open(stats_path) as fh, open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, cfg)), "w") as out:
        out.write(fh.read())
    rows = list(csv.DictReader(open(stats_path)))
    kern = {}
    if pmc:
        f, w, m = counters(pmc + "_fetch"), counters(pmc + "_write"), counters(pmc + "_mfma")
        for k in sorted(set(f) | set(w) | set(m)):
            rec = {}
            if k in f or k in w:
                fe, wr = f.get(k, {}).get("FETCH_SIZE", 0.0), w.get(k, {}).get("WRITE_SIZE", 0.0)
                rec.update(FETCH_SIZE_KB=fe, WRITE_SIZE_KB=wr, hbm_bytes_corrected=int((2 * fe + wr) * 1024))
            if k in m and m[k].get("GRBM_GUI_ACTIVE"):
                mm = m[k]
                rec["mfma_busy_frac"] = mm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (N_SIMD * mm["GRBM_GUI_ACTIVE"] / N_XCD)
                rec["mfma_flop_issued"] = 512.0 * (mm.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + mm.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0))
            kern[k] = rec
        traffic["configs"][cfg] = kern
    js = os.path.join(src, sdir + ".json")
    line = open(js).read().strip() if os.path.exists(js) else ""
    if line:
        with open(os.path.join(dst, "%s_%s_bench.json" % (tag, cfg)), "w") as out:
            out.write(line + "\n")
    with open(os.path.join(dst, "%s_%s_summary.md" % (tag, cfg)), "w") as md:
        md.write("# %s / %s: rocprofv3 --kernel-trace --stats of `python bench.py ...`\n\n%s, 1 MI355X.\n" % (tag, cfg, desc))
        md.write("Durations: averages over all launches of the kernel-trace run (warm-up, timed steps, the\n"
                 "sync-API loop and the roofline leg).  HBM bytes / matrix-core utilisation: separate PMC passes\n"
                 "(see %s_traffic.json for the method" % tag.split("_")[0] + " and the gfx950 FETCH_SIZE correction).\n\n")
        md.write("| kernel | calls | avg us | % of GPU time | HBM MB / launch (PMC) | MFMA busy (PMC) | MFMA GFLOP issued / launch |\n"
                 "|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            k = short(r["Name"])
            rec = kern.get(k, {})
            hb = rec.get("hbm_bytes_corrected")
            mf = rec.get("mfma_busy_frac")
            fl = rec.get("mfma_flop_issued")
            md.write("| `%s` | %s | %.1f | %s | %s | %s | %s |\n" % (
                k[:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"],
                "%.2f" % (hb / 1e6) if hb else "-", "%.0f %%" % (100 * mf) if mf else "-",
                "%.2f" % (fl / 1e9) if fl else "-"))
        if line:
            md.write("\nbench line of the kernel-trace run:\n\n```\n%s\n```\n" % line[:2500])
    print("wrote", cfg)
json.dump(traffic, open(os.path.join(dst, "%s_traffic.json" % tag.split("_")[0]), "w"), indent=1)
