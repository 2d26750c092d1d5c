#!/usr/bin/env python
"""Headline benchmark: training images/sec (fwd + bwd + update) of the MNIST CNN
(params/mnist.prms, 28x28x1 synthetic, batch 4096 per GPU) on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* come from the launcher (torch itself
is not imported: the hot path is libtheanet_hip.so + RCCL).  Rank 0 prints ONE JSON line.
N > 1 defaults to STRONG scaling as BASELINE.json configs[2] states it ("bs4096 sharded 8 ways"):
the global batch stays 4096 and every rank trains on 4096/N rows, one flat gradient all-reduce per
step.  --scaling weak keeps 4096 rows per GPU instead (global 4096*N).
"""
import argparse
import ast
import copy
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def load_prms(name):
    with open(os.path.join(ROOT, "params", name)) as fh:
        return ast.literal_eval(fh.read())


def synthetic(rows, c, hw):
    x = np.random.default_rng(0).random((rows, c, hw, hw), dtype=np.float32)
    y = np.random.default_rng(1).integers(0, 10, rows).astype(np.int32)
    return x, y


CPU_LEG = r"""
import copy, json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from theanet_amd import NeuralNet
prms, batch, c, hw, budget = %(prms)r, %(batch)d, %(c)d, %(hw)d, %(budget)f
x = np.random.default_rng(0).random((2 * batch, c, hw, hw), dtype=np.float32)
y = np.random.default_rng(1).integers(0, 10, 2 * batch).astype(np.int32)
net = NeuralNet(copy.deepcopy(prms["layers"]), dict(prms["training_params"]))
fn = net.get_trin_model(x, y)
t0 = time.perf_counter(); fn(0); est = time.perf_counter() - t0          # warm-up + step-time estimate
fn(1)
t0, n = time.perf_counter(), 0
while n == 0 or (time.perf_counter() - t0 + est < budget and n < 500):
    fn.enqueue(n %% 2); n += 1
cost = float(fn.fetch()[0])
dt = time.perf_counter() - t0
print("CPULEG " + json.dumps({"steps": n, "seconds": dt, "cost": cost}))
"""


def cpu_baseline(prms, hw, c, batch, budget_s=18.0):
    """The timed CPU baseline: this build's C++/OpenMP backend behind the same C-ABI (theanet_amd/csrc_cpu:
    im2col + blocked SGEMM conv, C loops for pooling, SGEMM for the fully-connected layers -- the algorithms
    Theano's CPU path uses; Theano itself cannot be installed), driving the SAME net at the SAME batch
    size through the same NeuralNet host code, on this box's host cores, in subprocesses with
    THEANET_BACKEND=cpu.  Whole training steps; the OpenMP thread count is SWEPT (8, 16, 32, 64, 128, capped at
    the core count; threads pinned to cores) and the best rate is reported with the sweep beside it: round 2 ran
    128 threads on a 256-core host and was slower than 8 threads of the build container.  A CPU restatement of
    the reference path, not the reference: baseline only."""
    import subprocess
    p = copy.deepcopy({k: v for k, v in prms.items() if not k.startswith("_")})
    p["training_params"]["BATCH_SZ"] = batch
    p["training_params"].pop("DTYPE", None)                     # the reference's floatX: float32
    nproc = os.cpu_count() or 1
    sweep = sorted({min(t, nproc) for t in (8, 16, 32, 64, 128)})
    per_leg = budget_s / len(sweep)
    results, err = {}, ""
    for threads in sweep:
        env = dict(os.environ, THEANET_BACKEND="cpu", OMP_NUM_THREADS=str(threads), OMP_PLACES="cores",
                   OMP_PROC_BIND="close", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        code = CPU_LEG % {"root": ROOT, "prms": p, "batch": batch, "c": c, "hw": hw, "budget": per_leg}
        try:
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            err = "timeout at %d threads" % threads
            continue
        line = [l for l in r.stdout.splitlines() if l.startswith("CPULEG ")]
        if r.returncode != 0 or not line:
            err = (r.stderr or r.stdout)[-300:]
            continue
        rec = json.loads(line[-1][7:])
        results[threads] = (batch * rec["steps"] / rec["seconds"], rec["steps"], rec["seconds"])
    if not results:
        return {"value": None, "unit": "images/sec", "cores": sweep[-1], "kind": "port",
                "sample": "CPU backend leg failed: " + err}
    best = max(results, key=lambda t: results[t][0])
    return {"value": results[best][0], "unit": "images/sec", "cores": best, "threads": best, "nproc": nproc,
            "kind": "port", "thread_sweep_images_per_sec": {str(t): round(v[0], 1) for t, v in sorted(results.items())},
            "sample": "C++/OpenMP CPU backend behind the same C-ABI (lib/libtheanet_cpu.so, THEANET_BACKEND=cpu; "
                      "a CPU restatement of the reference path -- Theano is not installable), %s fwd+bwd+update at "
                      "batch %d; OpenMP threads swept over %s (OMP_PLACES=cores), best = %d threads: %d steps in %.1f s"
                      % (prms.get("_name", "net"), batch, sweep, best, results[best][1], results[best][2])}


PMC_FILE = "r06_traffic.json"


def _pmc_table(config):
    """{kernel: {hbm_bytes_corrected, mfma_busy_frac, ...}} of one profiled configuration (profiles/r0N_traffic.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as fh:
            return json.load(fh)["configs"].get(config, {})
    except (OSError, KeyError, ValueError):
        return {}


DEFAULTS = {"mnist.prms": (4096, 28, 1), "cifar_like.prms": (2048, 32, 3),
            "wide6.prms": (128, 64, 3), "3flat.prms": (4096, 28, 1)}
# BASELINE.json configs[3] and configs[4] (one GPU's share of it), reported by the default N=1 run beside the headline
OTHER_CONFIGS = (("cifar_like.prms", "f32"), ("cifar_like.prms", "f16"), ("wide6.prms", "f32"), ("wide6.prms", "f16"))


def plan_batches(prms_name, batch, scaling, world_size):
    """(global batch, rows per GPU, scaling label).  mnist / cifar_like: N > 1 shards the stated batch (strong
    scaling, BASELINE configs[2]) unless --scaling weak.  wide6 is stated per node -- "bs1024, 8 x MI355X",
    configs[4] -- i.e. 128 images per GPU: its default keeps 128 per GPU at every N (weak by definition)."""
    dB = DEFAULTS.get(prms_name, (4096, 28, 1))[0]
    base = batch or dB
    if prms_name == "wide6.prms" and not batch:
        return base * world_size, base, "weak"
    if world_size == 1:
        return base, base, "weak"                 # N = 1: the two coincide
    if scaling == "strong":
        assert base % world_size == 0, "batch %d does not divide over %d GPUs" % (base, world_size)
        return base, base // world_size, "strong"
    return base * world_size, base, "weak"


def build(prms_name, global_batch, per_gpu, img, dtype, group=None, n_batches=None):
    from theanet_amd import NeuralNet
    prms = load_prms(prms_name)
    prms["_name"] = prms_name
    _, dimg, C = DEFAULTS.get(prms_name, (4096, 28, 1))
    img = img or dimg
    C = prms["layers"][0][1].get("num_maps", C)
    prms["layers"][0][1]["img_sz"] = img
    tr = prms["training_params"]
    tr["SEED"] = 555555
    tr["BATCH_SZ"] = global_batch
    if dtype == "f16":
        tr["DTYPE"] = "float16"
    if dtype == "b3":
        tr["MATMUL"] = "bf16x3"
    net = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr))
    if group is not None:
        net._dev_group = group              # a second net of the same launch: ONE communicator per process
    if n_batches is None:
        n_batches = max(2, 65536 // global_batch) if per_gpu * img * img * C < (1 << 24) else 2
    x, y = synthetic(n_batches * global_batch, C, img)
    fn = net.get_trin_model(x, y)
    return prms, tr, net, fn, n_batches, img, C


def weak_leg(ctx, args, world, group, per_gpu):
    """The weak-scaling form of an N-rank run: ``per_gpu`` rows per rank (global batch N * per_gpu), same harness as
    the headline region (set-up steps, warm-up, exactly --steps enqueued steps between barriers, max over ranks)."""
    gb = per_gpu * world.size
    prms, tr, net, fn, n_batches, img, C = build(args.prms, gb, per_gpu, args.img, args.dtype, group=group)
    n_setup = settle(ctx, fn, n_batches, group)
    for i in range(args.warmup):
        fn.enqueue(i % n_batches)
    ctx.sync()
    group.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        fn.enqueue(i % n_batches)
    ctx.sync()
    group.barrier()
    dt = group.rdzv.gather_max(time.perf_counter() - t0)
    cost = float(fn.fetch()[0])
    assert np.isfinite(cost), "training diverged (weak leg)"
    rec = {"value": gb * args.steps / dt, "unit": "images/sec", "scaling": "weak", "global_batch": gb,
           "rows_per_gpu": per_gpu, "steps": args.steps, "setup_steps": n_setup, "ms_per_step": 1e3 * dt / args.steps,
           "dp_schedule": getattr(net, "dp_schedule", None), "final_cost": cost}
    del fn, net
    return rec


def settle(ctx, fn, n_batches, group, min_seconds=0.2):
    """Untimed set-up steps (the twin net of the two-steps-in-flight schedule is built on the first call; the GPU
    comes out of seconds of idle while the net was built: 48 + 5 steps = 10 ms of load left the first 20 timed
    steps at 193-196 us, 180 ms of load at 180-182 us, the steady state being 177 -- tools/probe_start.py: a 50 ms
    host pause alone costs the next 20 steps 25 %).  So the set-up runs for at least 0.2 s, synchronised.  Every
    rank must take the same number of steps (each one is a collective): the first 48 are timed and the count
    that fills 0.2 s is the maximum over ranks.  Returns the number of steps taken."""
    for i in range(32):                 # (the first call builds the twin net: not representative)
        fn.enqueue(i % n_batches)
    ctx.sync()
    t_setup = time.perf_counter()
    for i in range(32, 48):
        fn.enqueue(i % n_batches)
    ctx.sync()
    per_step = max((time.perf_counter() - t_setup) / 16, 1e-6)
    more = float(min(20000, max(0, int(min_seconds / per_step) - 48)))
    if group is not None:
        more = group.rdzv.gather_max(more)
    n_setup = 48 + 16 * ((int(more) + 15) // 16)
    for i in range(48, n_setup):
        fn.enqueue(i % n_batches)
        if i % 16 == 15:
            ctx.sync()
    return n_setup


def conv_roofline(ctx, net, fn, n_batches, nsteps, prms_name, dtype):
    """conv nets (cifar_like, wide6): every conv product of the step, grouped into forward and backward, each as
    the sum over its launches -> "fraction of the conv roofline" (conv-layer FLOPs / conv-kernel
    time / MFMA peak; peak = fp32 MFMA for fp32 operands, fp16 MFMA for fp16 operands).  HIP events on the
    stream the kernels run on, one step at a time.  Fused conv+act+pool blocks count with their conv FLOPs only
    (pooling, masks and activations ride along).  Returns the two records, slowest first."""
    from theanet_amd import roofline
    from theanet_amd.layer import ConvLayer
    convs = [l for l in net.tr_layers if isinstance(l, ConvLayer)]
    first_param = next(l for l in net.tr_layers if getattr(l, "params", None))
    fl_of = lambda l: 2 * l.batch_sz * l.out_sz ** 2 * l.num_maps * l.num_prev_maps * l.filter_sz ** 2
    groups = (("conv forward, all conv layers (conv_tile / convpool kernels)", net.CONV_FWD_OPS,
               sum(fl_of(l) for l in convs)),
              ("conv backward, all conv layers (weight + input gradients)", net.CONV_BWD_OPS,
               sum(fl_of(l) * (1 if l is first_param else 2) for l in convs)))
    recs = []
    f16 = dtype == "f16"
    peak = roofline.MFMA_F16_PEAK_TFLOPS if f16 else roofline.MFMA_F32_PEAK_TFLOPS
    # DTYPE float16: one launch per conv layer and direction, in layer order (forward) / reverse order (gradients):
    # the per-launch HIP-event times give one record per layer and direction (`per_layer`)
    order = {"tn_c8_conv_fwd": ("forward", convs), "tn_c8_conv_wgrad": ("weight gradient", convs[::-1]),
             "tn_c8_conv_dgrad": ("input gradient", [l for l in convs[::-1] if l is not first_param])}
    for label, ops, fl in groups:
        ms, launches, per_layer = 0.0, 0, []
        for op in ops:
            ctx.time_calls(op, 0)
            for i in range(nsteps):
                ctx.new_step()
                fn.enqueue(i % n_batches)
            ctx.sync()
            times = ctx.collect_times_ms()
            ms += float(np.sum(times)) / nsteps
            launches += len(times) // nsteps
            L = len(times) // nsteps
            if f16 and op in order and L == len(order[op][1]) and L * nsteps == len(times):
                avg = np.asarray(times).reshape(nsteps, L).mean(axis=0)
                for lyr, t_ms in zip(order[op][1], avg):
                    gf = fl_of(lyr) / 1e9
                    per_layer.append({"layer": "conv%d %d->%d @%dx%d%s" % (convs.index(lyr) + 1, lyr.num_prev_maps, lyr.num_maps,
                                                                          lyr.in_sz, lyr.in_sz,
                                                                          " +pool" if getattr(lyr, "fused_pool", None) else ""),
                                      "direction": order[op][0], "ms": float(t_ms), "gflop": gf,
                                      "tflops": gf / float(t_ms), "frac": gf / float(t_ms) / peak})
        if not launches or not fl:
            continue
        ach = fl / (ms * 1e-3) / 1e12
        rec = {"kernel": label, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
               "frac": ach / peak, "traffic": None, "ms_per_step": ms, "launches_per_step": launches,
               "flops_per_step": fl}
        if f16:
            rec["frac_of_fp32_peak"] = ach / roofline.MFMA_F32_PEAK_TFLOPS
            if "backward" in label:
                rec["note"] = ("weight-gradient launches take half the CUs by design (DESIGN.md 4.3): timed alone here the "
                               "other half idles, in the step the other stream's launches run there")
        if per_layer:
            rec["per_layer"] = per_layer
        recs.append(rec)
    recs.sort(key=lambda r: -r["ms_per_step"])
    return recs


def other_config_leg(ctx, prms_name, dtype, steps):
    """One of BASELINE.json's other single-GPU configurations through the same harness, in-process: set-up steps,
    a timed region of enqueue-only steps bracketed by stream syncs (>= 0.25 s or ``steps``), then the conv
    roofline legs.  No CPU baseline, no sync-API leg."""
    import gc
    from theanet_amd import roofline
    gb, per_gpu, _ = plan_batches(prms_name, 0, "weak", 1)
    t_build = time.perf_counter()
    prms, tr, net, fn, n_batches, img, C = build(prms_name, gb, per_gpu, 0, dtype)
    n_setup = settle(ctx, fn, n_batches, None, 0.15)
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(8):
        fn.enqueue(i % n_batches)
    ctx.sync()
    est = (time.perf_counter() - t0) / 8
    n = int(max(steps, min(2000, 0.25 / max(est, 1e-6))))
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(n):
        fn.enqueue(i % n_batches)
    ctx.sync()
    dt = time.perf_counter() - t0
    cost = float(fn.fetch()[0])
    legs = conv_roofline(ctx, net, fn, n_batches, 6, prms_name, dtype)
    step_flops = roofline.net_step_flops(net)
    first = net.tr_layers[0]
    stage = "elastic stage on" if type(first).__name__ == "ElasticLayer" and first.active else \
        "no input distortion (%s)" % type(first).__name__
    rec = {"config": {"workload": "params/%s %dx%dx%d synthetic, batch %d on 1 GPU, %s"
                                  % (prms_name, img, img, C, gb, stage),
                      "global_batch": gb,
                      "schedule": "two steps in flight" if type(fn).__name__ == "_PipeTrainFn" and fn._twin is not None
                      else "one step at a time",
                      "step_gflop_algorithmic": step_flops / 1e9,
                      "step_tflops_algorithmic": step_flops / (dt / n) / 1e12},
           "dtype": DTYPE_LABEL[dtype], "steps": n, "setup_steps": n_setup, "ms_per_step": 1e3 * dt / n,
           "value": gb * n / dt, "unit": "images/sec", "final_cost": cost,
           "conv_roofline": legs, "measured_in": "conv legs: one step at a time, HIP events on the kernels' stream",
           "seconds_incl_build": None}
    assert np.isfinite(cost), "training diverged (%s %s)" % (prms_name, dtype)
    del fn, net
    gc.collect()
    rec["seconds_incl_build"] = time.perf_counter() - t_build
    return rec


def shard_leg(ctx, steps, rows=512):
    """What ONE rank of BASELINE configs[2] does per step (mnist.prms, batch 4096 sharded 8 ways = 512 rows), timed on
    this GPU through the data-parallel code path: TN_DP_FORCE=1 builds the net with a 1-rank RCCL communicator, so the
    step is the N > 1 step -- two steps in flight, the gradient bucket's ncclAllReduce on the communication stream, the
    update waiting for its event -- with the collective's launch cost inside and its wire time absent.  The ratio
    t(4096 rows) / t(512 rows) is the CEILING of the 8-GPU strong-scaling factor (a free interconnect); no scaling
    curve is measured here."""
    import gc
    old = os.environ.get("TN_DP_FORCE")
    os.environ["TN_DP_FORCE"] = "1"
    try:
        prms, tr, net, fn, n_batches, img, C = build("mnist.prms", rows, rows, 0, "f32")
        n_setup = settle(ctx, fn, n_batches, None, 0.15)
        ctx.sync()
        t0 = time.perf_counter()
        for i in range(16):
            fn.enqueue(i % n_batches)
        ctx.sync()
        est = (time.perf_counter() - t0) / 16
        n = int(max(steps, min(6000, 0.3 / max(est, 1e-6))))
        ctx.sync()
        t0 = time.perf_counter()
        for i in range(n):
            fn.enqueue(i % n_batches)
        ctx.sync()
        dt = time.perf_counter() - t0
        cost = float(fn.fetch()[0])
        assert np.isfinite(cost), "training diverged (shard leg)"
        rec = {"cfg": "mnist.prms", "rows": rows, "dtype": "f32", "ms_per_step": round(1e3 * dt / n, 5), "steps": n,
               "setup_steps": n_setup, "value": rows * n / dt,
               "schedule": "two steps in flight" if type(fn).__name__ == "_PipeTrainFn" and fn._twin is not None
               else "one step at a time",
               "dp": "1-rank RCCL communicator (TN_DP_FORCE=1): %s schedule, %d collective(s) per step on the communication "
                     "stream" % (getattr(net, "dp_schedule", "?"), 2 if getattr(net, "_dp_bucket", None) is not None else 1),
               "what": "one rank's share of BASELINE configs[2] (batch 4096 sharded 8 ways), wire time absent"}
        del fn, net
        gc.collect()
        return rec
    finally:
        if old is None:
            os.environ.pop("TN_DP_FORCE", None)
        else:
            os.environ["TN_DP_FORCE"] = old


DTYPE_LABEL = {"f32": "f32", "f16": "f16 (fp16 tensors and MFMA operands, fp32 accumulate, fp32 master weights)",
               "b3": "f32 tensors; dense products as six bf16 MFMA products of exactly split operands (MATMUL 'bf16x3', opt-in)"}


def dry_multi(args):
    """--dry-multi N: what an N-rank launch of this command line WOULD run -- per-rank row shards, the flat
    gradient buffer every rank all-reduces, the schedule -- built without a communicator (and without a second
    process), so the scaling command line can be exercised where there is one GPU or none (THEANET_BACKEND=cpu)."""
    from theanet_amd import comm
    N = args.dry_multi
    gb, per_gpu, scaling = plan_batches(args.prms, args.batch, args.scaling, N)
    ranks = []
    plan = None
    for r in (0, N - 1):
        comm._world = comm.World(r, N, r, dry=True)
        prms, tr, net, fn, n_batches, img, C = build(args.prms, gb, per_gpu, args.img, args.dtype)
        ranks.append({"rank": r, "rows_of_minibatch": [net.shard_lo, net.shard_lo + net.local_bsz],
                      "first_dataset_row_of_minibatch_3": comm.minibatch_row0(3, gb, N, r)})
        if plan is None:
            tensors = []
            for i, lyr in enumerate(net.tr_layers):
                for p, g in zip(lyr.params, lyr.grads or ()):
                    tensors.append({"layer": i, "type": type(lyr).__name__, "shape": list(p.shape),
                                    "offset_floats": (g.ptr - net.flat_grads.ptr) // 4})
            bk = getattr(net, "_dp_bucket", None)
            buckets = [{"floats": net.n_flat, "offset_floats": 0, "issued": "end of the backward pass",
                        "holds": "every gradient + the cost"}] if bk is None else \
                [{"floats": net.n_flat - bk[1], "offset_floats": bk[1],
                  "issued": "right after the dense layers' backward pass (layer %d down), beside the conv backward" % bk[0],
                  "holds": "dense-layer gradients + the cost"},
                 {"floats": bk[1], "offset_floats": 0, "issued": "end of the backward pass",
                  "holds": "conv-layer gradients"}]
            for b_ in buckets:
                b_["bytes"] = 4 * b_["floats"]
                b_["algorithm"] = {"rsag": "direct reduce-scatter + all-gather (tn_allreduce_sum_rsag: ncclReduceScatter + "
                                           "ncclAllGather in place, every element summed once at its owner)",
                                   "allreduce": "one ncclAllReduce (latency-bound bucket: the library's own algorithm)"
                                   }[comm.collective_algo(b_["floats"], N)]
            plan = {"flat_gradient_buffer": {"floats_reduced_per_step": net.n_flat, "bytes": 4 * net.n_flat,
                                             "cost_slot": net.n_flat - 1, "tensors": tensors},
                    "allreduce_buckets": buckets,
                    "allreduce_stream": "the context's communication stream (tn_allreduce_sum_async / _rsag); consumer = "
                                        "the update that opens the same stream's next step",
                    "multi_gpu_measurement": "none in any round (no multi-GPU lease): the algorithm per bucket is a design "
                                             "choice from the xGMI figures of SURVEY.md 8e, not a measured one",
                    "schedule": "pipelined (two steps in flight, all-reduce buckets on the communication stream)"
                    if type(fn).__name__ == "_PipeTrainFn" else
                    "one step at a time; all-reduce schedule autotuned among %s" %
                    (["plain"] + (["overlap"] if net._dp_cand else []) + (["delayed"] if net._dp_can_delay else [])),
                    "n_batches_resident": n_batches}
        del fn, net
    comm._world = None
    print(json.dumps({"dry_multi": N, "prms": args.prms, "dtype": args.dtype, "scaling": scaling, "global_batch": gb,
                      "rows_per_gpu": per_gpu, "ranks_shown": ranks, **plan,
                      "launch": "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 "
                                "--master-port P bench.py --gpus %d --steps K --warmup W" % (N, N)}))


SELFCHECK_PLAIN = {"TN_PIPELINE": "0", "TN_DP_PIPELINE": "0", "TN_DP_BUCKETS": "0", "TN_DP_ALGO": "allreduce", "TN_DP_OVERLAP": "0"}


def dp_selfcheck_child():
    """One rank of the start-up check of an N > 1 run (its own process and communicator, so that a hang here cannot hang
    the benchmark): cifar_like.prms (two gradient buckets) at 8 rows per rank, 6 steps under the DEFAULT schedule -- two
    steps in flight, the buckets on the communication stream, every bucket forced into its reduce-scatter + all-gather
    form -- and 6 steps under the PLAIN one (one step at a time, one ncclAllReduce after the backward pass) from the same
    SEED, device RNG on.  Replicas must agree bit for bit within a schedule (comm.agree raises otherwise); the two
    schedules sum in different orders, so their weight checksums are compared to 1e-5.  Rank 0 prints the record (one
    line starting with "{": the only kind this script lets through to its stdout)."""
    from theanet_amd import comm
    world = comm.get_world()
    hang = os.environ.get("TN_TEST_SELFCHECK_HANG")
    sums = {}
    for label, env in (("default", {"TN_DP_RSAG_MIN_BYTES": "4096"}), ("plain", SELFCHECK_PLAIN)):
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            gb = 8 * world.size
            prms, tr, net, fn, n_batches, img, C = build("cifar_like.prms", gb, 8, 0, "f32", n_batches=6)
            for i in range(6):
                fn.enqueue(i % n_batches)
                if hang is not None and int(hang) == world.rank and label == "default" and i == 3:
                    time.sleep(3600)            # (tests: a rank that stops issuing collectives)
            cost = float(fn.fetch()[0])
            chk = float(sum(np.float64(w.astype(np.float64).sum()) for l in net.tr_layers for w in l.get_wts()))
            comm.agree(chk, "the weights after 6 steps of the %s schedule (checksum)" % label, net._group().rdzv)
            net._group().verify_order()
            sums[label] = {"checksum": chk, "cost": cost, "schedule": str(getattr(net, "dp_schedule", "?")),
                           "collectives": net._group().n_issued}
            del fn, net
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    a, b = sums["default"]["checksum"], sums["plain"]["checksum"]
    ok = abs(a - b) <= 1e-5 * max(1.0, abs(b)) and np.isfinite(a)
    if world.rank == 0:
        print(json.dumps({"selfcheck": {"ok": bool(ok), "rel_diff": abs(a - b) / max(1.0, abs(b)), **sums}}), flush=True)
    return 0 if ok else 3


def dp_selfcheck(world):
    """N > 1, before this process touches the GPU: every rank runs dp_selfcheck_child() in a CHILD process (rendezvous on
    MASTER_PORT + 110) and waits for it at most TN_BENCH_SELFCHECK_TIMEOUT seconds (default 180).  A child that fails,
    disagrees or does not come back (it is killed by its exact pid) on ANY rank makes EVERY rank fall back, loudly, to
    the plain schedule for the measured run.  Returns the record for the JSON line (`dp_selfcheck`)."""
    import subprocess
    from theanet_amd import comm
    if os.environ.get("TN_BENCH_SELFCHECK", "1") == "0":
        return {"ok": None, "skipped": "TN_BENCH_SELFCHECK=0"}
    limit = float(os.environ.get("TN_BENCH_SELFCHECK_TIMEOUT", 180))
    env = dict(os.environ, MASTER_PORT=str(world.master_port + 110))
    t0 = time.perf_counter()
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--dp-selfcheck-child"], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    rec, why = None, ""
    try:
        out, err = child.communicate(timeout=limit)
        for l in out.splitlines():
            if l.startswith("{") and '"selfcheck"' in l:
                rec = json.loads(l)["selfcheck"]
        if child.returncode != 0:
            why = "rank %d: the check exited with %d: %s" % (world.rank, child.returncode, (err or out)[-300:].replace("\n", " | "))
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        why = "rank %d: the check did not finish within %.0f s (killed)" % (world.rank, limit)
    bad = comm.get_rendezvous().gather_max(1.0 if why else 0.0)
    out = {"ok": not bad, "seconds": round(time.perf_counter() - t0, 1), "what": "cifar_like.prms, 8 rows per rank, 6 steps: "
           "pipelined + bucketed + reduce-scatter/all-gather on the communication stream against one step at a time with "
           "one all-reduce, same SEED; replicas bit-identical within a schedule, checksums of the two schedules to 1e-5"}
    if rec:
        out.update({k: rec[k] for k in ("rel_diff", "default", "plain") if k in rec})
    if bad:
        out["error"] = why or "another rank's check failed or timed out"
        out["fallback"] = "plain schedule for the measured run: " + " ".join("%s=%s" % kv for kv in SELFCHECK_PLAIN.items())
        os.environ.update(SELFCHECK_PLAIN)
        print("bench.py: WARNING: the data-parallel self-check FAILED (%s): the measured run uses the plain schedule "
              "(one step at a time, one ncclAllReduce per step)" % out["error"], file=sys.stderr, flush=True)
    return out


REAL_STDOUT_FD = None        # the launcher's stdout (set under __main__: fd 1 itself is pointed at stderr for the run)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prms", default="mnist.prms")
    ap.add_argument("--batch", type=int, default=0, help="batch of the N=1 workload (default: 4096 for mnist)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = the global batch stays --batch, every rank takes batch/N rows "
                         "(BASELINE configs[2]); weak = --batch rows per GPU")
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32",
                    help="f16: fp16 operands / fp32 accumulation for the conv products (DTYPE float16)")
    ap.add_argument("--img", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel roofline leg")
    ap.add_argument("--no-weak-leg", action="store_true",
                    help="N > 1, strong scaling: skip the weak-scaling leg (value_weak) the same launch appends")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the cifar_like / wide6 legs the default N=1 mnist run appends (other_configs)")
    ap.add_argument("--dry-multi", type=int, default=0, metavar="N",
                    help="print the plan of an N-rank run (shards, flat gradient layout, schedule) and exit")
    ap.add_argument("--sequential", action="store_true",
                    help="one step at a time (TN_PIPELINE=0) instead of two steps in flight")
    ap.add_argument("--time-op", default="", help="C-ABI function to bracket with HIP events, "
                    "e.g. tn_fc_wgrad:1 (nth call inside a step)")
    ap.add_argument("--dp-selfcheck-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.dp_selfcheck_child:
        sys.exit(dp_selfcheck_child())

    from theanet_amd import comm, roofline
    from theanet_amd.device import get_context

    if args.sequential:
        os.environ["TN_PIPELINE"] = "0"
    if args.dry_multi:
        return dry_multi(args)
    world = comm.get_world()
    assert world.size == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    global_batch, per_gpu, scaling = plan_batches(args.prms, args.batch, args.scaling, world.size)
    selfcheck = dp_selfcheck(world) if world.size > 1 else None

    # N > 1: a rank stuck in a collective is not an exception.  A watchdog thread (ctypes calls release the interpreter
    # lock) prints whatever has been measured so far -- with "error": "rank stuck in <phase>" -- and leaves, so that the
    # first multi-GPU run yields a line whatever happens.  TN_BENCH_TIMEOUT seconds (default 600) for everything up to
    # and including the train-loop leg.
    import threading
    progress = {"phase": "building the net", "line": None}

    def stuck():
        ph = progress["phase"]
        if world.rank == 0:
            rec = progress["line"] or {"metric": "training images/sec (fwd+bwd+update) MNIST-CNN bs4096, 1/2/4/8 MI355X",
                                       "value": None, "unit": "images/sec", "n_gpus": world.size, "steps": args.steps,
                                       "warmup": args.warmup, "higher_is_better": True, "scaling": scaling}
            rec = dict(rec, error="rank stuck in: %s (no progress within %s s; TN_BENCH_TIMEOUT)" % (ph, limit_main),
                       dp_selfcheck=selfcheck)
            os.write(REAL_STDOUT_FD if REAL_STDOUT_FD is not None else 1, (json.dumps(rec) + "\n").encode())
        os._exit(0 if progress["line"] else 4)

    limit_main = float(os.environ.get("TN_BENCH_TIMEOUT", 600))
    dog_main = threading.Timer(limit_main, stuck) if world.size > 1 else None
    if dog_main is not None:
        dog_main.daemon = True
        dog_main.start()

    ctx = get_context()
    prms, tr, net, fn, n_batches, img, C = build(args.prms, global_batch, per_gpu, args.img, args.dtype)
    group = net._group() if world.size > 1 else None

    def barrier():
        ctx.sync()
        if group is not None:
            group.barrier()

    # data-parallel runs first let the schedule autotune finish (untimed set-up steps: it times a few
    # steps of each all-reduce schedule and every rank adopts the fastest, NeuralNet._dp_tune_tick)
    tune_steps = 0
    while getattr(net, "_dp_tune", None) is not None and tune_steps < 1000:
        fn.enqueue(tune_steps % n_batches)
        tune_steps += 1
    progress["phase"] = "set-up steps (schedule %s)" % getattr(net, "dp_schedule", "single GPU")
    setup_steps = tune_steps + settle(ctx, fn, n_batches, group)
    progress["phase"] = "warm-up steps"
    for i in range(args.warmup):
        fn.enqueue(i % n_batches)
    barrier()
    progress["phase"] = "the timed region (%d steps, schedule %s)" % (args.steps, getattr(net, "dp_schedule", "single GPU"))
    hang = os.environ.get("TN_TEST_BENCH_HANG")
    t0 = time.perf_counter()
    for i in range(args.steps):
        fn.enqueue(i % n_batches)
        if hang is not None and int(hang) == world.rank and i == args.steps // 2:
            time.sleep(3600)                    # (tests: a rank that stops issuing its collectives)
    barrier()
    dt = time.perf_counter() - t0
    if group is not None:
        dt = group.rdzv.gather_max(dt)
    cost = fn.fetch()[0]
    assert np.isfinite(cost), "training diverged"
    # (what the watchdog prints should a LATER leg hang: the headline is measured)
    progress["line"] = {"metric": "training images/sec (fwd+bwd+update) MNIST-CNN bs4096, 1/2/4/8 MI355X",
                        "value": tr["BATCH_SZ"] * args.steps / dt, "unit": "images/sec", "n_gpus": world.size,
                        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": DTYPE_LABEL[args.dtype],
                        "data": "synthetic", "config": {"workload": "params/%s, global batch %d = %d images/GPU/step x %d"
                                                        % (args.prms, tr["BATCH_SZ"], per_gpu, world.size),
                                                        "dp_schedule": getattr(net, "dp_schedule", None)}}
    progress["phase"] = "the legs behind the timed region (sustained / sync API / train loop)"

    # The driver's --steps can make the timed region a few milliseconds: a second, longer loop of the
    # same enqueue-only steps (>= 0.5 s) is reported beside it as `sustained`.
    sustained = None
    if dt < 0.05:
        n_sus = int(min(50000, max(args.steps, 0.6 / max(dt / args.steps, 1e-6))))
        barrier()
        t1 = time.perf_counter()
        for i in range(n_sus):
            fn.enqueue(i % n_batches)
        barrier()
        dt_sus = time.perf_counter() - t1
        if group is not None:
            dt_sus = group.rdzv.gather_max(dt_sus)
        sustained = {"steps": n_sus, "seconds": dt_sus, "ms_per_step": 1e3 * dt_sus / n_sus,
                     "value": tr["BATCH_SZ"] * n_sus / dt_sus}
    # The drop-in call fn(i) of the reference's loop (train.py:211: cost, features, logprob come back
    # every step, i.e. a device sync + a D2H of B x n_out floats per step) -- its throughput, untimed above.
    n_sync = int(min(2000, max(20, 0.3 / max(dt / args.steps, 1e-6))))
    barrier()
    t1 = time.perf_counter()
    for i in range(n_sync):
        fn(i % n_batches)
    barrier()
    dt_sync = time.perf_counter() - t1
    if group is not None:
        dt_sync = group.rdzv.gather_max(dt_sync)
    value_sync_api = tr["BATCH_SZ"] * n_sync / dt_sync
    # What THIS build's train.py loop does instead: step_cost(i) -- the step is enqueued and the costs that have arrived
    # (each step's 4 bytes travel to page-locked memory by themselves) are summed a few calls late; nothing waits.
    n_loop = int(min(50000, max(args.steps, 0.3 / max(dt / args.steps, 1e-6))))
    barrier()
    t1 = time.perf_counter()
    tot, seen = 0.0, 0
    for i in range(n_loop):
        for _, c in fn.step_cost(i % n_batches):
            tot += float(c)
            seen += 1
    for _, c in fn.drain_costs():
        tot += float(c)
        seen += 1
    barrier()
    dt_loop = time.perf_counter() - t1
    if group is not None:
        dt_loop = group.rdzv.gather_max(dt_loop)
    assert seen == n_loop and np.isfinite(tot), "the cost ring lost a step"
    value_train_loop = tr["BATCH_SZ"] * n_loop / dt_loop
    if dog_main is not None:
        dog_main.cancel()

    # ---- per-kernel roofline leg: HIP events (on the stream the kernel runs on) around the
    # heavy kernels of the SAME workload; the dominant one (largest share of the step) is
    # reported as `roofline`, the rest as `roofline_others`.
    roof, others = None, []
    if args.no_roofline or os.environ.get("THEANET_BACKEND", "hip") == "cpu":
        pass                                    # the roofline leg times HIP kernels
    elif args.prms == "mnist.prms":
        conv2, fc1 = net.tr_layers[3], net.tr_layers[5]
        B_ = per_gpu
        # algorithmic FLOPs / bytes per launch
        cb_flops = 2 * 2 * B_ * conv2.out_sz ** 2 * conv2.num_maps * conv2.num_prev_maps * 9   # wgrad + dgrad
        pooled = conv2.num_maps * conv2.fused_pool.out_sz ** 2
        # x and g read, dx written (f32), pooling mask read (u8); leaky-relu: y is not read
        cb_bytes = B_ * (4 * (2 * conv2.num_prev_maps * conv2.in_sz ** 2 + pooled) + pooled)
        fl_fc, by_fc = roofline.kernel_cost("fc_fwd", B=B_, n_in=fc1.n_in, n_out=fc1.n_out)
        specs = [("tn_convblock_bwd_mask", 1,
                  "convblock_bwd_mask_mfma (conv2 block backward from the pooling mask: wgrad + dgrad "
                  "on 16x16x4 f32 MFMA, dz in LDS only)", cb_flops, cb_bytes),
                 ("tn_fc_fwd_dropout", 1, "gemm_f32_dma (fc1 forward 4096x720x500, LDS-DMA staged, bias + act + "
                  "inline dropout epilogue)", fl_fc, by_fc + B_ * fc1.n_out),
                 ("tn_fc_bwd", 1, "gemm_f32_pair_dma (fc1 weight gradient, split-K, + input gradient in one "
                  "launch)", 2 * fl_fc, 2 * by_fc)]
        if args.time_op:
            op, nth = (args.time_op.split(":") + ["1"])[:2]
            specs = [(op, int(nth), op, 0, 0)]
        for op, nth, label, fl, by in specs:
            ctx.time_calls(op, nth)
            for i in range(min(args.steps, 40)):
                ctx.new_step()
                fn.enqueue(i % n_batches)
            ctx.sync()
            times = ctx.collect_times_ms()
            if not times:
                continue
            avg_ms = float(np.mean(times))
            ach = fl / (avg_ms * 1e-3) / 1e12
            others.append({"kernel": label, "bound": "mfma", "achieved": ach,
                           "peak": roofline.MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / roofline.MFMA_F32_PEAK_TFLOPS, "traffic": None,
                           "avg_launch_ms": avg_ms, "flops_per_launch": fl, "bytes_per_launch": by,
                           "algorithmic_GBps": by / (avg_ms * 1e-3) / 1e9})
        if others:
            others.sort(key=lambda r: -r["avg_launch_ms"])
            roof, others = others[0], others[1:]
            # the per-kernel leg always runs one step at a time: with two steps in flight the launches
            # of the two streams share the GPU and a kernel's own duration cannot be separated
            roof["measured_in"] = "one-step-at-a-time schedule (bench.py --sequential)"
            # HBM traffic / matrix-core utilisation of these kernels: measured offline with rocprofv3 --pmc
            # (separate FETCH_SIZE / WRITE_SIZE / MFMA passes, gfx950 FETCH correction) by
            # tools/collect_profiles.sh, tabulated in profiles/r0N_traffic.json
            pmc = _pmc_table("mnist_bs4096")
            for rec in [roof] + others:
                key = rec["kernel"].split(" ")[0]
                for name, vals in pmc.items():
                    if name.startswith(key):
                        rec["traffic"] = vals.get("hbm_bytes_corrected")
                        rec["mfma_busy_frac_pmc"] = vals.get("mfma_busy_frac")
                        rec["traffic_source"] = "profiles/%s (rocprofv3 --pmc)" % PMC_FILE
    else:
        legs = conv_roofline(ctx, net, fn, n_batches, min(args.steps, 10), args.prms, args.dtype)
        if legs:
            roof, others = legs[0], legs[1:]
            pmc = _pmc_table("%s_%s" % (args.prms.split(".")[0], args.dtype))
            conv = {k: v for k, v in pmc.items() if k.startswith(("conv_tile", "convpool", "conv_", "c8_"))}
            if conv:
                roof["pmc_per_kernel"] = {k: {"hbm_MB_per_launch": round(v.get("hbm_bytes_corrected", 0) / 1e6, 2),
                                              "mfma_busy_frac": round(v.get("mfma_busy_frac", 0.0), 3)}
                                          for k, v in conv.items()}
                roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc, per launch)" % PMC_FILE

    value = tr["BATCH_SZ"] * args.steps / dt
    step_flops = roofline.net_step_flops(net) * world.size
    first = net.tr_layers[0]
    stage = "elastic stage on" if type(first).__name__ == "ElasticLayer" and first.active else \
        "no input distortion (%s)" % type(first).__name__
    line = {
        "metric": "training images/sec (fwd+bwd+update) MNIST-CNN bs4096, 1/2/4/8 MI355X",
        "value": value, "unit": "images/sec", "n_gpus": world.size, "steps": args.steps,
        "warmup": args.warmup, "setup_steps": setup_steps, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None,
        "dtype": DTYPE_LABEL[args.dtype],
        "data": "synthetic",
        "value_sync_api": value_sync_api,
        "sync_api": {"steps": n_sync, "ms_per_step": 1e3 * dt_sync / n_sync,
                     "what": "fn(i) returning [cost, features, logprob] every step as the reference's train.py:211 does"},
        "value_train_loop": value_train_loop,
        "train_loop": {"steps": n_loop, "ms_per_step": 1e3 * dt_loop / n_loop,
                       "what": "this build's train.py loop: fn.step_cost(i) -- every step's cost summed on the host a few "
                               "calls late (page-locked ring), the NaN guard on it; nothing waits for the GPU"},
        "sustained": sustained,
        "config": {"workload": "params/%s %dx%dx%d synthetic, global batch %d = %d images/GPU/step x %d, %s"
                               % (args.prms, img, img, C, tr["BATCH_SZ"], per_gpu, world.size, stage),
                   "global_batch": tr["BATCH_SZ"], "parallelism": "dp%d" % world.size,
                   "schedule": "two steps in flight (exact: the update applies the old velocity)"
                   if type(fn).__name__ == "_PipeTrainFn" and fn._twin is not None else "one step at a time",
                   "dp_schedule": getattr(net, "dp_schedule", None) if world.size > 1 else None,
                   "dp_schedule_us_per_step": {k: 1e3 * v for k, v in getattr(net, "dp_tuned_ms", {}).items()}
                   if world.size > 1 and isinstance(getattr(net, "dp_tuned_ms", None), dict) else None,
                   "step_gflop_algorithmic": step_flops / 1e9,
                   "step_tflops_algorithmic": step_flops / (dt / args.steps) / 1e12},
        "roofline": roof,
        "roofline_others": others,
        "final_cost": float(cost),
    }
    if selfcheck is not None:
        line["dp_selfcheck"] = selfcheck
    # N > 1: `value` is STRONG scaling, as BASELINE configs[2] states it (the global batch sharded N ways).  The same
    # launch then times the WEAK form (the stated batch per GPU, N times the global batch) and reports it beside it:
    # for mnist.prms the strong figure is bounded by the kernels' fixed costs at 4096/N rows,
    # the weak one is what the interconnect and the schedule allow.  Every rank takes part (the steps are collectives).
    if world.size > 1 and scaling == "strong" and not args.no_weak_leg:
        import gc
        import threading
        del fn, net
        gc.collect()
        # The headline is measured; the extra leg must not be able to cost it its line.  An exception is caught; a HANG
        # (a rank stuck in a collective) is not an exception: a watchdog thread prints the line as it stands and leaves
        # (ctypes calls release the interpreter lock, so the thread runs while the main thread sits in the library).
        limit = float(os.environ.get("TN_BENCH_WEAK_TIMEOUT", 240))

        def give_up():
            if world.rank == 0:
                line["weak"] = {"error": "the weak-scaling leg did not finish within %.0f s; skipped" % limit}
                os.write(REAL_STDOUT_FD if REAL_STDOUT_FD is not None else 1, (json.dumps(line) + "\n").encode())
            os._exit(0)

        dog = threading.Timer(limit, give_up)
        dog.daemon = True
        dog.start()
        try:
            wk = weak_leg(ctx, args, world, group, global_batch)
            line["value_weak"] = wk["value"]
            line["weak"] = wk
        except Exception as e:      # (an exception on every rank alike: the headline keeps its line)
            line["weak"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
        finally:
            dog.cancel()
    if world.rank != 0:
        return
    # BASELINE.json's other single-GPU configurations (north_star: images/sec "on synthetic 28x28x1 and 32x32x3
    # batches ... as fraction of the conv roofline"; the >= 50 % fp32-MFMA target on 3x3 convs; configs[4]'s fp16
    # path), timed by THIS run after the headline: the default N = 1 mnist run only.
    if world.size == 1 and args.prms == "mnist.prms" and not args.batch and not args.no_other_configs \
            and not args.no_roofline and os.environ.get("THEANET_BACKEND", "hip") != "cpu":
        import gc
        del fn, net
        gc.collect()
        line["other_configs"] = []
        for name, dt_ in OTHER_CONFIGS:
            try:
                line["other_configs"].append(other_config_leg(ctx, name, dt_, args.steps))
            except Exception as e:          # a leg must never cost the headline its line
                line["other_configs"].append({"config": {"workload": "params/%s %s" % (name, dt_)},
                                              "error": "%s: %s" % (type(e).__name__, str(e)[-300:])})
        # the headline workload once more with the opt-in MATMUL 'bf16x3' (NOT the headline: fp32-grade accuracy, other
        # bits); reported whichever way it compares
        try:
            b3 = other_config_leg(ctx, "mnist.prms", "b3", args.steps)
            line["value_bf16x3"] = b3["value"]
            line["bf16x3"] = {k: b3[k] for k in ("ms_per_step", "steps", "dtype", "final_cost")}
        except Exception as e:
            line["bf16x3"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
        # the 512-row shard of the 8-way strong-scaling run, through the data-parallel step
        try:
            line["shard_leg"] = shard_leg(ctx, args.steps)
            t_full = (sustained or {}).get("ms_per_step") or line["ms_per_step"]
            line["strong_scaling_ceiling_8"] = {
                "value": t_full / line["shard_leg"]["ms_per_step"],
                "what": "t(4096 rows) / t(512 rows) on ONE GPU = the most an 8-GPU strong-scaling run of configs[2] can gain "
                        "with a free all-reduce; NOT a measured scaling curve (none exists: no multi-GPU lease)",
                "t_4096_ms": round(t_full, 5), "t_512_ms": line["shard_leg"]["ms_per_step"]}
        except Exception as e:
            line["shard_leg"] = {"cfg": "mnist.prms", "rows": 512, "error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
    if world.size == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(prms, img, C, tr["BATCH_SZ"])
    # The compact figures once more, LAST in the line: a reader that keeps only the tail of the output (the round
    # driver keeps 2000 characters) still sees every configuration's step time.
    def frac(rec, key):
        for leg in rec.get("conv_roofline") or []:
            if key in str(leg.get("kernel", "")) or key in str(leg.get("what", "")):
                return leg.get("frac")
        return None
    legs = [{"cfg": args.prms + (" b%d" % args.batch if args.batch else ""), "dtype": args.dtype,
             "ms_per_step": round(line["ms_per_step"], 5),
             "sustained_ms": round(sustained["ms_per_step"], 5) if isinstance(sustained, dict) and sustained.get("ms_per_step") else None}]
    for rec in line.get("other_configs", []):
        wl = str(rec.get("config", {}).get("workload", "")).replace("params/", "").split(" ")[0]
        legs.append({"cfg": wl, "dtype": str(rec.get("dtype", "?"))[:3], "ms_per_step": round(rec["ms_per_step"], 5)
                     if "ms_per_step" in rec else None, "conv_fwd_frac": frac(rec, "forward"), "conv_bwd_frac": frac(rec, "backward"),
                     "error": rec.get("error")})
    if isinstance(line.get("shard_leg"), dict):
        sl = line["shard_leg"]
        legs.append({"cfg": "mnist.prms", "rows": 512, "dtype": "f32", "ms_per_step": sl.get("ms_per_step"),
                     "dp": "1-rank RCCL, two steps in flight" if "error" not in sl else None, "error": sl.get("error"),
                     "strong_scaling_ceiling_8": round(line["strong_scaling_ceiling_8"]["value"], 3)
                     if "strong_scaling_ceiling_8" in line else None})
    line["legs"] = legs
    line["tail"] = {"ms_per_step": round(line["ms_per_step"], 5), "value": round(line["value"], 1),
                    "value_train_loop": round(value_train_loop, 1) if value_train_loop else None,
                    "value_sync_api": round(value_sync_api, 1) if value_sync_api else None,
                    "roofline_frac": (roof or {}).get("frac") if isinstance(roof, dict) else None}
    print(json.dumps(line))


if __name__ == "__main__":
    # stdout carries exactly ONE line (the JSON record).  Everything else -- the library's progress
    # prints ("Compiling training function..." like the reference) and whatever native libraries write
    # to file descriptor 1 (RCCL's version banner) -- is sent to stderr: fd 1 is pointed at fd 2 for
    # the whole run and the record goes out through a private duplicate of the real stdout.
    import contextlib
    import io

    sys.stdout.flush()
    _real_fd = REAL_STDOUT_FD = os.dup(1)
    os.dup2(2, 1)
    _buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(_buf):
            main()
    finally:
        sys.stdout.flush()
        _lines = _buf.getvalue().splitlines()
        for _line in _lines:
            if not _line.startswith("{"):
                print(_line, file=sys.stderr)
        sys.stderr.flush()
        with os.fdopen(_real_fd, "w") as _out:
            for _line in _lines:
                if _line.startswith("{"):
                    _out.write(_line + "\n")
