"""Upper bound of finer-grained concurrency: K INDEPENDENT nets at batch B on K streams (two contexts with two
streams each) against the two-stream figure at batch 4096 (numbers only: the nets of a context share
scratch, so the results are not meaningful)."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_prms, synthetic
from theanet_amd import NeuralNet
from theanet_amd import device


def build(B, nctx):
    prms = load_prms("mnist.prms")
    prms["layers"][0][1]["img_sz"] = 28
    tr = prms["training_params"]; tr["SEED"] = 555555; tr["BATCH_SZ"] = B
    x, y = synthetic(4 * B, 1, 28)
    lanes = []
    for c in range(nctx):
        device._context = None
        ctx = device.get_context()
        for k in range(2):
            net = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr))
            lanes.append((ctx, k, net.get_trin_model(x, y)))
    return lanes


def run(lanes, steps, B):
    ctxs = {id(c): c for c, _, _ in lanes}.values()
    def loop(n):
        for i in range(n):
            for ctx, k, fn in lanes:
                ctx.call("tn_stream_select", k)
                fn.enqueue(i % 4)
        for ctx in ctxs:
            ctx.call("tn_stream_select", 0); ctx.sync()
    loop(20)
    t0 = time.perf_counter()
    loop(steps)
    dt = time.perf_counter() - t0
    n = steps * len(lanes)
    print("%d lanes x batch %d: %.1f us per lane-step, %.2f M images/s" % (len(lanes), B, 1e6 * dt / n, n * B / dt / 1e6))


os.environ["TN_PIPELINE"] = "0"
for B, nctx in ((4096, 1), (2048, 2), (1024, 2), (4096, 2)):
    lanes = build(B, nctx)
    run(lanes, 200, B); run(lanes, 200, B)
