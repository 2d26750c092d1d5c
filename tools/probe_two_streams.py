"""Upper bound of cross-step pipelining: two INDEPENDENT nets stepped alternately on the context's two
streams vs one net on one stream (numbers only; the nets share the scratch buffer, so results of the
two-stream run are not meaningful)."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_prms, synthetic
from theanet_amd import NeuralNet
from theanet_amd.device import get_context
prms = load_prms("mnist.prms")
B = 4096
prms["layers"][0][1]["img_sz"] = 28
tr = prms["training_params"]; tr["SEED"] = 555555; tr["BATCH_SZ"] = B
ctx = get_context()
x, y = synthetic(4 * B, 1, 28)
nets = [NeuralNet(copy.deepcopy(prms["layers"]), dict(tr)) for _ in range(2)]
fns = [n.get_trin_model(x, y) for n in nets]
def run(two, steps):
    for i in range(20):
        for k, fn in enumerate(fns if two else fns[:1]):
            if two: ctx.call("tn_stream_select", k)
            fn.enqueue(i % 4)
    ctx.call("tn_stream_select", 0); ctx.sync()
    t0 = time.perf_counter()
    n = 0
    for i in range(steps):
        for k, fn in enumerate(fns if two else fns[:1]):
            if two: ctx.call("tn_stream_select", k)
            fn.enqueue(i % 4); n += 1
    ctx.call("tn_stream_select", 0); ctx.sync()
    dt = time.perf_counter() - t0
    print("%s: %.1f us per step (%d steps)" % ("two streams" if two else "one stream", 1e6 * dt / n, n))
run(False, 400); run(True, 200); run(False, 400); run(True, 200)
