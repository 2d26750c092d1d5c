"""How long K pipelined steps take when they start from an empty GPU (sync, K x enqueue, sync)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_prms, synthetic
from theanet_amd import NeuralNet
from theanet_amd.device import get_context
B = 4096
prms = load_prms("mnist.prms"); prms["layers"][0][1]["img_sz"] = 28
tr = prms["training_params"]; tr["SEED"] = 555555; tr["BATCH_SZ"] = B
x, y = synthetic(16 * B, 1, 28)
net = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr))
fn = net.get_trin_model(x, y)
ctx = get_context()
for i in range(100): fn.enqueue(i % 16)
ctx.sync()
for K in (1, 2, 4, 8, 16, 32, 64, 128, 512):
    best = None
    for rep in range(5):
        ctx.sync()
        t0 = time.perf_counter()
        for i in range(K): fn.enqueue(i % 16)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        r = ((t2 - t0) * 1e6, (t1 - t0) * 1e6)
        best = r if best is None or r[0] < best[0] else best
    print("K=%4d  total %8.1f us (%6.1f per step)   host enqueue %8.1f us (%5.1f per step)" %
          (K, best[0], best[0] / K, best[1], best[1] / K))

print("-- like bench.py: (53 untimed steps, sync, 20 timed steps, sync) x 6, then 20-step bursts back to back")
for rep in range(6):
    for i in range(53): fn.enqueue(i % 16)
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(20): fn.enqueue(i % 16)
    ctx.sync()
    print("   after 53 untimed: %.1f us per step" % ((time.perf_counter() - t0) * 1e6 / 20))
for rep in range(6):
    t0 = time.perf_counter()
    for i in range(20): fn.enqueue(i % 16)
    ctx.sync()
    print("   back to back:     %.1f us per step" % ((time.perf_counter() - t0) * 1e6 / 20))
import gc
print("-- with a host pause of 50 ms before each burst")
for rep in range(4):
    time.sleep(0.05)
    t0 = time.perf_counter()
    for i in range(20): fn.enqueue(i % 16)
    ctx.sync()
    print("   after 50 ms idle: %.1f us per step" % ((time.perf_counter() - t0) * 1e6 / 20))
