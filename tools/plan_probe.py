"""Host enqueue time per step against the GPU's, and whether the step is replayed from its recorded plan (B=512 TN_DP_FORCE=1: the
shard of the strong-scaling run through the data-parallel step).  python tools/plan_probe.py   (env: B, TN_DP_FORCE)"""
import ast, copy, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from theanet_amd import NeuralNet
from theanet_amd.device import get_context
B = int(os.environ.get("B", 512))
prms = ast.literal_eval(open("params/mnist.prms").read())
prms["layers"][0][1]["img_sz"] = 28
tr = dict(prms["training_params"]); tr.update(SEED=555555, BATCH_SZ=B)
net = NeuralNet(copy.deepcopy(prms["layers"]), tr)
rng = np.random.default_rng(0)
x = rng.random((16 * B, 1, 28, 28), dtype=np.float32); y = rng.integers(0, 10, 16 * B).astype(np.int32)
fn = net.get_trin_model(x, y)
ctx = get_context()
for i in range(64): fn.enqueue(i % 16)
ctx.sync()
pl = fn._plan
print(type(fn).__name__, "plan ready", pl.ready, "off", pl.off, "why", getattr(pl, "why", None), "period", getattr(pl, "period", None))
for n in (50, 300, 2000):
    ctx.sync(); t0 = time.perf_counter()
    for i in range(n): fn.enqueue(i % 16)
    t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    print("n=%d host enqueue %.1f us/step; with final sync %.1f us/step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
