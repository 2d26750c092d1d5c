"""Where the time of the drop-in call fn(i) -> [cost, features, logprob] goes (run on the GPU box)."""
import ast, copy, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theanet_amd import NeuralNet
from theanet_amd.device import get_context
prms = ast.literal_eval(open(os.path.join(os.path.dirname(__file__), "..", "params", "mnist.prms")).read())
prms["layers"][0][1]["img_sz"] = 28
tr = dict(prms["training_params"]); tr.update(SEED=555555, BATCH_SZ=int(os.environ.get("B", 4096)))
B = tr["BATCH_SZ"]
net = NeuralNet(copy.deepcopy(prms["layers"]), tr)
rng = np.random.default_rng(0)
x = rng.random((16 * B, 1, 28, 28), dtype=np.float32); y = rng.integers(0, 10, 16 * B).astype(np.int32)
fn = net.get_trin_model(x, y)
ctx = get_context()
for i in range(20): fn(i % 16)
n = 300
T = np.zeros(4)
for i in range(n):
    t0 = time.perf_counter(); fn.enqueue(i % 16)
    t1 = time.perf_counter(); ctx.sync()
    t2 = time.perf_counter(); out = fn.fetch()
    t3 = time.perf_counter()
    T += [t1 - t0, t2 - t1, t3 - t2, t3 - t0]
print("enqueue %.1f  wait %.1f  fetch %.1f  total %.1f us/step" % tuple(T / n * 1e6))
t0 = time.perf_counter()
for i in range(n): fn(i % 16)
print("fn(i): %.1f us/step" % ((time.perf_counter() - t0) / n * 1e6))
# pieces of fetch() after a call that sent the outputs ahead
import theanet_amd.neuralnet as nn
X = None
T = np.zeros(6)
for i in range(n):
    fn._want = True
    fn.enqueue(i % 16)
    fn._want = False
    X = fn._last if getattr(fn, "_seq", None) is None else fn._seq.net
    t0 = time.perf_counter()
    if getattr(fn, "_seq", None) is None and X._cost_pending:
        X.ctx.call("tn_stream_select", fn.nets.index(X)); fn._finish_cost(X); X.ctx.call("tn_stream_select", 0)
    t1 = time.perf_counter(); X.ctx.sync()
    t2 = time.perf_counter(); c = X.d_cost.get_value()[0]
    t3 = time.perf_counter(); X.ctx.call("tn_copy_sync")
    t4 = time.perf_counter(); a = X._early["logprob"].array.copy(); X._early["live"] = False
    t5 = time.perf_counter()
    T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0]
print("finish-cost launch %.1f  sync %.1f  cost d2h %.1f  copy_sync %.1f  host copy %.1f  total %.1f" % tuple(T / n * 1e6))
