import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.gpu_util import load_prms
from theanet_amd import NeuralNet
B = int(os.environ.get("B", 4096))
prms = load_prms("mnist.prms", 28, batch=B)
rng = np.random.RandomState(5)
x = rng.rand(B * 4, 1, 28, 28).astype(np.float32); y = rng.randint(0, 10, B * 4).astype(np.int32)
net = NeuralNet(copy.deepcopy(prms["layers"]), dict(prms["training_params"]))
fn = net.get_trin_model(x, y)
for i in range(300): fn.enqueue(i % 4)
for i in range(20): fn(i % 4)
n = int(os.environ.get("NSTEPS", 2000))
seen, tot, ks = 0, 0.0, []
for i in range(n):
    for k, c in fn.step_cost(i % 4):
        seen += 1; tot += float(c); ks.append(k)
        if not np.isfinite(c): print("non-finite at", k, c, hex(int(np.float32(c).view(np.uint32))))
for k, c in fn.drain_costs():
    seen += 1; tot += float(c); ks.append(k)
    if not np.isfinite(c): print("non-finite (drain) at", k, c)
print("seen", seen, "of", n, "tot", tot, "in order", ks == list(range(len(ks))), "plan", fn._plan.ready, fn._plan.why, type(fn).__name__, getattr(fn, "_seq", None) is None)
