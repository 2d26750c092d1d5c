"""Host-side cost of enqueueing one training step (no synchronisation inside the loop)."""
import ast, copy, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theanet_amd import NeuralNet
from theanet_amd.device import get_context
prms = ast.literal_eval(open(os.path.join(os.path.dirname(__file__), "..", "params", "mnist.prms")).read())
prms["layers"][0][1]["img_sz"] = 28
tr = dict(prms["training_params"]); tr.update(SEED=555555, BATCH_SZ=int(os.environ.get("B", 4096)))
net = NeuralNet(copy.deepcopy(prms["layers"]), tr)
rng = np.random.default_rng(0)
x = rng.random((16 * int(os.environ.get("B", 4096)), 1, 28, 28), dtype=np.float32); y = rng.integers(0, 10, 16 * int(os.environ.get("B", 4096))).astype(np.int32)
fn = net.get_trin_model(x, y)
ctx = get_context()
for i in range(20): fn.enqueue(i % 16)
ctx.sync()
n = 300
t0 = time.perf_counter()
for i in range(n): fn.enqueue(i % 16)
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print("host enqueue %.1f us/step; with final sync %.1f us/step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
