"""cProfile of the host side of enqueue-only training steps (run on the GPU box)."""
import ast, copy, cProfile, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theanet_amd import NeuralNet
from theanet_amd.device import get_context
prms = ast.literal_eval(open(os.path.join(os.path.dirname(__file__), "..", "params", "mnist.prms")).read())
prms["layers"][0][1]["img_sz"] = 28
B = int(os.environ.get("B", 512))
tr = dict(prms["training_params"]); tr.update(SEED=555555, BATCH_SZ=B)
net = NeuralNet(copy.deepcopy(prms["layers"]), tr)
rng = np.random.default_rng(0)
x = rng.random((16 * B, 1, 28, 28), dtype=np.float32); y = rng.integers(0, 10, 16 * B).astype(np.int32)
fn = net.get_trin_model(x, y)
ctx = get_context()
for i in range(20): fn.enqueue(i % 16)
ctx.sync()
pr = cProfile.Profile()
pr.enable()
for i in range(2000):
    fn.enqueue(i % 16)
    if i % 8 == 7: ctx.sync()
pr.disable()
ctx.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
