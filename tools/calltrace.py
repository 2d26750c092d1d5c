"""Prints the C-ABI calls of one steady-state training step (mnist.prms, B=4096 by default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_prms  # noqa
from theanet_amd import NeuralNet
from theanet_amd.device import get_context
prms = load_prms(os.environ.get("PRMS", "mnist.prms"))
B, img = int(os.environ.get("B", 4096)), int(os.environ.get("IMG", 28))
C = prms["layers"][0][1].get("num_maps", 1)
prms["layers"][0][1]["img_sz"] = img
prms["training_params"]["BATCH_SZ"] = B
prms["training_params"]["SEED"] = 555555
x = np.random.default_rng(0).random((2 * B, C, img, img), np.float32)
y = np.random.default_rng(1).integers(0, 10, 2 * B).astype(np.int32)
net = NeuralNet(prms["layers"], prms["training_params"])
fn = net.get_trin_model(x, y)
for i in range(3):
    fn.enqueue(i % 2)
ctx = get_context()
orig = ctx.call
def logged(name, *a):
    print("  ", name)
    return orig(name, *a)
ctx.call = logged
fn.enqueue(1)
ctx.call = orig
ctx.sync()
