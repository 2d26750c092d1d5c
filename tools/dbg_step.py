import sys, os, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests.gpu_util import load_prms
from theanet_amd import NeuralNet
B = int(sys.argv[1])
prms = load_prms("mnist.prms", 28, batch=B)
net = NeuralNet(prms["layers"], prms["training_params"])
rng = np.random.default_rng(0)
x = rng.random((2*B, 1, 28, 28), dtype=np.float32)
y = np.random.default_rng(1).integers(0, 10, 2*B).astype(np.int32)
fn = net.get_trin_model(x, y)
orig = net.ctx.call
def call(name, *a):
    print("call", name, flush=True)
    orig(name, *a); net.ctx.lib.tn_sync(net.ctx.h)
net.ctx.call = call
w0 = net.tr_layers[5].get_wts()[0]
print("step0", flush=True)
c0 = fn(0)[0]
print("cost0", c0, flush=True)
w = net.tr_layers[5].get_wts()[0]
for i in range(1, 6):
    print("step", i, flush=True)
    print(fn(i % 2)[0], flush=True)
print("done")
