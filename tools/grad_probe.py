"""One training step of params/mnist.prms against the float64 oracle, gradient by gradient:
    python tools/grad_probe.py BATCH ELASTIC(0|1) [p]      (p: conv kernels de-symmetrised as in GOLD-B, perturbed_init)
Prints cost, max |dlogprob| and, per parametrised tensor, max |g - g_oracle| / max |g_oracle| (then the same for the
velocities after the update).  With ELASTIC=1 and the reference's own init (+-1/sqrt(fan_in)) conv1's gradient differs by
0.5-7 % from ANY other implementation's: nearest-mode zoom duplicates pixels, pooling-window members whose patches are
permutations of one another are mathematically equal sums, and the tie is broken by the order of summation (DESIGN.md
section 2).  With `p` the agreement is 1e-6.  Works on the GPU and with THEANET_BACKEND=cpu."""
import copy, sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import theanet_oracle as O
from tests.gpu_util import load_prms
from theanet_amd import NeuralNet
B = int(sys.argv[1]); el = int(sys.argv[2])
prms = load_prms("mnist.prms", 28, batch=B)
if not el:
    prms["layers"][0] = ("ElasticLayer", {"img_sz": 28, "invert_image": True})
tr = prms["training_params"]
x = np.random.default_rng(0).random((B, 1, 28, 28), dtype=np.float32)
y = np.random.default_rng(1).integers(0, 10, B).astype(np.int32)
from tests.golden.make_golden import perturbed_init
allwts = perturbed_init(prms) if len(sys.argv) > 3 else None
net = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr), copy.deepcopy(allwts))
ora = O.OracleNet(copy.deepcopy(prms["layers"]), dict(tr), allwts=copy.deepcopy(allwts), dtype=np.float64)
fn = net.get_trin_model(x, y)
draws = {}
if el:
    d0 = ora.L[0].stage.draw((B, 1, 28, 28)); draws[0] = d0
    net.tr_layers[0].inject(**{k: getattr(d0, k) for k in d0.__slots__})
m5 = ora.L[5].mask_rv.draw((B, 500)); draws[5] = m5
net.tr_layers[5].drop.inject(m5)
cost, _, lp = fn(0)
cw, lpw, grads, _ = ora.grads(x, y, draws)
print(cost, cw, np.abs(lp - lpw).max())
for i, lyr in enumerate(net.tr_layers):
    for j, g in enumerate(lyr.grads or ()):
        gg = g.get_value(); gw = grads[i][j]
        print(i, j, np.abs(gg - gw).max() / np.abs(gw).max())
vel = net.get_init_params(with_opt_state=True)["opt_state"]["velocities"]
ora.train_step(x, y, draws)
for i, (lyr, ol) in enumerate(zip(net.tr_layers, ora.L)):
    for j in range(len(lyr.params or ())):
        v, vw = vel[i][j].astype(np.float64), ol.vel[j]
        print("vel", i, j, np.abs(v - vw).max() / np.abs(vw).max())
