"""Regenerates tests/golden/torch_xchk.npz (run in the BUILD container: `python
tests/golden/make_torch_xchk.py`): the whole params/mnist.prms net -- forward, every gradient and a
3-step weight trajectory -- computed by an INDEPENDENT second implementation, torch CPU autograd in
float64, written straight from the reference's layer definitions:

  conv      theanet/layer/convpool.py:54-72   nnconv.conv2d(filter_flip=True) -> F.conv2d on flipped kernels
  relu{ii}  theanet/layer/layer.py:31-38      max(0,z) + min(0,z)*ii/100
  pool      theanet/layer/convpool.py:106-107 pool_2d(ignore_border=False) -> F.max_pool2d(ceil_mode=True)
  hidden    theanet/layer/hidden.py:30-32     act(x.W + b) * mask            (mask injected, no rescale)
  softmax   theanet/layer/outlayers.py:87-95  log(softmax(x.W + b)); cost = -mean logprob[n, y_n] (:50-51)
  update    theanet/layer/layer.py:82-86      v' = m v + (1-m) g ; p' = p - rate*lr*v  (OLD velocity)

It uses none of oracle/'s arithmetic (only its seed-chain initial weights, which KAT-6 pins by hash, and
numpy's RNG for inputs/masks).  The oracle and the HIP path are both checked against this file, so the
oracle's "parity unpinned" status rests on two independent restatements agreeing on a whole training
trajectory, not only on single ops.  The ElasticLayer is replaced by an InputLayer here (its gather has
no torch counterpart; it is pinned by injected-draw tests against the oracle), dropout masks are drawn
with numpy and injected, data is tie-free uniform noise."""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import theanet_oracle as O  # noqa: E402  (initial weights only)

B, STEPS, SEED = 8, 3, 555555


def sub_index(size, n=4096):
    """Fixed pseudo-random subsample used for tensors too large to commit whole (as make_golden.py)."""
    return np.random.RandomState(size % (2 ** 31)).choice(size, n, replace=False)


def put(out, name, arr):
    arr = np.asarray(arr)
    if arr.size > 20000:
        out[name + "@sub"] = arr.reshape(-1)[sub_index(arr.size)]
        out[name + "@sum"] = np.asarray(arr.sum(dtype=np.float64))
        out[name + "@abs"] = np.asarray(np.abs(arr).sum(dtype=np.float64))
    else:
        out[name] = arr


def leaky(z, s):
    return torch.clamp(z, min=0) + torch.clamp(z, max=0) * s


def forward(p, x, mask):
    W1, b1, W2, b2, W3, b3, W4, b4 = p
    a = leaky(F.conv2d(x, torch.flip(W1, (2, 3)), b1), .10)
    a = F.max_pool2d(a, 2, ceil_mode=True)
    a = leaky(F.conv2d(a, torch.flip(W2, (2, 3)), b2), .05)
    a = F.max_pool2d(a, 2, ceil_mode=True)
    h = leaky(a.flatten(1) @ W3 + b3, .01) * mask
    return F.log_softmax(h @ W4 + b4, dim=1)


def main():
    with open(os.path.join(ROOT, "params", "mnist.prms")) as fh:
        prms = ast.literal_eval(fh.read())
    prms["layers"][0] = ("InputLayer", {"img_sz": 28})
    tr = dict(prms["training_params"], SEED=SEED, BATCH_SZ=B)
    init = O.OracleNet(prms["layers"], dict(tr), dtype=np.float64)
    params = [torch.tensor(np.array(w, np.float64), requires_grad=True)
              for l in init.L for w in l.params]
    rng = np.random.RandomState(2024)
    x = rng.rand(STEPS * B, 1, 28, 28)
    y = rng.randint(0, 10, STEPS * B)
    masks = (rng.rand(STEPS, B, 500) < .5).astype(np.float64)
    out = {"x": x.astype(np.float32), "y": y.astype(np.int32), "masks": masks.astype(np.uint8)}
    for i, p in enumerate(params):
        put(out, "w0_%d" % i, p.detach().numpy().astype(np.float32))
    vel = [torch.zeros_like(p) for p in params]
    m, rate = .95, 1.0
    lr = tr["INIT_LEARNING_RATE"] / (1 + 0 / tr["EPOCHS_TO_HALF_RATE"])
    lr = float(np.float32(lr))                   # the learning rate is a float32 device scalar
    for s in range(STEPS):
        xs = torch.tensor(x[s * B:(s + 1) * B].astype(np.float32).astype(np.float64))
        ys = torch.tensor(y[s * B:(s + 1) * B])
        for p in params:
            p.grad = None
        lp = forward(params, xs, torch.tensor(masks[s]))
        cost = F.nll_loss(lp, ys)
        cost.backward()
        out["logprob_%d" % s] = lp.detach().numpy()
        out["cost_%d" % s] = np.asarray(cost.item())
        if s == 0:
            for i, p in enumerate(params):
                put(out, "grad0_%d" % i, p.grad.numpy().copy())
        with torch.no_grad():
            for p, v in zip(params, vel):
                g = p.grad
                p -= rate * lr * v                # OLD velocity (layer.py:86)
                v.mul_(m).add_((1 - m) * g)       # layer.py:82-84
    for i, p in enumerate(params):
        put(out, "w%d_%d" % (STEPS, i), p.detach().numpy())
    np.savez_compressed(os.path.join(HERE, "torch_xchk.npz"), **out)
    print("wrote torch_xchk.npz:", sorted(out))


if __name__ == "__main__":
    main()
