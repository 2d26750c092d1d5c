"""Synthetic MNIST-shaped dataset module (the data contract of the reference's
data/mnist.py:54: module attributes ``training_x, training_y, testing_x,
testing_y``; x is 4-D NCHW float, y integer labels).

Images are class-dependent so that training visibly learns: each class has a
fixed random 28x28 prototype; a sample is its prototype plus uniform noise,
clipped to [0, 1].  Sizes via THEANET_SYNTH_TRAIN / THEANET_SYNTH_TEST.
"""
import os

import numpy as np


def make(n, rng, protos, noise=.35):
    y = rng.integers(0, protos.shape[0], n)
    x = protos[y] + noise * (rng.random((n,) + protos.shape[1:], dtype=np.float32) - .5)
    return np.clip(x, 0, 1).astype(np.float32), y.astype(np.int64)


_n_tr = int(os.environ.get("THEANET_SYNTH_TRAIN", 8192))
_n_te = int(os.environ.get("THEANET_SYNTH_TEST", 2048))
_c = int(os.environ.get("THEANET_SYNTH_CHANNELS", 1))
_hw = int(os.environ.get("THEANET_SYNTH_SIZE", 28))
_rng = np.random.default_rng(2016)
_protos = (_rng.random((10, _c, _hw, _hw), dtype=np.float32) > .7).astype(np.float32)
training_x, training_y = make(_n_tr, _rng, _protos)
testing_x, testing_y = make(_n_te, _rng, _protos)
if os.environ.get("THEANET_SYNTH_AUX"):
    # per-sample side input for AuxConcatLayer / SoftAuxLayer nets (reference train.py:133-137): two candidate 2-D
    # locations per sample, the second one class-dependent
    def _aux(n, y):
        a = _rng.random((n, 2, 2), dtype=np.float32)
        a[:, 1, 0] = y / 10.0
        return a
    training_aux, testing_aux = _aux(_n_tr, training_y), _aux(_n_te, testing_y)
