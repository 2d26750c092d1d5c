"""MNIST data module with the reference's contract (data/mnist.py:54).  Reads
``mnist.pkl.gz`` from this directory if present (train+valid merged into 60000
training rows like data/mnist.py:45-49); there is no network here, so a missing
file is an error rather than a download -- use ``data.synthetic`` instead."""
import gzip
import os
import pickle

import numpy as np

_f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mnist.pkl.gz")
if not os.path.isfile(_f):
    raise FileNotFoundError(
        _f + " not found and this environment cannot download it; "
        "run `python train.py synthetic params/mnist.prms` for an MNIST-shaped stand-in")
with gzip.open(_f, "rb") as fh:
    _u = pickle._Unpickler(fh)
    _u.encoding = "latin1"
    (_trx, _try), (_vx, _vy), (_tex, _tey) = _u.load()
training_x = np.vstack((_trx, _vx)).reshape((-1, 1, 28, 28))
training_y = np.concatenate((_try, _vy))
testing_x = _tex.reshape((-1, 1, 28, 28))
testing_y = _tey
