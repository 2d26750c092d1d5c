"""
ORACLE (test infrastructure only -- never imported by the product path).

Restatement of the slice of Theano's ``shared_randomstreams.RandomStreams`` that
the reference touches.  Theano is a third-party, un-vendored, un-pinned
dependency of the reference (/root/reference/setup.py:14-17) and is not
installable here, so this restates its *published* algorithm (Theano 0.8-1.0,
``theano/tensor/shared_randomstreams.py`` + ``raw_random.py``):

  * ``RandomStreams(seed)`` keeps ``gen_seedgen = numpy.random.RandomState(seed)``.
  * every random variable created through ``gen()`` -- in graph construction
    order -- owns a private ``RandomState(int(gen_seedgen.randint(2**30)))``
    whose state advances once per function call.
  * draws are the plain numpy legacy calls ``rs.uniform(low, high, size)``,
    ``rs.normal(avg, std, size)``, ``rs.binomial(n, p, size)``; float draws are
    made in float64 and cast to the variable's dtype (floatX = float32 unless
    asked otherwise); binomial defaults to int64.

Reference call sites: theanet/layer/dropout.py:10-12,
theanet/layer/inlayers.py:72-73,81,94,101,107,112,141.

PARITY UNPINNED: the reference has no test that pins these streams and Theano
cannot be run here; the numbers below are pinned only against this restatement.
"""
import numpy as np


class _RandomVariable:
    """One Theano random variable: private RandomState + a draw recipe."""

    def __init__(self, seed, kind, args, size, dtype):
        self.rs = np.random.RandomState(seed)
        self.kind, self.args, self.size, self.dtype = kind, args, size, dtype

    def draw(self, size=None):
        size = self.size if size is None else size
        if self.kind == "uniform":
            low, high = self.args
            v = self.rs.uniform(low=low, high=high, size=size)
        elif self.kind == "normal":
            avg, std = self.args
            v = self.rs.normal(avg, std, size=size)
        elif self.kind == "binomial":
            n, p = self.args
            v = self.rs.binomial(n, p, size=size)
        else:  # pragma: no cover
            raise NotImplementedError(self.kind)
        return np.asarray(v, dtype=self.dtype)


class RandomStreams:
    """``tt.shared_randomstreams.RandomStreams(seed)`` look-alike (see module doc)."""

    def __init__(self, seed=None):
        self.gen_seedgen = np.random.RandomState(seed)

    def _gen(self, kind, args, size, dtype):
        seed = int(self.gen_seedgen.randint(2 ** 30))
        return _RandomVariable(seed, kind, args, size, dtype)

    def uniform(self, size=None, low=0.0, high=1.0, dtype="float32"):
        return self._gen("uniform", (low, high), size, dtype)

    def normal(self, size=None, avg=0.0, std=1.0, dtype="float32"):
        return self._gen("normal", (avg, std), size, dtype)

    def binomial(self, size=None, n=1, p=0.5, dtype="int64"):
        return self._gen("binomial", (n, p), size, dtype)
