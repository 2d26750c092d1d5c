"""Dropout -- host mirror of theanet/layer/dropout.py.

mask ~ Bernoulli(1 - pdrop), output * mask, NO 1/(1-p) rescale in training;
the test version multiplies by (1 - pdrop) (:28-31).  The stream seed is drawn
from ``rand_gen.randint(1e6)`` at construction exactly like the reference (:10)
so the numpy seed chain stays in step; the mask itself comes from an on-device
Philox4x32-10 generator keyed by (seed, step, GLOBAL element index) -- so it does
not depend on how the batch is sharded over GPUs -- or is injected for parity
tests (``inject_mask``).
"""
import numpy as np

from .. import _lib
from .layer import Layer


class DropStream:
    """Device-side replacement of ``RandomStreams(seed).binomial(n=1, p=1-pdrop)``."""

    def __init__(self, ctx, shape, pdrop, rand_gen=None):
        self.ctx, self.shape, self.pdrop = ctx, tuple(shape), float(pdrop)
        self.seed = int(rand_gen.randint(1e6)) if rand_gen is not None \
            else int(np.random.randint(0, 1e6))
        self.mask = ctx.empty(shape, np.uint8)
        self.injected = False
        self.ready = False          # mask of the current step already generated (side stream)
        self.d_step = None          # device step counter (set by the net)
        self.elem0 = 0              # global index of this shard's first element

    def inject(self, mask):
        """Parity hook: use this host mask (0/1) instead of the device RNG.
        Pass None to return to the generator."""
        if mask is None:
            self.injected = False
        else:
            self.mask.set_value(np.asarray(mask).reshape(self.shape).astype(np.uint8))
            self.injected = True

    def generate(self):
        if self.injected:
            return
        if self.ready:              # produced ahead of time by NeuralNet._train_step
            self.ready = False
            return
        self.ctx.call("tn_dropout_mask", self.mask.ptr, self.mask.size, self.pdrop, self.seed,
                      0, self.d_step.ptr if self.d_step is not None else None, self.elem0)


def drop_output(layer, output, pdrop, rand_gen=None):
    """dropout.py:9-13 -- attaches a DropStream to ``layer`` for ``output``."""
    layer.drop = DropStream(output.ctx, output.shape, pdrop, rand_gen)
    return layer.drop


class DropOutLayer(Layer):
    def __init__(self, inpt, rand_gen=None, n_in=None, pdrop=0):
        self.ctx = inpt.ctx
        self.inpt = inpt
        self.params = []
        self.n_in, self.n_out = n_in, n_in
        self.pdrop = pdrop
        self.test_scale = 1.0
        self.drop = None
        if pdrop:
            drop_output(self, inpt, pdrop, rand_gen)
            self.output = self.ctx.empty(inpt.shape)
        else:
            self.output = inpt
        self.gin = None
        self.representation = "Drop:{:.0%} Out:{:3d}".format(pdrop, n_in)

    def TestVersion(self, inpt):
        test_version = DropOutLayer(inpt, n_in=self.n_in, pdrop=0)
        if self.pdrop:
            test_version.test_scale = 1 - self.pdrop
            test_version.output = self.ctx.empty(inpt.shape)
        return test_version

    def forward(self, train=True):
        if self.drop is not None:
            self.drop.generate()
            self.ctx.call("tn_scale_mask", self.inpt.ptr, self.drop.mask.ptr, 1.0,
                          self.output.ptr, self.inpt.size, None, _lib.TN_ACT_LINEAR, 0.0)
        elif self.test_scale != 1.0:
            self.ctx.call("tn_scale_mask", self.inpt.ptr, None, float(self.test_scale),
                          self.output.ptr, self.inpt.size, None, _lib.TN_ACT_LINEAR, 0.0)

    def backward(self, gout, need_gin, below):
        if not need_gin:
            return None
        b_out, b_act, b_prm, b_mask = below.act_info()
        fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
        if self.drop is None and not fuse and b_mask is None:
            return gout
        if self.gin is None:
            self.gin = self.ctx.empty(self.inpt.shape)
        src = gout
        if b_mask is not None:      # layer below is a Hidden layer with its own dropout
            self.ctx.call("tn_scale_mask", src.ptr, b_mask.ptr, 1.0, self.gin.ptr, self.inpt.size,
                          None, _lib.TN_ACT_LINEAR, 0.0)
            src = self.gin
        self.ctx.call("tn_scale_mask", src.ptr, self.drop.mask.ptr if self.drop else None, 1.0,
                      self.gin.ptr, self.inpt.size, b_out.ptr if fuse else None, b_act, b_prm)
        return self.gin
