"""Layer base class, activations and the momentum-SGD/max-norm update.

Host-side mirror of /root/reference/theanet/layer/layer.py: same activation
names (:27-54), same ``reg`` keys and update algebra (:70-107), same L1/L2 cost
(:109-117).  Instead of returning symbolic Theano update pairs the methods
enqueue HIP kernels on the context stream.
"""
import numpy as np

from .. import _lib
from .weights import borrow


class Activation:
    """Callable-by-name activation descriptor: ``kind``/``prm`` select the fused
    epilogue inside the HIP kernels (include/theanet_hip.h ``enum tn_act``)."""

    def __init__(self, name, kind, prm=0.0):
        self.name, self.kind, self.prm = name, kind, float(prm)

    def __str__(self):
        return self.name


activation_list = [
    Activation('sigmoid', _lib.TN_ACT_SIGMOID),
    Activation('softplus', _lib.TN_ACT_SOFTPLUS),
    Activation('softmax', None),
    Activation('linear', _lib.TN_ACT_LINEAR),
    Activation('scaled_tanh', _lib.TN_ACT_SCALED_TANH),
    Activation('relu', _lib.TN_ACT_LEAKY, 0.0),
    Activation('tanh', _lib.TN_ACT_TANH),
] + [
    Activation('relu{:02d}'.format(i), _lib.TN_ACT_LEAKY, i / 100)
    for i in range(100)
]


def activation_by_name(name):
    """layer.py:41-54 -- unknown names raise NotImplementedError."""
    if name in ("Softmax", "softmax"):
        return activation_list[2]
    for act in activation_list:
        if name == str(act):
            return act
    raise NotImplementedError("Unknown Activation Specified: " + name)


class Layer:
    """Base class.  Sub-classes set: params (list of DeviceArray), output
    (DeviceArray), representation, and optionally reg / actvn / mask."""

    params = ()
    grads = None          # views into the net's flat gradient buffer
    accumulated_updates = None   # velocity buffers (reference name, layer.py:72)

    def __str__(self):
        return self.representation

    _wts_hook = None           # set by NeuralNet: brings the weights up to date before they are read

    def get_wts(self):
        if self._wts_hook is not None:
            self._wts_hook()
        return [borrow(p) for p in self.params]

    # -- how the layer ABOVE must turn d(cost)/d(output) into d(cost)/d(z) -----------
    def act_info(self):
        """(output buffer or None, act kind, act param, mask or None).  The kernel
        that produces the gradient w.r.t. this layer's output fuses
        ``* act'(output) * mask`` so no separate elementwise pass is needed."""
        return None, _lib.TN_ACT_LINEAR, 0.0, None

    def has_updates(self):
        return bool(self.params) and hasattr(self, "reg") and bool(self.reg['rate'])

    # -- layer.py:70-107 --------------------------------------------------------------
    def get_updates(self, d_lr, gscale=1.0):
        """Enqueue v' = m v + (1-m) g ; p' = p - rate*lr*v_old ; maxnorm(p')."""
        if not self.has_updates():
            return
        ctx = self.params[0].ctx
        reg = self.reg
        for p, v, g in zip(self.params, self.accumulated_updates, self.grads):
            ctx.call("tn_sgd_update", p.ptr, v.ptr, g.ptr, p.size, float(reg['momentum']),
                     float(reg['rate']), d_lr.ptr, float(reg['L1']), float(reg['L2']),
                     float(gscale))
        self.apply_maxnorm()

    def apply_maxnorm(self):
        """layer.py:88-103: clip (1-D) / per-column (2-D) / per-kernel (4-D) norm projection
        of the freshly updated parameters."""
        if not self.has_updates() or not self.reg['maxnorm']:
            return
        ctx = self.params[0].ctx
        mx = float(self.reg['maxnorm'])
        for p in self.params:
            if p.ndim == 1:
                ctx.call("tn_maxnorm", p.ptr, 1, p.shape[0], 1, mx)
            elif p.ndim == 2:
                ctx.call("tn_maxnorm", p.ptr, 2, p.shape[0], p.shape[1], mx)
            elif p.ndim == 4:
                ctx.call("tn_maxnorm", p.ptr, 4, p.shape[0], int(np.prod(p.shape[1:])), mx)

    # -- layer.py:109-117 ---------------------------------------------------------------
    def get_wtcost(self, d_cost):
        """Enqueue cost += L1*sum|p| + L2*sum p^2 over all params of the layer."""
        reg = getattr(self, "reg", None)
        if not reg or not self.params or not (reg['L1'] or reg['L2']):
            return
        ctx = self.params[0].ctx
        for p in self.params:
            ctx.call("tn_wtcost", p.ptr, p.size, float(reg['L1']), float(reg['L2']), d_cost.ptr, 1)

    # default: nothing to do
    def forward(self, train=True):
        pass

    def backward(self, gout, need_gin, below):
        return None
