"""Auxiliary-input layers -- host mirror of theanet/layer/auxiliary.py:14-160.

A second, small input per sample (for the reference's use: two candidate 2-D locations of a glyph,
a (N, 2, 2) tensor) is pushed through a two-layer perceptron (``LocationInfo``: relu50 then relu01) and
either concatenated to the features of the layer below (``AuxConcatLayer``) or added, through a "cross"
weight matrix, to the logits of a softmax layer (``SoftAuxLayer``).  In training the two candidates are
mixed with one U(0,1) draw per sample; the test version averages them.  The dense maps run on the
library's tn_fc_* ops (any shape), the mix and the concatenation on two small kernels
(tn_aux_mix, tn_copy_cols)."""
import numpy as np

from .. import _lib
from .hidden import HiddenLayer
from .layer import Layer, activation_by_name
from .outlayers import HEAD_SOFTMAX, OutputLayer, loss_code
from .weights import init_wb


class AuxSlot:
    """Stand-in for the reference's symbolic ``AuxiallaryInput`` (auxiliary.py:23): a window of ``batch``
    rows into a device-resident (N, 2, d) array, moved by the step functions."""

    def __init__(self, batch):
        self.batch = batch
        self.data = None
        self.row0 = 0
        self.row_global0 = 0

    def bind(self, data):
        self.data = data


class LocationInfo:
    """auxiliary.py:14-58."""

    def __init__(self, wts, rand_gen=None, n_aux=(5, 9), boost=1, test_version=False, batch_sz=None, ctx=None):
        self.ctx, self.batch_sz = ctx, batch_sz
        self.aux_inpt = AuxSlot(batch_sz)
        self.test_version, self.boost = test_version, boost
        self.seed, self.d_step, self._inj = 0, None, None
        if not test_version:                                   # :25-27: the stream seed consumes the seed chain
            self.seed = int(rand_gen.randint(1e6)) if rand_gen is not None else int(np.random.randint(0, 1e6))
        n_aux_hid, n_aux_out = n_aux
        self.act1, self.act2 = activation_by_name("relu50"), activation_by_name("relu01")
        loc1_wts = None if wts is None else wts[:2]
        self.w1, self.b1 = init_wb(loc1_wts, rand_gen, (2, n_aux_hid), n_aux_hid,
                                   n_aux_hid + 2, n_aux_hid + 2, "relu50", 'Loc1')
        loc2_wts = None if wts is None else wts[2:4]
        self.w2, self.b2 = init_wb(loc2_wts, rand_gen, (n_aux_hid, n_aux_out), n_aux_out,
                                   n_aux_out + n_aux_hid, n_aux_out + n_aux_hid, "relu01", 'Loc2')
        self.params = [self.w1, self.b1, self.w2, self.b2]
        B = batch_sz
        self.mix = ctx.empty((B, 2))
        self.hid = ctx.empty((B, n_aux_hid))
        self.output = ctx.empty((B, n_aux_out))
        self.n_hid, self.n_out = n_aux_hid, n_aux_out
        self._dz2 = self._dhid = self._dmix = None
        self._ws1 = self._ws2 = None

    def inject(self, u=None):
        """Replace the device RNG by explicit uniforms (batch,) -- the reference's ``srs.uniform`` draw."""
        self._inj = None if u is None else self.ctx.array(np.ascontiguousarray(np.asarray(u, np.float32).reshape(-1)))

    def forward(self, train=True):
        s, c, B = self.aux_inpt, self.ctx, self.batch_sz
        mixing = train and not self.test_version
        c.call("tn_aux_mix", s.data.ptr, int(s.row0), self.mix.ptr, B, 2, float(self.boost), 1 if mixing else 0,
               self._inj.ptr if (self._inj is not None and mixing) else None, self.seed, 0,
               self.d_step.ptr if self.d_step is not None else None, int(s.row_global0))
        c.call("tn_fc_fwd", self.mix.ptr, self.w1.ptr, self.b1.ptr, self.hid.ptr, B, 2, self.n_hid,
               self.act1.kind, self.act1.prm, None)
        c.call("tn_fc_fwd", self.hid.ptr, self.w2.ptr, self.b2.ptr, self.output.ptr, B, self.n_hid, self.n_out,
               self.act2.kind, self.act2.prm, None)

    def backward(self, gout, grads):
        """gout = d cost / d output (before this net's last activation); grads = [dW1, db1, dW2, db2]."""
        c, B = self.ctx, self.batch_sz
        if self._dz2 is None:
            self._dz2, self._dhid, self._dmix = c.empty(self.output.shape), c.empty(self.hid.shape), c.empty(self.mix.shape)
            self._ws2 = c.empty(((c.lib.tn_fc_wgrad_ws_bytes(B, self.n_hid, self.n_out) + 3) // 4,))
            self._ws1 = c.empty(((c.lib.tn_fc_wgrad_ws_bytes(B, 2, self.n_hid) + 3) // 4,))
        c.call("tn_scale_mask", gout.ptr, None, 1.0, self._dz2.ptr, self._dz2.size, self.output.ptr,
               self.act2.kind, self.act2.prm)
        c.call("tn_fc_bwd", self.hid.ptr, self._dz2.ptr, self.w2.ptr, grads[2].ptr, grads[3].ptr, self._dhid.ptr,
               B, self.n_hid, self.n_out, self._ws2.ptr, self.hid.ptr, self.act1.kind, self.act1.prm, None)
        c.call("tn_fc_wgrad", self.mix.ptr, self._dhid.ptr, grads[0].ptr, grads[1].ptr, B, 2, self.n_hid,
               self._ws1.ptr)


_AUX_TYPES = {"LocationInfo": LocationInfo}


class AuxConcatLayer(Layer):
    """auxiliary.py:64-101: output = concatenate(inpt, aux MLP output).  The layer has no ``reg``: like in the
    reference (layer.py:74-75) its aux weights are never updated, the gradient only passes through to the
    layer below."""

    def __init__(self, inpt, wts, rand_gen, n_in, n_aux, aux_type, boost=1, test_version=False):
        self.ctx = inpt.ctx
        self.inpt = inpt.flatten(2)
        self.batch_sz = self.inpt.shape[0]
        if wts is not None and len(wts) == 0:
            wts = None
        self.aux = _AUX_TYPES[aux_type](wts, rand_gen, n_aux=n_aux, boost=boost, test_version=test_version,
                                        batch_sz=self.batch_sz, ctx=self.ctx)
        self.aux_inpt = self.aux.aux_inpt
        self.n_aux, self.n_in = n_aux, n_in
        self.n_out = n_aux[-1] + n_in
        self.aux_type, self.boost = aux_type, boost
        self.params = self.aux.params
        self.output = self.ctx.empty((self.batch_sz, self.n_out))
        self.gin = None
        self.representation = "AuxConcat In:{:3d} Aux:{} Out:{:3d} ".format(n_in, n_aux, self.n_out)

    def TestVersion(self, te_inpt):
        return AuxConcatLayer(te_inpt, self.params, None, self.n_in, self.n_aux, self.aux_type,
                              boost=self.boost, test_version=True)

    def forward(self, train=True):
        self.aux.forward(train)
        B = self.batch_sz
        self.ctx.call("tn_copy_cols", self.inpt.ptr, self.n_in, 0, self.output.ptr, self.n_out, 0, self.n_in, B,
                      None, 0, 0.0)
        self.ctx.call("tn_copy_cols", self.aux.output.ptr, self.n_aux[-1], 0, self.output.ptr, self.n_out,
                      self.n_in, self.n_aux[-1], B, None, 0, 0.0)

    def backward(self, gout, need_gin, below):
        if not need_gin:
            return None
        b_out, b_act, b_prm, b_mask = below.act_info()
        if self.gin is None:
            self.gin = self.ctx.empty(self.inpt.shape)
        fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
        self.ctx.call("tn_copy_cols", gout.ptr, self.n_out, 0, self.gin.ptr, self.n_in, 0, self.n_in, self.batch_sz,
                      b_out.ptr if fuse else None, b_act, b_prm)
        if b_mask is not None:        # a Hidden layer with dropout right below: its mask, after the copy
            self.ctx.call("tn_scale_mask", self.gin.ptr, b_mask.ptr, 1.0, self.gin.ptr, self.gin.size,
                          None, _lib.TN_ACT_LINEAR, 0.0)
        return self.gin


class SoftAuxLayer(HiddenLayer, OutputLayer):
    """auxiliary.py:104-160: probs = softmax(x.W + b + cross_b + aux_out . cross_w); every one of its eight
    parameter tensors follows the layer's ``reg``."""

    def __init__(self, inpt, wts, rand_gen, n_in, n_out, n_aux, aux_type, reg=(), loss="nll", boost=1,
                 test_version=False):
        if wts is not None and len(wts) == 0:
            wts = None
        hidden_wts = None if wts is None else wts[:2]
        HiddenLayer.__init__(self, inpt, hidden_wts, rand_gen, n_in, n_out, actvn='linear', reg=reg, pdrop=0)
        aux_wts = None if wts is None else wts[2:6]
        self.aux = _AUX_TYPES[aux_type](aux_wts, rand_gen, n_aux=n_aux, boost=boost, test_version=test_version,
                                        batch_sz=self.batch_sz, ctx=self.ctx)
        cross_wts = None if wts is None else wts[6:]
        n_aux_hid, n_aux_out = n_aux
        self.cross_w, self.cross_b = init_wb(cross_wts, rand_gen, (n_aux_out, n_out), n_out,
                                             n_aux_out + n_out, n_aux_out + n_out, 'softmax', 'SoftAuxCross')
        self.hidden_output = self.logits = self.output
        self.aux_inpt = self.aux.aux_inpt
        self.n_aux, self.aux_type, self.boost = n_aux, aux_type, boost
        self.loss = loss
        self.params = self.params + self.aux.params + [self.cross_w, self.cross_b]
        self.representation = "SoftAux In:{:3d} Aux:{} Out:{:3d}" \
            "\n\t  L1:{L1} L2:{L2} Momentum:{momentum} Max Norm:{maxnorm} " \
            "Rate:{rate}".format(n_in, n_aux, n_out, **self.reg)
        self._alloc_head(self.n_out)
        self.probs = self.features = self.logprob
        self.kind = 'SOFTMAX'
        self._cross = self.ctx.empty((self.batch_sz, self.n_out))
        self._daux = self._ws_c = None

    def TestVersion(self, inpt):
        return SoftAuxLayer(inpt, self.params, rand_gen=None, n_in=self.n_in, n_out=self.n_out, n_aux=self.n_aux,
                            aux_type=self.aux_type, boost=self.boost, test_version=True)

    def act_info(self):
        return None, _lib.TN_ACT_LINEAR, 0.0, None

    def forward(self, train=True, y=None, y_row0=0, d_row0=None, cost_scale=None, below=None):
        c, B = self.ctx, self.batch_sz
        c.call("tn_fc_fwd", self.inpt.ptr, self.w.ptr, self.b.ptr, self.logits.ptr, B, self.n_in, self.n_out,
               _lib.TN_ACT_LINEAR, 0.0, None)
        self.aux.forward(train)
        c.call("tn_fc_fwd", self.aux.output.ptr, self.cross_w.ptr, self.cross_b.ptr, self._cross.ptr, B,
               self.n_aux[-1], self.n_out, _lib.TN_ACT_LINEAR, 0.0, None)
        c.call("tn_axpby", self.logits.ptr, self._cross.ptr, B * self.n_out, 1.0, 1.0)
        self._head_rows(HEAD_SOFTMAX, getattr(self, "_loss", None) or loss_code(self.loss or "nll"), self.logits,
                        train, y, y_row0, d_row0)

    def backward(self, gout, need_gin, below):
        c, B = self.ctx, self.batch_sz
        if self._daux is None:
            self._daux = c.empty(self.aux.output.shape)
            self._ws_c = c.empty(((c.lib.tn_fc_wgrad_ws_bytes(B, self.n_aux[-1], self.n_out) + 3) // 4,))
        if self.has_updates():
            c.call("tn_fc_bwd", self.aux.output.ptr, gout.ptr, self.cross_w.ptr, self.grads[6].ptr, self.grads[7].ptr,
                   self._daux.ptr, B, self.n_aux[-1], self.n_out, self._ws_c.ptr, None, 0, 0.0, None)
            self.aux.backward(self._daux, self.grads[2:6])
        return HiddenLayer.backward(self, gout, need_gin, below)
