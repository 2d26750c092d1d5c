"""Fully-connected layer -- host mirror of theanet/layer/hidden.py.

a = act(x . W + b), W is (n_in, n_out); optional non-inverted dropout on the
output (:31-32); the test version drops the mask and scales by (1 - pdrop)
(:50-55).  Note the reference's init quirk: fan_in = fan_out = n_in + n_out (:21-27).
"""
from .. import _lib
from ..device import C8Array
from .dropout import drop_output
from .layer import Layer, activation_by_name
from .weights import init_wb


class HiddenLayer(Layer):

    def __init__(self, inpt, wts,
                 rand_gen=None,
                 n_in=None,
                 n_out=None,
                 pdrop=0,
                 actvn='relu01',
                 reg=()):
        assert wts is not None or rand_gen is not None

        try:
            fan_in_out = n_in + n_out
        except TypeError:
            fan_in_out = None

        self.w, self.b = init_wb(wts, rand_gen, (n_in, n_out), (n_out,),
                                 fan_in_out, fan_in_out, actvn, 'Hid')
        n_in, n_out = self.w.shape
        self.ctx = self.w.ctx

        self.act = activation_by_name(actvn)
        # DTYPE float16: the layer above the conv stack consumes the c8 tensor as it is stored (tn_c8_fc_*: fp16
        # operands through the NCHW row map, fp32 output); further dense layers are fp32 like the reference
        self.c8 = getattr(inpt, "c8", None)
        if self.c8 is not None:
            c, h, wd = self.c8
            assert c * h * wd == n_in, (self.c8, n_in)
            assert self.ctx.lib.tn_c8_fc_supported(inpt.shape[0], c, h * wd, n_out), (
                "DTYPE float16: a dense layer on {} maps of {}x{} needs a multiple of 64 inputs (c8) and of 32 outputs "
                "(got {})".format(c, h, wd, n_out))
            self.inpt = inpt
        else:
            self.inpt = inpt.flatten(2)
            assert self.inpt.shape[1] == n_in, (self.inpt.shape, n_in)
        self.batch_sz = self.inpt.shape[0]
        self.output = self.ctx.empty((self.batch_sz, n_out))
        self.drop = None
        self.test_scale = 1.0
        if pdrop:
            drop_output(self, self.output, pdrop, rand_gen)
        self.gin = None
        self.wgrad_ws = None

        self.params = [self.w, self.b]
        self.n_in, self.n_out = n_in, n_out
        self.actvn = actvn
        self.pdrop = pdrop
        self.reg = {"L1": 0, "L2": 0,
                    "momentum": .95,
                    "maxnorm": 0,
                    "rate": 1}
        self.reg.update(reg)

        self.representation = (
            "Hidden In:{:3d} Out:{:3d} Act:{} Drop%:{}"
            "\n\t  L1:{L1} L2:{L2} Momentum:{momentum} Max Norm:{maxnorm} "
            "Rate:{rate}".format(n_in, n_out, actvn, pdrop, **self.reg))

    def TestVersion(self, inpt):
        test_version = HiddenLayer(inpt, (self.w, self.b),
                                   pdrop=0,
                                   actvn=self.actvn)
        test_version.test_scale = 1 - self.pdrop
        return test_version

    def act_info(self):
        return (self.output, self.act.kind, self.act.prm,
                self.drop.mask if self.drop is not None else None)

    def forward(self, train=True):
        drop = self.drop
        if self.c8 is not None:
            c, h, wd = self.c8
            if drop is not None and not drop.injected and not drop.ready:
                # the mask is drawn by the product's finishing kernel (and kept for the backward pass)
                self.ctx.call("tn_c8_fc_fwd_dropout", self.inpt.ptr, self.w.ptr, self.b.ptr, self.output.ptr, self.batch_sz,
                              c, h * wd, self.n_out, self.act.kind, self.act.prm, drop.mask.ptr, drop.pdrop, drop.seed, 0,
                              drop.d_step.ptr if drop.d_step is not None else None, drop.elem0)
            else:
                if drop is not None:
                    drop.generate()
                self.ctx.call("tn_c8_fc_fwd", self.inpt.ptr, self.w.ptr, self.b.ptr, self.output.ptr, self.batch_sz, c, h * wd,
                              self.n_out, self.act.kind, self.act.prm, drop.mask.ptr if drop is not None else None)
        elif drop is not None and not drop.injected and not drop.ready:
            # the mask is drawn inside the layer's own launch (and kept for the backward pass)
            self.ctx.call("tn_fc_fwd_dropout", self.inpt.ptr, self.w.ptr, self.b.ptr, self.output.ptr,
                          self.batch_sz, self.n_in, self.n_out, self.act.kind, self.act.prm,
                          drop.mask.ptr, drop.pdrop, drop.seed, 0,
                          drop.d_step.ptr if drop.d_step is not None else None, drop.elem0)
        else:
            if drop is not None:
                drop.generate()
            self.ctx.call("tn_fc_fwd", self.inpt.ptr, self.w.ptr, self.b.ptr, self.output.ptr,
                          self.batch_sz, self.n_in, self.n_out, self.act.kind, self.act.prm,
                          drop.mask.ptr if drop is not None else None)
        if self.test_scale != 1.0:
            self.ctx.call("tn_scale_mask", self.output.ptr, None, float(self.test_scale),
                          self.output.ptr, self.output.size, None, _lib.TN_ACT_LINEAR, 0.0)

    def backward(self, gout, need_gin, below):
        """gout = d cost / d z (activation gradient and dropout mask already applied)."""
        if self.c8 is not None:
            c, h, wd = self.c8
            if self.has_updates():
                self.ctx.call("tn_c8_fc_wgrad", self.inpt.ptr, gout.ptr, self.grads[0].ptr, self.grads[1].ptr,
                              self.batch_sz, c, h * wd, self.n_out)
            if not need_gin:
                return None
            if self.gin is None:
                self.gin = C8Array(self.ctx, self.batch_sz, c, h, wd)
            b_out, b_act, b_prm, b_mask = below.act_info()
            assert b_mask is None
            fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
            self.ctx.call("tn_c8_fc_dgrad", gout.ptr, self.w.ptr, self.gin.ptr, self.batch_sz, c, h * wd, self.n_out,
                          b_out.ptr if fuse else None, b_act, b_prm)
            return self.gin
        if self.has_updates() and need_gin:
            # weight gradient and input gradient only share dz: one op, one launch
            if self.wgrad_ws is None:
                nbytes = self.ctx.lib.tn_fc_wgrad_ws_bytes(self.batch_sz, self.n_in, self.n_out)
                self.wgrad_ws = self.ctx.empty((nbytes + 3) // 4)
            if self.gin is None:
                self.gin = self.ctx.empty(self.inpt.shape)
            b_out, b_act, b_prm, b_mask = below.act_info()
            fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
            self.ctx.call("tn_fc_bwd", self.inpt.ptr, gout.ptr, self.w.ptr, self.grads[0].ptr,
                          self.grads[1].ptr, self.gin.ptr, self.batch_sz, self.n_in, self.n_out,
                          self.wgrad_ws.ptr, b_out.ptr if fuse else None, b_act, b_prm,
                          b_mask.ptr if b_mask is not None else None)
            return self.gin
        if self.has_updates():
            if self.wgrad_ws is None:
                nbytes = self.ctx.lib.tn_fc_wgrad_ws_bytes(self.batch_sz, self.n_in, self.n_out)
                self.wgrad_ws = self.ctx.empty((nbytes + 3) // 4)
            self.ctx.call("tn_fc_wgrad", self.inpt.ptr, gout.ptr, self.grads[0].ptr,
                          self.grads[1].ptr, self.batch_sz, self.n_in, self.n_out,
                          self.wgrad_ws.ptr)
        if not need_gin:
            return None
        if self.gin is None:
            self.gin = self.ctx.empty(self.inpt.shape)
        b_out, b_act, b_prm, b_mask = below.act_info()
        fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
        self.ctx.call("tn_fc_dgrad", gout.ptr, self.w.ptr, self.gin.ptr, self.batch_sz, self.n_in,
                      self.n_out, b_out.ptr if fuse else None, b_act, b_prm,
                      b_mask.ptr if b_mask is not None else None)
        return self.gin
