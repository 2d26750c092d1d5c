"""Output head -- host mirror of theanet/layer/outlayers.py (Softmax + 'nll').

logprob = log_softmax(x.W + b); features = logprob; y_preds = argmax (first
maximum); cost('nll') = -mean(logprob[n, y_n]) (:50-51); error statistics
(:69-80).  One fused HIP kernel produces logprob, the per-row loss, argmax,
P(label) and d cost / d logits.  The other heads of the reference (hinge, exp,
truncated nll, centered) are outside the accelerated path.
"""
import numpy as np

from .. import _lib
from .hidden import HiddenLayer


class OutputLayer(object):
    def cost(self, y):
        """Validates the loss name like outlayers.py:12-36; the value itself is
        produced on the device by ``forward``."""
        if self.loss == "nll":
            return self.neg_log_likli(y)
        raise NotImplementedError("Loss : " + str(self.loss))

    def neg_log_likli(self, y):
        return self.d_cost

    def features_and_predictions(self):
        return self.features, self.y_preds

    def sym_and_oth_err_rate(self, y):
        return self.d_stats


class SoftmaxLayer(HiddenLayer, OutputLayer):
    def __init__(self, inpt, wts, rand_gen=None, n_in=None, n_out=None,
                 reg=(),
                 loss="nll"):
        HiddenLayer.__init__(self, inpt, wts, rand_gen, n_in, n_out,
                             actvn='Softmax', reg=reg,
                             pdrop=0)
        ctx, B = self.ctx, self.batch_sz
        self.logits = self.output                      # x.W + b  (linear epilogue)
        self.logprob = ctx.empty((B, self.n_out))
        self.probs = self.logprob                      # exp() taken on the host when asked
        self.features = self.logprob
        self.y_preds = ctx.empty((B,), np.int32)
        self.rowloss = ctx.empty((B,))
        self.rowp = ctx.empty((B,))
        self.dlogits = ctx.empty((B, self.n_out))
        self.d_cost = None                             # device scalar, owned by the net
        self.cost_ws = None
        self.d_stats = ctx.empty((2,))
        self.kind = 'SOFTMAX'
        self.loss = loss
        self.labels = None                             # (DeviceArray int32, row0) bound by the net
        self.inv_batch = 1.0 / B
        self.representation = "Softmax In:{:3d} Out:{:3d} Loss:{}" \
            "\n\t  L1:{L1} L2:{L2} Momentum:{momentum} Max Norm:{maxnorm} " \
            "Rate:{rate}""".format(self.n_in, self.n_out,
                                   self.loss, **self.reg)

    def TestVersion(self, inpt):
        return SoftmaxLayer(inpt, (self.w, self.b), loss=None)

    def act_info(self):
        return None, _lib.TN_ACT_LINEAR, 0.0, None

    def backward(self, gout, need_gin, below):
        if getattr(self, "_bwd_done", False):       # produced by the fused forward of this step
            self._bwd_done = False
            return self.gin if need_gin else None
        return HiddenLayer.backward(self, gout, need_gin, below)

    def forward(self, train=True, y=None, y_row0=0, d_row0=None, cost_scale=None, below=None):
        """Logits GEMM + the fused softmax/NLL row kernel (+ the cost scalar when training:
        cost = cost_scale * sum_n -logprob[n, y_n], reduced inside the same launch).

        ``below`` (training only): the layer under this one -- the forward then also produces this
        layer's weight gradients and the gradient w.r.t. its input (one op, tn_fc_softmax_train);
        ``backward`` returns that result."""
        have_y = y is not None
        self._bwd_done = False
        if have_y and train and below is not None and cost_scale is None and self.has_updates():
            if self.wgrad_ws is None:
                nbytes = self.ctx.lib.tn_fc_wgrad_ws_bytes(self.batch_sz, self.n_in, self.n_out)
                self.wgrad_ws = self.ctx.empty((nbytes + 3) // 4)
            if self.gin is None:
                self.gin = self.ctx.empty(self.inpt.shape)
            b_out, b_act, b_prm, b_mask = below.act_info()
            fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
            self.ctx.call("tn_fc_softmax_train", self.inpt.ptr, self.w.ptr, self.b.ptr, self.logits.ptr,
                          self.batch_sz, self.n_in, self.n_out, y.ptr, int(y_row0),
                          d_row0.ptr if d_row0 is not None else None, self.logprob.ptr,
                          self.rowloss.ptr, self.y_preds.ptr, self.rowp.ptr, self.dlogits.ptr,
                          float(self.inv_batch), self.grads[0].ptr, self.grads[1].ptr, self.gin.ptr,
                          self.wgrad_ws.ptr, b_out.ptr if fuse else None, b_act, b_prm,
                          b_mask.ptr if b_mask is not None else None)
            self._bwd_done = True
            return
        if not (have_y and train and cost_scale is not None and self.d_cost is not None):
            # affine map + softmax / NLL rows as ONE op (one launch for <= 16 classes)
            self.ctx.call("tn_fc_softmax_nll", self.inpt.ptr, self.w.ptr, self.b.ptr, self.logits.ptr,
                          self.batch_sz, self.n_in, self.n_out, y.ptr if have_y else None, int(y_row0),
                          d_row0.ptr if d_row0 is not None else None, self.logprob.ptr,
                          self.rowloss.ptr if have_y else None, self.y_preds.ptr,
                          self.rowp.ptr if have_y else None,
                          self.dlogits.ptr if (have_y and train) else None, float(self.inv_batch))
            return
        # explicit in-launch cost reduction (kept for callers that ask for it; the training step
        # lets the cost ride in the update launch instead)
        self.ctx.call("tn_fc_fwd", self.inpt.ptr, self.w.ptr, self.b.ptr, self.logits.ptr,
                      self.batch_sz, self.n_in, self.n_out, _lib.TN_ACT_LINEAR, 0.0, None)
        if self.cost_ws is None:
            n = self.ctx.lib.tn_softmax_cost_ws_bytes(self.batch_sz)
            self.cost_ws = self.ctx.zeros(((n + 3) // 4,))
        self.ctx.call("tn_softmax_nll_cost", self.logits.ptr, y.ptr, int(y_row0),
                      d_row0.ptr if d_row0 is not None else None, self.logprob.ptr,
                      self.rowloss.ptr, self.y_preds.ptr, self.rowp.ptr, self.dlogits.ptr,
                      self.batch_sz, self.n_out, float(self.inv_batch), float(cost_scale),
                      self.d_cost.ptr, self.cost_ws.ptr)
