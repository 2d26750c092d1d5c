"""Output heads -- host mirror of theanet/layer/outlayers.py.

SoftmaxLayer + 'nll' is the hot path: logprob = log_softmax(x.W + b); features = logprob; y_preds =
argmax (first maximum); cost('nll') = -mean(logprob[n, y_n]) (:50-51); error statistics (:69-80) -- one
fused HIP kernel produces logprob, the per-row loss, argmax, P(label) and d cost / d logits (and, in
training, the layer's whole backward).  The other losses of SoftmaxLayer ('nllsq', 'nllNN', 'hinge',
'exp'; :12-64) and the other heads (ExpLossLayer :105-126, HingeLayer :129-147, CenteredOutLayer
:153-224) run the affine map (tn_fc_fwd, with the head's activation) followed by one row kernel
(tn_head_rows: log-probabilities, predictions, second error statistic, per-row loss and d cost / d z).
"""
import numpy as np

from .. import _lib
from .hidden import HiddenLayer


HEAD_SOFTMAX, HEAD_EXPLOSS, HEAD_HINGE, HEAD_LOGIT, HEAD_RBF = range(5)
LOSS_NLL, LOSS_NLLSQ, LOSS_NLLTRUNC, LOSS_HINGE, LOSS_EXP = range(5)


def loss_code(loss):
    """(code, parameter) of a loss name, with the reference's parsing (outlayers.py:12-36): 'nll', 'nllsq',
    'nll<NN>' = negative log-likelihood truncated at probability NN/100 (anything unparsable after 'nll':
    plain nll), 'hinge', 'exp'; unknown names raise NotImplementedError."""
    if loss == "nll":
        return LOSS_NLL, 0.0
    if loss == "nllsq":
        return LOSS_NLLSQ, 0.0
    if isinstance(loss, str) and loss.startswith("nll"):
        try:
            threshold = float(np.clip(int(loss[-2:]) / 100, 0, 1))
        except ValueError:
            print("Did not understand {}, using plain NLL".format(loss))
            threshold = 1.0
        print("Using threshold: ", threshold)
        with np.errstate(divide="ignore"):
            return LOSS_NLLTRUNC, float(np.log(threshold))
    if loss == "hinge":
        return LOSS_HINGE, 0.0
    if loss == "exp":
        return LOSS_EXP, 0.0
    raise NotImplementedError("Loss : " + str(loss))


class OutputLayer(object):
    def cost(self, y):
        """Validates the loss name like outlayers.py:12-36; the value itself is
        produced on the device by ``forward``."""
        self._loss = loss_code(self.loss)
        return self.d_cost

    def neg_log_likli(self, y):
        return self.d_cost

    def _alloc_head(self, ncols, feat_cols=None):
        """Buffers every head exposes to NeuralNet: logprob (B, ncols), y_preds, per-row loss, per-row second
        statistic, d cost / d z (``dlogits``), the two error-rate scalars."""
        ctx, B = self.ctx, self.batch_sz
        self.logprob = ctx.empty((B, ncols))
        self.y_preds = ctx.empty((B,), np.int32)
        self.rowloss = ctx.empty((B,))
        self.rowp = ctx.empty((B,))
        self.dlogits = ctx.empty((B, self.n_out))
        self.d_cost = None                             # device scalar, owned by the net
        self.d_stats = ctx.empty((2,))
        self.inv_batch = 1.0 / B

    def _head_rows(self, head, loss, a, train, y, y_row0, d_row0, feat=None, centers=None, ncls=0, dcenters=None,
                   junk_dist=0.0):
        have_y = y is not None
        code, prm = loss
        self.ctx.call("tn_head_rows", head, code, prm, a.ptr, centers.ptr if centers is not None else None, ncls,
                      y.ptr if have_y else None, int(y_row0), d_row0.ptr if d_row0 is not None else None,
                      feat.ptr if feat is not None else None, self.logprob.ptr,
                      self.rowloss.ptr if have_y else None, self.y_preds.ptr, self.rowp.ptr if have_y else None,
                      self.dlogits.ptr if (have_y and train) else None,
                      dcenters.ptr if (dcenters is not None and have_y and train) else None,
                      self.batch_sz, a.shape[1], float(self.inv_batch), float(junk_dist), self.act.kind or 0,
                      self.act.prm)

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
        if self.loss not in (None, "nll"):
            # the other losses on the softmax head (:38-64): affine map, then the generic row kernel
            self.ctx.call("tn_fc_fwd", self.inpt.ptr, self.w.ptr, self.b.ptr, self.logits.ptr,
                          self.batch_sz, self.n_in, self.n_out, _lib.TN_ACT_LINEAR, 0.0, None)
            self._head_rows(HEAD_SOFTMAX, getattr(self, "_loss", None) or loss_code(self.loss), self.logits, train,
                            y, y_row0, d_row0)
            return
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


class _LinearHead(HiddenLayer, OutputLayer):
    """Common part of ExpLossLayer / HingeLayer: a linear HiddenLayer followed by tn_head_rows."""
    HEAD, KIND, LOSS, NAME = None, None, None, None

    def __init__(self, inpt, wts, rand_gen=None, n_in=None, n_out=None, reg=()):
        HiddenLayer.__init__(self, inpt, wts, rand_gen, n_in, n_out, actvn='linear', reg=reg, pdrop=0)
        self._alloc_head(self.n_out)
        self.kind, self.loss = self.KIND, self.LOSS
        self._loss = loss_code(self.LOSS)
        self.probs = self.logprob
        self.representation = self.NAME + " In:{:3d} Out:{:3d} Loss:{}" \
            "\n\t  L1:{L1} L2:{L2} Momentum:{momentum} Max Norm:{maxnorm} " \
            "Rate:{rate}""".format(self.n_in, self.n_out, self.loss, **self.reg)

    def TestVersion(self, inpt):
        return type(self)(inpt, (self.w, self.b))

    def act_info(self):
        return None, _lib.TN_ACT_LINEAR, 0.0, None

    def forward(self, train=True, y=None, y_row0=0, d_row0=None, cost_scale=None, below=None):
        self.ctx.call("tn_fc_fwd", self.inpt.ptr, self.w.ptr, self.b.ptr, self.output.ptr,
                      self.batch_sz, self.n_in, self.n_out, _lib.TN_ACT_LINEAR, 0.0, None)
        self._head_rows(self.HEAD, self._loss, self.output, train, y, y_row0, d_row0, feat=self._feat)


class ExpLossLayer(_LinearHead):
    """outlayers.py:105-126: y_preds = argmax(x.W + b); output = that minus its row mean (= features);
    probs = softmax(output); cost = mean exp(-output[n, y_n])."""
    HEAD, KIND, LOSS, NAME = HEAD_EXPLOSS, 'ExpLoss', 'exp', 'ExpLoss'

    def __init__(self, inpt, wts, rand_gen=None, n_in=None, n_out=None, reg=()):
        _LinearHead.__init__(self, inpt, wts, rand_gen, n_in, n_out, reg)
        self.features = self._feat = self.ctx.empty((self.batch_sz, self.n_out))


class HingeLayer(_LinearHead):
    """outlayers.py:129-147: logprob = probs = features = x.W + b; cost = mean over ALL (n, c) of
    max(0, out[n,c] + 1 - out[n, y_n]); the second error statistic is mean out[n, y_n]."""
    HEAD, KIND, LOSS, NAME = HEAD_HINGE, 'Hinge', 'hinge', 'SVM'

    def __init__(self, inpt, wts, rand_gen=None, n_in=None, n_out=None, reg=()):
        _LinearHead.__init__(self, inpt, wts, rand_gen, n_in, n_out, reg)
        self._feat = None
        self.features = self.logprob


activs = {'LOGIT': 'sigmoid', 'RBF': 'scaled_tanh'}


class CenteredOutLayer(HiddenLayer, OutputLayer):
    """outlayers.py:153-224: a hidden layer (sigmoid for LOGIT, scaled_tanh for RBF) whose features are compared
    with one center per class.  LOGIT: logprob[n,k] = sum_f log(c v' + (1-c)(1-v')), v' = v(1-2e)+e; the second
    error statistic is the bit error rate.  RBF: probs = softmax(-[squared distances, junk_dist]) over
    n_classes + 1 columns; centers may be learned.
    Deviations, both where the reference cannot run: it never sets ``self.loss`` on this layer (its ``cost``
    raises AttributeError) -- 'nll' = -mean logprob[n, y_n] is used; and ``get_wts`` includes the centers, so a
    checkpoint can rebuild the layer (the reference reads ``wts[3]`` of a 2- or 3-element list)."""

    def __init__(self, inpt, wts, centers, rand_gen=None,
                 n_in=None, n_features=None, n_classes=None,
                 kind='LOGIT', learn_centers=False, junk_dist=np.inf,
                 reg=()):
        assert kind in activs
        assert n_in or wts
        assert n_features or wts or centers is not None
        assert n_classes or centers is not None
        assert kind == 'RBF' or not learn_centers
        HiddenLayer.__init__(self, inpt, wts, rand_gen, n_in, n_out=n_features,
                             actvn=activs[kind], pdrop=0, reg=reg)
        from .weights import is_shared_var
        if centers is None:
            if kind == 'LOGIT':
                centers_vals = rand_gen.binomial(n=1, p=.5, size=(n_classes, n_features))
            else:
                centers_vals = rand_gen.uniform(low=0, high=1, size=(n_classes, n_features))
            centers = np.asarray(centers_vals, dtype=np.float32)
        self.centers = centers if is_shared_var(centers) else self.ctx.array(np.asarray(centers, np.float32))
        self.centers.name = 'centers'
        self.learn_centers = learn_centers
        if learn_centers:
            self.params.append(self.centers)
        n_classes, n_features = self.centers.shape
        assert n_features == self.n_out
        self.n_classes, self.n_features = n_classes, n_features
        self.kind, self.junk_dist, self.loss = kind, junk_dist, 'nll'
        self._loss = loss_code('nll')
        self.features = self.output
        self._alloc_head(n_classes + (1 if kind == 'RBF' else 0))
        self.probs = self.logprob
        self.representation = ('CenteredOut Kind:{} In:{:3d} Hidden:{:3d} '
                               'Out:{:3d} learn_centers:{} junk_dist:{}'.format(
            kind, self.n_in, n_features, n_classes, learn_centers, junk_dist))

    def TestVersion(self, inpt):
        return CenteredOutLayer(inpt, (self.w, self.b), self.centers,
                                kind=self.kind, junk_dist=self.junk_dist)

    def get_wts(self):
        if self._wts_hook is not None:
            self._wts_hook()
        return [self.w.get_value(), self.b.get_value(), self.centers.get_value()]

    def act_info(self):
        return None, _lib.TN_ACT_LINEAR, 0.0, None

    def forward(self, train=True, y=None, y_row0=0, d_row0=None, cost_scale=None, below=None):
        self.ctx.call("tn_fc_fwd", self.inpt.ptr, self.w.ptr, self.b.ptr, self.output.ptr,
                      self.batch_sz, self.n_in, self.n_out, self.act.kind, self.act.prm, None)
        dcent = None
        if self.learn_centers and train and y is not None and self.grads is not None:
            dcent = self.grads[2]
            dcent.fill_bytes(0)
        jd = self.junk_dist if np.isfinite(self.junk_dist) else 3.0e38
        self._head_rows(HEAD_LOGIT if self.kind == 'LOGIT' else HEAD_RBF, self._loss, self.output, train, y, y_row0,
                        d_row0, centers=self.centers, ncls=self.n_classes, dcenters=dcent, junk_dist=jd)

    def backward(self, gout, need_gin, below):
        """gout = dlogits = d cost / d z of the hidden map (the row kernel already applied act')."""
        if not self.learn_centers:
            return HiddenLayer.backward(self, gout, need_gin, below)
        # the generic FC backward writes grads[0], grads[1]; the centers' gradient came from the row kernel
        return HiddenLayer.backward(self, gout, need_gin, below)
