"""
ORACLE -- CPU (numpy) restatement of theanet's convolutional training hot path.

THIS IS TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product (``theanet_amd``)
never does and fails loudly when its HIP library is missing.

What it restates (all paths relative to /root/reference):
  theanet/layer/layer.py      activations :27-54, momentum-SGD + maxnorm :70-107,
                              L1/L2 cost :109-117
  theanet/layer/convpool.py   ConvLayer :14-95, PoolLayer :97-127, MeanLayer :129-144
  theanet/layer/hidden.py     HiddenLayer :11-55
  theanet/layer/dropout.py    drop_output :9-13, DropOutLayer :15-31
  theanet/layer/outlayers.py  SoftmaxLayer :83-102, losses :12-64, error rates :69-80,
                              ExpLossLayer :105-126, HingeLayer :129-147,
                              CenteredOutLayer :153-224
  theanet/layer/color.py      ColorLayer :9-52
  theanet/layer/auxiliary.py  LocationInfo :14-58, AuxConcatLayer :64-101, SoftAuxLayer :104-160
  theanet/layer/weights.py    init_wb :25-81
  theanet/layer/inlayers.py   InputLayer :12-26, ElasticLayer :29-163
  theanet/neuralnet.py        NeuralNet :60-111,:113-201,:203-241,:257-277,:303-311
  extras/deformer.py          transform :7-18

The arithmetic of the reference lives in Theano (third-party, un-vendored,
un-pinned -- setup.py:14-17; README.md:15-17 installs git master), which cannot
be installed here.  The Theano semantics restated below are its published ones
(Theano 0.8-1.0): conv2d(filter_flip=True) = true convolution; subsample keeps
every s-th output of the stride-1 result; pool_2d mode='max', stride = window,
no padding; MaxPoolGrad credits every tied maximum; Maximum/Minimum gradients
use eq(out, x) (ties feed both branches); log(softmax(x)) is the stable
log-softmax; simultaneous ``updates`` read pre-step values.

PARITY UNPINNED against the reference itself for everything except the two pieces of it
that run without Theano: ``deformer_transform`` (pinned against the reference's own
extras/deformer.py:7-18 executed in the build container, fixture tests/golden/deformer.npz)
and ``init_wb`` + the numpy seed chain (pinned against the reference's own draw lines,
theanet/layer/weights.py:51-65, compiled in place: fixture tests/golden/init_ref.npz holds
the initial weights of mnist.prms and one layer per scale / bias rule, reproduced bit for
bit).  The reference holds no golden vectors, no asserts (tests/test_elastic.py has none)
and the rest of it cannot run.
The remaining functions are pinned by analytic known-answer tests, float64
finite-difference gradient checks (whole nets, every head / loss, mid-net Color and
Elastic layers) and by an INDEPENDENT second implementation: a whole mnist.prms
training trajectory (forward, all gradients, three old-velocity updates) computed by
torch CPU autograd (tests/golden/make_torch_xchk.py -> torch_xchk.npz), which this
module reproduces to 1e-9 (tests/test_oracle_kat.py).

Beyond the reference (it is float32-only, weights.py:8): DTYPE 'float16' = the STORED-fp16 mode of the fp16-resident
kernels (OracleNet): every tensor of the conv stack -- activations going up, gradients coming down -- is rounded to IEEE
half when it is stored (gradients as r16(GRAD_SCALE * g) / GRAD_SCALE), weights when they are used, products exact,
sums float64 (the device: fp32), bias / activation / pooling / ties on the unrounded sums, activation derivatives from
the stored output; the dense layer on top of the stack takes the halfs as its input operand and is fp32 from there on.
``r16`` and the ``f16=True``
modes of conv2d_fwd / conv2d_bwd specify the build's DTYPE='float16' (fp16-rounded
operands, exact products, float64 sums).
"""
import math

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

from .randomstreams import RandomStreams

# --------------------------------------------------------------------------- #
# activations  (layer.py:27-54)
# --------------------------------------------------------------------------- #


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def activation(name):
    """Return (f, df) for an activation name; df(z) = dact/dz with Theano's tie rules.

    relu{ii}: max(0,z) + min(0,z)*ii/100 (layer.py:36-38); Theano's Maximum /
    Minimum gradient is eq(out, x), so at z == 0 both branches fire: 1 + ii/100.
    """
    if name in ("Softmax", "softmax"):
        raise ValueError("softmax is handled by SoftmaxLayer")
    if name == "linear":
        return (lambda z: z), (lambda z: np.ones_like(z))
    if name == "sigmoid":
        return _sigmoid, (lambda z: _sigmoid(z) * (1 - _sigmoid(z)))
    if name == "softplus":
        return (lambda z: np.logaddexp(z, 0).astype(z.dtype)), _sigmoid
    if name == "tanh":
        return np.tanh, (lambda z: 1 - np.tanh(z) ** 2)
    if name == "scaled_tanh":
        return ((lambda z: (1.7 * np.tanh(2 * z / 3)).astype(z.dtype)),
                (lambda z: (1.7 * 2 / 3 * (1 - np.tanh(2 * z / 3) ** 2)).astype(z.dtype)))
    if name == "relu":
        return ((lambda z: np.maximum(0, z)),
                (lambda z: (z >= 0).astype(z.dtype)))
    if len(name) == 6 and name.startswith("relu") and name[4:].isdigit():
        ii = int(name[4:])

        def f(z):
            return (np.maximum(0, z) + np.minimum(0, z) * ii / 100).astype(z.dtype)

        def df(z):
            return ((z >= 0).astype(z.dtype) +
                    (z <= 0).astype(z.dtype) * z.dtype.type(ii / 100))
        return f, df
    raise NotImplementedError("Unknown Activation Specified: " + name)


def act_grad_from_out(name, a):
    """act'(z) expressed through the layer's STORED output a = act(z) -- how the device's backward kernels take it
    (documented deviation: for slope 0 an exact a == 0 reads as z < 0)."""
    if name == "linear":
        return np.ones_like(a)
    s = leaky_slope(name)
    if s is not None:
        tie = 1.0 + s if s > 0 else 0.0
        return np.where(a > 0, 1.0, np.where(a < 0, s, tie))
    if name == "tanh":
        return 1 - a * a
    if name == "sigmoid":
        return a * (1 - a)
    if name == "softplus":
        return 1 - np.exp(-a)
    if name == "scaled_tanh":
        return 1.7 * 2 / 3 * (1 - (a / 1.7) ** 2)
    raise NotImplementedError(name)


def leaky_slope(name):
    """(slope for z<0) of a relu-family activation, or None (for the HIP fused epilogues)."""
    if name == "relu":
        return 0.0
    if len(name) == 6 and name.startswith("relu") and name[4:].isdigit():
        return int(name[4:]) / 100
    if name == "linear":
        return 1.0
    return None


# --------------------------------------------------------------------------- #
# weight init  (weights.py:25-81)
# --------------------------------------------------------------------------- #


def init_wb(rand_gen, size_w, size_b, fan_in, fan_out, actvn, dtype=np.float32):
    if len(size_w) == 4:
        w = 2. * rand_gen.randint(2, size=size_w) - 1          # weights.py:52
        w /= np.sqrt(fan_in)                                   # :54
    else:
        w = rand_gen.uniform(low=-1, high=1, size=size_w)      # :56
        w *= np.sqrt(6 / (fan_in + fan_out))                   # :57
    w = np.asarray(w, dtype=dtype)                             # :59
    b = np.zeros(size_b, dtype=dtype)                          # :60
    if actvn == 'sigmoid':
        w *= 4                                                 # :62-63
    if actvn in ('softplus', 'relu') or actvn.startswith('relu0'):
        b += .5                                                # :64-65
    return w, b


# --------------------------------------------------------------------------- #
# conv  (convpool.py:14-95)
# --------------------------------------------------------------------------- #


def conv_geometry(in_sz, f, stride, mode):
    """(pad_lo, pad_hi, out_sz).  out_sz follows the reference formula
    (convpool.py:57-70) and is asserted equal to what Theano computes."""
    if mode == "valid":
        pad_lo = pad_hi = 0
        out = in_sz - f + 1
    elif mode == "same":
        assert stride == 1, "For Same mode stride should be 1"
        shift = (f - 1) // 2          # full conv cropped [shift:in+shift] (convpool.py:59-60)
        pad_lo, pad_hi = f - 1 - shift, shift
        out = in_sz
    else:
        raise NotImplementedError(
            "mode 'full': reference out_sz = in+f+1 is wrong (convpool.py:63-64)")
    out //= stride
    theano_out = (in_sz + pad_lo + pad_hi - f) // stride + 1
    assert out == theano_out, (
        "stride {} does not divide the stride-1 output ({}): reference out_sz {} != "
        "Theano's {}".format(stride, in_sz + pad_lo + pad_hi - f + 1, out, theano_out))
    return pad_lo, pad_hi, out


def _windows(x, f, stride, pad_lo, pad_hi):
    xp = np.pad(x, ((0, 0), (0, 0), (pad_lo, pad_hi), (pad_lo, pad_hi)))
    win = sliding_window_view(xp, (f, f), axis=(2, 3))      # N,C,Ho1,Wo1,f,f
    return xp, win[:, :, ::stride, ::stride]


def r16(a, scale=1.0):
    """fp16 operand rounding of the DTYPE='float16' mode (NOT in the reference, which is float32-only:
    weights.py:8; BASELINE.json configs[4]): round-to-nearest-even to IEEE half (scale*a, a power-of-two
    scale being exact), widened to float64 so that the products and sums that follow are exact /
    float64 -- the device accumulates the same exact products in fp32."""
    return (np.asarray(a, np.float64) * scale).astype(np.float16).astype(np.float64) / scale


def conv2d_fwd(x, W, b, stride=1, mode="valid", f16=False):
    """z[n,k,i,j] = b[k] + sum_{c,u,v} xpad[n,c,i*s+u,j*s+v] * W[k,c,f-1-u,f-1-v].
    f16: both operands rounded to fp16 first, float64 accumulation (see r16)."""
    f = W.shape[2]
    out_dtype = x.dtype
    if f16:
        x, W = r16(x), r16(W)
    pad_lo, pad_hi, _ = conv_geometry(x.shape[2], f, stride, mode)
    _, win = _windows(x, f, stride, pad_lo, pad_hi)
    Wf = W[:, :, ::-1, ::-1]
    z = np.tensordot(win, Wf, axes=([1, 4, 5], [1, 2, 3]))   # N,Ho,Wo,K
    z = np.ascontiguousarray(z.transpose(0, 3, 1, 2)) + b[None, :, None, None]
    return z.astype(out_dtype)


def conv2d_bwd(x, W, dz, stride=1, mode="valid", need_dx=True, f16=False, grad_scale=1.0):
    """Returns (dx or None, dW, db) for z = conv2d_fwd(x, W, b).
    f16: the operands of both gradient products are rounded to fp16 (dz as r16(grad_scale*dz)/grad_scale);
    db sums the unrounded dz."""
    f = W.shape[2]
    H = x.shape[2]
    out_dtype = x.dtype
    db = dz.sum(axis=(0, 2, 3)).astype(out_dtype)
    if f16:
        x, W, dz = r16(x), r16(W), r16(dz, grad_scale)
    pad_lo, pad_hi, _ = conv_geometry(H, f, stride, mode)
    xp, win = _windows(x, f, stride, pad_lo, pad_hi)
    # dWf[k,c,u,v] = sum_{n,i,j} dz[n,k,i,j] * win[n,c,i,j,u,v]
    dWf = np.tensordot(dz, win, axes=([0, 2, 3], [0, 2, 3]))  # K,C,f,f
    dW = np.ascontiguousarray(dWf[:, :, ::-1, ::-1]).astype(out_dtype)
    dx = None
    if need_dx:
        Wf = W[:, :, ::-1, ::-1]
        dxp = np.zeros_like(xp)
        Ho, Wo = dz.shape[2], dz.shape[3]
        for u in range(f):
            for v in range(f):
                # dxp[n,c,i*s+u,j*s+v] += sum_k dz[n,k,i,j] * Wf[k,c,u,v]
                contrib = np.tensordot(dz, Wf[:, :, u, v], axes=([1], [0]))  # N,Ho,Wo,C
                dxp[:, :, u:u + stride * Ho:stride, v:v + stride * Wo:stride] += \
                    contrib.transpose(0, 3, 1, 2)
        dx = np.ascontiguousarray(
            dxp[:, :, pad_lo:pad_lo + H, pad_lo:pad_lo + H]).astype(out_dtype)
    return dx, dW, db


# --------------------------------------------------------------------------- #
# pool / mean  (convpool.py:97-144)
# --------------------------------------------------------------------------- #


def pool_out_sz(in_sz, p, ignore_border):
    return in_sz // p if ignore_border else math.ceil(in_sz / p)


def _pool_windows(x, p, ignore_border, fill):
    N, C, H, W = x.shape
    Ho = pool_out_sz(H, p, ignore_border)
    Wo = pool_out_sz(W, p, ignore_border)
    xp = np.full((N, C, Ho * p, Wo * p), fill, dtype=x.dtype)
    h, w = min(H, Ho * p), min(W, Wo * p)
    xp[:, :, :h, :w] = x[:, :, :h, :w]
    return xp.reshape(N, C, Ho, p, Wo, p), (h, w)


def pool_fwd(x, p, ignore_border=False):
    win, _ = _pool_windows(x, p, ignore_border, -np.inf)
    return win.max(axis=(3, 5))


def pool_bwd(x, dy, p, ignore_border=False):
    """Theano MaxPoolGrad: every element equal to its window max receives dy."""
    N, C, H, W = x.shape
    win, (h, w) = _pool_windows(x, p, ignore_border, -np.inf)
    mx = win.max(axis=(3, 5), keepdims=True)
    g = (win == mx).astype(x.dtype) * dy[:, :, :, None, :, None]
    g = g.reshape(N, C, win.shape[2] * p, win.shape[4] * p)
    dx = np.zeros_like(x)
    dx[:, :, :h, :w] = g[:, :, :h, :w]
    return dx


def mean_fwd(x):
    return x.mean(axis=(2, 3), dtype=x.dtype)


def mean_bwd(x, dy):
    return np.broadcast_to(dy[:, :, None, None] / x.dtype.type(x.shape[2] * x.shape[3]),
                           x.shape).astype(x.dtype)


# --------------------------------------------------------------------------- #
# softmax + nll  (outlayers.py:50-51, 69-80, 87-95)
# --------------------------------------------------------------------------- #


def log_softmax(z):
    m = z.max(axis=1, keepdims=True)
    e = z - m
    return (e - np.log(np.exp(e).sum(axis=1, keepdims=True))).astype(z.dtype)


def nll(logprob, y):
    return -logprob[np.arange(len(y)), y].mean(dtype=logprob.dtype)


def nll_dlogits(logprob, y):
    g = np.exp(logprob)
    g[np.arange(len(y)), y] -= 1
    return (g / logprob.dtype.type(len(y))).astype(logprob.dtype)


# --------------------------------------------------------------------------- #
# the other losses and output heads  (outlayers.py:12-64, 105-224)
# --------------------------------------------------------------------------- #


def parse_loss(loss):
    """outlayers.py:12-36 -> (name, threshold)."""
    if loss in ("nll", "nllsq", "hinge", "exp"):
        return loss, None
    if isinstance(loss, str) and loss.startswith("nll"):
        try:
            threshold = float(np.clip(int(loss[-2:]) / 100, 0, 1))     # :22-24
        except ValueError:
            threshold = 1.0                                           # :25-27
        return "nlltrunc", threshold
    raise NotImplementedError("Loss : " + str(loss))


def softmax_head(z, y, loss="nll"):
    """SoftmaxLayer with any of its losses (outlayers.py:83-102 + :38-64).  Returns
    (logprob, preds, second_stat_rows, cost, dz) with dz = d cost / d z."""
    lp = log_softmax(z)
    p = np.exp(lp)
    B, n = z.shape
    r = np.arange(B)
    name, thr = parse_loss(loss)
    onehot = np.zeros_like(z)
    onehot[r, y] = 1
    if name == "nll":
        cost, glp = -lp[r, y].mean(), -np.ones(B) / B                              # :50-51
    elif name == "nllsq":
        cost, glp = (lp[r, y] ** 2).mean(), 2 * lp[r, y] / B                       # :41-42
    elif name == "nlltrunc":
        with np.errstate(divide="ignore"):
            t = np.log(thr) - lp[r, y]                                             # :44-48
        cost, glp = np.maximum(0, t).mean(), -(t >= 0).astype(z.dtype) / B
    if name in ("nll", "nllsq", "nlltrunc"):
        dz = glp[:, None] * (onehot - p)
    else:
        if name == "hinge":                                                        # :60-62 on self.output = probs
            m = p + 1 - p[r, y][:, None]
            cost = np.maximum(0, m).mean()
            gp = (m >= 0).astype(z.dtype) / (B * n)
            gp[r, y] = 0
            gp[r, y] = -gp.sum(1)          # d/dp_y of every other margin (the c == y margin is the constant 1)
        else:                                                                      # :38-39
            cost = np.exp(-p[r, y]).mean()
            gp = np.zeros_like(z)
            gp[r, y] = -np.exp(-p[r, y]) / B
        dz = p * (gp - (gp * p).sum(1, keepdims=True))
    return lp, p.argmax(1), p[r, y], cost, dz


def exploss_head(z, y):
    """ExpLossLayer (outlayers.py:105-126): preds from the raw output (:111), output -= row mean (:112),
    probs = softmax(output) (:115), features = output (:117), cost = mean exp(-output[n, y_n]) (:38-39)."""
    B, n = z.shape
    r = np.arange(B)
    o = z - z.mean(1, keepdims=True)
    lp = log_softmax(o)
    e = np.exp(-o[r, y])
    go = np.zeros_like(z)
    go[r, y] = -e / B
    dz = go - go.mean(1, keepdims=True)
    return lp, z.argmax(1), np.exp(lp)[r, y], e.mean(), dz, o


def hinge_head(z, y):
    """HingeLayer (outlayers.py:129-147): logprob = probs = features = output; cost = mean over ALL (n, c) of
    max(0, out + 1 - out[n, y_n]) (:60-62); second statistic = probs[n, y_n] = the raw output (:77-78)."""
    B, n = z.shape
    r = np.arange(B)
    m = z + 1 - z[r, y][:, None]
    g = (m >= 0).astype(z.dtype) / (B * n)
    g[r, y] = 0
    g[r, y] = -g.sum(1)
    return z.copy(), z.argmax(1), z[r, y], np.maximum(0, m).mean(), g


def centered_head(v, centers, y, kind, junk_dist=np.inf):
    """CenteredOutLayer on the hidden features v (outlayers.py:186-210).  Loss: -mean logprob[n, y_n] (the
    reference never sets ``loss`` on this layer; see theanet_amd/layer/outlayers.py).  Returns (logprob, preds,
    second_stat_rows, cost, dv, dcenters)."""
    B, nf = v.shape
    r = np.arange(B)
    c = centers[None, :, :]
    if kind == "LOGIT":
        eps = .001                                                                  # :198
        vp = (v * (1 - 2 * eps) + eps)[:, None, :]                                  # :199
        bitprob = c * vp + (1 - c) * (1 - vp)                                       # :200
        lp = np.log(bitprob).sum(2)                                                 # :201
        stat = (bitprob[r, y] < .5).mean(1)                                         # :73-74 (mean over n and f)
        dv = -(1 - 2 * eps) * (2 * centers[y] - 1) / bitprob[r, y] / B
        return lp, lp.argmax(1), stat, -lp[r, y].mean(), dv, np.zeros_like(centers)
    d = v[:, None, :] - c
    dists = (d ** 2).sum(2)                                                         # :205
    full = np.concatenate([dists, np.full((B, 1), junk_dist)], axis=1)              # :206-207
    lp = log_softmax(-full)                                                         # :208-209
    p = np.exp(lp)
    onehot = np.zeros_like(full)
    onehot[r, y] = 1
    gd = ((onehot - p) / B)[:, :-1]            # d cost / d dists
    dv = 2 * (gd[:, :, None] * d).sum(1)
    dc = -2 * (gd[:, :, None] * d).sum(0)
    return lp, p.argmax(1), p[r, y], -lp[r, y].mean(), dv, dc


# --------------------------------------------------------------------------- #
# update  (layer.py:70-117)
# --------------------------------------------------------------------------- #


def maxnorm_project(p, maxnorm):
    """layer.py:88-103 (applied to the already-updated parameter)."""
    t = p.dtype.type
    if not maxnorm:
        return p
    if p.ndim == 1:
        return np.clip(p, -maxnorm, maxnorm).astype(p.dtype)
    if p.ndim == 2:
        n = np.sqrt((p * p).sum(axis=0, dtype=p.dtype))
        scale = (t(1e-7) + np.clip(n, 0, maxnorm)) / (t(1e-7) + n)
        return (p * scale[None, :]).astype(p.dtype)
    if p.ndim == 4:
        n = np.sqrt((p * p).sum(axis=(1, 2, 3), dtype=p.dtype))
        scale = (t(1e-7) + np.clip(n, 0, maxnorm)) / (t(1e-7) + n)
        return (p * scale[:, None, None, None]).astype(p.dtype)
    return p


def sgd_update(p, v, g, lr, reg):
    """One simultaneous Theano update (layer.py:82-105): returns (p', v').

    v' = m*v + (1-m)*g ;  p' = p - rate*lr*v_OLD ; maxnorm(p').
    g must already contain the L1/L2 gradient terms."""
    t = p.dtype.type
    m = t(reg["momentum"])
    v_new = (m * v + (t(1.) - m) * g).astype(p.dtype)
    p_new = (p - t(reg["rate"]) * t(lr) * v).astype(p.dtype)
    return maxnorm_project(p_new, reg["maxnorm"]), v_new


def wtcost(params, reg):
    """layer.py:109-117."""
    t = params[0].dtype.type
    return (t(reg["L1"]) * sum(np.abs(p).sum(dtype=p.dtype) for p in params) +
            t(reg["L2"]) * sum((p * p).sum(dtype=p.dtype) for p in params))


def wtcost_grad(p, reg):
    t = p.dtype.type
    return (t(reg["L1"]) * np.sign(p) + t(2 * reg["L2"]) * p).astype(p.dtype)


# --------------------------------------------------------------------------- #
# elastic input stage  (inlayers.py:29-163)
# --------------------------------------------------------------------------- #


def elastic_filter(sigma, dtype=np.float32):
    """inlayers.py:86-91: (2s+1)^2 gaussian, radius = sigma, NOT renormalised."""
    var = sigma ** 2
    filt = np.array([[np.exp(-.5 * (i * i + j * j) / var)
                      for i in range(-sigma, sigma + 1)]
                     for j in range(-sigma, sigma + 1)], dtype=dtype)
    filt /= 2 * np.pi * var
    return filt


def _tconst(v):
    """Theano's typing of a Python/numpy scalar constant under floatX=float32:
    ints stay ints; a float becomes float32 when that is lossless, else float64."""
    if isinstance(v, (int, np.integer)):
        return int(v)
    return np.float32(v) if float(np.float32(v)) == float(v) else np.float64(v)


class ElasticDraws:
    """The random inputs of one ElasticLayer call (what the HIP stage accepts
    injected for parity runs).  All float32 as Theano delivers them."""
    __slots__ = ("transln", "noise", "origin_u", "zoom_u", "theta_u", "flipmask")

    def __init__(self):
        for s in self.__slots__:
            setattr(self, s, None)


def elastic_field(h, w, prm, draws, coord_dtype=np.float64):
    """Target sampling coordinates (2,h,w) *before* clipping (inlayers.py:77-118).

    prm: dict(translation, zoom, magnitude, sigma, angle).  draws: ElasticDraws.
    Coordinates are float64 on the Theano CPU path (int64 indices + float32 ->
    float64 upcast), while every random input arrives rounded to float32.
    """
    ct = coord_dtype
    target = np.indices((h, w)).astype(ct)                              # :77
    if prm["translation"]:
        target = target + (_tconst(prm["translation"]) * draws.transln).astype(ct)  # :81-82
    if prm["magnitude"]:
        sigma = prm["sigma"]
        filt = elastic_filter(sigma)                                    # float32 filter
        elast = (_tconst(prm["magnitude"]) * draws.noise).astype(np.float32)  # (2,h,w)
        # signal.conv2d 'full' then crop [sigma:h+sigma] == zero-padded 'same' (:95-96);
        # the gaussian is symmetric so convolution == correlation.
        pad = np.pad(elast, ((0, 0), (sigma, sigma), (sigma, sigma)))
        win = sliding_window_view(pad, filt.shape, axis=(1, 2))         # 2,h,w,k,k
        # float32 inputs, float64 accumulation, result rounded to float32: within one
        # float32 ulp of ANY summation order of Theano's float32 conv2d, and reproducible
        sm = np.einsum("chwuv,uv->chw", win.astype(np.float32), filt[::-1, ::-1],
                       dtype=np.float64).astype(np.float32)
        target = target + sm.astype(ct)                                 # :97
    if prm["zoom"] - 1 or prm["angle"]:
        origin = (draws.origin_u.astype(ct) *
                  np.array((h, w)).reshape((2, 1, 1)))                  # :101-102
        target = target - origin
        if prm["zoom"] - 1:
            zoomer = np.exp(_tconst(np.log(prm["zoom"])) * draws.zoom_u)         # :107
            target = target * zoomer.astype(ct)
        if prm["angle"]:
            theta = _tconst(prm["angle"] * np.pi / 180) * np.float32(draws.theta_u)  # :112
            c, s = np.cos(theta), np.sin(theta)
            rotate = np.array([[c, -s], [s, c]])                        # :114
            # tensordot(rotate, target, axes=(0,0)): out[j] = sum_i rotate[i,j]*target[i]
            target = np.tensordot(rotate.astype(ct), target, axes=((0,), (0,)))  # :115
        target = target + origin                                        # :118
    return target


def elastic_apply(x, target, nearest, flipmask=None):
    """Clip + resample + flip noise (inlayers.py:121-142).  x already inverted."""
    h, w = x.shape[2], x.shape[3]
    transy = np.clip(target[0], 0, h - 1 - .001)
    transx = np.clip(target[1], 0, w - 1 - .001)
    if nearest:
        vert = np.rint(transy).astype(np.int64)      # tt.iround (half-to-even, Theano >= 0.9)
        horz = np.rint(transx).astype(np.int64)
        out = x[:, :, vert, horz]
    else:
        topp = transy.astype(np.int32)
        left = transx.astype(np.int32)
        fy = (transy - topp).astype(x.dtype)
        fx = (transx - left).astype(x.dtype)
        one = x.dtype.type(1)
        out = (x[:, :, topp, left] * (one - fy) * (one - fx) +
               x[:, :, topp, left + 1] * (one - fy) * fx +
               x[:, :, topp + 1, left] * fy * (one - fx) +
               x[:, :, topp + 1, left + 1] * fy * fx)
    if flipmask is not None:
        one = x.dtype.type(1)
        out = (one - out) * flipmask + out * (one - flipmask)
    return out.astype(x.dtype)


class ElasticStage:
    """ElasticLayer with its RandomStreams (inlayers.py:29-155)."""

    def __init__(self, img_sz, num_maps=1, translation=0, zoom=1, magnitude=0, sigma=1,
                 pflip=0, angle=0, rand_gen=None, invert_image=False, nearest=False):
        assert zoom > 0
        self.prm = dict(translation=translation, zoom=zoom, magnitude=magnitude,
                        sigma=sigma, angle=angle, pflip=pflip)
        self.img_sz, self.num_maps = img_sz, num_maps
        self.invert, self.nearest = invert_image, nearest
        self.active = bool(magnitude or translation or pflip or angle) or zoom != 1
        self.rv = {}
        if not self.active:
            return
        srs = RandomStreams(rand_gen.randint(1e6) if rand_gen else None)   # :72-73
        h = w = img_sz
        if translation:
            self.rv["transln"] = srs.uniform((2, 1, 1), -1)               # :81
        if magnitude:
            self.rv["noise"] = srs.normal((2, h, w))                      # :94
        if zoom - 1 or angle:
            self.rv["origin_u"] = srs.uniform((2, 1, 1), .25, .75)        # :101
            if zoom - 1:
                self.rv["zoom_u"] = srs.uniform((2, 1, 1), -1)            # :107
            if angle:
                self.rv["theta_u"] = srs.uniform(None, low=-1)            # :112
        if pflip:
            self.rv["flipmask"] = srs.binomial(None, n=1, p=pflip, dtype="float32")  # :141

    def draw(self, x_shape):
        d = ElasticDraws()
        for k, rv in self.rv.items():
            setattr(d, k, rv.draw(x_shape) if k == "flipmask" else rv.draw())
        return d

    def forward(self, x, draws=None, train=True):
        if self.invert:
            x = x.dtype.type(1) - x                                       # :63-64
        if not (train and self.active):
            return x, None
        if draws is None:
            draws = self.draw(x.shape)
        target = elastic_field(x.shape[2], x.shape[3], self.prm, draws)
        out = elastic_apply(x, target, self.nearest,
                            draws.flipmask if self.prm["pflip"] else None)
        return out, target


# --------------------------------------------------------------------------- #
# extras/deformer.py:7-18  (per-image elastic deformation, float64)
# --------------------------------------------------------------------------- #


def elastic_apply_bwd(g, target, nearest, flipmask=None, invert=False):
    """Gradient of ``elastic_apply(1 - x if invert else x, ...)`` w.r.t. x (Theano's grad through the
    advanced-indexing gather = scatter-add, inlayers.py:126-142, :63-64)."""
    N, C, h, w = g.shape
    g = np.asarray(g, np.float64)
    if flipmask is not None:
        g = g * (1 - 2 * np.asarray(flipmask, np.float64))          # d/dout [(1-out) m + out (1-m)]
    dx = np.zeros((N, C, h * w))
    gf = g.reshape(N, C, h * w)
    if target is None:
        dx += gf
    else:
        transy = np.clip(target[0], 0, h - 1 - .001)
        transx = np.clip(target[1], 0, w - 1 - .001)
        if nearest:
            idx = (np.rint(transy).astype(np.int64) * w + np.rint(transx).astype(np.int64)).reshape(-1)
            for k, wt in ((idx, 1.0),):
                np.add.at(dx, (slice(None), slice(None), k), gf * wt)
        else:
            topp, left = transy.astype(np.int64), transx.astype(np.int64)
            fy = (transy - topp).astype(np.float32).astype(np.float64).reshape(-1)
            fx = (transx - left).astype(np.float32).astype(np.float64).reshape(-1)
            i00 = (topp * w + left).reshape(-1)
            for off, wt in ((0, (1 - fy) * (1 - fx)), (1, (1 - fy) * fx), (w, fy * (1 - fx)), (w + 1, fy * fx)):
                np.add.at(dx, (slice(None), slice(None), i00 + off), gf * wt)
    if invert:
        dx = -dx
    return dx.reshape(N, C, h, w)


class ColorStage:
    """ColorLayer with its RandomStreams (color.py:9-52): three uniform(-1, 1) variables of shape
    (batch, num_maps), created in the order balance, gamma, gamma."""

    def __init__(self, img_sz, num_maps=3, rand_gen=None, balance=1, gamma=1, maxval=1):
        self.img_sz, self.num_maps = img_sz, num_maps
        self.balance, self.gamma, self.maxval = balance, gamma, maxval
        self.active = not (gamma == 1 and balance == 1)               # :27-29
        if not self.active:
            return
        assert gamma > 0 and balance > 0
        srs = RandomStreams(rand_gen.randint(1e6) if rand_gen else None)   # :31-32
        self.rv = [srs.uniform(None, -1), srs.uniform(None, -1), srs.uniform(None, -1)]   # :34 (one per pos_rand)

    def draw(self, batch):
        return np.stack([rv.draw((batch, self.num_maps)) for rv in self.rv])

    def factors(self, u):
        lnb, lng = np.log(self.balance), np.log(self.gamma)
        return [np.exp(ln * np.asarray(uk, np.float32).astype(np.float64))[:, :, None, None]
                for ln, uk in zip((lnb, lng, lng), u)]

    def forward(self, x, u=None, train=True):
        if not self.active or not train:
            return x, None
        if u is None:
            u = self.draw(x.shape[0])
        b, g1, g2 = [f.astype(np.float32).astype(x.dtype) for f in self.factors(u)]   # .astype(float_x), :35
        o1 = x / x.dtype.type(self.maxval) * b                                        # :38-39
        o2 = np.clip(o1, 0, 1)                                                        # :40
        o3 = o2 ** g1                                                                 # :41
        out = (1 - (1 - o3) ** g2) * x.dtype.type(self.maxval)                        # :42-44
        return out.astype(x.dtype), (b, g1, g2, o1, o2, o3)

    def backward(self, g, saved):
        b, g1, g2, o1, o2, o3 = saved
        with np.errstate(divide="ignore", invalid="ignore"):
            d = g2 * (1 - o3) ** (g2 - 1) * g1 * o2 ** (g1 - 1) * b
        d = np.where((o1 >= 0) & (o1 <= 1), d, 0.0)                  # theano Clip.grad is inclusive
        return g * d


def gaussian_kernel1d(sigma, truncate=2.0):
    """scipy.ndimage.gaussian_filter1d kernel: radius int(truncate*sigma + .5), normalised."""
    r = int(truncate * float(sigma) + 0.5)
    xk = np.arange(-r, r + 1)
    k = np.exp(-0.5 / (sigma * sigma) * xk ** 2)
    return k / k.sum()


def gaussian_filter_nearest(a, sigma, truncate=2.0):
    """Separable gaussian (axis 0 then axis 1), edge replicated (mode='nearest')."""
    k = gaussian_kernel1d(sigma, truncate)
    r = len(k) // 2
    out = np.asarray(a, dtype=np.float64)
    for axis in (0, 1):
        pad = [(0, 0), (0, 0)]
        pad[axis] = (r, r)
        p = np.pad(out, pad, mode="edge")
        win = sliding_window_view(p, len(k), axis=axis)
        out = win @ k
    return out


def map_coordinates_linear(img, coords, cval=0.0):
    """scipy map_coordinates(order=1, mode='constant'): bilinear, taps outside -> cval,
    and any coordinate outside [0, n-1] is cval outright (scipy >= 1.6 'constant')."""
    H, W = img.shape
    y, x = coords
    y0 = np.floor(y).astype(np.int64)
    x0 = np.floor(x).astype(np.int64)
    fy, fx = y - y0, x - x0

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], cval)

    out = (tap(y0, x0) * (1 - fy) * (1 - fx) + tap(y0, x0 + 1) * (1 - fy) * fx +
           tap(y0 + 1, x0) * fy * (1 - fx) + tap(y0 + 1, x0 + 1) * fy * fx)
    outside = (y < 0) | (y > H - 1) | (x < 0) | (x > W - 1)
    return np.where(outside, cval, out)


def deformer_transform(img, scale, sigma, cval=0, noise=None, rng=np.random):
    """extras/deformer.py:7-18.  ``noise`` = the U(-1,1) field (2,H,W) (drawn from
    ``rng`` -- the reference uses global np.random -- when None)."""
    if noise is None:
        noise = rng.uniform(-1, 1, (2,) + img.shape)
    trans = np.indices(img.shape) + scale * noise
    trans = np.stack([gaussian_filter_nearest(t, sigma, truncate=2) for t in trans])
    return map_coordinates_linear(np.asarray(img, np.float64), trans, cval), trans


# --------------------------------------------------------------------------- #
# the net  (neuralnet.py)
# --------------------------------------------------------------------------- #

DEFAULT_REG = {"L1": 0, "L2": 0, "momentum": .95, "rate": 1, "maxnorm": 0}


class LocationInfoStage:
    """auxiliary.py:14-58: the two candidate locations (B, 2, 2) mixed with one U(0,1) draw per sample (train)
    or averaged (test), times ``boost``, through relu50(2 -> n_hid) and relu01(n_hid -> n_out)."""

    def __init__(self, wts, rand_gen, n_aux, boost, dtype):
        self.boost, self.dtype = boost, dtype
        self.rv = None
        if rand_gen is not None or wts is None:
            srs = RandomStreams(rand_gen.randint(1e6) if rand_gen else None)          # :25-26
            self.rv = srs.uniform(None)                                                # :27
        nh, no = n_aux
        if wts is None:
            self.params = list(init_wb(rand_gen, (2, nh), nh, nh + 2, nh + 2, "relu50", dtype)) + \
                list(init_wb(rand_gen, (nh, no), no, no + nh, no + nh, "relu01", dtype))  # :39-54
        else:
            self.params = [np.array(w, dtype=dtype) for w in wts]

    def draw(self, batch):
        return self.rv.draw((batch,))

    def forward(self, aux, u, train):
        aux = np.asarray(aux, self.dtype)
        if train:
            u = np.asarray(u, np.float32).astype(self.dtype)[:, None]
            loc = aux[:, 0, :] * u + aux[:, 1, :] * (1 - u)                              # :28
        else:
            loc = aux.mean(axis=1)                                                     # :31
        loc = loc * self.dtype.type(self.boost)                                        # :33
        w1, b1, w2, b2 = self.params
        z1 = loc @ w1 + b1
        h = activation("relu50")[0](z1)
        z2 = h @ w2 + b2
        return activation("relu01")[0](z2), (loc, z1, h, z2)

    def backward(self, gout, saved):
        """gout = d cost / d output -> [dW1, db1, dW2, db2]"""
        loc, z1, h, z2 = saved
        w1, b1, w2, b2 = self.params
        dz2 = gout * activation("relu01")[1](z2)
        dz1 = (dz2 @ w2.T) * activation("relu50")[1](z1)
        return [loc.T @ dz1, dz1.sum(0), h.T @ dz2, dz2.sum(0)]


class _L:
    """One layer's static description + parameters."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.params = []
        self.vel = None
        self.reg = None
        self.__dict__.update(kw)


class OracleNet:
    """NeuralNet restatement: builds from the same (layers, training_params, allwts)
    triple, consumes ``RandomState(SEED)`` in the reference's order, and exposes a
    train step / test pass with optionally injected random draws."""

    def __init__(self, layers, training_params, allwts=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.tr_prms = training_params
        self.layers_spec = layers
        self.rand_gen = (np.random.RandomState(training_params['SEED'])
                         if allwts is None else None)              # neuralnet.py:65-68
        self.batch_sz = training_params['BATCH_SZ']
        if 'CUR_EPOCH' not in training_params:
            training_params['CUR_EPOCH'] = 0                       # :108-109
        # DTYPE='float16' (this build's extension; the reference is float32-only): conv products on
        # fp16-rounded operands, dz scaled by GRAD_SCALE before rounding
        self._aux = None
        self.f16 = training_params.get('DTYPE', 'float32') == 'float16'
        self.grad_scale = float(training_params.get('GRAD_SCALE', 4096.))
        self.L = []
        for i, (ltype, largs) in enumerate(layers):
            wts = allwts[i] if allwts else None
            self._append(ltype, dict(largs), wts, first=(i == 0))
        self.set_rate()

    # -- construction (neuralnet.py:87-201) ---------------------------------
    def _prev_maps(self):
        prev = self.L[-1]
        if prev.kind == "DropOut":
            prev = self.L[-2]                                      # :125-128
        return prev.num_maps, prev.out_sz

    def _init(self, wts, size_w, size_b, fan_in, fan_out, actvn):
        if wts is None or len(wts) == 0:
            return list(init_wb(self.rand_gen, size_w, size_b, fan_in, fan_out,
                                actvn, self.dtype))
        return [np.array(wts[0], dtype=self.dtype), np.array(wts[1], dtype=self.dtype)]

    def _append(self, ltype, a, wts, first):
        dt = self.dtype
        if first:
            assert ltype in ("InputLayer", "ElasticLayer", "ColorLayer"), \
                "First layer needs to be Input or Elastic or Color Layer"
        if ltype == "InputLayer":
            self.L.append(_L("Input", num_maps=a.get("num_maps", 1), out_sz=a["img_sz"],
                             n_out=a.get("num_maps", 1) * a["img_sz"] ** 2))
        elif ltype == "ElasticLayer":
            if not first:
                a.pop("num_maps", None), a.pop("img_sz", None)
                a["num_maps"], a["img_sz"] = self._prev_maps()
            st = ElasticStage(rand_gen=self.rand_gen, **a)
            self.L.append(_L("Elastic", stage=st, num_maps=st.num_maps, out_sz=st.img_sz,
                             n_out=st.num_maps * st.img_sz ** 2))
        elif ltype == "ColorLayer":
            if not first:
                a.pop("num_maps", None), a.pop("img_sz", None)
                a["num_maps"], a["img_sz"] = self._prev_maps()
            st = ColorStage(rand_gen=self.rand_gen, **a)
            self.L.append(_L("Color", stage=st, num_maps=st.num_maps, out_sz=st.img_sz,
                             n_out=st.num_maps * st.img_sz ** 2))
        elif ltype == "ConvLayer":
            C, in_sz = self._prev_maps()
            K, f, s = a["num_maps"], a["filter_sz"], a["stride"]
            mode, actvn = a.get("mode", "valid"), a.get("actvn", "relu50")
            _, _, out_sz = conv_geometry(in_sz, f, s, mode)
            l = _L("Conv", C=C, in_sz=in_sz, num_maps=K, f=f, stride=s, mode=mode,
                   actvn=actvn, out_sz=out_sz, n_out=K * out_sz ** 2)
            l.params = self._init(wts, (K, C, f, f), (K,), C * f * f, K * f * f, actvn)
            l.reg = dict(DEFAULT_REG, **dict(a.get("reg", ())))
            self.L.append(l)
        elif ltype == "PoolLayer":
            C, in_sz = self._prev_maps()
            p, ib = a["pool_sz"], a.get("ignore_border", False)
            out_sz = pool_out_sz(in_sz, p, ib)
            self.L.append(_L("Pool", num_maps=C, in_sz=in_sz, p=p, ignore_border=ib,
                             out_sz=out_sz, n_out=C * out_sz ** 2))
        elif ltype == "MeanLayer":
            C, in_sz = self._prev_maps()
            self.L.append(_L("Mean", num_maps=C, in_sz=in_sz, out_sz=1, n_out=C))
        elif ltype == "DropOutLayer":
            n_in = self.L[-1].n_out
            pdrop = a.get("pdrop", 0)
            l = _L("DropOut", n_out=n_in, pdrop=pdrop, mask_rv=None)
            if pdrop:
                srs = RandomStreams(self.rand_gen.randint(1e6) if self.rand_gen else None)
                l.mask_rv = srs.binomial(None, n=1, p=1 - pdrop)
            self.L.append(l)
        elif ltype == "CenteredOutLayer":
            n_in = self.L[-1].n_out
            kind = a.get("kind", "LOGIT")
            actvn = {"LOGIT": "sigmoid", "RBF": "scaled_tanh"}[kind]                # outlayers.py:150
            centers = None
            if wts is not None and len(wts) > 2:
                centers = np.array(wts[2], dtype=dt)
            nf = a.get("n_features") or (wts[0].shape[1] if wts is not None and len(wts) else centers.shape[1])
            l = _L("Centered", n_in=n_in, n_out=nf, actvn=actvn, pdrop=0, mask_rv=None, ckind=kind,
                   learn_centers=a.get("learn_centers", False), junk_dist=a.get("junk_dist", np.inf), loss="nll")
            fio = n_in + nf
            l.params = self._init(wts[:2] if wts is not None and len(wts) else None, (n_in, nf), (nf,), fio, fio, actvn)
            if centers is None:                                                     # :171-179
                ncls = a["n_classes"]
                if kind == "LOGIT":
                    centers = self.rand_gen.binomial(n=1, p=.5, size=(ncls, nf))
                else:
                    centers = self.rand_gen.uniform(low=0, high=1, size=(ncls, nf))
                centers = np.asarray(np.asarray(centers, np.float32), dt)
            l.centers = centers
            if l.learn_centers:
                l.params.append(l.centers)                                          # :186-187
            l.reg = dict(DEFAULT_REG, **dict(a.get("reg", ())))
            self.L.append(l)
        elif ltype == "AuxConcatLayer":
            n_in = self.L[-1].n_out
            assert a.get("aux_type") == "LocationInfo"
            st = LocationInfoStage(wts if wts is not None and len(wts) else None, self.rand_gen,
                                   a["n_aux"], a.get("boost", 1), dt)
            l = _L("AuxConcat", n_in=n_in, n_out=n_in + a["n_aux"][-1], aux=st)          # auxiliary.py:64-101
            l.params = st.params
            self.L.append(l)                                                           # (no reg: never updated)
        elif ltype == "SoftAuxLayer":
            n_in, n_out = self.L[-1].n_out, a["n_out"]
            assert a.get("aux_type") == "LocationInfo"
            have = wts is not None and len(wts)
            l = _L("SoftAux", n_in=n_in, n_out=n_out, actvn="linear", pdrop=0, mask_rv=None, loss=a.get("loss", "nll"))
            fio = n_in + n_out
            l.params = self._init(wts[:2] if have else None, (n_in, n_out), (n_out,), fio, fio, "linear")   # :116-119
            l.aux = LocationInfoStage(wts[2:6] if have else None, self.rand_gen, a["n_aux"], a.get("boost", 1), dt)
            nao = a["n_aux"][-1]
            cross = self._init(wts[6:] if have else None, (nao, n_out), (n_out,), nao + n_out, nao + n_out, "softmax")
            l.params = l.params + l.aux.params + cross                                 # :142-144
            l.reg = dict(DEFAULT_REG, **dict(a.get("reg", ())))
            self.L.append(l)
        elif ltype in ("HiddenLayer", "SoftmaxLayer", "ExpLossLayer", "HingeLayer"):
            n_in = self.L[-1].n_out
            n_out = a["n_out"]
            if ltype == "SoftmaxLayer":
                actvn, pdrop = "Softmax", 0
            elif ltype in ("ExpLossLayer", "HingeLayer"):
                actvn, pdrop = "linear", 0
            else:
                actvn, pdrop = a.get("actvn", "relu01"), a.get("pdrop", 0)
            l = _L({"SoftmaxLayer": "Softmax", "ExpLossLayer": "ExpLoss", "HingeLayer": "Hinge"}.get(ltype, "Hidden"),
                   n_in=n_in, n_out=n_out, actvn=actvn, pdrop=pdrop, mask_rv=None,
                   loss=a.get("loss", "nll"))
            fio = n_in + n_out                                      # hidden.py:21-27
            l.params = self._init(wts, (n_in, n_out), (n_out,), fio, fio, actvn)
            l.reg = dict(DEFAULT_REG, **dict(a.get("reg", ())))
            if pdrop:                                               # dropout.py:10-12
                srs = RandomStreams(self.rand_gen.randint(1e6) if self.rand_gen else None)
                l.mask_rv = srs.binomial(None, n=1, p=1 - pdrop)
            self.L.append(l)
        else:
            raise NotImplementedError("Unknown Layer Type" + ltype)

    # -- learning-rate schedule (neuralnet.py:303-311) ------------------------
    def set_rate(self):
        self.cur_learn_rate = np.float32(
            self.tr_prms['INIT_LEARNING_RATE'] /
            (1 + self.tr_prms['CUR_EPOCH'] / self.tr_prms['EPOCHS_TO_HALF_RATE']))

    def inc_epoch_set_rate(self):
        self.tr_prms['CUR_EPOCH'] += 1
        self.set_rate()

    def get_wts(self):
        return [[p.copy() for p in l.params] for l in self.L]

    # -- forward --------------------------------------------------------------
    def forward(self, x, train, draws=None, keep=False, aux=None):
        """draws: {layer_index: ElasticDraws | mask ndarray | aux mixing draw}.  Returns (logprob, cache)."""
        draws = draws or {}
        aux = self._aux if aux is None else aux
        cache = []
        h = np.asarray(x, dtype=self.dtype)
        for i, l in enumerate(self.L):
            c = {"in": h}
            if l.kind == "Input":
                pass
            elif l.kind == "Elastic":
                h = h.reshape(h.shape[0], l.num_maps, l.out_sz, l.out_sz)
                c["draws"] = draws.get(i)
                if train and l.stage.active and c["draws"] is None:
                    c["draws"] = l.stage.draw(h.shape)
                h, c["target"] = l.stage.forward(h, c["draws"], train)
            elif l.kind == "Color":
                h = h.reshape(h.shape[0], l.num_maps, l.out_sz, l.out_sz)
                h, c["saved"] = l.stage.forward(h, draws.get(i), train)
            elif l.kind == "Conv":
                z = conv2d_fwd(h, l.params[0], l.params[1], l.stride, l.mode, f16=self.f16)
                c["z"] = z
                h = activation(l.actvn)[0](z)
                if self.f16:
                    c["exact"] = h                              # what pooling (ties included) sees
                    h = r16(h).astype(self.dtype)               # what is stored
            elif l.kind == "Pool":
                if self.f16:
                    c["in"] = cache[-1]["exact"]                # max and ties on the unrounded activation
                    h = r16(pool_fwd(c["in"], l.p, l.ignore_border)).astype(self.dtype)
                else:
                    h = pool_fwd(h, l.p, l.ignore_border)
            elif l.kind == "Mean":
                h = mean_fwd(h)
            elif l.kind == "DropOut":
                if l.pdrop:
                    if train:
                        m = draws.get(i)
                        if m is None:
                            m = l.mask_rv.draw(h.shape)
                        c["mask"] = np.asarray(m, dtype=self.dtype).reshape(h.shape)
                        h = h * c["mask"]
                    else:
                        h = h * self.dtype.type(1 - l.pdrop)       # dropout.py:28-31
            elif l.kind == "AuxConcat":
                h = h.reshape(h.shape[0], -1)
                ao, c["aux_saved"] = l.aux.forward(aux, draws.get(i), train)
                h = np.concatenate([h, ao], axis=1)                 # auxiliary.py:83
            elif l.kind == "SoftAux":
                h = h.reshape(h.shape[0], -1)
                c["in"] = h
                l.aux.params = l.params[2:6]
                ao, c["aux_saved"] = l.aux.forward(aux, draws.get(i), train)
                c["aux_out"] = ao
                z = (h @ l.params[0] + l.params[1] + l.params[7] + ao @ l.params[6]).astype(self.dtype)   # :138-139
                c["z"] = z
                h = log_softmax(z)
            elif l.kind in ("Hidden", "Softmax", "ExpLoss", "Hinge", "Centered"):
                h = h.reshape(h.shape[0], -1)                       # flatten(2), neuralnet.py:169
                c["in"] = h
                c["c8"] = self.f16 and i > 0 and self.L[i - 1].kind in ("Conv", "Pool")     # fp16 operands: the stack's halfs
                if c["c8"]:
                    z = (r16(h) @ r16(l.params[0]) + l.params[1]).astype(self.dtype)
                else:
                    z = (h @ l.params[0] + l.params[1]).astype(self.dtype)
                c["z"] = z
                if l.kind == "Softmax":
                    h = log_softmax(z)
                elif l.kind in ("ExpLoss", "Hinge"):
                    h = z                                           # the head itself is applied in head_eval
                elif l.kind == "Centered":
                    h = activation(l.actvn)[0](z)
                else:
                    h = activation(l.actvn)[0](z)
                    if l.pdrop:
                        if train:
                            m = draws.get(i)
                            if m is None:
                                m = l.mask_rv.draw(h.shape)
                            c["mask"] = np.asarray(m, dtype=self.dtype).reshape(h.shape)
                            h = h * c["mask"]                       # hidden.py:31-32
                        else:
                            h = h * self.dtype.type(1 - l.pdrop)    # hidden.py:50-55
            c["out"] = h
            cache.append(c)
        return h, cache

    # -- backward -------------------------------------------------------------
    def backward(self, cache, y):
        """Gradients of cost = nll + sum wtcost w.r.t. every parameter."""
        grads = [None] * len(self.L)
        g = None
        first_param = min(i for i, l in enumerate(self.L) if l.params)
        for i in range(len(self.L) - 1, -1, -1):
            l, c = self.L[i], cache[i]
            dcent = None
            if l.kind == "Softmax":
                if l.loss in (None, "nll"):
                    dz = nll_dlogits(c["out"], y)       # d cost / d logits
                else:
                    dz = softmax_head(c["z"], y, l.loss)[4].astype(self.dtype)
            elif l.kind in ("ExpLoss", "Hinge", "Centered"):
                hd = self.head(c["out"], y)
                dz = hd["dA"]
                if l.kind == "Centered":
                    dz = dz * activation(l.actvn)[1](c["z"])
                    dcent = hd["dcenters"]
                dz = dz.astype(self.dtype)
            elif l.kind == "Hidden":
                if "mask" in c:
                    g = g * c["mask"]
                dz = (g * activation(l.actvn)[1](c["z"])).astype(self.dtype)
            if l.kind == "Hidden" and c.get("c8"):
                # dense layer on the fp16-resident stack: both operands of its three products are halfs
                dz16 = r16(dz, self.grad_scale)
                grads[i] = [(r16(c["in"]).T @ dz16).astype(self.dtype) + wtcost_grad(l.params[0], l.reg),
                            dz16.sum(axis=0).astype(self.dtype) + wtcost_grad(l.params[1], l.reg)]
                g = self._f16_down(dz16 @ r16(l.params[0]).T, i, cache) if i > first_param else None
            elif l.kind in ("Softmax", "Hidden", "ExpLoss", "Hinge", "Centered"):
                xin = c["in"]
                dW = (xin.T @ dz).astype(self.dtype)
                db = dz.sum(axis=0, dtype=self.dtype)
                grads[i] = [dW + wtcost_grad(l.params[0], l.reg),
                            db + wtcost_grad(l.params[1], l.reg)]
                if l.kind == "Centered" and l.learn_centers:
                    grads[i].append(dcent.astype(self.dtype) + wtcost_grad(l.params[2], l.reg))
                g = (dz @ l.params[0].T).astype(self.dtype) if i > first_param else None
            elif l.kind == "SoftAux":
                dz = (nll_dlogits(c["out"], y) if l.loss in (None, "nll")
                      else softmax_head(c["z"], y, l.loss)[4]).astype(self.dtype)
                ga = l.aux.backward(dz @ l.params[6].T, c["aux_saved"])
                gr = [c["in"].T @ dz, dz.sum(0)] + ga + [c["aux_out"].T @ dz, dz.sum(0)]
                grads[i] = [np.asarray(gg, self.dtype) + wtcost_grad(p, l.reg) for gg, p in zip(gr, l.params)]
                g = (dz @ l.params[0].T).astype(self.dtype) if i > first_param else None
            elif l.kind == "AuxConcat":
                grads[i] = [np.zeros_like(p) for p in l.params]     # no reg: never updated (layer.py:74-75)
                g = g[:, :l.n_in]
            elif l.kind == "DropOut":
                if "mask" in c:
                    g = g.reshape(c["mask"].shape) * c["mask"]
            elif l.kind == "Mean":
                g = mean_bwd(c["in"], g.reshape(c["out"].shape))
            elif l.kind == "Pool":
                g = pool_bwd(c["in"], g.reshape(c["out"].shape), l.p, l.ignore_border)
            elif l.kind == "Conv" and self.f16:
                # what arrives IS dz as stored (halfs, act' applied by its producer from this block's stored output)
                dz = np.asarray(g, np.float64).reshape(c["z"].shape)
                g, dW, db = conv2d_bwd(c["in"], l.params[0], dz, l.stride, l.mode,
                                       need_dx=i > first_param, f16=True, grad_scale=self.grad_scale)
                grads[i] = [dW + wtcost_grad(l.params[0], l.reg),
                            db + wtcost_grad(l.params[1], l.reg)]
                if g is not None:
                    g = self._f16_down(g, i, cache)
            elif l.kind == "Conv":
                dz = (g.reshape(c["z"].shape) * activation(l.actvn)[1](c["z"])).astype(self.dtype)
                g, dW, db = conv2d_bwd(c["in"], l.params[0], dz, l.stride, l.mode,
                                       need_dx=i > first_param, f16=self.f16, grad_scale=self.grad_scale)
                grads[i] = [dW + wtcost_grad(l.params[0], l.reg),
                            db + wtcost_grad(l.params[1], l.reg)]
            elif l.kind == "Input" or i <= first_param:
                g = None
            elif l.kind == "Elastic":
                st, d = l.stage, c.get("draws")
                g = elastic_apply_bwd(g.reshape(c["out"].shape), c.get("target") if st.active else None, st.nearest,
                                      getattr(d, "flipmask", None) if d is not None else None,
                                      st.invert).astype(self.dtype)
            elif l.kind == "Color":
                g = g.reshape(c["out"].shape)
                if c.get("saved") is not None:
                    g = l.stage.backward(g, c["saved"]).astype(self.dtype)
            if g is None:
                break
        return grads

    def _f16_down(self, g, i, cache):
        """Stored-fp16 mode: the gradient layer i hands to the layer below, as the device stores it: times act' of the
        block below taken from ITS stored output (a Pool layer stands for its conv block), rounded to half at the
        gradient scale."""
        below = self.L[i - 1]
        out = np.asarray(cache[i - 1]["out"], np.float64)
        g = np.asarray(g, np.float64).reshape(out.shape)
        if below.kind == "Pool":
            g = g * act_grad_from_out(self.L[i - 2].actvn, out)
        elif below.kind == "Conv":
            g = g * act_grad_from_out(below.actvn, out)
        return r16(g, self.grad_scale)

    # -- public steps ---------------------------------------------------------
    def head(self, h, y):
        """Everything the output head derives from the last layer's output h (logprob for Softmax, the
        linear output for ExpLoss / Hinge, the hidden features for CenteredOut)."""
        l = self.L[-1]
        if l.kind in ("Softmax", "SoftAux"):
            lp, preds, stat, cost, dA = softmax_head(h, y, l.loss or "nll")   # log_softmax(logprob) == logprob
            return dict(logprob=lp, preds=preds, stat=stat, cost=cost, dA=dA, feats=lp)
        if l.kind == "ExpLoss":
            lp, preds, stat, cost, dA, o = exploss_head(h, y)
            return dict(logprob=lp, preds=preds, stat=stat, cost=cost, dA=dA, feats=o)
        if l.kind == "Hinge":
            lp, preds, stat, cost, dA = hinge_head(h, y)
            return dict(logprob=lp, preds=preds, stat=stat, cost=cost, dA=dA, feats=lp)
        centers = l.params[2] if l.learn_centers else l.centers
        lp, preds, stat, cost, dA, dc = centered_head(h, centers, y, l.ckind, l.junk_dist)
        return dict(logprob=lp, preds=preds, stat=stat, cost=cost, dA=dA, dcenters=dc, feats=h)

    def cost(self, h, y):
        l = self.L[-1]
        c = nll(h, y) if (l.kind in ("Softmax", "SoftAux") and l.loss in (None, "nll")) else self.head(h, y)["cost"]
        for l in self.L:
            if l.params and l.reg:
                c = c + wtcost(l.params, l.reg)
        return self.dtype.type(c)

    def set_aux(self, aux):
        """The auxiliary input of the minibatch about to be processed (neuralnet.py:216-226)."""
        self._aux = aux

    def grads(self, x, y, draws=None):
        logprob, cache = self.forward(x, True, draws)        # (the head's input for the non-Softmax heads)
        return self.cost(logprob, y), logprob, self.backward(cache, y), cache

    def train_step(self, x, y, draws=None):
        """One call of the function built by get_trin_model (neuralnet.py:203-241):
        returns [cost, features, logprob] and applies the simultaneous updates."""
        cost, logprob, grads, _ = self.grads(x, y, draws)
        feats = logprob
        if self.L[-1].kind not in ("Softmax", "SoftAux"):
            hd = self.head(logprob, y)
            feats, logprob = hd["feats"], hd["logprob"]
        for l, g in zip(self.L, grads):
            if not l.params or not l.reg or not l.reg["rate"]:       # layer.py:74-75 (no reg: AuxConcatLayer)
                continue
            if l.vel is None:
                l.vel = [np.zeros_like(p) for p in l.params]
            for j in range(len(l.params)):
                l.params[j], l.vel[j] = sgd_update(l.params[j], l.vel[j], g[j],
                                                   self.cur_learn_rate, l.reg)
        return cost, feats, logprob

    def test(self, x, y):
        """get_test_model outputs (neuralnet.py:257-277; outlayers.py:69-80)."""
        logprob, _ = self.forward(x, False)
        if self.L[-1].kind not in ("Softmax", "SoftAux"):
            hd = self.head(logprob, y)
            return np.mean(hd["preds"] != y), np.mean(hd["stat"]), hd["logprob"], hd["preds"]
        preds = logprob.argmax(axis=1)
        sym_err = np.mean(preds != y)
        p_mle = np.exp(logprob)[np.arange(len(y)), y].mean()
        return sym_err, p_mle, logprob, preds
