"""DTYPE='float16' parity (BASELINE.json configs[4]: "fp16 inputs / fp32 accum MFMA").

Whole nets (NeuralNet with DTYPE float16 = fp16-RESIDENT tensors, tests/test_gpu_c8.py has the ops) against the float64
oracle in its stored-fp16 mode (oracle.theanet_oracle.OracleNet, DTYPE 'float16').

Tolerances.  The device and the oracle multiply the SAME fp16-rounded operands (products of two
halfs are exact in fp32), so a single product differs only by the fp32 accumulation: 2e-5 relative to
the largest entry.  Through a whole net the activations that feed the next layer's rounding differ by
fp32 round-off, and a value on an fp16 rounding boundary may round the other way (one fp16 ulp =
4.9e-4 relative on that operand): logprob 2e-3 rel + 2e-4 abs, weights after two steps 2e-3 rel,
argmax exact.  The reference itself is float32-only (weights.py:8), so this mode has no reference
counterpart: the oracle mode is the specification."""
import copy

import numpy as np
import pytest

from oracle import theanet_oracle as O
from tests.gpu_util import assert_close, ctx, load_prms

pytestmark = pytest.mark.gpu

GS = 4096.0


def _inject_draws(net, ora, B, C, img):
    draws = {}
    for i, l in enumerate(ora.L):
        if l.kind == "Elastic" and l.stage.active:
            d = l.stage.draw((B, C, img, img))
            draws[i] = d
            net.tr_layers[i].inject(**{k: getattr(d, k) for k in d.__slots__})
        if getattr(l, "mask_rv", None) is not None:
            m = l.mask_rv.draw((B, l.n_out))
            draws[i] = m
            net.tr_layers[i].drop.inject(m)
    return draws


@pytest.mark.parametrize("name,img,B", [("cifar_like.prms", 32, 16), ("wide6.prms", 64, 4), ("wide6.prms", 32, 6)])
def test_f16_nets_match_f16_oracle(name, img, B):
    """Two training steps (forward, every gradient, momentum update, maxnorm) in DTYPE float16 against
    the float64 oracle in its stored-fp16 mode."""
    from theanet_amd import NeuralNet
    prms = load_prms(name, img, batch=B)
    tr = dict(prms["training_params"], DTYPE="float16", GRAD_SCALE=GS)
    rng = np.random.RandomState(1)
    x = rng.rand(2 * B, 3, img, img).astype(np.float32)
    y = rng.randint(0, 10, 2 * B).astype(np.int32)
    net = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr))
    assert all(l.f16 for l in net.tr_layers if hasattr(l, "f16"))
    ora = O.OracleNet(copy.deepcopy(prms["layers"]), dict(tr), dtype=np.float64)
    ora32 = O.OracleNet(copy.deepcopy(prms["layers"]), dict(tr, DTYPE="float32"), dtype=np.float64)
    fn = net.get_trin_model(x, y)
    for s in range(2):
        draws = _inject_draws(net, ora, B, 3, img)
        cost_w, lp_w, _ = ora.train_step(x[s * B:(s + 1) * B], y[s * B:(s + 1) * B], draws)
        cost, _, lp = fn(s)
        assert_close(lp, lp_w, 2e-3, 2e-4, what="%s f16 logprob step %d" % (name, s))
        assert_close(cost, cost_w, 2e-3, 2e-4, what="%s f16 cost step %d" % (name, s))
        np.testing.assert_array_equal(lp.argmax(1), lp_w.argmax(1))
        if s == 0:      # the mode is not a no-op: the fp32 oracle is measurably further away
            lp32 = ora32.forward(x[:B], True, draws)[0]
            assert np.abs(lp - lp_w).max() < .5 * np.abs(lp32 - lp_w).max() + 1e-6
    for i, (lyr, ol) in enumerate(zip(net.tr_layers, ora.L)):
        for j, w in enumerate(lyr.get_wts()):
            assert_close(w, ol.params[j], 2e-3, 2e-6, what="%s f16 w %d %d" % (name, i, j))


def _full_size(name, img, B, rows, dtype, steps):
    """BASELINE batch size: properties that need no oracle + an oracle cross-check of the first `rows`
    rows of a test-mode forward pass with the trained weights."""
    from theanet_amd import NeuralNet
    prms = load_prms(name, img, batch=B)
    tr = dict(prms["training_params"], DTYPE=dtype, GRAD_SCALE=GS)
    rng = np.random.default_rng(0)
    x = rng.random((2 * B, 3, img, img), dtype=np.float32)
    y = np.random.default_rng(1).integers(0, 10, 2 * B).astype(np.int32)
    net = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr))
    fn = net.get_trin_model(x, y)
    conv = [l for l in net.tr_layers if getattr(l, "params", None)][1]
    w0 = conv.get_wts()[0]
    cost0, _, lp = fn(0)
    np.testing.assert_array_equal(conv.get_wts()[0], w0)            # step 0 applies the zero velocity (layer.py:86)
    assert np.isfinite(cost0) and abs(cost0 - np.log(10)) < 1.5
    np.testing.assert_allclose(np.exp(lp).sum(1), 1, rtol=1e-4)
    costs = [fn(i % 2)[0] for i in range(1, steps)]
    assert np.isfinite(costs).all() and min(costs[-2:]) < cost0, (cost0, costs)
    assert not np.array_equal(conv.get_wts()[0], w0)
    ora = O.OracleNet(copy.deepcopy(prms["layers"]), dict(tr), allwts=net.get_init_params()["allwts"])
    tfn = net.get_test_model(x, y, preds_feats=True)
    sym, pm, feats, preds = tfn(1)
    _, _, lp_w, preds_w = ora.test(x[B:B + rows], y[B:B + rows])
    tol = (2e-3, 2e-4) if dtype == "float16" else (1e-4, 2e-5)
    assert_close(feats[:rows], lp_w, *tol, what="%s %s test logprob rows 0..%d" % (name, dtype, rows - 1))
    np.testing.assert_array_equal(preds[:rows], preds_w)
    assert 0 <= sym <= 1 and 0 < pm <= 1


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_full_size_wide6_64x64_b128(dtype):
    """BASELINE.json configs[4] at its per-GPU size (1024 / 8 GPUs): 64x64x3, 128 images."""
    _full_size("wide6.prms", 64, 128, 16, dtype, 8)


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_full_size_cifar_like_b2048(dtype):
    """BASELINE.json configs[3] at its stated size: 32x32x3, 2048 images, elastic stage on."""
    _full_size("cifar_like.prms", 32, 2048, 256, dtype, 12)


def _directional_derivative(name, img, B, dtype, tol):
    """Every gradient of a full-size step through a size-independent property: along the direction D = the step's own
    gradient (all parameters), the training cost must change by |g|^2 per unit step -- (cost(W + eps D) - cost(W - eps D))
    / 2 eps against sum(g^2), with the distortion field and the dropout masks of the step held fixed (same stream
    seeds, same step counter).  The gradient is read back as the velocity after one step from rest, v = (1 - m) g
    (layer.py:82-84)."""
    from theanet_amd import NeuralNet
    prms = load_prms(name, img, batch=B)
    tr = dict(prms["training_params"], DTYPE=dtype, GRAD_SCALE=GS)
    x = np.random.default_rng(3).random((2 * B, 3, img, img), dtype=np.float32)
    y = np.random.default_rng(4).integers(0, 10, 2 * B).astype(np.int32)
    base = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr))
    p0 = base.get_init_params(with_opt_state=True)
    W0, st0 = p0["allwts"], p0["opt_state"]
    c0 = float(base.get_trin_model(x, y)(0)[0])
    vel = base.get_init_params(with_opt_state=True)["opt_state"]["velocities"]
    g = [[np.asarray(v, np.float64) / (1.0 - lyr.reg['momentum']) for v in row] if lyr.has_updates() else
         [np.zeros_like(w, np.float64) for w in ws] for lyr, row, ws in zip(base.tr_layers, vel, W0)]
    gg = sum(float((a * a).sum()) for row in g for a in row)
    assert np.isfinite(gg) and gg > 0
    eps = 0.02 / gg                                    # the cost moves by about +-0.02

    def cost_at(sign):
        W = [[(w.astype(np.float64) + sign * eps * a).astype(np.float32) for w, a in zip(ws, row)] for ws, row in zip(W0, g)]
        net = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr), allwts=W)
        net.load_opt_state(st0)
        return float(net.get_trin_model(x, y)(0)[0])

    assert abs(cost_at(0) - c0) <= 1e-6 * abs(c0), "the rebuilt net does not repeat the step"
    fd = (cost_at(+1) - cost_at(-1)) / (2 * eps)
    assert abs(fd - gg) <= tol * gg, "%s %s: directional derivative %.6g, |g|^2 %.6g" % (name, dtype, fd, gg)


@pytest.mark.parametrize("dtype,tol", [("float32", 5e-3), ("float16", 1e-2)])
def test_full_size_gradients_wide6_64x64_b128(dtype, tol):
    _directional_derivative("wide6.prms", 64, 128, dtype, tol)


@pytest.mark.parametrize("dtype,tol", [("float32", 5e-3), ("float16", 1e-2)])
def test_full_size_gradients_cifar_like_b2048(dtype, tol):
    _directional_derivative("cifar_like.prms", 32, 2048, dtype, tol)


def test_full_size_wide6_64x64_b1024_on_one_gpu():
    """BASELINE.json configs[4] as stated for the node (bs 1024) on ONE GPU, fp16-resident."""
    _full_size("wide6.prms", 64, 1024, 4, "float16", 8)
