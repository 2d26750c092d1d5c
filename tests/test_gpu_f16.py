"""DTYPE='float16' parity (BASELINE.json configs[4]: "fp16 inputs / fp32 accum MFMA").

Whole nets (NeuralNet with DTYPE float16 = fp16-RESIDENT tensors, tests/test_gpu_c8.py has the ops) against the float64
oracle in its stored-fp16 mode (oracle.theanet_oracle.OracleNet, DTYPE 'float16').  The first part of this file keeps
the C-ABI's operand-rounded products (tn_set_matmul_dtype(1) + tn_conv2d_* / tn_convpool_*: fp32 tensors in HBM, both
operands rounded while staged; theanet_amd/csrc/conv_tile16.hip) against oracle.theanet_oracle.r16 /
conv2d_fwd(f16=True) / conv2d_bwd(f16=True).

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
from tests.gpu_util import act_code, assert_close, call, ctx, dev, empty, load_prms

pytestmark = pytest.mark.gpu

GS = 4096.0


@pytest.fixture
def f16_mode():
    ctx().set_matmul_dtype("float16", GS)
    yield
    ctx().set_matmul_dtype("float32")


def _scaled_tol(want, rel=2e-5):
    return rel * float(np.abs(want).max())


F16_CONV_CASES = [
    # N, C, H, K, mode, act          (3x3, stride 1)
    (2, 64, 64, 64, "same", "relu10"),      # wide6 conv2
    (3, 64, 32, 128, "same", "relu10"),     # wide6 conv3
    (2, 128, 32, 128, "same", "tanh"),      # wide6 conv4
    (3, 128, 16, 256, "same", "relu10"),    # wide6 conv5
    (2, 256, 16, 256, "same", "relu10"),    # wide6 conv6
    (5, 32, 16, 64, "same", "relu10"),      # cifar_like conv2
    (9, 64, 8, 128, "same", "relu05"),      # cifar_like conv3 (several images per tile, ragged group)
    (3, 3, 64, 64, "same", "relu10"),       # first layers: C = 3 (one chunk, 13 zero channels)
    (4, 3, 32, 32, "same", "relu10"),
    (2, 20, 32, 48, "same", "sigmoid"),     # ragged channel / filter counts
    (2, 40, 16, 24, "same", "relu10"),
]


def _setup(case, seed=0):
    N, C, H, K, mode, act = case
    rng = np.random.RandomState(seed)
    x = rng.randn(N, C, H, H).astype(np.float32)
    W = (rng.randn(K, C, 3, 3) / np.sqrt(C * 9)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    pad_lo, _, out = O.conv_geometry(H, 3, 1, mode)
    return x, W, b, pad_lo, out


@pytest.mark.parametrize("case", F16_CONV_CASES)
def test_conv_f16_fwd(case, f16_mode):
    N, C, H, K, mode, act = case
    x, W, b, pad_lo, out = _setup(case)
    z = O.conv2d_fwd(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64), 1, mode, f16=True)
    want = O.activation(act)[0](z)
    a = empty((N, K, out, out))
    kind, prm = act_code(act)
    call("tn_conv2d_fwd", dev(x).ptr, dev(W).ptr, dev(b).ptr, a.ptr, N, C, H, H, K, 3, 1, pad_lo, out, out,
         kind, prm)
    assert_close(a.get_value(), want, rtol=1e-5, atol=_scaled_tol(z), what="f16 conv fwd %s" % (case,))
    # and it is NOT the fp32 product (the rounding is really applied)
    z32 = O.conv2d_fwd(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64), 1, mode)
    assert np.abs(z32 - z).max() > 5 * _scaled_tol(z)


@pytest.mark.parametrize("case", F16_CONV_CASES)
def test_conv_f16_wgrad_dgrad(case, f16_mode):
    N, C, H, K, mode, act = case
    x, W, b, pad_lo, out = _setup(case, 1)
    rng = np.random.RandomState(2)
    # gradients of realistic size: without the grad scale most of these would be fp16 subnormals
    dz = (rng.randn(N, K, out, out) * 3e-6).astype(np.float32)
    dx_w, dW_w, db_w = O.conv2d_bwd(x.astype(np.float64), W.astype(np.float64), dz.astype(np.float64), 1, mode,
                                    f16=True, grad_scale=GS)
    dW, db, dx = empty(W.shape), empty((K,)), empty(x.shape)
    xd, dzd, Wd = dev(x), dev(dz), dev(W)
    call("tn_conv2d_wgrad", xd.ptr, dzd.ptr, dW.ptr, db.ptr, N, C, H, H, K, 3, 1, pad_lo, out, out)
    assert_close(dW.get_value(), dW_w, rtol=1e-5, atol=_scaled_tol(dW_w), what="f16 conv dW %s" % (case,))
    assert_close(db.get_value(), db_w, rtol=1e-5, atol=_scaled_tol(db_w), what="f16 conv db %s" % (case,))
    if C * 9 > 32:
        call("tn_conv2d_dgrad", dzd.ptr, Wd.ptr, dx.ptr, N, C, H, H, K, 3, 1, pad_lo, out, out, None, 0, 0.0)
        assert_close(dx.get_value(), dx_w, rtol=1e-5, atol=_scaled_tol(dx_w), what="f16 conv dx %s" % (case,))
        prev_a = rng.randn(*x.shape).astype(np.float32)
        prev_a[0, 0, 0, :3] = 0
        kind, prm = act_code("relu10")
        call("tn_conv2d_dgrad", dzd.ptr, Wd.ptr, dx.ptr, N, C, H, H, K, 3, 1, pad_lo, out, out,
             dev(prev_a).ptr, kind, prm)
        g = np.where(prev_a > 0, 1.0, np.where(prev_a < 0, .1, 1.1))
        assert_close(dx.get_value(), dx_w * g, rtol=1e-5, atol=_scaled_tol(dx_w), what="f16 conv dx*act' %s" % (case,))


def test_conv_f16_valid_mode_fwd_dgrad(f16_mode):
    """'valid' layers: forward (pad 0) and input gradient (pad 2) run on the tile kernel too (the gathered
    tensor's rows must be a multiple of 4 pixels: x for the forward, dz for the input gradient)."""
    N, C, K = 2, 16, 32
    rng = np.random.RandomState(5)
    W = (rng.randn(K, C, 3, 3) / 12).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    H = 20
    x = rng.randn(N, C, H, H).astype(np.float32)
    z = O.conv2d_fwd(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64), 1, "valid", f16=True)
    a = empty(z.shape)
    call("tn_conv2d_fwd", dev(x).ptr, dev(W).ptr, dev(b).ptr, a.ptr, N, C, H, H, K, 3, 1, 0, H - 2, H - 2, 0, 0.0)
    assert_close(a.get_value(), z, rtol=1e-5, atol=_scaled_tol(z), what="f16 valid fwd")
    H = 18
    x = rng.randn(N, C, H, H).astype(np.float32)
    dz = (rng.randn(N, K, H - 2, H - 2) * 1e-5).astype(np.float32)
    dx_w, _, _ = O.conv2d_bwd(x.astype(np.float64), W.astype(np.float64), dz.astype(np.float64), 1, "valid",
                              f16=True, grad_scale=GS)
    dx = empty(x.shape)
    call("tn_conv2d_dgrad", dev(dz).ptr, dev(W).ptr, dx.ptr, N, C, H, H, K, 3, 1, 0, H - 2, H - 2, None, 0, 0.0)
    assert_close(dx.get_value(), dx_w, rtol=1e-5, atol=_scaled_tol(dx_w), what="f16 valid dx")


def test_f16_unsupported_shape_is_an_error_not_a_fallback(f16_mode):
    from theanet_amd import _lib
    x, W, b = dev(np.zeros((1, 8, 12, 12), np.float32)), dev(np.zeros((16, 8, 5, 5), np.float32)), dev(np.zeros(16, np.float32))
    a = empty((1, 16, 8, 8))
    with pytest.raises(_lib.BackendError, match="no fp16-operand kernel"):
        call("tn_conv2d_fwd", x.ptr, W.ptr, b.ptr, a.ptr, 1, 8, 12, 12, 16, 5, 1, 0, 8, 8, 0, 0.0)
    assert ctx().lib.tn_conv_f16_supported(1, 8, 12, 12, 16, 5, 1, 0, 8, 8) == 0
    assert ctx().lib.tn_conv_f16_supported(4, 64, 32, 32, 64, 3, 1, 1, 32, 32) == 7
    with pytest.raises(_lib.BackendError, match="power of two"):
        call("tn_set_matmul_dtype", 1, 1000.0)


F16_BLOCK_CASES = [
    # N, C, H, K, act      conv 3x3 'same' + act + 2x2 max-pool
    (2, 64, 64, 64, "relu10"),
    (3, 128, 32, 128, "relu10"),
    (2, 256, 16, 256, "tanh"),
    (5, 3, 32, 32, "relu10"),       # cifar_like first block (weight gradient on the small-C kernel)
    (4, 32, 16, 64, "relu10"),
    (6, 64, 8, 128, "relu10"),
]


@pytest.mark.parametrize("case", F16_BLOCK_CASES)
def test_convpool_block_f16(case, f16_mode):
    N, C, H, K, act = case
    rng = np.random.RandomState(7)
    x = rng.randn(N, C, H, H).astype(np.float32)
    W = (rng.randn(K, C, 3, 3) / np.sqrt(C * 9)).astype(np.float32)
    b = (rng.randn(K) * .1).astype(np.float32)
    assert ctx().lib.tn_convpool_f16_supported(N, C, H, H, K, 3, 1, 1, H, H, 2, H // 2, H // 2)
    x64, W64, b64 = x.astype(np.float64), W.astype(np.float64), b.astype(np.float64)
    z = O.conv2d_fwd(x64, W64, b64, 1, "same", f16=True)
    fwd, bwd = O.activation(act)
    a_w = fwd(z)
    y_w = O.pool_fwd(a_w, 2)
    y, mask = empty(y_w.shape), empty(y_w.shape, np.uint8)
    kind, prm = act_code(act)
    geom = (N, C, H, H, K, 3, 1, H, H, 2, H // 2, H // 2, kind, prm)
    xd, Wd, bd = dev(x), dev(W), dev(b)
    call("tn_convpool_fwd_mask", xd.ptr, Wd.ptr, bd.ptr, y.ptr, mask.ptr, *geom)
    assert_close(y.get_value(), y_w, rtol=1e-5, atol=_scaled_tol(z), what="f16 block fwd %s" % (case,))
    # backward from the device's own pooled output / mask (ties, if any, are the device's)
    g = (rng.randn(*y_w.shape) * 1e-5).astype(np.float32)
    yv = y.get_value().astype(np.float64)
    m = mask.get_value()
    dz = np.zeros_like(z)
    for di in range(2):
        for dj in range(2):
            bit = (m >> (2 * di + dj)) & 1
            if act == "tanh":
                ga = g.astype(np.float64) * (1 - yv * yv)
            else:
                slope = float(act[4:]) / 100
                ga = g.astype(np.float64) * np.where(m & 16, 1.0, np.where(m & 32, slope, 1 + slope))
            dz[:, :, di::2, dj::2] = np.where(bit, ga, 0.0)
    need_dx = C * 9 > 32
    dx_w, dW_w, db_w = O.conv2d_bwd(x64, W64, dz, 1, "same", need_dx=need_dx, f16=True, grad_scale=GS)
    dW, db = empty(W.shape), empty((K,))
    dx = empty(x.shape) if need_dx else None
    call("tn_convpool_bwd_mask_dx", xd.ptr, Wd.ptr, dev(g).ptr, y.ptr, mask.ptr, dx.ptr if need_dx else None,
         dW.ptr, db.ptr, *geom, None, 0, 0.0)
    # dz = g * act'(y) is formed in fp32 on the device and in float64 here: a value on an fp16 rounding
    # boundary may round the other way (one half ulp = 4.9e-4 of that operand), hence 1e-4 of the largest entry
    assert_close(dW.get_value(), dW_w, rtol=1e-5, atol=_scaled_tol(dW_w, 1e-4), what="f16 block dW %s" % (case,))
    assert_close(db.get_value(), db_w, rtol=1e-5, atol=_scaled_tol(db_w), what="f16 block db %s" % (case,))
    if need_dx:
        assert_close(dx.get_value(), dx_w, rtol=1e-5, atol=_scaled_tol(dx_w, 1e-4), what="f16 block dx %s" % (case,))


# --------------------------------------------------------------------------------------------------
# whole nets
# --------------------------------------------------------------------------------------------------

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
    _full_size("wide6.prms", 64, 128, 4, dtype, 8)


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_full_size_cifar_like_b2048(dtype):
    """BASELINE.json configs[3] at its stated size: 32x32x3, 2048 images, elastic stage on."""
    _full_size("cifar_like.prms", 32, 2048, 32, dtype, 12)


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
