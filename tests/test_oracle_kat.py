"""Known-answer tests that pin the oracle's Theano semantics
and float64 finite-difference gradient checks.  CPU only."""
import ast
import hashlib
import os

import numpy as np
import pytest

from oracle import theanet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kat1_conv_is_true_convolution():
    # delta input: a true convolution reproduces the kernel un-flipped ('full');
    # 'valid' on a 3x3 delta-at-centre picks W[1,1]; an off-centre delta picks the
    # mirrored tap.
    W = np.arange(1, 10, dtype=np.float64).reshape(1, 1, 3, 3)
    b = np.zeros(1)
    x = np.zeros((1, 1, 3, 3)); x[0, 0, 0, 0] = 1          # delta at top-left
    z = O.conv2d_fwd(x, W, b, 1, "valid")
    assert z.shape == (1, 1, 1, 1)
    # z = sum x[u,v] * W[2-u,2-v] = W[2,2]
    assert z[0, 0, 0, 0] == 9
    x = np.zeros((1, 1, 5, 5)); x[0, 0, 2, 2] = 1
    z = O.conv2d_fwd(x, W, b, 1, "same")
    # true convolution of a centred delta returns the kernel itself
    np.testing.assert_array_equal(z[0, 0, 1:4, 1:4], W[0, 0])


def test_kat1b_same_mode_even_filter_crop():
    # same = full[shift:in+shift], shift=(f-1)//2 (convpool.py:57-61), f=4 -> shift=1
    rng = np.random.RandomState(0)
    x = rng.randn(1, 1, 6, 6); W = rng.randn(1, 1, 4, 4)
    from scipy.signal import convolve2d
    full = convolve2d(x[0, 0], W[0, 0], mode="full")
    z = O.conv2d_fwd(x, W, np.zeros(1), 1, "same")
    np.testing.assert_allclose(z[0, 0], full[1:7, 1:7], atol=1e-12)


def test_conv_stride_mismatch_raises():
    with pytest.raises(AssertionError):
        O.conv_geometry(9, 3, 2, "valid")
    assert O.conv_geometry(10, 3, 2, "valid") == (0, 0, 4)


def test_conv_stride_subsample_values():
    rng = np.random.RandomState(1)
    x = rng.randn(2, 3, 10, 10); W = rng.randn(4, 3, 3, 3); b = rng.randn(4)
    z1 = O.conv2d_fwd(x, W, b, 1, "valid")
    z2 = O.conv2d_fwd(x, W, b, 2, "valid")
    np.testing.assert_allclose(z2, z1[:, :, ::2, ::2])


def test_kat2_pool_ceil_mode_and_ties():
    x = np.arange(25, dtype=np.float64).reshape(1, 1, 5, 5)
    y = O.pool_fwd(x, 2, ignore_border=False)
    np.testing.assert_array_equal(y[0, 0], [[6, 8, 9], [16, 18, 19], [21, 23, 24]])
    y = O.pool_fwd(x, 2, ignore_border=True)
    np.testing.assert_array_equal(y[0, 0], [[6, 8], [16, 18]])
    # ties: every maximal element receives the full gradient (Theano MaxPoolGrad)
    x = np.ones((1, 1, 4, 4)); dy = np.array([[[[1., 2.], [3., 4.]]]])
    dx = O.pool_bwd(x, dy, 2)
    np.testing.assert_array_equal(dx[0, 0], np.kron(dy[0, 0], np.ones((2, 2))))
    # partial window gradient
    x = np.arange(9, dtype=np.float64).reshape(1, 1, 3, 3); dy = np.ones((1, 1, 2, 2))
    dx = O.pool_bwd(x, dy, 2)
    np.testing.assert_array_equal(dx[0, 0], [[0, 0, 0], [0, 1, 1], [0, 1, 1]])


def test_kat3_leaky_relu_value_and_grad_at_zero():
    f, df = O.activation("relu10")
    z = np.array([-2., 0., 3.], dtype=np.float32)
    np.testing.assert_allclose(f(z), [-0.2, 0, 3], rtol=1e-6)
    np.testing.assert_allclose(df(z), [0.1, 1.1, 1.0], rtol=1e-6)
    f, df = O.activation("relu")
    np.testing.assert_array_equal(df(z), [0, 1, 1])
    with pytest.raises(NotImplementedError):
        O.activation("nope")


def test_kat4_momentum_uses_old_velocity():
    reg = dict(O.DEFAULT_REG, momentum=.5)
    p, v = np.array([1.], np.float64), np.array([0.], np.float64)
    g = np.array([2.], np.float64)
    traj = []
    for _ in range(3):
        p, v = O.sgd_update(p, v, g, 0.1, reg)
        traj.append((p[0], v[0]))
    # step0: p unchanged (v_old=0), v=1 ; step1: p=1-.1*1=.9, v=1.5 ; step2: p=.9-.15=.75
    np.testing.assert_allclose(traj, [(1.0, 1.0), (0.9, 1.5), (0.75, 1.75)])


def test_kat5_maxnorm():
    p = np.array([-3., .5, 2.])
    np.testing.assert_array_equal(O.maxnorm_project(p, 1.), [-1, .5, 1])
    p = np.array([[3., .3], [4., .4]])                 # column norms 5, .5
    q = O.maxnorm_project(p, 1.)
    np.testing.assert_allclose(np.sqrt((q * q).sum(0)), [1, .5], rtol=1e-6)
    p = np.zeros((2, 1, 2, 2)); p[0] = 2.; p[1] = .1   # kernel norms 4, .2
    q = O.maxnorm_project(p, 1.)
    np.testing.assert_allclose(np.sqrt((q * q).sum((1, 2, 3))), [1, .2], rtol=1e-6)


def _mnist_prms():
    with open(os.path.join(ROOT, "params", "mnist.prms")) as fh:
        return ast.literal_eval(fh.read())


def test_kat6_seed_chain_init_hashes():
    prms = _mnist_prms()
    prms["layers"][0][1]["img_sz"] = 28
    prms["training_params"]["SEED"] = 555555
    net = O.OracleNet(prms["layers"], prms["training_params"])
    shapes = [[p.shape for p in l.params] for l in net.L]
    assert shapes == [[], [(4, 1, 3, 3), (4,)], [], [(20, 4, 3, 3), (20,)], [],
                      [(720, 500), (500,)], [(500, 10), (10,)]]
    # conv weights are +-1/sqrt(fan_in); biases: relu10 -> 0, relu05/relu01 -> .5, Softmax -> 0
    assert set(np.unique(net.L[1].params[0])) <= {np.float32(-1 / 3), np.float32(1 / 3)}
    assert np.all(net.L[1].params[1] == 0) and np.all(net.L[3].params[1] == .5)
    assert np.all(net.L[5].params[1] == .5) and np.all(net.L[6].params[1] == 0)
    assert abs(net.L[5].params[0]).max() <= np.sqrt(6 / (2 * 1220))
    golden = np.load(os.path.join(ROOT, "tests", "golden", "kat.npz"))
    for i, l in enumerate(net.L):
        for j, p in enumerate(l.params):
            h = hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest()
            assert h == str(golden["init_sha_%d_%d" % (i, j)])


def _check_against_init_ref(name, arr, ref):
    if name in ref.files:
        assert np.array_equal(arr, ref[name]), name
    else:                                   # big tensor: digest + every 37th value
        assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == str(ref[name + "_sha256"]), name
        assert np.array_equal(arr.ravel()[::37], ref[name + "_every37"]), name


def test_init_matches_the_reference_drawn_fixture():
    """tests/golden/init_ref.npz was drawn by the REFERENCE's own lines (theanet/layer/weights.py:51-65 compiled
    from /root/reference by tests/golden/make_golden.py::make_init_ref): the oracle's restatement of init_wb and of
    the seed chain must reproduce it bit for bit -- mnist.prms under SEED 555555 and one layer per bias / scale
    rule (sigmoid x4, softplus / relu / relu0x -> b = .5, relu10 / tanh -> 0)."""
    ref = np.load(os.path.join(ROOT, "tests", "golden", "init_ref.npz"))
    prms = _mnist_prms()
    prms["layers"][0][1]["img_sz"] = 28
    prms["training_params"]["SEED"] = 555555
    net = O.OracleNet(prms["layers"], prms["training_params"])
    seen = 0
    for i, l in enumerate(net.L):
        for p, n in zip(l.params, "Wb"):
            assert p.dtype == np.float32
            _check_against_init_ref("mnist_%d_%s" % (i, n), p, ref)
            seen += 1
    assert seen == 8
    for j, (act, size_w) in enumerate([("sigmoid", (7, 5)), ("softplus", (3, 2, 3, 3)), ("relu", (6, 4)),
                                       ("relu10", (2, 3, 5, 5)), ("relu05", (9, 2)), ("tanh", (4, 1, 3, 3))]):
        assert str(ref["case%d_act" % j]) == act
        fan_in = int(np.prod(size_w[1:])) if len(size_w) == 4 else size_w[0]
        w, b = O.init_wb(np.random.RandomState(1000 + j), size_w, (size_w[0] if len(size_w) == 4 else size_w[1],),
                         fan_in, size_w[-1], act)
        assert np.array_equal(w, ref["case%d_W" % j]) and np.array_equal(b, ref["case%d_b" % j]), act


def test_elastic_gaussian_matches_the_reference_lines():
    """The gaussian of the elastic stage as the REFERENCE's own lines build it (inlayers.py:87-91, run in place by
    tests/golden/make_golden.py): float64 exponentials cast to float32, then a float32 division."""
    ref = np.load(os.path.join(ROOT, "tests", "golden", "train_helpers.npz"))
    for sigma in (1, 2, 3, 4, 8):
        want = ref["elastic_filt_sigma%d" % sigma]
        got = O.elastic_filter(sigma)
        assert got.dtype == np.float32 and got.shape == (2 * sigma + 1, 2 * sigma + 1)
        assert np.array_equal(got, want), sigma


def test_kat7_lr_schedule():
    prms = _mnist_prms()
    prms["layers"][0][1]["img_sz"] = 28
    prms["training_params"]["SEED"] = 1
    net = O.OracleNet(prms["layers"], prms["training_params"])
    rates = []
    for _ in range(4):
        rates.append(float(net.cur_learn_rate))
        net.inc_epoch_set_rate()
    np.testing.assert_allclose(rates, [.1, .05, .1 / 3, .025], rtol=1e-6)


# ---- float64 finite differences through the whole net ------------------------------

def _tiny_net(dtype, with_same=False):
    layers = [
        ("InputLayer", {"img_sz": 9, "num_maps": 2}),
        ("ConvLayer", {"num_maps": 3, "filter_sz": 3, "stride": 1, "actvn": "relu10",
                       "mode": "same" if with_same else "valid",
                       "reg": {"L1": .01, "L2": .02}}),
        ("PoolLayer", {"pool_sz": 2}),
        ("ConvLayer", {"num_maps": 4, "filter_sz": 2, "stride": 1, "actvn": "tanh"}),
        ("PoolLayer", {"pool_sz": 2, "ignore_border": True}),
        ("HiddenLayer", {"n_out": 7, "pdrop": .5, "actvn": "scaled_tanh",
                         "reg": {"L2": .03}}),
        ("SoftmaxLayer", {"n_out": 5}),
    ]
    tr = {"SEED": 7, "BATCH_SZ": 4, "INIT_LEARNING_RATE": .1, "EPOCHS_TO_HALF_RATE": 1}
    return O.OracleNet(layers, tr, dtype=dtype)


@pytest.mark.parametrize("same", [False, True])
def test_fd_gradients_float64(same):
    net = _tiny_net(np.float64, same)
    rng = np.random.RandomState(3)
    x = rng.rand(4, 2, 9, 9)
    y = rng.randint(0, 5, 4)
    mask = (rng.rand(4, 7) > .5).astype(np.float64)
    draws = {5: mask}
    cost, _, grads, _ = net.grads(x, y, draws)
    eps = 1e-6
    for i, l in enumerate(net.L):
        for j, p in enumerate(l.params):
            flat = p.reshape(-1)
            for idx in rng.choice(flat.size, size=min(6, flat.size), replace=False):
                old = flat[idx]
                flat[idx] = old + eps
                cp = net.cost(net.forward(x, True, draws)[0], y)
                flat[idx] = old - eps
                cm = net.cost(net.forward(x, True, draws)[0], y)
                flat[idx] = old
                fd = (cp - cm) / (2 * eps)
                an = grads[i][j].reshape(-1)[idx]
                assert abs(fd - an) <= 1e-6 + 1e-5 * abs(fd), (i, j, idx, fd, an)


def test_torch_cross_check_forward_backward():
    """Independent second implementation (torch CPU autograd) on tie-free data."""
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    rng = np.random.RandomState(5)
    x = rng.rand(3, 2, 8, 8); W = rng.randn(4, 2, 3, 3); b = rng.randn(4)
    for mode, pad in (("valid", 0), ("same", 1)):
        z = O.conv2d_fwd(x, W, b, 1, mode)
        xt = torch.tensor(x, requires_grad=True); Wt = torch.tensor(W, requires_grad=True)
        zt = F.conv2d(xt, torch.flip(Wt, (2, 3)), torch.tensor(b), padding=pad)
        np.testing.assert_allclose(z, zt.detach().numpy(), atol=1e-12)
        dz = rng.randn(*z.shape)
        zt.backward(torch.tensor(dz))
        dx, dW, db = O.conv2d_bwd(x, W, dz, 1, mode)
        np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-12)
        np.testing.assert_allclose(dW, Wt.grad.numpy(), atol=1e-11)
        np.testing.assert_allclose(db, dz.sum((0, 2, 3)), atol=1e-12)
    # ceil-mode max pool + its gradient
    x = rng.rand(2, 3, 7, 7)
    xt = torch.tensor(x, requires_grad=True)
    yt = F.max_pool2d(xt, 2, ceil_mode=True)
    np.testing.assert_array_equal(O.pool_fwd(x, 2), yt.detach().numpy())
    dy = rng.randn(*yt.shape); yt.backward(torch.tensor(dy))
    np.testing.assert_allclose(O.pool_bwd(x, dy, 2), xt.grad.numpy())
    # log-softmax + nll gradient
    z = rng.randn(6, 5); y = rng.randint(0, 5, 6)
    zt = torch.tensor(z, requires_grad=True)
    lt = F.nll_loss(F.log_softmax(zt, 1), torch.tensor(y)); lt.backward()
    lp = O.log_softmax(z)
    np.testing.assert_allclose(O.nll(lp, y), lt.item(), rtol=1e-12)
    np.testing.assert_allclose(O.nll_dlogits(lp, y), zt.grad.numpy(), atol=1e-12)


def _xchk_net(dtype):
    import ast
    with open(os.path.join(ROOT, "params", "mnist.prms")) as fh:
        prms = ast.literal_eval(fh.read())
    prms["layers"][0] = ("InputLayer", {"img_sz": 28})
    tr = dict(prms["training_params"], SEED=555555, BATCH_SZ=8)
    return prms["layers"], tr


def xchk_compare(gold, name, arr, rtol, atol):
    """Compare ``arr`` with fixture entry ``name`` (large tensors are stored as a fixed subsample plus
    their sum and absolute sum, see tests/golden/make_torch_xchk.py::put)."""
    arr = np.asarray(arr, np.float64)
    if name in gold.files:
        np.testing.assert_allclose(arr, gold[name], rtol=rtol, atol=atol, err_msg=name)
        return
    idx = np.random.RandomState(arr.size % (2 ** 31)).choice(arr.size, 4096, replace=False)
    np.testing.assert_allclose(arr.reshape(-1)[idx], gold[name + "@sub"], rtol=rtol, atol=atol, err_msg=name)
    scale = float(gold[name + "@abs"])
    assert abs(arr.sum() - float(gold[name + "@sum"])) <= 10 * rtol * scale + atol, name
    assert abs(np.abs(arr).sum() - scale) <= 10 * rtol * scale + atol, name


def test_whole_net_trajectory_matches_the_torch_fixture():
    """tests/golden/torch_xchk.npz (made by tests/golden/make_torch_xchk.py: torch CPU autograd,
    float64, written from the reference's layer definitions without the oracle's arithmetic) pins the
    oracle on a WHOLE training trajectory of the mnist.prms net: logprobs and cost of three steps,
    every gradient of the first, the weights after the third (old-velocity update)."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "torch_xchk.npz"))
    layers, tr = _xchk_net(np.float64)
    net = O.OracleNet(layers, tr, dtype=np.float64)
    for i, w in enumerate([w for l in net.L for w in l.params]):     # same seed chain -> same initial weights
        xchk_compare(gold, "w0_%d" % i, w.astype(np.float32), 0, 0)
    x, y, masks = gold["x"].astype(np.float64), gold["y"], gold["masks"].astype(np.float64)
    B = 8
    for s in range(3):
        xs, ys = x[s * B:(s + 1) * B], y[s * B:(s + 1) * B]
        if s == 0:
            cost, lp, grads, _ = net.grads(xs, ys, {5: masks[0]})
            for i, g in enumerate([g for gl in grads if gl is not None for g in gl]):
                xchk_compare(gold, "grad0_%d" % i, g, 1e-9, 1e-13)
        cost, lp, _ = net.train_step(xs, ys, {5: masks[s]})
        np.testing.assert_allclose(lp, gold["logprob_%d" % s], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(cost, gold["cost_%d" % s], rtol=1e-10)
    for i, w in enumerate([w for l in net.L for w in l.params]):
        xchk_compare(gold, "w3_%d" % i, w, 1e-9, 1e-13)


def test_oracle_gradients_through_mid_net_color_and_elastic_layers_fd():
    """Finite differences (float64) through ColorLayer and ElasticLayers in the middle of a net: the
    oracle's hand-written backward of the clip / pow chain and of the gather (scatter-add, flip and
    inversion signs) -- what Theano's grad does through color.py:38-44 and inlayers.py:63-142."""
    import copy
    layers = [
        ("InputLayer", {"img_sz": 10, "num_maps": 2}),
        ("ConvLayer", {"num_maps": 3, "filter_sz": 3, "stride": 1, "mode": "same", "actvn": "sigmoid"}),
        ("ColorLayer", {"balance": 1.4, "gamma": 1.7}),
        ("ElasticLayer", {"translation": 1.2, "zoom": 1.15, "magnitude": 15, "sigma": 2, "pflip": .1, "angle": 8}),
        ("ConvLayer", {"num_maps": 4, "filter_sz": 3, "stride": 1, "actvn": "tanh"}),
        ("ElasticLayer", {"translation": 1, "nearest": True, "invert_image": True}),
        ("SoftmaxLayer", {"n_out": 5}),
    ]
    tr = {"SEED": 3, "BATCH_SZ": 4, "INIT_LEARNING_RATE": .1, "EPOCHS_TO_HALF_RATE": 1}
    net = O.OracleNet(copy.deepcopy(layers), dict(tr), dtype=np.float64)
    rng = np.random.RandomState(0)
    x = rng.rand(4, 2, 10, 10)
    y = rng.randint(0, 5, 4)
    draws = {2: net.L[2].stage.draw(4), 3: net.L[3].stage.draw((4, 3, 10, 10)), 5: net.L[5].stage.draw((4, 4, 8, 8))}
    cost, _, grads, _ = net.grads(x, y, draws)
    eps = 1e-6
    for i in (1, 4):                       # conv below the color/elastic pair, conv between the elastic layers
        for j, p in enumerate(net.L[i].params):
            flat = p.reshape(-1)
            for idx in rng.choice(flat.size, min(6, flat.size), replace=False):
                old = flat[idx]
                flat[idx] = old + eps
                cp = net.cost(net.forward(x, True, draws)[0], y)
                flat[idx] = old - eps
                cm = net.cost(net.forward(x, True, draws)[0], y)
                flat[idx] = old
                fd = (cp - cm) / (2 * eps)
                an = grads[i][j].reshape(-1)[idx]
                assert abs(fd - an) <= 1e-7 + 2e-5 * abs(fd), (i, j, idx, fd, an)


@pytest.mark.parametrize("variant", ["concat", "softaux"])
def test_oracle_gradients_of_aux_input_layers_fd(variant):
    """Finite differences (float64) through AuxConcatLayer / SoftAuxLayer and their LocationInfo perceptron
    (auxiliary.py:14-160): the eight parameter tensors of the SoftAux layer, the layers below both."""
    import copy
    layers = [("InputLayer", {"img_sz": 6, "num_maps": 1}),
              ("HiddenLayer", {"n_out": 7, "actvn": "tanh"}),
              ("AuxConcatLayer", {"n_aux": (5, 4), "aux_type": "LocationInfo", "boost": 2}),
              ("SoftmaxLayer", {"n_out": 4})]
    if variant == "softaux":
        layers[2:] = [("SoftAuxLayer", {"n_out": 4, "n_aux": (4, 3), "aux_type": "LocationInfo", "boost": 1.5})]
    tr = {"SEED": 5, "BATCH_SZ": 5, "INIT_LEARNING_RATE": .1, "EPOCHS_TO_HALF_RATE": 1}
    net = O.OracleNet(copy.deepcopy(layers), dict(tr), dtype=np.float64)
    rng = np.random.RandomState(1)
    x, y = rng.rand(5, 1, 6, 6), rng.randint(0, 4, 5)
    net.set_aux(rng.rand(5, 2, 2))
    draws = {2: net.L[2].aux.draw(5)}
    for p in net.L[2].params[2:6] if variant == "softaux" else net.L[2].params:
        p += .05 * rng.rand(*p.shape)                          # away from the relu kinks, biases non-zero
    cost, _, grads, _ = net.grads(x, y, draws)
    eps, worst = 1e-6, 0.0
    check = [1, 2] if variant == "softaux" else [1, 3]          # AuxConcat's own weights have no gradient slot
    for i in check:
        for j, p in enumerate(net.L[i].params):
            flat = p.reshape(-1)
            for idx in rng.choice(flat.size, min(5, flat.size), replace=False):
                old = flat[idx]
                flat[idx] = old + eps
                cp = net.cost(net.forward(x, True, draws)[0], y)
                flat[idx] = old - eps
                cm = net.cost(net.forward(x, True, draws)[0], y)
                flat[idx] = old
                fd, an = (cp - cm) / (2 * eps), grads[i][j].reshape(-1)[idx]
                assert abs(fd - an) <= 1e-7 + 2e-5 * abs(fd), (i, j, idx, fd, an)
