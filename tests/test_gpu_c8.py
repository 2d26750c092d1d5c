"""The fp16-RESIDENT kernels of DTYPE 'float16' (BASELINE.json configs[4]: "fp16 inputs / fp32 accum MFMA"), op by op
through the C-ABI: c8 conv forward (+ fused 2x2 max-pool and mask), input gradient (plain and gathered from a pooled
gradient + mask), weight gradient (plain and gathered), pack / unpack, and the dense products on a c8 input
(include/theanet_hip.h tn_c8_*; theanet_amd/csrc/conv_c8.hip, fc_c8.hip).

Specification = the stored-fp16 arithmetic restated in numpy (tests/c8_util.py, the same statement as
oracle.theanet_oracle's DTYPE float16 mode): operands are halfs, products exact, float64 sums standing in for the
device's fp32 accumulation, one rounding to half when a tensor is stored.  Tolerances: a stored fp16 tensor may differ
by one rounding (half an ulp = 4.9e-4 relative) where the fp32 and the float64 sums fall on different sides of a
rounding boundary: 1e-3 of the largest entry; fp32 results (weight / bias gradients, dense outputs): 2e-5 of the largest
entry (accumulation order only); pooling masks bit-exact except at provable near-ties (every mismatching byte is
checked against the fp32 accumulation bound, _assert_masks_equal_up_to_provable_near_ties).  The reference itself is
float32-only (weights.py:8)."""
import numpy as np
import pytest

from tests import c8_util as U
from tests.gpu_util import ctx, dev, empty, call

pytestmark = pytest.mark.gpu

GS = 1024.0
LEAKY, SLOPE = 1, .1          # TN_ACT_LEAKY (include/theanet_hip.h enum tn_act)


@pytest.fixture
def f16_mode():
    ctx().set_matmul_dtype("float16", GS)
    yield
    ctx().set_matmul_dtype("float32")


def _c8(a):
    return dev(U.to_c8(a).view(np.uint16))


def _rel(got, want):
    return float(np.abs(got - want).max() / np.abs(want).max())


def _assert_masks_equal_up_to_provable_near_ties(gotm, bits, a, x, W16, b, C):
    """Index work is bit-exact: every mask byte equals the specification's -- except where the device's fp32 sums and the
    specification's float64 sums may PROVABLY order two window elements differently: a window bit may differ only if
    that element lies within the fp32 accumulation bound of the window maximum, a sign bit only if the maximum lies
    within it of zero.  Bound per pre-activation: (9 C + 2) u * (sum |x| |w| + |b|), u = 2^-24 (n fp32 additions of exact
    fp16 x fp16 products); the leaky ReLU is monotone with slope <= 1, so the bound carries over to the activations."""
    bad = gotm != bits
    if not bad.any():
        return
    assert bad.mean() < 1e-3, "mask mismatches are not rare: %g" % bad.mean()
    absum = U.conv_same(np.abs(x), np.abs(W16)) + np.abs(b)[None, :, None, None]
    tol = (9 * C + 2) * 2.0 ** -24 * absum
    N, K, H, _ = a.shape
    aw = a.reshape(N, K, H // 2, 2, H // 2, 2)
    tw = tol.reshape(N, K, H // 2, 2, H // 2, 2)
    m, tmax = aw.max(axis=(3, 5)), tw.max(axis=(3, 5))
    for n, k, i, j in zip(*np.nonzero(bad)):
        diff = int(gotm[n, k, i, j]) ^ int(bits[n, k, i, j])
        for e in range(4):
            if (diff >> e) & 1:
                gap = m[n, k, i, j] - aw[n, k, i, e >> 1, j, e & 1]
                assert gap <= 2 * tmax[n, k, i, j], ("window bit differs away from a tie", (n, k, i, j, e), gap, tmax[n, k, i, j])
        if diff & 0x30:
            assert abs(m[n, k, i, j]) <= tmax[n, k, i, j], ("sign bit differs away from zero", (n, k, i, j), m[n, k, i, j])
        assert diff & ~0x3f == 0, ("unused mask bits set", (n, k, i, j), gotm[n, k, i, j])


C8_CASES = [  # N, C, H, K
    (3, 16, 16, 32), (2, 64, 64, 64), (5, 24, 8, 40), (2, 128, 32, 128), (2, 8, 64, 64), (3, 64, 32, 128),
    (3, 3, 64, 64), (37, 3, 32, 32), (9, 256, 16, 256), (33, 40, 8, 72),
    (520, 3, 32, 32), (131, 5, 64, 40), (2100, 8, 16, 32),      # first layers with enough pixels for 512-pixel weight-gradient tiles
    (40, 32, 16, 64), (21, 32, 8, 64), (6, 24, 32, 96),           # 32-channel layers (conv2 of cifar_like): the sixteen-wave weight gradient with two step subsets
]


@pytest.mark.parametrize("case", C8_CASES)
def test_c8_conv_ops(case, f16_mode):
    from theanet_amd import _lib
    assert _lib.TN_ACT_LEAKY == LEAKY
    N, C, H, K = case
    rng = np.random.RandomState(0)
    lib = ctx().lib
    assert lib.tn_c8_conv_supported(N, C, H, H, K, 3, 1, 1) and lib.tn_c8_conv_wgrad_supported(N, C, H, H, K)
    x = U.r16(rng.randn(N, C, H, H))
    W = (rng.randn(K, C, 3, 3) / np.sqrt(9 * C)).astype(np.float32)
    b = (rng.randn(K) * .1).astype(np.float32)
    W16 = U.r16(W)
    a = U.leaky(U.conv_same(x, W16) + b[None, :, None, None], SLOPE)
    xd, Wd, bd = _c8(x), dev(W), dev(b)
    K8, C8, Hp = K // 8, (C + 7) // 8, H // 2
    # forward
    out = empty((N, K8, H, H, 8), np.uint16)
    call("tn_c8_conv_fwd", xd.ptr, Wd.ptr, bd.ptr, out.ptr, None, N, C, H, H, K, LEAKY, SLOPE, 0, None)
    assert _rel(U.from_c8(out.get_value().view(np.float16), K), U.r16(a)) < 1e-3
    # forward + 2x2 max-pool + mask (ties: every window element equal to the maximum)
    pm, bits = U.pool2(a)
    outp, mk = empty((N, K8, Hp, Hp, 8), np.uint16), empty((N, K8, Hp, Hp, 8), np.uint8)
    call("tn_c8_conv_fwd", xd.ptr, Wd.ptr, bd.ptr, outp.ptr, mk.ptr, N, C, H, H, K, LEAKY, SLOPE, 1, None)
    assert _rel(U.from_c8(outp.get_value().view(np.float16), K), U.r16(pm)) < 1e-3
    gotm = mk.get_value().transpose(0, 1, 4, 2, 3).reshape(N, K, Hp, Hp)
    _assert_masks_equal_up_to_provable_near_ties(gotm, bits, a, x, W16, b, C)
    # input gradient: dz (halfs at the gradient scale) -> dx * act'(output of the layer below), stored as halfs
    dz = U.r16(GS * rng.randn(N, K, H, H) * 1e-3)
    prev = U.r16(rng.randn(N, C, H, H))
    prev[0, 0, 0, :2] = 0                              # exact zeros: the tie derivative 1 + slope
    dxw = U.conv_same_dgrad(dz, W16) * U.leaky_grad_from_out(prev, SLOPE)
    dzd, pd = _c8(dz), _c8(prev)
    dxo = empty((N, C8, H, H, 8), np.uint16)
    call("tn_c8_conv_dgrad", dzd.ptr, Wd.ptr, dxo.ptr, N, C, H, H, K, pd.ptr, LEAKY, SLOPE, 0, None, None)
    assert _rel(U.from_c8(dxo.get_value().view(np.float16), C), U.r16(dxw)) < 1e-3
    # ... of a pooled block: dz = (window bit of the device's own mask) ? pooled gradient : 0
    g = U.r16(GS * rng.randn(N, K, Hp, Hp) * 1e-3)
    gd = _c8(g)
    dzp = U.unpool_dz(g, gotm)
    call("tn_c8_conv_dgrad", gd.ptr, Wd.ptr, dxo.ptr, N, C, H, H, K, pd.ptr, LEAKY, SLOPE, 1, mk.ptr, None)
    dxw2 = U.conv_same_dgrad(dzp, W16) * U.leaky_grad_from_out(prev, SLOPE)
    assert _rel(U.from_c8(dxo.get_value().view(np.float16), C), U.r16(dxw2)) < 1e-3
    # weight / bias gradient (fp32 results, scale removed), plain and gathered
    gW, gb = empty((K, C, 3, 3)), empty((K,))
    for pooled, src, dzz in ((0, dzd, dz), (1, gd, dzp)):
        call("tn_c8_conv_wgrad", xd.ptr, src.ptr, gW.ptr, gb.ptr, N, C, H, H, K, pooled, mk.ptr if pooled else None)
        assert _rel(gW.get_value(), U.conv_same_wgrad(x, dzz) / GS) < 2e-5
        assert _rel(gb.get_value(), dzz.sum(axis=(0, 2, 3)) / GS) < 2e-5


def _wgrad_blas(x, dz):
    """U.conv_same_wgrad as nine matrix products (float64): the large shapes below in a second instead of a minute."""
    N, C, H, Wd = x.shape
    K = dz.shape[1]
    xp = np.zeros((N, C, H + 2, Wd + 2))
    xp[:, :, 1:-1, 1:-1] = x
    d2 = dz.astype(np.float64).transpose(1, 0, 2, 3).reshape(K, -1)
    dW = np.zeros((K, C, 3, 3))
    for u in range(3):
        for v in range(3):
            dW[:, :, 2 - u, 2 - v] = d2 @ xp[:, :, u:u + H, v:v + Wd].transpose(1, 0, 2, 3).reshape(C, -1).T
    return dW


@pytest.mark.parametrize("case", [(44, 128, 32, 128), (70, 64, 64, 64), (19, 32, 64, 96), (2049, 16, 32, 32), (21, 40, 32, 72)])
def test_c8_wgrad_rolling_ring_over_many_tiles(case, f16_mode):
    """The weight gradient's rolling x ring (c8_wgrad_kernel ROLL: rows of >= 32 pixels, one image band per tile) with SEVERAL
    tiles per slab: the ring wraps (four regions), slabs start in the middle of an image (the row above comes from the
    tile in front of the slab) and cross image boundaries (zero row), the last slab is short, channel / filter planes
    beyond the tensors (re-read, never stored); plain and gathered from a pooled gradient with a random mask.  convpool.py:54-56 (CorrMM_gradWeights); numbers as in test_c8_conv_ops."""
    N, C, H, K = case
    rng = np.random.RandomState(11)
    assert ctx().lib.tn_c8_conv_wgrad_supported(N, C, H, H, K)
    x = U.r16(rng.randn(N, C, H, H))
    dz = U.r16(GS * rng.randn(N, K, H, H) * 1e-3)
    Hp = H // 2
    g = U.r16(GS * rng.randn(N, K, Hp, Hp) * 1e-3)
    bits = rng.randint(1, 16, (N, K, Hp, Hp)).astype(np.uint8)
    mk = dev(np.ascontiguousarray(bits.reshape(N, K // 8, 8, Hp, Hp).transpose(0, 1, 3, 4, 2)))
    dzp = U.unpool_dz(g, bits)
    xd, dzd, gd = _c8(x), _c8(dz), _c8(g)
    gW, gb = empty((K, C, 3, 3)), empty((K,))
    for pooled, src, dzz in ((0, dzd, dz), (1, gd, dzp)):
        call("tn_c8_conv_wgrad", xd.ptr, src.ptr, gW.ptr, gb.ptr, N, C, H, H, K, pooled, mk.ptr if pooled else None)
        assert _rel(gW.get_value(), _wgrad_blas(x, dzz) / GS) < 2e-5
        assert _rel(gb.get_value(), dzz.sum(axis=(0, 2, 3)) / GS) < 2e-5


def test_c8_generic_activation_and_pack_roundtrip(f16_mode):
    """An activation outside the leaky-ReLU family takes the generic epilogue; pack / unpack are exact on halfs."""
    from theanet_amd.layer.layer import activation_by_name
    N, C, H, K = 3, 16, 16, 24
    rng = np.random.RandomState(3)
    x = U.r16(rng.randn(N, C, H, H))
    W = (rng.randn(K, C, 3, 3) / 12).astype(np.float32)
    b = (rng.randn(K) * .1).astype(np.float32)
    act = activation_by_name("tanh")
    out = empty((N, K // 8, H, H, 8), np.uint16)
    call("tn_c8_conv_fwd", _c8(x).ptr, dev(W).ptr, dev(b).ptr, out.ptr, None, N, C, H, H, K, act.kind, act.prm, 0, None)
    want = np.tanh(U.conv_same(x, U.r16(W)) + b[None, :, None, None])
    assert _rel(U.from_c8(out.get_value().view(np.float16), K), U.r16(want)) < 1e-3
    # pack rows 2.. of an fp32 NCHW dataset (x scale), unpack back
    data = rng.randn(N + 2, 3, H, H).astype(np.float32)
    packed = empty((N, 1, H, H, 8), np.uint16)
    call("tn_c8_pack", dev(data).ptr, 2, packed.ptr, N, 3, H * H, 2.0)
    raw = packed.get_value().view(np.float16)
    np.testing.assert_array_equal(U.from_c8(raw, 3), (2.0 * data[2:]).astype(np.float16).astype(np.float32))
    assert not raw[..., 3:].any()                      # channels beyond C are zero
    back = empty((N, 3, H, H))
    call("tn_c8_unpack", packed.ptr, back.ptr, N, 3, H * H, .5)
    np.testing.assert_array_equal(back.get_value(), (2.0 * data[2:]).astype(np.float16).astype(np.float32) * .5)


def _rowmap(C, HW):
    C8 = (C + 7) // 8
    k = np.arange(C8 * HW * 8)
    cell, e = k >> 3, k & 7
    o, p = cell // HW, cell % HW
    ch = o * 8 + e
    return np.where(ch < C, ch * HW + p, -1)


@pytest.mark.parametrize("case", [(5, 16, 4, 32), (37, 24, 16, 96), (128, 40, 8, 160), (200, 64, 1, 64), (300, 128, 16, 512),
                                  (2000, 128, 16, 1024)])      # (the last: one K slab, the forward finishes its own outputs)
def test_c8_fc_ops(case, f16_mode):
    """Dense products on a c8 input: W (C*HW, n_out) keeps the reference's NCHW-flattened row order (neuralnet.py:168-173)
    and is walked through the row map."""
    B, C, HW, N = case
    rng = np.random.RandomState(1)
    assert ctx().lib.tn_c8_fc_supported(B, C, HW, N)
    rm = _rowmap(C, HW)
    Kc, n_in, ok = len(rm), C * HW, rm >= 0
    x = np.zeros((B, Kc)); x[:, ok] = U.r16(rng.randn(B, n_in))[:, rm[ok]]
    W = (rng.randn(n_in, N) / np.sqrt(n_in)).astype(np.float32)
    b = (rng.randn(N) * .1).astype(np.float32)
    mask = (rng.rand(B, N) < .5).astype(np.uint8)
    Wp = np.zeros((Kc, N)); Wp[ok] = U.r16(W)[rm[ok]]
    z = x @ Wp + b
    xd, Wd, bd = dev(x.astype(np.float16).view(np.uint16)), dev(W), dev(b)
    a = empty((B, N))
    call("tn_c8_fc_fwd", xd.ptr, Wd.ptr, bd.ptr, a.ptr, B, C, HW, N, LEAKY, SLOPE, dev(mask).ptr)
    assert _rel(a.get_value(), U.leaky(z, SLOPE) * mask) < 2e-5
    # the same product with the mask drawn in the launch: the bits of tn_dropout_mask, the masked output of the call above
    want, got_mask, a2 = empty((B * N,), np.uint8), empty((B * N,), np.uint8), empty((B, N))
    for elem0 in (1000, 1003):               # (1003: an output quad straddles two Philox blocks)
        call("tn_dropout_mask", want.ptr, B * N, .3, 99, 5, None, elem0)
        call("tn_c8_fc_fwd_dropout", xd.ptr, Wd.ptr, bd.ptr, a2.ptr, B, C, HW, N, LEAKY, SLOPE, got_mask.ptr, .3, 99, 5, None,
             elem0)
        assert np.array_equal(got_mask.get_value(), want.get_value())
        call("tn_c8_fc_fwd", xd.ptr, Wd.ptr, bd.ptr, a.ptr, B, C, HW, N, LEAKY, SLOPE, want.ptr)
        assert np.array_equal(a2.get_value(), a.get_value())
    dz = (rng.randn(B, N) * 1e-3).astype(np.float32)
    dz16 = U.r16(GS * dz)
    y = U.r16(rng.randn(B, Kc)); y[0, :3] = 0
    dxw = (dz16 @ Wp.T) * U.leaky_grad_from_out(y, SLOPE)
    dzd = dev(dz)
    dxo = empty((B, Kc), np.uint16)
    call("tn_c8_fc_dgrad", dzd.ptr, Wd.ptr, dxo.ptr, B, C, HW, N, dev(y.astype(np.float16).view(np.uint16)).ptr, LEAKY, SLOPE)
    got = dxo.get_value().view(np.float16).astype(np.float64)
    assert _rel(got[:, ok], U.r16(dxw)[:, ok]) < 1e-3
    dWw = np.zeros((n_in, N)); dWw[rm[ok]] = (x.T @ dz16)[ok] / GS
    gW, gb = empty((n_in, N)), empty((N,))
    call("tn_c8_fc_wgrad", xd.ptr, dzd.ptr, gW.ptr, gb.ptr, B, C, HW, N)
    assert _rel(gW.get_value(), dWw) < 2e-5
    assert _rel(gb.get_value(), dz16.sum(0) / GS) < 2e-5


def test_c8_unsupported_shapes_are_errors_not_fallbacks(f16_mode):
    from theanet_amd import _lib
    lib = ctx().lib
    assert not lib.tn_c8_conv_supported(4, 16, 16, 16, 20, 3, 1, 1)       # filters not a multiple of 8
    assert not lib.tn_c8_conv_supported(4, 16, 12, 12, 16, 3, 1, 1)       # rows of 12 pixels
    assert not lib.tn_c8_conv_supported(4, 16, 16, 16, 16, 5, 1, 2)       # 5x5
    assert not lib.tn_c8_fc_supported(4, 10, 1, 32)                        # 16 c8 inputs: not a multiple of 64
    x, W, b = empty((4, 2, 16, 16, 8), np.uint16), dev(np.zeros((20, 16, 3, 3), np.float32)), dev(np.zeros(20, np.float32))
    with pytest.raises(_lib.BackendError, match="multiple of 8"):
        call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, x.ptr, None, 4, 16, 16, 16, 20, LEAKY, SLOPE, 0, None)
    from theanet_amd import NeuralNet
    tp = {"SEED": 1, "BATCH_SZ": 4, "INIT_LEARNING_RATE": .1, "EPOCHS_TO_HALF_RATE": 1, "DTYPE": "float16"}
    with pytest.raises(AssertionError, match="DTYPE float16"):
        NeuralNet([("InputLayer", {"img_sz": 16, "num_maps": 3}),
                   ("ConvLayer", {"num_maps": 20, "filter_sz": 3, "stride": 1, "mode": "same"}),
                   ("SoftmaxLayer", {"n_out": 10})], dict(tp))
    with pytest.raises(AssertionError, match="DTYPE float16"):          # a dense layer must follow the conv stack
        NeuralNet([("InputLayer", {"img_sz": 16, "num_maps": 3}),
                   ("ConvLayer", {"num_maps": 16, "filter_sz": 3, "stride": 1, "mode": "same"}),
                   ("SoftmaxLayer", {"n_out": 10})], dict(tp))
    ctx().set_matmul_dtype("float32")
