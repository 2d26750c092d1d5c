"""Per-kernel parity: every HIP op (through the C-ABI) against the numpy oracle on the
same seeded inputs.  Tolerance for fp32 kernels: rtol 1e-4 / atol 1e-5 (north_star:
1e-4 rel on logits); index / mask / argmax outputs are bit-exact."""
import os

import numpy as np
import pytest

from oracle import theanet_oracle as O
from tests.gpu_util import act_code, assert_close, call, ctx, dev, empty

pytestmark = pytest.mark.gpu

CONV_CASES = [
    # N, C, H, K, f, stride, mode, act
    (3, 1, 28, 4, 3, 1, "valid", "relu10"),
    (2, 4, 13, 20, 3, 1, "valid", "relu05"),
    (2, 3, 12, 8, 3, 1, "same", "tanh"),
    (2, 2, 11, 6, 5, 1, "valid", "relu"),
    (2, 3, 10, 5, 3, 2, "valid", "sigmoid"),
    (1, 5, 9, 33, 3, 1, "same", "relu10"),
    (2, 2, 8, 3, 4, 1, "same", "linear"),
    (2, 3, 9, 4, 2, 1, "valid", "scaled_tanh"),
    (2, 2, 7, 3, 1, 1, "valid", "softplus"),
    (2, 9, 8, 16, 3, 1, "same", "relu10"),
    # implicit-im2col MFMA path (C*f*f >= 32, K >= 16)
    (3, 16, 12, 32, 3, 1, "same", "relu10"),
    (2, 32, 9, 64, 3, 1, "valid", "relu10"),
    (2, 17, 10, 70, 3, 1, "same", "tanh"),          # ragged: Kd = 153 (not % 4), K = 70
    (1, 8, 14, 16, 5, 1, "same", "relu05"),
    (5, 64, 8, 96, 3, 1, "same", "relu10"),
    (2, 24, 7, 40, 2, 1, "valid", "linear"),
    # LDS-resident-tile MFMA path (3x3, row length % 4 == 0): whole small images per block, row
    # tiles of larger ones, ragged filter / channel counts, partial image groups and lane tiles
    (2, 8, 16, 48, 3, 1, "valid", "relu10"),        # forward on the tile path
    (2, 8, 18, 32, 3, 1, "valid", "relu10"),        # dgrad on the tile path (padding 2)
    (3, 20, 32, 64, 3, 1, "same", "relu10"),
    (2, 16, 64, 64, 3, 1, "same", "tanh"),
    (1, 40, 36, 24, 3, 1, "same", "relu05"),
    (9, 10, 4, 33, 3, 1, "same", "sigmoid"),
    # first layers (C <= 4, >= 16 filters): forward on the tile kernel with two channel pairs, weight
    # gradient with the (channel, tap) pairs as GEMM columns
    (3, 3, 16, 32, 3, 1, "same", "relu10"),
    (2, 1, 32, 16, 3, 1, "same", "tanh"),
    (5, 2, 8, 40, 3, 1, "same", "relu05"),
]


def _conv_setup(case, seed=0):
    N, C, H, K, f, s, mode, act = case
    rng = np.random.RandomState(seed)
    x = rng.randn(N, C, H, H).astype(np.float32)
    W = (rng.randn(K, C, f, f) / np.sqrt(C * f * f)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    pad_lo, _, out = O.conv_geometry(H, f, s, mode)
    return x, W, b, pad_lo, out


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(case):
    N, C, H, K, f, s, mode, act = case
    x, W, b, pad_lo, out = _conv_setup(case)
    want = O.activation(act)[0](O.conv2d_fwd(x, W, b, s, mode))
    a = empty((N, K, out, out))
    kind, prm = act_code(act)
    call("tn_conv2d_fwd", dev(x).ptr, dev(W).ptr, dev(b).ptr, a.ptr, N, C, H, H, K, f, s, pad_lo,
         out, out, kind, prm)
    assert_close(a.get_value(), want, what="conv fwd %s" % (case,))


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad_dgrad(case):
    N, C, H, K, f, s, mode, act = case
    x, W, b, pad_lo, out = _conv_setup(case, 1)
    rng = np.random.RandomState(2)
    dz = rng.randn(N, K, out, out).astype(np.float32)
    dx_w, dW_w, db_w = O.conv2d_bwd(x.astype(np.float64), W.astype(np.float64),
                                    dz.astype(np.float64), s, mode)
    dW, db, dx = empty(W.shape), empty((K,)), empty(x.shape)
    dxd, dzd, Wd = dev(x), dev(dz), dev(W)
    call("tn_conv2d_wgrad", dxd.ptr, dzd.ptr, dW.ptr, db.ptr, N, C, H, H, K, f, s, pad_lo, out, out)
    # entries near zero are sums of N*Ho*Wo products that cancel: the absolute floor follows the
    # largest entry (fp32 accumulation), never below 1e-4
    assert_close(dW.get_value(), dW_w, atol=max(1e-4, 2e-6 * np.abs(dW_w).max()), what="conv dW %s" % (case,))
    assert_close(db.get_value(), db_w, atol=max(1e-4, 2e-6 * np.abs(db_w).max()), what="conv db %s" % (case,))
    call("tn_conv2d_dgrad", dzd.ptr, Wd.ptr, dx.ptr, N, C, H, H, K, f, s, pad_lo, out, out,
         None, 0, 0.0)
    assert_close(dx.get_value(), dx_w, atol=1e-4, what="conv dx %s" % (case,))
    # fused activation gradient of the layer below
    prev_a = rng.randn(*x.shape).astype(np.float32)
    prev_a[0, 0, 0, :3] = 0          # exact zeros exercise the tie rule
    kind, prm = act_code("relu10")
    call("tn_conv2d_dgrad", dzd.ptr, Wd.ptr, dx.ptr, N, C, H, H, K, f, s, pad_lo, out, out,
         dev(prev_a).ptr, kind, prm)
    g = np.where(prev_a > 0, 1.0, np.where(prev_a < 0, .1, 1.1))
    assert_close(dx.get_value(), dx_w * g, atol=1e-4, what="conv dx*act' %s" % (case,))


CONVPOOL_CASES = [
    # N, C, H, K, f, mode, act, ignore_border
    (5, 1, 28, 4, 3, "valid", "relu10", False),
    (3, 4, 13, 20, 3, "valid", "relu05", False),     # 11 -> 6: partial last window
    (3, 4, 13, 20, 3, "valid", "relu05", True),      # 11 -> 5: last row/col in no window
    (2, 3, 16, 9, 3, "same", "relu10", False),
    (2, 2, 12, 5, 5, "valid", "tanh", False),
    (2, 1, 10, 3, 5, "same", "relu", False),
    (70, 2, 9, 6, 3, "same", "sigmoid", False),
    # conv2 of mnist.prms at longer, odd batches; 20 / 28 / 12 filters, a transcendental kind; 11 -> 6: partial last windows
    (65, 4, 13, 20, 3, "valid", "relu05", False),
    (200, 4, 13, 28, 3, "valid", "relu10", False),
    (131, 4, 13, 12, 3, "valid", "tanh", False),
]


@pytest.mark.parametrize("case", CONVPOOL_CASES)
def test_convpool_fused_fwd_bwd(case):
    N, C, H, K, f, mode, act, ib = case
    assert ctx().lib.tn_convpool_supported(C, f, 1, 2)
    rng = np.random.RandomState(N + K)
    x = rng.randn(N, C, H, H).astype(np.float32)
    W = (rng.randn(K, C, f, f) / np.sqrt(C * f * f)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    pad_lo, _, Ho = O.conv_geometry(H, f, 1, mode)
    Hp = O.pool_out_sz(Ho, 2, ib)
    fa, dfa = O.activation(act)
    x64, W64, b64 = x.astype(np.float64), W.astype(np.float64), b.astype(np.float64)
    z = O.conv2d_fwd(x64, W64, b64, 1, mode)
    a = fa(z)
    want_y = O.pool_fwd(a, 2, ib)
    kind, prm = act_code(act)
    xd, Wd, bd = dev(x), dev(W), dev(b)
    y = empty((N, K, Hp, Hp))
    geom = (N, C, H, H, K, f, pad_lo, Ho, Ho, 2, Hp, Hp, kind, prm)
    call("tn_convpool_fwd", xd.ptr, Wd.ptr, bd.ptr, y.ptr, *geom)
    assert_close(y.get_value(), want_y, what="convpool fwd %s" % (case,))
    g = rng.randn(N, K, Hp, Hp).astype(np.float32)
    da = O.pool_bwd(a, g.astype(np.float64), 2, ib)
    dz_w = da * dfa(z)
    dx_w, dW_w, db_w = O.conv2d_bwd(x64, W64, dz_w, 1, mode)
    dz, dW, db = empty((N, K, Ho, Ho)), empty(W.shape), empty((K,))
    dz.fill_bytes(0xff)
    call("tn_convpool_bwd", xd.ptr, Wd.ptr, bd.ptr, dev(g).ptr, dz.ptr, dW.ptr, db.ptr, *geom)
    assert_close(dz.get_value(), dz_w, atol=1e-5, what="convpool dz %s" % (case,))
    assert_close(dW.get_value(), dW_w, atol=2e-4, what="convpool dW %s" % (case,))
    assert_close(db.get_value(), db_w, atol=2e-4, what="convpool db %s" % (case,))
    dW.fill_bytes(0)
    call("tn_convpool_bwd", xd.ptr, Wd.ptr, bd.ptr, dev(g).ptr, None, dW.ptr, db.ptr, *geom)
    assert_close(dW.get_value(), dW_w, atol=2e-4, what="convpool dW (no dz) %s" % (case,))


@pytest.mark.parametrize("case", [
    (9, 4, 13, 20, 3, "valid", "relu05", False),      # mnist.prms conv2 (partial last window)
    (5, 4, 13, 20, 3, "valid", "relu05", True),       # ignore_border: last row/col in no window
    (6, 3, 12, 16, 3, "same", "relu10", False),
    (3, 2, 10, 30, 3, "same", "tanh", False),
    (2, 4, 8, 21, 3, "valid", "relu", False),
])
def test_convblock_lds_backward(case):
    N, C, H, K, f, mode, act, ib = case
    pad_lo, _, Ho = O.conv_geometry(H, f, 1, mode)
    assert ctx().lib.tn_convblock_supported(C, K, f, 1, 2, Ho, Ho) > 0
    rng = np.random.RandomState(N * 7 + K)
    x = rng.randn(N, C, H, H).astype(np.float32)
    W = (rng.randn(K, C, f, f) / np.sqrt(C * f * f)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    Hp = O.pool_out_sz(Ho, 2, ib)
    fa, dfa = O.activation(act)
    x64, W64, b64 = x.astype(np.float64), W.astype(np.float64), b.astype(np.float64)
    z = O.conv2d_fwd(x64, W64, b64, 1, mode)
    g = rng.randn(N, K, Hp, Hp).astype(np.float32)
    dz_w = O.pool_bwd(fa(z), g.astype(np.float64), 2, ib) * dfa(z)
    dx_w, dW_w, db_w = O.conv2d_bwd(x64, W64, dz_w, 1, mode)
    kind, prm = act_code(act)
    dx, dW, db = empty(x.shape), empty(W.shape), empty((K,))
    geom = (N, C, H, H, K, f, pad_lo, Ho, Ho, 2, Hp, Hp, kind, prm)
    call("tn_convblock_bwd", dev(x).ptr, dev(W).ptr, dev(b).ptr, dev(g).ptr, dx.ptr, dW.ptr, db.ptr, *geom)
    assert_close(dx.get_value(), dx_w, atol=2e-5, what="convblock dx %s" % (case,))
    assert_close(dW.get_value(), dW_w, atol=2e-4, what="convblock dW %s" % (case,))
    assert_close(db.get_value(), db_w, atol=2e-4, what="convblock db %s" % (case,))
    dW.fill_bytes(0)
    call("tn_convblock_bwd", dev(x).ptr, dev(W).ptr, dev(b).ptr, dev(g).ptr, None, dW.ptr, db.ptr, *geom)
    assert_close(dW.get_value(), dW_w, atol=2e-4, what="convblock dW (no dx) %s" % (case,))


@pytest.mark.parametrize("case", [
    (9, 4, 13, 20, 3, "valid", "relu05"),       # mnist.prms conv2 (partial last window)
    (70, 4, 13, 20, 3, "valid", "relu05"),      # more images than one block's waves
    (6, 3, 12, 16, 3, "same", "relu10"),
    (3, 2, 10, 30, 3, "same", "tanh"),
    (2, 4, 8, 21, 3, "valid", "relu"),
    (5, 1, 14, 7, 3, "valid", "sigmoid"),
    (4, 1, 12, 5, 3, "valid", "relu10"),        # weight gradient on the 16-block MFMA: one channel, 5 filters (no remainder product)
    (3, 4, 10, 32, 3, "valid", "relu05"),       # ... the largest filter count (8 filter quads + the remainder columns)
])
def test_convblock_mask_backward(case):
    """tn_convpool_fwd_mask + tn_convblock_bwd_mask (matrix-core backward driven by the forward's
    pooling mask) against the oracle's conv -> act -> pool backward."""
    N, C, H, K, f, mode, act = case
    pad_lo, _, Ho = O.conv_geometry(H, f, 1, mode)
    Hp = O.pool_out_sz(Ho, 2, False)
    assert ctx().lib.tn_convblock_mask_supported(C, K, f, 1, 2, H, H, pad_lo, Ho, Ho, Hp, Hp)
    rng = np.random.RandomState(N * 11 + K)
    x = rng.randn(N, C, H, H).astype(np.float32)
    W = (rng.randn(K, C, f, f) / np.sqrt(C * f * f)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    fa, dfa = O.activation(act)
    x64, W64, b64 = x.astype(np.float64), W.astype(np.float64), b.astype(np.float64)
    z = O.conv2d_fwd(x64, W64, b64, 1, mode)
    a = fa(z)
    want_y = O.pool_fwd(a, 2, False)
    g = rng.randn(N, K, Hp, Hp).astype(np.float32)
    dz_w = O.pool_bwd(a, g.astype(np.float64), 2, False) * dfa(z)
    dx_w, dW_w, db_w = O.conv2d_bwd(x64, W64, dz_w, 1, mode)
    kind, prm = act_code(act)
    xd, Wd, bd, gd = dev(x), dev(W), dev(b), dev(g)
    y = empty((N, K, Hp, Hp))
    mask = empty((N, K, Hp, Hp), np.uint8)
    geom = (N, C, H, H, K, f, pad_lo, Ho, Ho, 2, Hp, Hp, kind, prm)
    call("tn_convpool_fwd_mask", xd.ptr, Wd.ptr, bd.ptr, y.ptr, mask.ptr, *geom)
    assert_close(y.get_value(), want_y, what="convpool fwd (mask) %s" % (case,))
    # the mask marks exactly the window elements equal to the pooled value
    m = mask.get_value()
    ap = np.full((N, K, 2 * Hp, 2 * Hp), -np.inf)
    ap[:, :, :Ho, :Ho] = a
    for r in range(4):
        sub = ap[:, :, (r >> 1)::2, (r & 1)::2]
        bit = (m >> r) & 1
        clear = np.abs(sub - want_y) > 1e-5          # away from float ties the bit is determined
        assert np.all(bit[clear] == 0), "mask bit %d set off the maximum" % r
    assert np.all((m & 15) > 0) and np.all(m < 64)
    dx, dW, db = empty(x.shape), empty(W.shape), empty((K,))
    call("tn_convblock_bwd_mask", xd.ptr, Wd.ptr, gd.ptr, y.ptr, mask.ptr, dx.ptr, dW.ptr, db.ptr, *geom)
    assert_close(dx.get_value(), dx_w, atol=2e-5, what="convblock(mask) dx %s" % (case,))
    assert_close(dW.get_value(), dW_w, atol=2e-4, what="convblock(mask) dW %s" % (case,))
    assert_close(db.get_value(), db_w, atol=2e-4, what="convblock(mask) db %s" % (case,))
    dW.fill_bytes(0)
    call("tn_convblock_bwd_mask", xd.ptr, Wd.ptr, gd.ptr, y.ptr, mask.ptr, None, dW.ptr, db.ptr, *geom)
    assert_close(dW.get_value(), dW_w, atol=2e-4, what="convblock(mask) dW (no dx) %s" % (case,))


@pytest.mark.parametrize("case", [
    (3, 16, 16, 32, "relu10"),          # whole 16x16 images per block, 32 filters (one filter tile)
    (2, 32, 32, 64, "tanh"),            # 8-row tiles; activation gradient from the pooled output
    (5, 64, 8, 96, "relu05"),           # 8x8 maps: 4 images per forward tile, 2 per wgrad tile; ragged K
    (2, 20, 64, 40, "relu10"),          # 64-wide rows, ragged channel / filter counts
])
def test_convpool_tile_block(case):
    """Wide conv + act + 2x2 max-pool block on the LDS-tile matrix-core kernels: the forward pools in
    its epilogue (tn_convpool_fwd_mask), the backward forms dz from the pooling mask while staging
    (tn_convpool_bwd_mask_dx), with the activation gradient of the layer below on dx."""
    N, C, H, K, act = case
    Hp = H // 2
    geom_i = (N, C, H, H, K, 3, 1, 1, H, H, 2, Hp, Hp)
    assert ctx().lib.tn_convpool_tile_supported(*geom_i)
    rng = np.random.RandomState(N * 13 + K)
    x = rng.randn(N, C, H, H).astype(np.float32)
    W = (rng.randn(K, C, 3, 3) / np.sqrt(C * 9)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    fa, dfa = O.activation(act)
    x64, W64, b64 = x.astype(np.float64), W.astype(np.float64), b.astype(np.float64)
    z = O.conv2d_fwd(x64, W64, b64, 1, "same")
    a = fa(z)
    want_y = O.pool_fwd(a, 2, False)
    g = rng.randn(N, K, Hp, Hp).astype(np.float32)
    dz_w = O.pool_bwd(a, g.astype(np.float64), 2, False) * dfa(z)
    dx_w, dW_w, db_w = O.conv2d_bwd(x64, W64, dz_w, 1, "same")
    kind, prm = act_code(act)
    xd, Wd, bd, gd = dev(x), dev(W), dev(b), dev(g)
    y, mask = empty((N, K, Hp, Hp)), empty((N, K, Hp, Hp), np.uint8)
    geom = (N, C, H, H, K, 3, 1, H, H, 2, Hp, Hp, kind, prm)
    call("tn_convpool_fwd_mask", xd.ptr, Wd.ptr, bd.ptr, y.ptr, mask.ptr, *geom)
    assert_close(y.get_value(), want_y, what="tile convpool fwd %s" % (case,))
    m = mask.get_value()
    for r in range(4):
        sub = a[:, :, (r >> 1)::2, (r & 1)::2]
        bit = (m >> r) & 1
        clear = np.abs(sub - want_y) > 1e-5
        assert np.all(bit[clear] == 0), "mask bit %d set off the maximum" % r
    assert np.all((m & 15) > 0) and np.all(m < 64)
    # without a mask (test graphs) the pooled output is the same
    y2 = empty((N, K, Hp, Hp))
    call("tn_convpool_fwd_mask", xd.ptr, Wd.ptr, bd.ptr, y2.ptr, None, *geom)
    np.testing.assert_array_equal(y2.get_value(), y.get_value())
    dx, dW, db = empty(x.shape), empty(W.shape), empty((K,))
    call("tn_convpool_bwd_mask_dx", xd.ptr, Wd.ptr, gd.ptr, y.ptr, mask.ptr, dx.ptr, dW.ptr, db.ptr, *geom,
         None, 0, 0.0)
    tolW = max(2e-4, 2e-6 * np.abs(dW_w).max())
    assert_close(dx.get_value(), dx_w, atol=1e-4, what="tile block dx %s" % (case,))
    assert_close(dW.get_value(), dW_w, atol=tolW, what="tile block dW %s" % (case,))
    assert_close(db.get_value(), db_w, atol=max(2e-4, 2e-6 * np.abs(db_w).max()), what="tile block db %s" % (case,))
    # activation gradient of the layer below in the epilogue; weight gradients only
    prev_a = rng.randn(*x.shape).astype(np.float32)
    pk, pp = act_code("relu10")
    call("tn_convpool_bwd_mask_dx", xd.ptr, Wd.ptr, gd.ptr, y.ptr, mask.ptr, dx.ptr, None, None, *geom,
         dev(prev_a).ptr, pk, pp)
    assert_close(dx.get_value(), dx_w * np.where(prev_a > 0, 1.0, .1), atol=1e-4, what="tile block dx*act' %s" % (case,))
    dW.fill_bytes(0)
    call("tn_convpool_bwd_mask_dx", xd.ptr, Wd.ptr, gd.ptr, y.ptr, mask.ptr, None, dW.ptr, db.ptr, *geom,
         None, 0, 0.0)
    assert_close(dW.get_value(), dW_w, atol=tolW, what="tile block dW (no dx) %s" % (case,))


@pytest.mark.parametrize("case", [
    (7, 1, 28, 4, "valid", "relu10", False),        # mnist.prms conv1
    (3, 1, 15, 5, "valid", "relu05", True),         # ignore_border: last row/col in no window
    (4, 3, 16, 6, "same", "tanh", False),
    (2, 4, 11, 9, "same", "relu", False),
    (3, 2, 9, 3, "valid", "sigmoid", False),
    # 'same' blocks with even power-of-two maps and >= 16 filters: without dz (first layer) the weight
    # gradient runs on the matrix core with the (channel, tap) pairs as GEMM columns (conv_tile.hip)
    (5, 3, 32, 32, "same", "relu10", False),        # cifar_like conv1
    (3, 2, 16, 40, "same", "tanh", False),          # two filter groups, the second ragged
    (7, 1, 8, 16, "same", "relu05", False),         # 8x8 maps: two images per tile, odd image count
    (2, 3, 64, 20, "same", "sigmoid", False),
])
def test_convpool_mask_backward(case):
    """tn_convpool_bwd_mask (window-per-thread backward from the pooling mask) vs the oracle."""
    N, C, H, K, mode, act, ib = case
    f = 3
    pad_lo, _, Ho = O.conv_geometry(H, f, 1, mode)
    Hp = O.pool_out_sz(Ho, 2, ib)
    rng = np.random.RandomState(N * 13 + K)
    x = rng.randn(N, C, H, H).astype(np.float32)
    W = (rng.randn(K, C, f, f) / np.sqrt(C * f * f)).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    fa, dfa = O.activation(act)
    x64, W64, b64 = x.astype(np.float64), W.astype(np.float64), b.astype(np.float64)
    z = O.conv2d_fwd(x64, W64, b64, 1, mode)
    a = fa(z)
    g = rng.randn(N, K, Hp, Hp).astype(np.float32)
    dz_w = O.pool_bwd(a, g.astype(np.float64), 2, ib) * dfa(z)
    _, dW_w, db_w = O.conv2d_bwd(x64, W64, dz_w, 1, mode)
    kind, prm = act_code(act)
    xd, Wd, bd, gd = dev(x), dev(W), dev(b), dev(g)
    y = empty((N, K, Hp, Hp))
    mask = empty((N, K, Hp, Hp), np.uint8)
    geom = (N, C, H, H, K, f, pad_lo, Ho, Ho, 2, Hp, Hp, kind, prm)
    call("tn_convpool_fwd_mask", xd.ptr, Wd.ptr, bd.ptr, y.ptr, mask.ptr, *geom)
    assert_close(y.get_value(), O.pool_fwd(a, 2, ib), what="convpool fwd (mask) %s" % (case,))
    yv, m = y.get_value(), mask.get_value()
    assert np.array_equal((m >> 4) & 1, yv > 0) and np.array_equal((m >> 5) & 1, yv < 0)
    dz, dW, db = empty((N, K, Ho, Ho)), empty(W.shape), empty((K,))
    dz.fill_bytes(0xff)
    call("tn_convpool_bwd_mask", xd.ptr, gd.ptr, y.ptr, mask.ptr, dz.ptr, dW.ptr, db.ptr, *geom)
    tolW, tolb = max(2e-4, 2e-6 * np.abs(dW_w).max()), max(2e-4, 2e-6 * np.abs(db_w).max())
    assert_close(dz.get_value(), dz_w, atol=1e-5, what="convpool(mask) dz %s" % (case,))
    assert_close(dW.get_value(), dW_w, atol=tolW, what="convpool(mask) dW %s" % (case,))
    assert_close(db.get_value(), db_w, atol=tolb, what="convpool(mask) db %s" % (case,))
    dW.fill_bytes(0)
    db.fill_bytes(0)
    call("tn_convpool_bwd_mask", xd.ptr, gd.ptr, y.ptr, mask.ptr, None, dW.ptr, db.ptr, *geom)
    assert_close(dW.get_value(), dW_w, atol=tolW, what="convpool(mask) dW (no dz) %s" % (case,))
    assert_close(db.get_value(), db_w, atol=tolb, what="convpool(mask) db (no dz) %s" % (case,))


def test_convpool_tie_rule():
    # constant image, zero weights -> every conv output equals the bias: all four tie
    x = np.ones((1, 1, 6, 6), np.float32)
    W = np.zeros((2, 1, 3, 3), np.float32)
    b = np.array([.5, -1.], np.float32)
    g = np.arange(8, dtype=np.float32).reshape(1, 2, 2, 2) + 1
    kind, prm = act_code("relu10")
    dz, dW, db = empty((1, 2, 4, 4)), empty(W.shape), empty((2,))
    call("tn_convpool_bwd", dev(x).ptr, dev(W).ptr, dev(b).ptr, dev(g).ptr, dz.ptr, dW.ptr, db.ptr,
         1, 1, 6, 6, 2, 3, 0, 4, 4, 2, 2, 2, kind, prm)
    want = np.kron(g[0], np.ones((2, 2), np.float32)) * np.array([1., .1])[:, None, None]
    assert_close(dz.get_value()[0], want, what="tie dz")
    assert_close(db.get_value(), want.sum((1, 2)), what="tie db")


def test_conv_big_batch_wgrad_is_deterministic():
    case = (64, 4, 13, 20, 3, 1, "valid", "relu05")
    N, C, H, K, f, s, mode, act = case
    x, W, b, pad_lo, out = _conv_setup(case, 3)
    dz = np.random.RandomState(4).randn(N, K, out, out).astype(np.float32)
    dW, db = empty(W.shape), empty((K,))
    xd, dzd = dev(x), dev(dz)
    res = []
    for _ in range(2):
        call("tn_conv2d_wgrad", xd.ptr, dzd.ptr, dW.ptr, db.ptr, N, C, H, H, K, f, s, pad_lo, out, out)
        res.append((dW.get_value(), db.get_value()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    _, dW_w, db_w = O.conv2d_bwd(x.astype(np.float64), W.astype(np.float64), dz.astype(np.float64), s,
                                 mode, need_dx=False)
    assert_close(res[0][0], dW_w, atol=2e-4, what="dW")
    assert_close(res[0][1], db_w, atol=2e-4, what="db")


@pytest.mark.parametrize("H,p,ib", [(26, 2, False), (11, 2, False), (11, 2, True), (7, 3, False),
                                    (5, 5, False), (9, 4, True)])
def test_pool_fwd_bwd(H, p, ib):
    rng = np.random.RandomState(H * 10 + p)
    x = rng.randn(3, 5, H, H).astype(np.float32)
    x[0, 0, :2, :2] = 7.0            # a tied window: every tie gets the gradient
    x[1, 1, 0, 0] = 0.0
    Ho = O.pool_out_sz(H, p, ib)
    y = empty((3, 5, Ho, Ho))
    xd = dev(x)
    call("tn_pool_fwd", xd.ptr, y.ptr, 15, H, H, p, Ho, Ho)
    want = O.pool_fwd(x, p, ib)
    np.testing.assert_array_equal(y.get_value(), want)
    dy = rng.randn(3, 5, Ho, Ho).astype(np.float32)
    dx = empty(x.shape)
    call("tn_pool_bwd", xd.ptr, y.ptr, dev(dy).ptr, dx.ptr, 15, H, H, p, Ho, Ho, 0, 0.0)
    np.testing.assert_array_equal(dx.get_value(), O.pool_bwd(x, dy, p, ib))
    kind, prm = act_code("relu05")
    call("tn_pool_bwd", xd.ptr, y.ptr, dev(dy).ptr, dx.ptr, 15, H, H, p, Ho, Ho, kind, prm)
    g = np.where(x > 0, 1.0, np.where(x < 0, .05, 1.05)).astype(np.float32)
    assert_close(dx.get_value(), O.pool_bwd(x, dy, p, ib) * g, what="pool bwd * act'")


def test_mean_fwd_bwd():
    rng = np.random.RandomState(0)
    x = rng.randn(4, 6, 5, 5).astype(np.float32)
    y = empty((4, 6))
    call("tn_mean_fwd", dev(x).ptr, y.ptr, 24, 25)
    assert_close(y.get_value(), O.mean_fwd(x))
    dy = rng.randn(4, 6).astype(np.float32)
    dx = empty(x.shape)
    call("tn_mean_bwd", dev(dy).ptr, dx.ptr, 24, 25, None, 0, 0.0)
    assert_close(dx.get_value(), O.mean_bwd(x, dy))


FC_CASES = [(64, 720, 500, "relu01"), (33, 500, 10, "linear"), (5, 7, 3, "tanh"),
            (128, 100, 64, "sigmoid"), (70, 33, 130, "relu10"), (256, 64, 457, "relu"),
            # few outputs, n_in % 4 == 0: the 16-byte-access kernels of fc_skinny.hip
            (300, 500, 10, "linear"), (129, 64, 16, "tanh"), (17, 8, 1, "relu05"), (515, 1028, 7, "sigmoid"),
            (1024, 128, 96, "relu01"),                   # long batch: split-K weight gradient
            (40, 2304, 100, "relu10"),                   # short and deep: split-K forward + finishing kernel
            # few 64 x 64 tiles, moderate reduction: gemm_f32_deep (waves split K, operands straight from global
            # memory): the 512-image shard of the sharded step, a K tail (500 = 31.25 tiles) with ragged M / N
            (512, 720, 500, "relu01"), (97, 500, 36, "relu10"),
            # many 64 x 64 tiles: the LDS-DMA kernel (gemm_f32_dma) -- ragged M / N (row and column clamps), K tails on
            # k-contiguous (n_out = 500) and row-contiguous (B = 1000) operands, 8 split-K slabs with column-sum blocks
            (2048, 720, 500, "relu01"), (1000, 500, 724, "relu10"), (1100, 96, 260, "tanh"),
            # the headline's own fc1 (BASELINE configs[1]: 4096 rows): the exact instantiations bench.py times
            (4096, 720, 500, "relu01")]


@pytest.mark.parametrize("B,n_in,n_out,act", FC_CASES)
def test_fc_fwd(B, n_in, n_out, act):
    rng = np.random.RandomState(B)
    x = rng.randn(B, n_in).astype(np.float32)
    W = (rng.randn(n_in, n_out) / np.sqrt(n_in)).astype(np.float32)
    b = rng.randn(n_out).astype(np.float32)
    mask = (rng.rand(B, n_out) > .5).astype(np.uint8)
    kind, prm = act_code(act)
    z = x.astype(np.float64) @ W.astype(np.float64) + b
    want = O.activation(act)[0](z)
    a = empty((B, n_out))
    xd, Wd, bd = dev(x), dev(W), dev(b)
    call("tn_fc_fwd", xd.ptr, Wd.ptr, bd.ptr, a.ptr, B, n_in, n_out, kind, prm, None)
    assert_close(a.get_value(), want, what="fc fwd")
    call("tn_fc_fwd", xd.ptr, Wd.ptr, bd.ptr, a.ptr, B, n_in, n_out, kind, prm, dev(mask).ptr)
    assert_close(a.get_value(), want * mask, what="fc fwd masked")


@pytest.mark.parametrize("B,n_in,n_out,act", FC_CASES)
def test_fc_bwd_paired(B, n_in, n_out, act):
    """tn_fc_bwd (weight + input gradient as one op) == tn_fc_wgrad + tn_fc_dgrad."""
    rng = np.random.RandomState(B + 2)
    x = rng.randn(B, n_in).astype(np.float32)
    W = (rng.randn(n_in, n_out) / np.sqrt(n_in)).astype(np.float32)
    dz = rng.randn(B, n_out).astype(np.float32)
    prev_a = rng.randn(B, n_in).astype(np.float32)
    pm = (rng.rand(B, n_in) > .5).astype(np.uint8)
    lib = ctx().lib
    ws = empty((lib.tn_fc_wgrad_ws_bytes(B, n_in, n_out) // 4 + 1,))
    dW, db, dx = empty((n_in, n_out)), empty((n_out,)), empty((B, n_in))
    kind, prm = act_code("relu01")
    call("tn_fc_bwd", dev(x).ptr, dev(dz).ptr, dev(W).ptr, dW.ptr, db.ptr, dx.ptr, B, n_in, n_out, ws.ptr,
         dev(prev_a * pm).ptr, kind, prm, dev(pm).ptr)
    # sums of B products of unit normals: absolute tolerance 1e-4 up to 2048 rows, 1e-6 of the largest entry beyond
    dW_w = x.astype(np.float64).T @ dz.astype(np.float64)
    assert_close(dW.get_value(), dW_w, atol=max(1e-4, 1e-6 * np.abs(dW_w).max()), what="pair dW")
    assert_close(db.get_value(), dz.astype(np.float64).sum(0), atol=1e-4, what="pair db")
    g = np.where(prev_a > 0, 1.0, .01) * pm
    assert_close(dx.get_value(), (dz.astype(np.float64) @ W.astype(np.float64).T) * g, atol=1e-4,
                 what="pair dx * act' * mask")
    call("tn_fc_bwd", dev(x).ptr, dev(dz).ptr, dev(W).ptr, dW.ptr, db.ptr, dx.ptr, B, n_in, n_out, ws.ptr,
         None, 0, 0.0, None)
    assert_close(dx.get_value(), dz.astype(np.float64) @ W.astype(np.float64).T, atol=1e-4, what="pair dx")


@pytest.mark.parametrize("B,n_in,n_out", [(4096, 500, 10), (37, 64, 16), (50, 24, 3), (21, 30, 40), (19, 7, 5)])
def test_fc_softmax_nll_fused(B, n_in, n_out):
    """tn_fc_softmax_nll == tn_fc_fwd (linear) + tn_softmax_nll, against the oracle."""
    rng = np.random.RandomState(B + n_out)
    x = rng.randn(B, n_in).astype(np.float32)
    W = (rng.randn(n_in, n_out) / np.sqrt(n_in)).astype(np.float32)
    b = rng.randn(n_out).astype(np.float32)
    y = rng.randint(0, n_out, size=B + 5).astype(np.int32)
    z = x.astype(np.float64) @ W.astype(np.float64) + b
    lp = O.log_softmax(z)
    lab = y[5:]
    logits, logprob, dz = empty((B, n_out)), empty((B, n_out)), empty((B, n_out))
    rowloss, rowp, pred = empty((B,)), empty((B,)), empty((B,), np.int32)
    call("tn_fc_softmax_nll", dev(x).ptr, dev(W).ptr, dev(b).ptr, logits.ptr, B, n_in, n_out,
         dev(y).ptr, 5, None, logprob.ptr, rowloss.ptr, pred.ptr, rowp.ptr, dz.ptr, 1.0 / B)
    assert_close(logits.get_value(), z, what="fused logits")
    assert_close(logprob.get_value(), lp, what="fused logprob")
    assert_close(rowloss.get_value(), -lp[np.arange(B), lab], what="fused rowloss")
    assert_close(rowp.get_value(), np.exp(lp[np.arange(B), lab]), what="fused P(label)")
    onehot = np.zeros((B, n_out)); onehot[np.arange(B), lab] = 1
    assert_close(dz.get_value(), (np.exp(lp) - onehot) / B, atol=1e-7, what="fused dlogits")
    got = pred.get_value()
    best = z.max(1)
    assert np.all(np.abs(z[np.arange(B), got] - best) <= 1e-5 * np.maximum(1, np.abs(best)))
    # labels optional (test graphs)
    call("tn_fc_softmax_nll", dev(x).ptr, dev(W).ptr, dev(b).ptr, logits.ptr, B, n_in, n_out,
         None, 0, None, logprob.ptr, None, pred.ptr, None, None, 1.0 / B)
    assert_close(logprob.get_value(), lp, what="fused logprob (no labels)")


@pytest.mark.parametrize("B,n_in,n_out", [(4096, 500, 10), (45, 64, 16), (70, 24, 3), (21, 30, 40)])
def test_fc_softmax_train_fused(B, n_in, n_out):
    """tn_fc_softmax_train: logits, log-softmax/NLL, input gradient and weight gradient of the softmax
    layer as one op, against the float64 oracle."""
    rng = np.random.RandomState(B + 3 * n_out)
    x = rng.randn(B, n_in).astype(np.float32)
    pm = (rng.rand(B, n_in) > .5).astype(np.uint8)
    x = x * pm                                       # output of a dropout layer: act(z) * mask
    W = (rng.randn(n_in, n_out) / np.sqrt(n_in)).astype(np.float32)
    b = rng.randn(n_out).astype(np.float32)
    y = rng.randint(0, n_out, size=B).astype(np.int32)
    z = x.astype(np.float64) @ W.astype(np.float64) + b
    lp = O.log_softmax(z)
    onehot = np.zeros((B, n_out)); onehot[np.arange(B), y] = 1
    dz_w = (np.exp(lp) - onehot) / B
    lib = ctx().lib
    ws = empty((lib.tn_fc_wgrad_ws_bytes(B, n_in, n_out) // 4 + 1,))
    logits, logprob, dz = empty((B, n_out)), empty((B, n_out)), empty((B, n_out))
    rowloss, rowp, pred = empty((B,)), empty((B,)), empty((B,), np.int32)
    dW, db, dx = empty((n_in, n_out)), empty((n_out,)), empty((B, n_in))
    kind, prm = act_code("relu01")
    xd = dev(x)
    call("tn_fc_softmax_train", xd.ptr, dev(W).ptr, dev(b).ptr, logits.ptr, B, n_in, n_out, dev(y).ptr, 0,
         None, logprob.ptr, rowloss.ptr, pred.ptr, rowp.ptr, dz.ptr, 1.0 / B, dW.ptr, db.ptr, dx.ptr,
         ws.ptr, xd.ptr, kind, prm, dev(pm).ptr)
    assert_close(logprob.get_value(), lp, what="train logprob")
    assert_close(rowloss.get_value(), -lp[np.arange(B), y], what="train rowloss")
    assert_close(dz.get_value(), dz_w, atol=1e-7, what="train dlogits")
    assert_close(dW.get_value(), x.astype(np.float64).T @ dz_w, atol=1e-5, what="train dW")
    assert_close(db.get_value(), dz_w.sum(0), atol=1e-5, what="train db")
    g = np.where(x > 0, 1.0, np.where(x < 0, .01, 1.01)) * pm
    assert_close(dx.get_value(), (dz_w @ W.astype(np.float64).T) * g, atol=1e-6, what="train dx")


@pytest.mark.parametrize("B,n_in,n_out", [(64, 720, 500), (70, 36, 132), (33, 50, 10), (40, 64, 101),
                                          (48, 4096, 128),        # split-K forward
                                          (512, 720, 500), (97, 500, 36)])   # gemm_f32_deep epilogue
def test_fc_fwd_dropout_matches_separate_mask(B, n_in, n_out):
    """tn_fc_fwd_dropout draws the mask inside the GEMM epilogue (or falls back to two launches):
    the mask must be bit-identical to tn_dropout_mask and the output the masked activation."""
    rng = np.random.RandomState(B + n_out)
    x = rng.randn(B, n_in).astype(np.float32)
    W = (rng.randn(n_in, n_out) / np.sqrt(n_in)).astype(np.float32)
    b = rng.randn(n_out).astype(np.float32)
    kind, prm = act_code("relu01")
    seed, step, elem0, p = 0xabcdef12345, 3, 4 * 25 * n_out, 0.5
    want_mask = empty((B, n_out), np.uint8)
    call("tn_dropout_mask", want_mask.ptr, B * n_out, p, seed, step, None, elem0)
    a, mask = empty((B, n_out)), empty((B, n_out), np.uint8)
    mask.fill_bytes(7)
    call("tn_fc_fwd_dropout", dev(x).ptr, dev(W).ptr, dev(b).ptr, a.ptr, B, n_in, n_out, kind, prm,
         mask.ptr, p, seed, step, None, elem0)
    m = want_mask.get_value()
    assert np.array_equal(mask.get_value(), m)
    assert 0.3 < m.mean() < 0.7
    z = x.astype(np.float64) @ W.astype(np.float64) + b
    assert_close(a.get_value(), O.activation("relu01")[0](z) * m, what="fc fwd + inline dropout")


@pytest.mark.parametrize("B,n_in,n_out,act", FC_CASES)
def test_fc_bwd(B, n_in, n_out, act):
    rng = np.random.RandomState(B + 1)
    x = rng.randn(B, n_in).astype(np.float32)
    W = (rng.randn(n_in, n_out) / np.sqrt(n_in)).astype(np.float32)
    dz = rng.randn(B, n_out).astype(np.float32)
    lib = ctx().lib
    ws = empty((lib.tn_fc_wgrad_ws_bytes(B, n_in, n_out) // 4 + 1,))
    dW, db, dx = empty((n_in, n_out)), empty((n_out,)), empty((B, n_in))
    xd, Wd, dzd = dev(x), dev(W), dev(dz)
    call("tn_fc_wgrad", xd.ptr, dzd.ptr, dW.ptr, db.ptr, B, n_in, n_out, ws.ptr)
    dW_w = x.astype(np.float64).T @ dz.astype(np.float64)
    assert_close(dW.get_value(), dW_w, atol=max(1e-4, 1e-6 * np.abs(dW_w).max()), what="fc dW")
    assert_close(db.get_value(), dz.astype(np.float64).sum(0), atol=1e-4, what="fc db")
    call("tn_fc_dgrad", dzd.ptr, Wd.ptr, dx.ptr, B, n_in, n_out, None, 0, 0.0, None)
    want = dz.astype(np.float64) @ W.astype(np.float64).T
    assert_close(dx.get_value(), want, atol=1e-4, what="fc dx")
    prev_a = rng.randn(B, n_in).astype(np.float32)
    pm = (rng.rand(B, n_in) > .5).astype(np.uint8)
    kind, prm = act_code("relu01")
    call("tn_fc_dgrad", dzd.ptr, Wd.ptr, dx.ptr, B, n_in, n_out, dev(prev_a * pm).ptr, kind, prm,
         dev(pm).ptr)
    g = np.where(prev_a > 0, 1.0, .01) * pm
    assert_close(dx.get_value(), want * g, atol=1e-4, what="fc dx * act' * mask")


def test_fc_wgrad_large_batch_split_k():
    B, n_in, n_out = 4096, 720, 500
    rng = np.random.RandomState(9)
    x = rng.rand(B, n_in).astype(np.float32)
    dz = (rng.randn(B, n_out) / B).astype(np.float32)
    ws = empty((ctx().lib.tn_fc_wgrad_ws_bytes(B, n_in, n_out) // 4 + 1,))
    dW, db = empty((n_in, n_out)), empty((n_out,))
    call("tn_fc_wgrad", dev(x).ptr, dev(dz).ptr, dW.ptr, db.ptr, B, n_in, n_out, ws.ptr)
    assert_close(dW.get_value(), x.astype(np.float64).T @ dz.astype(np.float64), atol=1e-5, what="dW")
    assert_close(db.get_value(), dz.astype(np.float64).sum(0), atol=1e-5, what="db")


@pytest.mark.parametrize("B,n", [(8, 10), (37, 457), (4096, 10), (5, 1)])
def test_softmax_nll(B, n):
    rng = np.random.RandomState(n)
    z = (3 * rng.randn(B, n)).astype(np.float32)
    z[0, :] = 1.5                                  # all tied: argmax must be the first index
    y = rng.randint(0, n, B + 3).astype(np.int32)
    lp, rl, pred, rp, dz = empty((B, n)), empty((B,)), empty((B,), np.int32), empty((B,)), empty((B, n))
    call("tn_softmax_nll", dev(z).ptr, dev(y).ptr, 3, None, lp.ptr, rl.ptr, pred.ptr, rp.ptr, dz.ptr,
         B, n, 1.0 / B)
    yy = y[3:]
    want = O.log_softmax(z.astype(np.float64))
    assert_close(lp.get_value(), want, what="logprob")
    np.testing.assert_array_equal(pred.get_value(), z.argmax(1))
    assert_close(rl.get_value(), -want[np.arange(B), yy], what="rowloss")
    assert_close(rp.get_value(), np.exp(want[np.arange(B), yy]), what="rowp")
    assert_close(dz.get_value(), O.nll_dlogits(want, yy), atol=1e-7, what="dlogits")
    cost = empty((1,))
    call("tn_reduce_sum", rl.ptr, B, 1.0 / B, cost.ptr, 0)
    assert_close(cost.get_value()[0], O.nll(want, yy), what="cost")
    st = empty((2,))
    call("tn_error_stats", pred.ptr, dev(y).ptr, 3, rp.ptr, B, st.ptr)
    assert_close(st.get_value(), [np.mean(z.argmax(1) != yy), np.exp(want[np.arange(B), yy]).mean()])


def test_sgd_update_and_maxnorm():
    rng = np.random.RandomState(0)
    for shape in [(500,), (720, 500), (20, 4, 3, 3)]:
        p = rng.randn(*shape).astype(np.float32)
        v = rng.randn(*shape).astype(np.float32)
        g = rng.randn(*shape).astype(np.float32)
        for reg in [dict(O.DEFAULT_REG), dict(O.DEFAULT_REG, L1=.01, L2=.02, momentum=.9, rate=.5),
                    dict(O.DEFAULT_REG, maxnorm=1.5)]:
            pd, vd, lr = dev(p), dev(v), dev(np.array([.1], np.float32))
            call("tn_sgd_update", pd.ptr, vd.ptr, dev(g).ptr, p.size, reg["momentum"], reg["rate"],
                 lr.ptr, reg["L1"], reg["L2"], 1.0)
            if reg["maxnorm"]:
                call("tn_maxnorm", pd.ptr, p.ndim, shape[0],
                     int(np.prod(shape[1:])) if p.ndim > 1 else 1, reg["maxnorm"])
            gg = g + O.wtcost_grad(p, reg)
            p_w, v_w = O.sgd_update(p, v, gg, .1, reg)
            assert_close(pd.get_value(), p_w, atol=1e-6, what="p %s %s" % (shape, reg))
            assert_close(vd.get_value(), v_w, atol=1e-6, what="v %s %s" % (shape, reg))
    cost = dev(np.array([2.0], np.float32))
    p = rng.randn(1000).astype(np.float32)
    call("tn_wtcost", dev(p).ptr, 1000, .01, .02, cost.ptr, 1)
    assert_close(cost.get_value()[0], 2 + .01 * np.abs(p).sum() + .02 * (p * p).sum())


def test_maxnorm_leaves_tensors_within_the_bound_untouched():
    """layer.py:88-103 with columns / kernels on both sides of the bound: those within it have a factor of exactly
    (1e-7 + n) / (1e-7 + n) = 1 and must come back bit for bit (the kernels skip them), the others are rescaled."""
    rng = np.random.RandomState(3)
    W = rng.randn(300, 130).astype(np.float32)
    W[:, ::2] *= 0.01                                   # every other column far inside the bound
    Wd = dev(W)
    call("tn_maxnorm", Wd.ptr, 2, 300, 130, 2.0)
    got = Wd.get_value()
    assert np.array_equal(got[:, ::2], W[:, ::2])
    nrm = np.sqrt((W.astype(np.float64) ** 2).sum(0))
    want = W * ((1e-7 + np.clip(nrm, 0, 2.0)) / (1e-7 + nrm))
    assert_close(got, want, atol=1e-6, what="2-D maxnorm")
    K = rng.randn(12, 3, 3, 3).astype(np.float32)
    K[::3] *= 0.01
    Kd = dev(K)
    call("tn_maxnorm", Kd.ptr, 4, 12, 27, 1.0)
    got = Kd.get_value()
    assert np.array_equal(got[::3], K[::3])
    nrm = np.sqrt((K.astype(np.float64) ** 2).sum((1, 2, 3)))
    assert_close(got, K * ((1e-7 + np.clip(nrm, 0, 1.0)) / (1e-7 + nrm))[:, None, None, None], atol=1e-6,
                 what="4-D maxnorm")


def test_maxnorm_multi_equals_per_tensor_calls():
    """tn_maxnorm_multi (every tensor of a net in one call: the 1-D / 4-D ones share a launch) gives the same bits as
    tn_maxnorm per tensor; maxnorm 0 and NULL entries are skipped."""
    rng = np.random.RandomState(5)
    shapes = [(20, 4, 3, 3), (20,), (300, 130), (130,), (64, 32, 3, 3), (64,), (7, 1, 5, 5), (700,)]
    mxs = [1.0, 0.5, 2.0, 0.3, 3.0, 0.0, 0.8, 0.9]
    hosts = [rng.randn(*sh).astype(np.float32) for sh in shapes]
    a, b = [dev(h) for h in hosts], [dev(h) for h in hosts]
    dt = np.dtype([('p', 'u8'), ('ndim', 'i4'), ('d0', 'i4'), ('rest', 'i4'), ('mx', 'f4')])
    tab = np.array([(d.ptr, len(sh), sh[0], 1 if len(sh) == 1 else int(np.prod(sh[1:])), mx)
                    for d, sh, mx in zip(a, shapes, mxs)], dtype=dt)
    call("tn_maxnorm_multi", tab.ctypes.data, len(tab))
    for d, sh, mx in zip(b, shapes, mxs):
        if mx:
            call("tn_maxnorm", d.ptr, len(sh), sh[0], 1 if len(sh) == 1 else int(np.prod(sh[1:])), mx)
    for x, y, h, mx in zip(a, b, hosts, mxs):
        assert np.array_equal(x.get_value(), y.get_value())
        assert mx or np.array_equal(x.get_value(), h)
        assert not mx or not np.array_equal(x.get_value(), h)


@pytest.mark.parametrize("mode", ["lazy", "pipe", "pipe_first_steps", "plain"])
def test_update_with_column_norm_rider_equals_update_then_maxnorm(mode):
    """tn_sgd_update_net_maxnorm (layer.py:82-103 as one call; the update launch leaves the dense matrices' column sums
    of squares) against tn_sgd_update_net + tn_maxnorm_multi: the same bits in weights and velocities, for tensors the
    tile walk takes (columns % 4 == 0) and tensors it does not, with columns on both sides of the bound."""
    from theanet_amd import _lib
    rng = np.random.RandomState(11)
    shapes = [(300, 132), (132,), (12, 3, 3, 3), (64, 130), (1030, 512), (512,)]
    mxs = [2.0, 0.5, 1.0, 1.5, 3.0, 0.0]
    P = [rng.randn(*sh).astype(np.float32) * (.1 if len(sh) == 2 else 1) for sh in shapes]
    for w in P:
        if w.ndim == 2:
            w[:, ::3] *= 40.0                            # every third column far outside the bound
    V = [rng.randn(*sh).astype(np.float32) * .1 for sh in shapes]
    G = [rng.randn(*sh).astype(np.float32) for sh in shapes]
    lr = dev(np.array([.1], np.float32))
    res = []
    for fused in (False, True):
        p, v, g = [dev(x) for x in P], [dev(x) for x in V], [dev(x) for x in G]
        p2 = [empty(sh) for sh in shapes]                # TN_UPD_PIPE: the stepping stream's own copy
        if mode.startswith("pipe"):
            dt = np.dtype([('p', 'u8'), ('psrc', 'u8'), ('v', 'u8'), ('g', 'u8'), ('n', 'u8'), ('m', 'f4'), ('rate', 'f4')])
            segs = np.array([(b.ptr, a.ptr, c.ptr, d.ptr, h.size, .9, .5) for a, b, c, d, h in zip(p, p2, v, g, P)], dtype=dt)
            out, m, flags = p2, _lib.TN_UPD_PIPE, 0 if mode == "pipe_first_steps" else 1
        else:
            dt = np.dtype([('p', 'u8'), ('v', 'u8'), ('g', 'u8'), ('n', 'u8'), ('m', 'f4'), ('rate', 'f4'), ('L1', 'f4'),
                           ('L2', 'f4')])
            segs = np.array([(a.ptr, c.ptr, d.ptr, h.size, .9, .5, .001, .002) for a, c, d, h in zip(p, v, g, P)], dtype=dt)
            out, m, flags = p, _lib.TN_UPD_LAZY if mode == "lazy" else _lib.TN_UPD_PLAIN, 0
        assert dt.itemsize == 48
        dsegs = dev(segs.view(np.uint8))
        mdt = np.dtype([('p', 'u8'), ('ndim', 'i4'), ('d0', 'i4'), ('rest', 'i4'), ('mx', 'f4')])
        tab = np.array([(d.ptr, len(sh), sh[0], 1 if len(sh) == 1 else int(np.prod(sh[1:])), mx)
                        for d, sh, mx in zip(out, shapes, mxs)], dtype=mdt)
        args = (m, dsegs.ptr, segs.ctypes.data, len(segs), max(h.size for h in P), lr.ptr, 1.0, None, 0, flags, None, 0, 0.0,
                None)
        if fused:
            call("tn_sgd_update_net_maxnorm", *args, tab.ctypes.data, len(tab))
        else:
            call("tn_sgd_update_net", *args)
            call("tn_maxnorm_multi", tab.ctypes.data, len(tab))
        res.append(([x.get_value() for x in out], [x.get_value() for x in v]))
    for a, b, w, mx in zip(res[0][0], res[1][0], P, mxs):
        assert np.array_equal(a, b)
        if w.ndim == 2:                                  # (and the projection happened)
            assert np.sqrt((a.astype(np.float64) ** 2).sum(0)).max() < mx * (1 + 1e-5)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", ["lazy", "pipe"])
def test_update_column_norm_rider_sums_pending_slabs(mode):
    """The tile walk of tn_sgd_update_net_maxnorm on a tensor whose gradient is still a stack of deferred split-K slabs
    (tn_defer_reductions window + tn_fc_wgrad, 8 slabs of the 720 x 500 product): slab sum, update, column sums and
    projection in the update's launches against the unfused calls -- gradient, velocity and weights bit for bit, the
    gradient also against the float64 product."""
    from theanet_amd import _lib
    B, n_in, n_out = 4096, 720, 500
    rng = np.random.RandomState(13)
    x = rng.rand(B, n_in).astype(np.float32)
    dz = (rng.randn(B, n_out) / B).astype(np.float32)
    P = [rng.randn(n_in, n_out).astype(np.float32) * .05, rng.randn(n_out).astype(np.float32)]
    P[0][:, ::3] *= 40.0
    V = [rng.randn(*w.shape).astype(np.float32) * .1 for w in P]
    xd, dzd, lr = dev(x), dev(dz), dev(np.array([.1], np.float32))
    ws = empty((ctx().lib.tn_fc_wgrad_ws_bytes(B, n_in, n_out) // 4 + 1,))
    res = []
    for fused in (False, True):
        p, v, g = [dev(w) for w in P], [dev(w) for w in V], [empty(w.shape) for w in P]
        p2 = [empty(w.shape) for w in P]
        if mode == "pipe":
            dt = np.dtype([('p', 'u8'), ('psrc', 'u8'), ('v', 'u8'), ('g', 'u8'), ('n', 'u8'), ('m', 'f4'), ('rate', 'f4')])
            segs = np.array([(b.ptr, a.ptr, c.ptr, d.ptr, h.size, .9, .5) for a, b, c, d, h in zip(p, p2, v, g, P)], dtype=dt)
            out, m, flags = p2, _lib.TN_UPD_PIPE, 1
        else:
            dt = np.dtype([('p', 'u8'), ('v', 'u8'), ('g', 'u8'), ('n', 'u8'), ('m', 'f4'), ('rate', 'f4'), ('L1', 'f4'),
                           ('L2', 'f4')])
            segs = np.array([(a.ptr, c.ptr, d.ptr, h.size, .9, .5, 0., .002) for a, c, d, h in zip(p, v, g, P)], dtype=dt)
            out, m, flags = p, _lib.TN_UPD_LAZY, 0
        dsegs = dev(segs.view(np.uint8))
        mdt = np.dtype([('p', 'u8'), ('ndim', 'i4'), ('d0', 'i4'), ('rest', 'i4'), ('mx', 'f4')])
        tab = np.array([(out[0].ptr, 2, n_in, n_out, 2.0), (out[1].ptr, 1, n_out, 1, .5)], dtype=mdt)
        call("tn_defer_reductions", 1)
        call("tn_fc_wgrad", xd.ptr, dzd.ptr, g[0].ptr, g[1].ptr, B, n_in, n_out, ws.ptr)
        args = (m, dsegs.ptr, segs.ctypes.data, 2, P[0].size, lr.ptr, 1.0, None, 0, flags, None, 0, 0.0, None)
        if fused:
            call("tn_sgd_update_net_maxnorm", *args, tab.ctypes.data, 2)
        else:
            call("tn_sgd_update_net", *args)
            call("tn_maxnorm_multi", tab.ctypes.data, 2)
        call("tn_defer_reductions", 0)
        res.append([a.get_value() for a in out] + [a.get_value() for a in v] + [a.get_value() for a in g])
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    assert_close(res[1][4], x.astype(np.float64).T @ dz.astype(np.float64), atol=1e-5, what="dW from the slab stack")
    assert np.sqrt((res[1][0].astype(np.float64) ** 2).sum(0)).max() < 2.0 * (1 + 1e-5)
    assert not np.array_equal(res[1][0], P[0])


def test_dropout_mask_statistics_and_sharding_invariance():
    n = 4096 * 500
    m = empty((n,), np.uint8)
    call("tn_dropout_mask", m.ptr, n, .5, 1234, 7, None, 0)
    full = m.get_value()
    assert abs(full.mean() - .5) < 2e-3
    # a shard that starts at element 1001 must reproduce the same bits
    call("tn_dropout_mask", m.ptr, 5000, .5, 1234, 7, None, 1001)
    np.testing.assert_array_equal(m.get_value()[:5000], full[1001:6001])
    # different step / seed -> different mask; device-side step is added to the value one
    step = dev(np.array([3], np.uint32))
    call("tn_dropout_mask", m.ptr, n, .5, 1234, 4, step.ptr, 0)
    np.testing.assert_array_equal(m.get_value(), full)
    call("tn_dropout_mask", m.ptr, n, .5, 1234, 8, None, 0)
    assert (m.get_value() != full).mean() > .4
    call("tn_dropout_mask", m.ptr, n, .3, 99, 0, None, 0)
    assert abs(m.get_value().mean() - .7) < 2e-3
    # rows/columns are not correlated
    mm = m.get_value().reshape(4096, 500).astype(np.float64)
    assert abs(np.corrcoef(mm[:, 0], mm[:, 1])[0, 1]) < .06


def test_scale_mask_gather_axpby():
    rng = np.random.RandomState(1)
    x = rng.randn(1000).astype(np.float32)
    m = (rng.rand(1000) > .5).astype(np.uint8)
    y = empty((1000,))
    call("tn_scale_mask", dev(x).ptr, dev(m).ptr, .5, y.ptr, 1000, None, 0, 0.0)
    assert_close(y.get_value(), x * .5 * m)
    src = rng.randn(50, 12).astype(np.float32)
    idx = rng.randint(0, 50, 20).astype(np.int32)
    dst = empty((20, 12))
    call("tn_gather_rows", dev(src).ptr, dev(idx).ptr, dst.ptr, 20, 48)
    np.testing.assert_array_equal(dst.get_value(), src[idx])
    yd = dev(x)
    call("tn_axpby", yd.ptr, dev(x[::-1].copy()).ptr, 1000, 2.0, -1.0)
    assert_close(yd.get_value(), 2 * x[::-1] - x)


def test_activation_gradient_at_an_exact_zero():
    """KAT of the activation derivative taken from the stored OUTPUT (documented deviations):
    leaky slopes s > 0 give Theano's tie value 1 + s at an exact 0 (grad of max(0,z) + min(0,z)*s, KAT-3 of the
    oracle); for slope 0 an output of exactly 0 is read as z < 0: gradient 0 where Theano gives 1 at z == 0 --
    the one documented, measure-zero deviation."""
    a = np.array([-.2, 0., 3.], np.float32)         # layer outputs
    g = np.ones(3, np.float32)
    for name, want in (("relu10", [.1, 1.1, 1.]), ("relu05", [.05, 1.05, 1.]), ("relu", [0., 0., 1.])):
        kind, prm = act_code(name)
        out = empty((3,))
        call("tn_scale_mask", dev(g).ptr, None, 1.0, out.ptr, 3, dev(a if name != "relu" else np.abs(a) * (a > 0)).ptr,
             kind, prm)
        np.testing.assert_allclose(out.get_value(), want, rtol=1e-6, err_msg=name)
