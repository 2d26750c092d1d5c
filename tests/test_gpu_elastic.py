"""Elastic / deformer stage parity against the oracle with INJECTED random draws, plus
statistical checks of the on-device generator."""
import os

import numpy as np
import pytest

from oracle import theanet_oracle as O
from tests.gpu_util import assert_close, call, ctx, dev, empty

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pack(d, h, w):
    v = np.zeros(8 + 2 * h * w, np.float32)
    if d.transln is not None:
        v[0:2] = d.transln.reshape(2)
    if d.origin_u is not None:
        v[2:4] = d.origin_u.reshape(2)
    if d.zoom_u is not None:
        v[4:6] = d.zoom_u.reshape(2)
    if d.theta_u is not None:
        v[6] = d.theta_u
    if d.noise is not None:
        v[8:] = d.noise.reshape(-1)
    return v


CASES = [
    (dict(translation=2, zoom=1.1, magnitude=60, sigma=15, pflip=.03, angle=5), True, 28, 1),
    (dict(translation=2, zoom=1.1, magnitude=30, sigma=4, pflip=0, angle=5), False, 32, 3),
    (dict(translation=3, zoom=1, magnitude=0, sigma=1, pflip=.1, angle=0), False, 17, 2),
    (dict(translation=0, zoom=1.3, magnitude=0, sigma=1, pflip=0, angle=0), True, 20, 1),
    (dict(translation=0, zoom=1, magnitude=12, sigma=3, pflip=0, angle=30), False, 24, 1),
]


@pytest.mark.parametrize("prm,nearest,hw,C", CASES)
def test_elastic_field_and_apply_match_oracle(prm, nearest, hw, C):
    st = O.ElasticStage(hw, num_maps=C, nearest=nearest, invert_image=True,
                        rand_gen=np.random.RandomState(5), **prm)
    rng = np.random.RandomState(0)
    x = rng.rand(10, C, hw, hw).astype(np.float32)
    d = st.draw((6, C, hw, hw))
    want, target = st.forward(x[2:8], d)
    draws = dev(_pack(d, hw, hw))
    idx, fy, fx = empty((hw * hw,), np.int32), empty((hw * hw,)), empty((hw * hw,))
    tgt = empty((2, hw, hw), np.float64)
    call("tn_elastic_field", draws.ptr, hw, hw, float(prm["translation"]), float(prm["zoom"]),
         float(prm["magnitude"]), prm["sigma"], float(prm["angle"]), int(nearest), idx.ptr, fy.ptr,
         fx.ptr, tgt.ptr)
    got_t = tgt.get_value()
    # the smoothed-noise plane is a float32 sum of up to 961 terms in a different order
    np.testing.assert_allclose(got_t, target, rtol=0, atol=2e-4)
    out = empty((6, C, hw, hw))
    fm = dev(d.flipmask.astype(np.uint8)) if prm["pflip"] else None
    row0 = dev(np.array([1], np.int64))
    call("tn_elastic_apply", dev(x).ptr, 1, row0.ptr, out.ptr, 6, C, hw, hw, 1, int(nearest),
         idx.ptr, fy.ptr, fx.ptr, 0.0, fm.ptr if fm is not None else None, 0, 0, None, 0)
    got = out.get_value()
    if nearest:
        # identical except where the coordinate sits within 2e-4 of a rounding boundary
        cy = np.clip(target[0], 0, hw - 1.001)
        cx = np.clip(target[1], 0, hw - 1.001)
        safe = (np.abs(cy - np.floor(cy) - .5) > 5e-4) & (np.abs(cx - np.floor(cx) - .5) > 5e-4)
        assert safe.mean() > .99, "%.3f %% of the pixels excluded (within 5e-4 of a rounding boundary)" % (
            100 * (1 - safe.mean()))
        np.testing.assert_array_equal(got[:, :, safe], want[:, :, safe],
                                      err_msg="compared %.2f %% of the pixels" % (100 * safe.mean()))
    else:
        assert_close(got, want, atol=2e-3, rtol=0, what="bilinear")
        assert np.abs(got - want).mean() < 2e-5


@pytest.mark.parametrize("prm,nearest,hw,C", [c for c in CASES if c[2] % 4 == 0])      # (the c8 stage takes rows of 16 bytes)
def test_c8_elastic_apply_matches_oracle(prm, nearest, hw, C):
    """tn_c8_elastic_apply (DTYPE float16: the stage writes the first conv layer's fp16 tensor directly) against the
    ORACLE's distortion stage with the same injected draws: fp16(oracle value) exactly for nearest-neighbour sampling
    (away from rounding boundaries of the coordinates), within the bilinear tolerance + half an fp16 ulp otherwise;
    channels beyond C are zero."""
    st = O.ElasticStage(hw, num_maps=C, nearest=nearest, invert_image=True,
                        rand_gen=np.random.RandomState(5), **prm)
    rng = np.random.RandomState(0)
    x = rng.rand(10, C, hw, hw).astype(np.float32)
    d = st.draw((6, C, hw, hw))
    want, target = st.forward(x[2:8], d)
    draws = dev(_pack(d, hw, hw))
    idx, fy, fx = empty((hw * hw,), np.int32), empty((hw * hw,)), empty((hw * hw,))
    call("tn_elastic_field", draws.ptr, hw, hw, float(prm["translation"]), float(prm["zoom"]),
         float(prm["magnitude"]), prm["sigma"], float(prm["angle"]), int(nearest), idx.ptr, fy.ptr, fx.ptr, None)
    fm = dev(d.flipmask.astype(np.uint8)) if prm["pflip"] else None
    row0 = dev(np.array([1], np.int64))
    C8 = (C + 7) // 8
    out16 = empty((6, C8, hw * hw, 8), np.uint16)
    call("tn_c8_elastic_apply", dev(x).ptr, 1, row0.ptr, out16.ptr, 6, C, hw, hw, 1, int(nearest),
         idx.ptr, fy.ptr, fx.ptr, 0.0, fm.ptr if fm is not None else None, 0, 0, None, 0)
    got = out16.get_value().view(np.float16).reshape(6, C8, hw * hw, 8).transpose(0, 1, 3, 2).reshape(6, C8 * 8, hw, hw)
    assert not got[:, C:].any()
    got = got[:, :C].astype(np.float32)
    if nearest:
        cy = np.clip(target[0], 0, hw - 1.001)
        cx = np.clip(target[1], 0, hw - 1.001)
        safe = (np.abs(cy - np.floor(cy) - .5) > 5e-4) & (np.abs(cx - np.floor(cx) - .5) > 5e-4)
        assert safe.mean() > .99
        np.testing.assert_array_equal(got[:, :, safe], want.astype(np.float16).astype(np.float32)[:, :, safe])
    else:
        assert_close(got, want, atol=2e-3 + 5e-4, rtol=0, what="bilinear, fp16")


def test_elastic_identity_and_invert():
    x = np.random.RandomState(0).rand(4, 2, 9, 9).astype(np.float32)
    out = empty((3, 2, 9, 9))
    call("tn_elastic_apply", dev(x).ptr, 1, None, out.ptr, 3, 2, 9, 9, 1, 1, None, None, None, 0.0,
         None, 0, 0, None, 0)
    np.testing.assert_array_equal(out.get_value(), np.float32(1) - x[1:4])
    call("tn_elastic_apply", dev(x).ptr, 0, None, out.ptr, 3, 2, 9, 9, 0, 0, None, None, None, 0.0,
         None, 0, 0, None, 0)
    np.testing.assert_array_equal(out.get_value(), x[:3])


def test_device_rng_statistics_and_flip_noise():
    h = w = 28
    n = ctx().lib.tn_elastic_draws_count(h, w)
    assert n == 8 + 2 * h * w
    d = empty((n,))
    vals = []
    for step in range(40):
        call("tn_elastic_draws", d.ptr, h, w, 4242, step, None)
        vals.append(d.get_value())
    v = np.stack(vals)
    assert np.all(np.abs(v[:, 0:2]) <= 1) and np.all(np.abs(v[:, 4:7]) <= 1)
    assert np.all((v[:, 2:4] >= .25) & (v[:, 2:4] <= .75))
    noise = v[:, 8:].ravel()
    assert abs(noise.mean()) < .02 and abs(noise.std() - 1) < .02
    assert abs((np.abs(noise) < 1).mean() - .6827) < .01
    assert not np.array_equal(v[0], v[1])
    call("tn_elastic_draws", d.ptr, h, w, 4242, 0, None)
    np.testing.assert_array_equal(d.get_value(), v[0])          # counter RNG: reproducible
    # flip noise: P(flip) = pflip, independent of sharding (row_global0)
    x = np.zeros((64, 1, 28, 28), np.float32)
    out = empty(x.shape)
    call("tn_elastic_apply", dev(x).ptr, 0, None, out.ptr, 64, 1, 28, 28, 0, 1, None, None, None,
         .03, None, 77, 5, None, 0)
    full = out.get_value()
    assert abs(full.mean() - .03) < .003
    out2 = empty((16, 1, 28, 28))
    call("tn_elastic_apply", dev(x).ptr, 32, None, out2.ptr, 16, 1, 28, 28, 0, 1, None, None, None,
         .03, None, 77, 5, None, 32)
    np.testing.assert_array_equal(out2.get_value(), full[32:48])


def test_deformer_matches_reference_fixture():
    g = np.load(os.path.join(G, "deformer.npz"))
    for k in range(4):
        scale, sigma, cval = g["prm%d" % k]
        img = g["imgs"][k].astype(np.float32)
        out = empty((1, 28, 28))
        call("tn_deformer_transform", dev(img[None]).ptr, out.ptr, 1, 28, 28, float(scale),
             float(sigma), float(cval), dev(g["noise%d" % k].astype(np.float32)[None]).ptr, 0, 0)
        # fixture = the reference's own transform(); inputs were rounded to float32 here
        assert_close(out.get_value()[0], g["out%d" % k], atol=2e-4, rtol=0, what="deformer %d" % k)


def test_deformer_device_rng_batch():
    rng = np.random.RandomState(0)
    imgs = rng.rand(32, 28, 28).astype(np.float32)
    out = empty(imgs.shape)
    call("tn_deformer_transform", dev(imgs).ptr, out.ptr, 32, 28, 28, 3.0, 2.0, 0.0, None, 11, 0)
    a = out.get_value()
    assert a.shape == imgs.shape and np.isfinite(a).all()
    assert .2 < np.corrcoef(a.ravel(), imgs.ravel())[0, 1] < .999   # deformed, not destroyed
    out2 = empty((8, 28, 28))
    call("tn_deformer_transform", dev(imgs[8:16]).ptr, out2.ptr, 8, 28, 28, 3.0, 2.0, 0.0, None, 11, 8)
    np.testing.assert_array_equal(out2.get_value(), a[8:16])          # keyed by global image index


def test_deformer_class_deforms_a_database_in_place():
    """extras/deformer.py:30-79: iterating a Deformer deforms the database batch by batch, in place, and
    yields the finished batch ids; keyed by (seed, global image index), so the batch size does not matter."""
    from theanet_amd.deformer import Deformer, transform
    rng = np.random.RandomState(3)
    data = rng.rand(24, 28 * 28).astype(np.float32)
    orig = data.copy()
    d = Deformer(data, 8, (28, 28), 3.0, 2.0, cval=0.0, seed=5)
    assert "Deformer" in str(d) and d.nBatches == 3
    assert list(d) == [0, 1, 2] and d.ndone == 3
    assert not np.array_equal(data, orig) and np.isfinite(data).all()
    other = orig.copy()
    assert list(Deformer(other, 12, (28, 28), 3.0, 2.0, cval=0.0, seed=5)) == [0, 1]
    np.testing.assert_array_equal(other, data)
    g = np.load(os.path.join(G, "deformer.npz"))                    # transform() against the reference's outputs
    scale, sigma, cval = g["prm0"]
    out = transform(g["imgs"][0].astype(np.float32), float(scale), float(sigma), float(cval), noise=g["noise0"])
    assert_close(out, g["out0"], atol=2e-4, rtol=0, what="transform()")


@pytest.mark.gpu
@pytest.mark.parametrize("hw,nearest,sigma", [(28, 1, 15), (13, 0, 4), (32, 1, 2)])
def test_elastic_field_gen_equals_two_launches(hw, nearest, sigma):
    """tn_elastic_field_gen (draws generated inside the field launch) is bit-identical to
    tn_elastic_draws followed by tn_elastic_field, and the 4-wide apply equals the scalar one."""
    h = w = hw
    lib = ctx().lib
    n = lib.tn_elastic_draws_count(h, w)
    seed, step = 0x1234_5678_9abc, 7
    prm = dict(translation=2.0, zoom=1.1, magnitude=60.0, sigma=sigma, angle=5.0)
    outs = []
    for fused in (False, True):
        draws = empty((n,))
        mi, fy, fx = empty((h * w,), np.int32), empty((h * w,)), empty((h * w,))
        tgt = empty((2 * h * w,), np.float64)
        args = (h, w, prm["translation"], prm["zoom"], prm["magnitude"], prm["sigma"], prm["angle"],
                nearest, mi.ptr, fy.ptr, fx.ptr, tgt.ptr)
        if fused:
            call("tn_elastic_field_gen", draws.ptr, seed, step, None, *args)
        else:
            call("tn_elastic_draws", draws.ptr, h, w, seed, step, None)
            call("tn_elastic_field", draws.ptr, *args)
        outs.append((draws.get_value(), mi.get_value(), tgt.get_value(),
                     fy.get_value() if not nearest else None, mi, fy, fx))
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    if not nearest:
        assert np.array_equal(a[3], b[3])
    # apply: odd pixel offset forces the scalar kernel (unaligned out), aligned call takes the 4-wide one
    N, C = 5, 2
    rng = np.random.RandomState(hw)
    x = rng.rand(N + 1, C, h, w).astype(np.float32)
    xd = dev(x)
    mi, fy, fx = b[4], b[5], b[6]
    out4 = empty((N, C, h, w))
    call("tn_elastic_apply", xd.ptr, 1, None, out4.ptr, N, C, h, w, 1, nearest, mi.ptr, fy.ptr, fx.ptr,
         0.03, None, seed, step, None, 3)
    big = empty((N * C * h * w + 1,))
    out1 = big.view(1, (N, C, h, w))
    call("tn_elastic_apply", xd.ptr, 1, None, out1.ptr, N, C, h, w, 1, nearest, mi.ptr, fy.ptr, fx.ptr,
         0.03, None, seed, step, None, 3)
    assert np.array_equal(out4.get_value(), big.get_value()[1:].reshape(N, C, h, w))


@pytest.mark.gpu
@pytest.mark.parametrize("hw,nearest,K,mode,act", [(28, 1, 4, "valid", "relu10"), (16, 0, 7, "same", "tanh"),
                                                    (12, 1, 16, "valid", "relu")])
def test_elastic_convpool_fused_equals_separate_ops(hw, nearest, K, mode, act):
    """tn_elastic_convpool_fwd_mask == tn_elastic_apply followed by tn_convpool_fwd_mask, bit for bit
    (resampled image, pooled output and pooling mask)."""
    from tests.gpu_util import act_code
    h = w = hw
    lib = ctx().lib
    pad_lo, _, Ho = O.conv_geometry(h, 3, 1, mode)
    Hp = O.pool_out_sz(Ho, 2, False)
    assert lib.tn_elastic_convpool_supported(h, w, K, 3, pad_lo, Ho, Ho, 2, Hp, Hp)
    n = lib.tn_elastic_draws_count(h, w)
    seed, step = 99, 3
    draws = empty((n,))
    mi, fy, fx = empty((h * w,), np.int32), empty((h * w,)), empty((h * w,))
    call("tn_elastic_field_gen", draws.ptr, seed, step, None, h, w, 2.0, 1.1, 20.0, 3, 5.0, nearest,
         mi.ptr, fy.ptr, fx.ptr, None)
    N = 9
    rng = np.random.RandomState(hw + K)
    x = rng.rand(N + 2, 1, h, w).astype(np.float32)
    Wt = (rng.randn(K, 1, 3, 3) / 3).astype(np.float32)
    b = rng.randn(K).astype(np.float32)
    kind, prm = act_code(act)
    xd, Wd, bd = dev(x), dev(Wt), dev(b)
    apply_args = (xd.ptr, 2, None)
    tail_args = (N, 1, h, w, 1, nearest, mi.ptr, fy.ptr, fx.ptr, 0.05, None, seed, step, None, 40)
    out_a = empty((N, 1, h, w))
    call("tn_elastic_apply", *apply_args, out_a.ptr, *tail_args)
    y_a, m_a = empty((N, K, Hp, Hp)), empty((N, K, Hp, Hp), np.uint8)
    geom = (N, 1, h, w, K, 3, pad_lo, Ho, Ho, 2, Hp, Hp, kind, prm)
    call("tn_convpool_fwd_mask", out_a.ptr, Wd.ptr, bd.ptr, y_a.ptr, m_a.ptr, *geom)
    out_b = empty((N, 1, h, w))
    y_b, m_b = empty((N, K, Hp, Hp)), empty((N, K, Hp, Hp), np.uint8)
    call("tn_elastic_convpool_fwd_mask", *apply_args, out_b.ptr, N, h, w, 1, nearest, mi.ptr, fy.ptr,
         fx.ptr, 0.05, None, seed, step, None, 40, Wd.ptr, bd.ptr, y_b.ptr, m_b.ptr, K, 3, pad_lo, Ho, Ho,
         2, Hp, Hp, kind, prm)
    assert np.array_equal(out_a.get_value(), out_b.get_value())
    assert np.array_equal(y_a.get_value(), y_b.get_value())
    assert np.array_equal(m_a.get_value(), m_b.get_value())


@pytest.mark.parametrize("C,hw,nearest,invert,mode", [
    (3, 32, False, 0, "philox"), (3, 32, True, 1, "mask"), (1, 28, False, 1, "none"), (8, 16, False, 0, "philox"),
    (11, 16, True, 0, "mask"), (4, 64, False, 0, "none"),
])
def test_elastic_apply_into_the_c8_tensor_equals_apply_then_pack(C, hw, nearest, invert, mode):
    """tn_c8_elastic_apply (DTYPE float16: the stage writes the first conv layer's fp16 tensor) stores exactly
    fp16(tn_elastic_apply's value) -- with sample maps, inversion, an injected flip mask or flips drawn on the device --
    and zero in the channels beyond C."""
    N, rng = 5, np.random.RandomState(3)
    x = rng.rand(N + 2, C, hw, hw).astype(np.float32)
    idx = dev(rng.randint(0, (hw - 1) * hw - 1, hw * hw).astype(np.int32) // hw * hw + rng.randint(0, hw - 1, hw * hw).astype(np.int32))
    fy, fx = dev(rng.rand(hw * hw).astype(np.float32)), dev(rng.rand(hw * hw).astype(np.float32))
    fm = dev((rng.rand(N, C, hw, hw) < .2).astype(np.uint8)) if mode == "mask" else None
    pflip = .15 if mode == "philox" else 0.0
    d_step = dev(np.array([7], np.uint32))
    args = (6, int(nearest), idx.ptr, fy.ptr, fx.ptr, pflip, fm.ptr if fm is not None else None, 12345, 3, d_step.ptr, 40)
    xd, row0 = dev(x), dev(np.array([1], np.int64))
    out = empty((N, C, hw, hw))
    call("tn_elastic_apply", xd.ptr, 1, row0.ptr, out.ptr, N, C, hw, hw, invert, *args[1:])
    C8 = (C + 7) // 8
    out16 = empty((N, C8, hw * hw, 8), np.uint16)
    call("tn_c8_elastic_apply", xd.ptr, 1, row0.ptr, out16.ptr, N, C, hw, hw, invert, *args[1:])
    want = np.zeros((N, C8 * 8, hw * hw), np.float16)
    want[:, :C] = out.get_value().reshape(N, C, hw * hw).astype(np.float16)
    want = want.reshape(N, C8, 8, hw * hw).transpose(0, 1, 3, 2)
    np.testing.assert_array_equal(out16.get_value().view(np.float16), want)
    if mode == "philox":                     # and the flips really happened (about pflip of the elements)
        plain = empty((N, C, hw, hw))
        call("tn_elastic_apply", xd.ptr, 1, row0.ptr, plain.ptr, N, C, hw, hw, invert, int(nearest), idx.ptr, fy.ptr, fx.ptr,
             0.0, None, 12345, 3, d_step.ptr, 40)
        assert 0.10 < np.mean(plain.get_value() != out.get_value()) < 0.20
