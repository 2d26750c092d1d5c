"""Pins for the oracle's elastic input stage.  deformer.npz holds outputs of the
REFERENCE's extras/deformer.py:7-18 (executed in the build container), so
``deformer_transform`` is pinned against the real reference; the in-graph
ElasticLayer restatement is pinned analytically (Theano cannot run)."""
import os

import numpy as np

from oracle import theanet_oracle as O
from oracle.randomstreams import RandomStreams

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_deformer_matches_reference_fixture():
    d = np.load(os.path.join(G, "deformer.npz"))
    for k in range(4):
        scale, sigma, cval = d["prm%d" % k]
        out, trans = O.deformer_transform(d["imgs"][k], scale, sigma, cval, noise=d["noise%d" % k])
        np.testing.assert_allclose(trans, d["trans%d" % k], rtol=0, atol=1e-11)
        np.testing.assert_allclose(out, d["out%d" % k], rtol=0, atol=1e-11)


def test_deformer_default_rng_is_global_numpy():
    d = np.load(os.path.join(G, "deformer.npz"))
    scale, sigma, cval = d["prm0"]
    np.random.seed(100)
    out, _ = O.deformer_transform(d["imgs"][0], scale, sigma, cval)
    np.testing.assert_allclose(out, d["out0"], rtol=0, atol=1e-11)


def test_elastic_inactive_is_identity_plus_invert():
    st = O.ElasticStage(8, invert_image=True)
    x = np.random.RandomState(0).rand(2, 1, 8, 8).astype(np.float32)
    out, tgt = st.forward(x)
    assert tgt is None
    np.testing.assert_array_equal(out, np.float32(1) - x)
    st = O.ElasticStage(8, translation=2, rand_gen=np.random.RandomState(1))
    out, _ = st.forward(x, train=False)       # TestVersion: no distortion (inlayers.py:157-163)
    np.testing.assert_array_equal(out, x)


def test_elastic_pure_translation_nearest():
    x = np.arange(36, dtype=np.float32).reshape(1, 1, 6, 6)
    d = O.ElasticDraws()
    d.transln = np.array([1., -.5], np.float32).reshape(2, 1, 1)   # *translation(2) -> (+2, -1)
    prm = dict(translation=2, zoom=1, magnitude=0, sigma=1, angle=0, pflip=0)
    tgt = O.elastic_field(6, 6, prm, d)
    out = O.elastic_apply(x, tgt, nearest=True)
    yy, xx = np.indices((6, 6))
    exp = x[0, 0][np.clip(yy + 2, 0, 5), np.clip(xx - 1, 0, 5)]
    # clip is to h-1-.001 = 4.999 -> rint 5
    np.testing.assert_array_equal(out[0, 0], exp)


def test_elastic_bilinear_half_pixel():
    x = np.arange(16, dtype=np.float32).reshape(1, 1, 4, 4)
    tgt = np.indices((4, 4)).astype(np.float64) + .5
    out = O.elastic_apply(x, tgt, nearest=False)
    # interior: mean of the 2x2 block
    assert out[0, 0, 0, 0] == np.float32((0 + 1 + 4 + 5) / 4)
    # last row/col clipped to 2.999 -> almost x[3]
    np.testing.assert_allclose(out[0, 0, 3, 3], 15, atol=1e-2)


def test_elastic_rotation_uses_transpose_and_origin():
    # angle only: theta = angle*pi/180*u ; tensordot over R's FIRST axis applies R^T
    d = O.ElasticDraws()
    d.origin_u = np.array([.5, .5], np.float32).reshape(2, 1, 1)
    d.theta_u = np.float32(1.0)
    prm = dict(translation=0, zoom=1, magnitude=0, sigma=1, angle=90, pflip=0)
    tgt = O.elastic_field(5, 5, prm, d)
    # origin (2.5,2.5); R^T with theta=90deg: y' = x-2.5 (s*x), x' = -(y-2.5) ... check one point
    y, x = 0.0, 0.0
    c, s = np.cos(np.pi / 2), np.sin(np.pi / 2)
    ey = c * (y - 2.5) + s * (x - 2.5) + 2.5
    ex = -s * (y - 2.5) + c * (x - 2.5) + 2.5
    np.testing.assert_allclose([tgt[0, 0, 0], tgt[1, 0, 0]], [ey, ex], atol=1e-12)


def test_elastic_gaussian_is_unnormalised_radius_sigma():
    f = O.elastic_filter(2)
    assert f.shape == (5, 5)
    np.testing.assert_allclose(f[2, 2], 1 / (2 * np.pi * 4), rtol=1e-6)
    assert f.sum() < 1        # truncated at 1 sigma, not renormalised
    # a constant noise plane in the interior is scaled by filt.sum()
    d = O.ElasticDraws(); d.noise = np.ones((2, 9, 9), np.float32)
    prm = dict(translation=0, zoom=1, magnitude=3, sigma=2, angle=0, pflip=0)
    tgt = O.elastic_field(9, 9, prm, d)
    np.testing.assert_allclose(tgt[0, 4, 4] - 4, 3 * f.sum(), rtol=1e-6)
    # zero padding at the border: corner sees only a quadrant
    np.testing.assert_allclose(tgt[0, 0, 0] - 0, 3 * f[2:, 2:].sum(), rtol=1e-6)


def test_flip_noise():
    x = np.full((1, 1, 2, 2), .25, np.float32)
    m = np.array([[1, 0], [0, 1]], np.float32).reshape(1, 1, 2, 2)
    out = O.elastic_apply(x, np.indices((2, 2)).astype(np.float64), True, flipmask=m)
    np.testing.assert_array_equal(out[0, 0], [[.75, .25], [.25, .75]])


def test_randomstreams_seeding_order():
    # each variable: RandomState(seedgen.randint(2**30)) in creation order; state advances per draw
    srs = RandomStreams(42)
    a = srs.uniform((3,), -1)
    b = srs.binomial((4,), n=1, p=.5)
    sg = np.random.RandomState(42)
    ra = np.random.RandomState(int(sg.randint(2 ** 30)))
    rb = np.random.RandomState(int(sg.randint(2 ** 30)))
    np.testing.assert_array_equal(a.draw(), ra.uniform(-1, 1, (3,)).astype(np.float32))
    np.testing.assert_array_equal(b.draw(), rb.binomial(1, .5, (4,)))
    np.testing.assert_array_equal(a.draw(), ra.uniform(-1, 1, (3,)).astype(np.float32))


def test_elastic_stage_stream_statistics():
    st = O.ElasticStage(28, translation=2, zoom=1.1, magnitude=60, sigma=15, pflip=.03,
                        angle=5, nearest=True, invert_image=True,
                        rand_gen=np.random.RandomState(3))
    x = np.random.RandomState(0).rand(64, 1, 28, 28).astype(np.float32)
    d = st.draw(x.shape)
    assert d.noise.shape == (2, 28, 28) and d.noise.dtype == np.float32
    assert abs(d.flipmask.mean() - .03) < .01
    out, tgt = st.forward(x, d)
    assert out.shape == x.shape and tgt.shape == (2, 28, 28)
    assert out.min() >= 0 and out.max() <= 1
