"""Quick GPU check of the c8 conv kernels against tests/c8_util.py (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.c8_util import *
from theanet_amd.device import get_context
from theanet_amd import _lib

ctx = get_context()
rng = np.random.RandomState(0)
ok_all = True
for (N, C, H, K) in [(3, 16, 16, 32), (2, 64, 64, 64), (5, 24, 8, 40), (2, 128, 32, 128), (2, 8, 64, 64), (3, 64, 32, 128),
                     (3, 3, 64, 64), (37, 3, 32, 32), (9, 256, 16, 256), (33, 40, 8, 72)]:
    x = r16(rng.randn(N, C, H, H))
    W = (rng.randn(K, C, 3, 3) / np.sqrt(9 * C)).astype(np.float32)
    b = rng.randn(K).astype(np.float32) * .1
    W16 = r16(W)
    z = conv_same(x, W16) + b[None, :, None, None]
    a = leaky(z, .1)
    dx_ = ctx.array(to_c8(x).view(np.uint16))
    dW_, db_ = ctx.array(W), ctx.array(b)
    K8 = K // 8
    out = ctx.empty((N, K8, H, H, 8), np.uint16)
    ctx.call("tn_c8_conv_fwd", dx_.ptr, dW_.ptr, db_.ptr, out.ptr, None, N, C, H, H, K, _lib.TN_ACT_LEAKY, .1, 0, None)
    got = from_c8(out.get_value().view(np.float16), K)
    err = np.abs(got - r16(a)).max() / np.abs(a).max()
    # pooled
    pm, bits = pool2(a)
    outp = ctx.empty((N, K8, H // 2, H // 2, 8), np.uint16)
    mk = ctx.empty((N, K8, H // 2, H // 2, 8), np.uint8)
    ctx.call("tn_c8_conv_fwd", dx_.ptr, dW_.ptr, db_.ptr, outp.ptr, mk.ptr, N, C, H, H, K, _lib.TN_ACT_LEAKY, .1, 1, None)
    gotp = from_c8(outp.get_value().view(np.float16), K)
    gotm = mk.get_value().transpose(0, 1, 4, 2, 3).reshape(N, K, H // 2, H // 2)
    errp = np.abs(gotp - r16(pm)).max() / np.abs(a).max()
    mism = (gotm != bits).mean()
    # dgrad
    gs = 1024.
    dz = r16(gs * rng.randn(N, K, H, H) * 1e-3)
    prev = r16(rng.randn(N, C, H, H))
    dxw = conv_same_dgrad(dz, W16) * leaky_grad_from_out(prev, .1)
    ddz = ctx.array(to_c8(dz).view(np.uint16))
    dprev = ctx.array(to_c8(prev).view(np.uint16))
    C8 = (C + 7) // 8
    dxo = ctx.empty((N, C8, H, H, 8), np.uint16)
    ctx.call("tn_c8_conv_dgrad", ddz.ptr, dW_.ptr, dxo.ptr, N, C, H, H, K, dprev.ptr, _lib.TN_ACT_LEAKY, .1, 0, None, None)
    gotdx = from_c8(dxo.get_value().view(np.float16), C)
    errd = np.abs(gotdx - r16(dxw)).max() / np.abs(dxw).max()
    # pooled dgrad: dz from (g, mask)
    g = r16(gs * rng.randn(N, K, H // 2, H // 2) * 1e-3)
    dg = ctx.array(to_c8(g).view(np.uint16))
    ctx.call("tn_c8_conv_dgrad", dg.ptr, dW_.ptr, dxo.ptr, N, C, H, H, K, dprev.ptr, _lib.TN_ACT_LEAKY, .1, 1, mk.ptr, None)
    gotdx2 = from_c8(dxo.get_value().view(np.float16), C)
    # the device's own mask may differ from numpy's on near-ties: use the device mask for the reference
    dzp_dev = unpool_dz(g, gotm)
    dxw2 = conv_same_dgrad(dzp_dev, W16) * leaky_grad_from_out(prev, .1)
    errd2 = np.abs(gotdx2 - r16(dxw2)).max() / np.abs(dxw2).max()
    # wgrad
    ctx.call("tn_set_matmul_dtype", 1, gs)
    dWw = conv_same_wgrad(x, dz) / gs
    dbw = dz.sum(axis=(0, 2, 3)) / gs
    gW, gb = ctx.zeros((K, C, 3, 3)), ctx.zeros((K,))
    sup = ctx.lib.tn_c8_conv_wgrad_supported(N, C, H, H, K)
    errw = errb = errw2 = errb2 = -1
    if sup:
        ctx.call("tn_c8_conv_wgrad", dx_.ptr, ddz.ptr, gW.ptr, gb.ptr, N, C, H, H, K, 0, None)
        errw = np.abs(gW.get_value() - dWw).max() / np.abs(dWw).max()
        errb = np.abs(gb.get_value() - dbw).max() / np.abs(dbw).max()
        ctx.call("tn_c8_conv_wgrad", dx_.ptr, dg.ptr, gW.ptr, gb.ptr, N, C, H, H, K, 1, mk.ptr)
        dWw2 = conv_same_wgrad(x, dzp_dev) / gs
        errw2 = np.abs(gW.get_value() - dWw2).max() / np.abs(dWw2).max()
        dbw2 = dzp_dev.sum(axis=(0, 2, 3)) / gs
        errb2 = np.abs(gb.get_value() - dbw2).max() / np.abs(dbw2).max()
    ctx.call("tn_set_matmul_dtype", 0, 1.0)
    print("N%d C%d H%d K%d: fwd %.2e pool %.2e maskmis %.1e dgrad %.2e pooled-dgrad %.2e wgrad(sup %d) %.2e db %.2e pooled-wgrad %.2e db %.2e"
          % (N, C, H, K, err, errp, mism, errd, errd2, sup, errw, errb, errw2, errb2))
    ok_all &= err < 1e-3 and errp < 1e-3 and mism < 1e-3 and errd < 1e-3 and errd2 < 1e-3 and errw < 1e-4 and errb < 1e-4 and errw2 < 1e-4 and errb2 < 1e-4
print("ALL OK" if ok_all else "FAILURES")
