#!/usr/bin/env python
"""Per-layer timings of the fp16-resident ("c8") conv kernels at the wide6 / cifar_like layer shapes.

    python tools/bench_c8.py [--iters N] [--n 128] [--only conv2]

us/launch, TFLOP/s and the fraction of the HBM roof (algorithmic bytes of the op on fp16 tensors / 6.3 TB/s) for
forward, pooled forward, input gradient (plain and gathered from a pooled gradient) and both weight gradients."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theanet_amd import _lib  # noqa: E402
from theanet_amd.device import get_context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--only", default="")
args = ap.parse_args()
ctx = get_context()
lib = ctx.lib
rng = np.random.default_rng(0)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    ctx.sync()
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    lib.tn_event_create(ctx.h, ctypes.byref(a))
    lib.tn_event_create(ctx.h, ctypes.byref(b))
    lib.tn_event_record(ctx.h, a)
    for _ in range(iters):
        fn()
    lib.tn_event_record(ctx.h, b)
    ms = ctypes.c_float()
    ctx.call("tn_event_elapsed_ms", a, b, ctypes.byref(ms))
    return ms.value * 1e3 / iters


def rnd16(shape, scale=1.0):
    return ctx.array((rng.standard_normal(shape) * scale).astype(np.float16).view(np.uint16))


SHAPES = [  # name, N-scale, C, H, K
    ("wide6 conv1", 1, 3, 64, 64), ("wide6 conv2", 1, 64, 64, 64), ("wide6 conv3", 1, 64, 32, 128),
    ("wide6 conv4", 1, 128, 32, 128), ("wide6 conv5", 1, 128, 16, 256), ("wide6 conv6", 1, 256, 16, 256),
    ("cifar conv1", 16, 3, 32, 32), ("cifar conv2", 16, 32, 16, 64), ("cifar conv3", 16, 64, 8, 128),
]
LEAKY = _lib.TN_ACT_LEAKY
ctx.call("tn_set_matmul_dtype", 1, 4096.0)
print("%-12s %-22s %8s %8s %8s" % ("layer", "op", "us", "TFLOP/s", "HBM frac"))
for name, ns, C, H, K in SHAPES:
    if args.only and args.only not in name:
        continue
    N = args.n * ns
    C8, K8, Hp = (C + 7) // 8, K // 8, H // 2
    if not lib.tn_c8_conv_supported(N, C, H, H, K, 3, 1, 1):
        print("%-12s unsupported" % name)
        continue
    x = rnd16((N, C8, H, H, 8))
    W = ctx.array((rng.standard_normal((K, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
    b = ctx.zeros((K,))
    y = ctx.empty((N, K8, H, H, 8), np.uint16)
    yp = ctx.empty((N, K8, Hp, Hp, 8), np.uint16)
    mk = ctx.empty((N, K8, Hp, Hp, 8), np.uint8)
    dz = rnd16((N, K8, H, H, 8), 1e-2)
    gp = rnd16((N, K8, Hp, Hp, 8), 1e-2)
    dx = ctx.empty((N, C8, H, H, 8), np.uint16)
    dW, db = ctx.empty((K, C, 3, 3)), ctx.empty((K,))
    flops = 2.0 * N * H * H * K * C * 9
    px = N * H * H
    cin, cout = 16.0 * C8 * px, 2.0 * K * px            # bytes of the input / output tensors (c8 cells / halfs)
    ops = [
        ("fwd", lambda: ctx.call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, y.ptr, None, N, C, H, H, K, LEAKY, .1, 0, None), cin + cout),
        ("fwd+pool", lambda: ctx.call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, yp.ptr, mk.ptr, N, C, H, H, K, LEAKY, .1, 1, None),
         cin + cout / 4 + cout / 8),
    ]
    if C >= 8:
        ops += [
            ("dgrad", lambda: ctx.call("tn_c8_conv_dgrad", dz.ptr, W.ptr, dx.ptr, N, C, H, H, K, x.ptr, LEAKY, .1, 0, None, None), cout + 2 * cin),
            ("dgrad (g, mask)", lambda: ctx.call("tn_c8_conv_dgrad", gp.ptr, W.ptr, dx.ptr, N, C, H, H, K, x.ptr, LEAKY, .1, 1, mk.ptr, None), cout / 4 + cout / 8 + 2 * cin),
        ]
    if lib.tn_c8_conv_wgrad_supported(N, C, H, H, K):
        ops += [
            ("wgrad", lambda: ctx.call("tn_c8_conv_wgrad", x.ptr, dz.ptr, dW.ptr, db.ptr, N, C, H, H, K, 0, None),
             cin + cout),
            ("wgrad (g, mask)", lambda: ctx.call("tn_c8_conv_wgrad", x.ptr, gp.ptr, dW.ptr, db.ptr, N, C, H, H, K, 1, mk.ptr), cin + cout / 4 + cout / 8),
        ]
    for op, fn, nbytes in ops:
        t = timeit(fn, args.iters)
        ctx.call("tn_defer_reductions", 0)
        print("%-12s %-22s %8.1f %8.0f %8.2f" % (name, op, t, flops / t / 1e6, nbytes / 6.3e12 * 1e6 / t))
    del x, y, yp, mk, dz, gp, dx
ctx.call("tn_set_matmul_dtype", 0, 1.0)
