#!/usr/bin/env python
"""Per-layer conv timings, fp32 vs fp16-operand kernels, at the wide6 / cifar_like layer shapes.

    python tools/bench_f16.py [--iters N] [--n 128]

Prints us/launch and TFLOP/s for forward, input gradient and weight gradient of every layer shape
(HIP events on the compute stream), plus the HBM-bound floor of each op (tensors read / written once
as fp32 at 6.3 TB/s achievable)."""
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


SHAPES = [  # name, N-scale, C, H, K
    ("wide6 conv1", 1, 3, 64, 64), ("wide6 conv2", 1, 64, 64, 64), ("wide6 conv3", 1, 64, 32, 128),
    ("wide6 conv4", 1, 128, 32, 128), ("wide6 conv5", 1, 128, 16, 256), ("wide6 conv6", 1, 256, 16, 256),
    ("cifar conv2", 16, 32, 16, 64), ("cifar conv3", 16, 64, 8, 128),
]
LEAKY = _lib.TN_ACT_LEAKY
print("%-14s %-6s %10s %10s %10s   (us/launch | TFLOP/s)   HBM floor us" % ("layer", "dtype", "fwd", "dgrad", "wgrad"))
for name, ns, C, H, K in SHAPES:
    if args.only and args.only not in name:
        continue
    N = args.n * ns
    x = ctx.array(rng.standard_normal((N, C, H, H)).astype(np.float32))
    W = ctx.array((rng.standard_normal((K, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
    b = ctx.zeros((K,))
    a = ctx.empty((N, K, H, H))
    dz = ctx.array((rng.standard_normal((N, K, H, H)) * 1e-4).astype(np.float32))
    dx = ctx.empty((N, C, H, H))
    dW, db = ctx.empty((K, C, 3, 3)), ctx.empty((K,))
    flops = 2.0 * N * H * H * K * C * 9
    floor = 4.0 * N * H * H * (C + K) / 6.3e12 * 1e6
    for dtype in ("float32", "float16"):
        ctx.set_matmul_dtype(dtype, 4096.0)
        geom = (N, C, H, H, K, 3, 1, 1, H, H)
        t_f = timeit(lambda: ctx.call("tn_conv2d_fwd", x.ptr, W.ptr, b.ptr, a.ptr, *geom, LEAKY, 0.1), args.iters)
        t_d = timeit(lambda: ctx.call("tn_conv2d_dgrad", dz.ptr, W.ptr, dx.ptr, *geom, None, 0, 0.0), args.iters) \
            if C * 9 > 32 else float("nan")
        t_w = timeit(lambda: ctx.call("tn_conv2d_wgrad", x.ptr, dz.ptr, dW.ptr, db.ptr, *geom), args.iters)
        if dtype == "float16" and H % 4 == 0 and lib.tn_convpool_f16_supported(N, C, H, H, K, 3, 1, 1, H, H, 2, H // 2, H // 2):
            yp = ctx.empty((N, K, H // 2, H // 2)); mk = ctx.empty((N, K, H // 2, H // 2), np.uint8)
            gp = ctx.array((rng.standard_normal((N, K, H // 2, H // 2)) * 1e-4).astype(np.float32))
            pg = (N, C, H, H, K, 3, 1, H, H, 2, H // 2, H // 2, LEAKY, 0.1)
            t_pf = timeit(lambda: ctx.call("tn_convpool_fwd_mask", x.ptr, W.ptr, b.ptr, yp.ptr, mk.ptr, *pg), args.iters)
            t_pb = timeit(lambda: ctx.call("tn_convpool_bwd_mask_dx", x.ptr, W.ptr, gp.ptr, yp.ptr, mk.ptr,
                                           dx.ptr if C * 9 > 32 else None, dW.ptr, db.ptr, *pg, None, 0, 0.0), args.iters)
            print("%-14s pooled block: forward %6.0f us, backward (dW, db%s) %6.0f us" % (
                name, t_pf, ", dx" if C * 9 > 32 else "", t_pb))
        print("%-14s %-6s %s   %.0f" % (name, "f16" if dtype == "float16" else "f32", " ".join(
            "%6.0f|%5.0f" % (t, flops / t / 1e6) for t in (t_f, t_d, t_w)), floor))
    ctx.set_matmul_dtype("float32")
    del x, W, a, dz, dx
