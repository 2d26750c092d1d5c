#!/usr/bin/env python
"""Per-op microbenchmark at the headline shapes (mnist.prms, B=4096): times each C-ABI op with
HIP events on the compute stream and prints us/launch plus achieved GB/s / TFLOP/s.

    python tools/opbench.py [filter-substring] [--iters N]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theanet_amd import _lib  # noqa: E402
from theanet_amd.device import get_context  # noqa: E402

B = int(os.environ.get("OPBENCH_B", 4096))
ctx = get_context()
lib = ctx.lib
rng = np.random.default_rng(0)


def dev(shape, dtype=np.float32, rand=True):
    if rand:
        a = rng.standard_normal(shape).astype(np.float32) if dtype == np.float32 else \
            (rng.random(shape) > .5).astype(dtype)
        return ctx.array(a, dtype)
    return ctx.zeros(shape, dtype)


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


LEAKY = _lib.TN_ACT_LEAKY
ops = {}


def op(name, flops=0, nbytes=0):
    def deco(f):
        ops[name] = (f, flops, nbytes)
        return f
    return deco


# ---- buffers -------------------------------------------------------------------------------
x0 = dev((B, 1, 28, 28))
W1, b1 = dev((4, 1, 3, 3)), dev((4,))
a1 = dev((B, 4, 26, 26)); p1 = dev((B, 4, 13, 13))
W2, b2 = dev((20, 4, 3, 3)), dev((20,))
a2 = dev((B, 20, 11, 11)); p2 = dev((B, 20, 6, 6))
g2 = dev((B, 20, 6, 6)); dz2 = dev((B, 20, 11, 11)); dW2, db2 = dev((20, 4, 3, 3)), dev((20,))
m2 = dev((B, 20, 6, 6), np.uint8, rand=False); m1 = dev((B, 4, 13, 13), np.uint8, rand=False)
g1 = dev((B, 4, 13, 13)); dz1 = dev((B, 4, 26, 26)); dW1, db1 = dev((4, 1, 3, 3)), dev((4,))
Wf, bf = dev((720, 500)), dev((500,))
h = dev((B, 500)); mask = dev((B, 500), np.uint8)
dh = dev((B, 500)); dWf, dbf = dev((720, 500)), dev((500,))
wsf = ctx.empty((lib.tn_fc_wgrad_ws_bytes(B, 720, 500) // 4 + 16,))
Ws, bs = dev((500, 10)), dev((10,))
lg = dev((B, 10)); dlg = dev((B, 10)); dWs, dbs = dev((500, 10)), dev((10,))
wss = ctx.empty((lib.tn_fc_wgrad_ws_bytes(B, 500, 10) // 4 + 16,))
y = ctx.array(rng.integers(0, 10, B).astype(np.int32))
lp, rl, pr, rp = dev((B, 10)), dev((B,)), ctx.empty((B,), np.int32), dev((B,))
xe = dev((B, 1, 28, 28))

c1 = (B, 1, 28, 28, 4, 3)
c2 = (B, 4, 13, 13, 20, 3)


@op("convpool1_fwd", 2 * B * 676 * 4 * 9, 4 * B * (784 + 676))
def _():
    ctx.call("tn_convpool_fwd", x0.ptr, W1.ptr, b1.ptr, p1.ptr, *c1, 0, 26, 26, 2, 13, 13, LEAKY, .1)


@op("convpool2_fwd", 2 * B * 121 * 20 * 36, 4 * B * (676 + 720))
def _():
    ctx.call("tn_convpool_fwd", p1.ptr, W2.ptr, b2.ptr, p2.ptr, *c2, 0, 11, 11, 2, 6, 6, LEAKY, .05)


@op("convpool1_bwd", 4 * B * 676 * 4 * 9, 4 * B * (784 + 676))
def _():
    ctx.call("tn_convpool_bwd", x0.ptr, W1.ptr, b1.ptr, g1.ptr, None, dW1.ptr, db1.ptr, *c1, 0, 26, 26,
             2, 13, 13, LEAKY, .1)


@op("convpool2_bwd", 4 * B * 121 * 20 * 36, 4 * B * (676 + 720 + 2420))
def _():
    ctx.call("tn_convpool_bwd", p1.ptr, W2.ptr, b2.ptr, g2.ptr, dz2.ptr, dW2.ptr, db2.ptr, *c2, 0, 11, 11,
             2, 6, 6, LEAKY, .05)


@op("convblock2_bwd", 6 * B * 121 * 20 * 36, 4 * B * (676 + 720 + 676))
def _():
    ctx.call("tn_convblock_bwd", p1.ptr, W2.ptr, b2.ptr, g2.ptr, g1.ptr, dW2.ptr, db2.ptr, *c2, 0, 11, 11,
             2, 6, 6, LEAKY, .05)


@op("convpool1_fwd_mask", 2 * B * 676 * 36, 4 * B * (784 + 676) + B * 676)
def _():
    ctx.call("tn_convpool_fwd_mask", x0.ptr, W1.ptr, b1.ptr, p1.ptr, m1.ptr, *c1, 0, 26, 26, 2, 13, 13, LEAKY, .1)


@op("convpool1_bwd_mask", 2 * B * 676 * 36, 4 * B * (784 + 676) + B * 676)
def _():
    ctx.call("tn_convpool_bwd_mask", x0.ptr, g1.ptr, p1.ptr, m1.ptr, None, dW1.ptr, db1.ptr, *c1, 0, 26, 26,
             2, 13, 13, LEAKY, .1)


@op("convpool2_fwd_mask", 2 * B * 121 * 20 * 36, 4 * B * (676 + 720) + B * 720)
def _():
    ctx.call("tn_convpool_fwd_mask", p1.ptr, W2.ptr, b2.ptr, p2.ptr, m2.ptr, *c2, 0, 11, 11, 2, 6, 6, LEAKY, .05)


@op("convblock2_bwd_mask", 4 * B * 121 * 20 * 36, 4 * B * (676 + 720 + 720 + 676) + B * 720)
def _():
    ctx.call("tn_convblock_bwd_mask", p1.ptr, W2.ptr, g2.ptr, p2.ptr, m2.ptr, g1.ptr, dW2.ptr, db2.ptr, *c2,
             0, 11, 11, 2, 6, 6, LEAKY, .05)


@op("conv2_dgrad", 2 * B * 121 * 20 * 36, 4 * B * (2420 + 676))
def _():
    ctx.call("tn_conv2d_dgrad", dz2.ptr, W2.ptr, g1.ptr, *c2, 1, 0, 11, 11, None, 0, 0.0)


@op("conv1_fwd_unfused", 2 * B * 676 * 36, 4 * B * (784 + 2704))
def _():
    ctx.call("tn_conv2d_fwd", x0.ptr, W1.ptr, b1.ptr, a1.ptr, *c1, 1, 0, 26, 26, LEAKY, .1)


@op("pool1_fwd_unfused", 0, 4 * B * (2704 + 676))
def _():
    ctx.call("tn_pool_fwd", a1.ptr, p1.ptr, B * 4, 26, 26, 2, 13, 13)


@op("fc1_fwd", 2 * B * 720 * 500, 4 * (B * 720 + 720 * 500 + B * 500))
def _():
    ctx.call("tn_fc_fwd", p2.ptr, Wf.ptr, bf.ptr, h.ptr, B, 720, 500, LEAKY, .01, mask.ptr)


@op("fc1_dgrad", 2 * B * 720 * 500, 4 * (B * 720 + 720 * 500 + B * 500))
def _():
    ctx.call("tn_fc_dgrad", dh.ptr, Wf.ptr, g2.ptr, B, 720, 500, None, 0, 0.0, None)


@op("fc1_wgrad", 2 * B * 720 * 500, 4 * (B * 720 + 720 * 500 + B * 500))
def _():
    ctx.call("tn_fc_wgrad", p2.ptr, dh.ptr, dWf.ptr, dbf.ptr, B, 720, 500, wsf.ptr)


@op("fc2_fwd", 2 * B * 500 * 10, 4 * (B * 500 + B * 10))
def _():
    ctx.call("tn_fc_fwd", h.ptr, Ws.ptr, bs.ptr, lg.ptr, B, 500, 10, 0, 0.0, None)


@op("fc2_dgrad", 2 * B * 500 * 10, 4 * (2 * B * 500 + B * 10) + B * 500)
def _():
    ctx.call("tn_fc_dgrad", dlg.ptr, Ws.ptr, dh.ptr, B, 500, 10, h.ptr, LEAKY, .01, mask.ptr)


@op("fc2_wgrad", 2 * B * 500 * 10, 4 * (B * 500 + B * 10))
def _():
    ctx.call("tn_fc_wgrad", h.ptr, dlg.ptr, dWs.ptr, dbs.ptr, B, 500, 10, wss.ptr)


@op("softmax_nll", 0, 4 * B * 40)
def _():
    ctx.call("tn_softmax_nll", lg.ptr, y.ptr, 0, None, lp.ptr, rl.ptr, pr.ptr, rp.ptr, dlg.ptr, B, 10, 1.0 / B)


@op("elastic_apply", 0, 8 * B * 784)
def _():
    ctx.call("tn_elastic_apply", x0.ptr, 0, None, xe.ptr, B, 1, 28, 28, 1, 1, None, None, None, .03, None,
             5, 0, None, 0)


@op("dropout_mask", 0, B * 500)
def _():
    ctx.call("tn_dropout_mask", mask.ptr, B * 500, .5, 77, 0, None, 0)


# ---- a wide6-like layer: 64 -> 64 maps, 3x3 same, 64x64 images, 128 images/GPU ------------
WB = int(os.environ.get("OPBENCH_WB", 128))
WC = int(os.environ.get("OPBENCH_WC", 64))      # input maps
WK = int(os.environ.get("OPBENCH_WK", 64))      # filters
WH = int(os.environ.get("OPBENCH_WH", 64))      # image side
wx = dev((WB, WC, WH, WH)); wW, wb_ = dev((WK, WC, 3, 3)), dev((WK,))
wa = dev((WB, WK, WH, WH)); wdz = dev((WB, WK, WH, WH)); wdx = dev((WB, WC, WH, WH))
wpa = dev((WB, WC, WH, WH))
wdW, wdb = dev((WK, WC, 3, 3)), dev((WK,))
cw = (WB, WC, WH, WH, WK, 3)
WFL = 2 * WB * WH * WH * WK * WC * 9
WBY = 4 * (WB * WH * WH * (WC + WK) + WK * WC * 9)


@op("wide_conv_fwd", WFL, WBY)
def _():
    ctx.call("tn_conv2d_fwd", wx.ptr, wW.ptr, wb_.ptr, wa.ptr, *cw, 1, 1, WH, WH, LEAKY, .1)


@op("wide_conv_dgrad", WFL, WBY)
def _():
    ctx.call("tn_conv2d_dgrad", wdz.ptr, wW.ptr, wdx.ptr, *cw, 1, 1, WH, WH, wpa.ptr, LEAKY, .1)


@op("wide_conv_wgrad", WFL, WBY)
def _():
    ctx.call("tn_conv2d_wgrad", wx.ptr, wdz.ptr, wdW.ptr, wdb.ptr, *cw, 1, 1, WH, WH)


# ---- calibration: launch boundary and GEMM fixed cost ------------------------------------------
tiny = dev((16,))
xk16, Wk16 = dev((B, 16)), dev((16, 500))


@op("empty_kernel", 0, 0)
def _():
    ctx.call("tn_axpby", tiny.ptr, tiny.ptr, 1, 1.0, 0.0)


@op("fc_fwd_K16", 2 * B * 16 * 500, 4 * (B * 16 + 16 * 500 + B * 500))
def _():
    ctx.call("tn_fc_fwd", xk16.ptr, Wk16.ptr, bf.ptr, h.ptr, B, 16, 500, LEAKY, .01, mask.ptr)


@op("fc_fwd_K16_nomask", 2 * B * 16 * 500, 4 * (B * 16 + 16 * 500 + B * 500))
def _():
    ctx.call("tn_fc_fwd", xk16.ptr, Wk16.ptr, bf.ptr, h.ptr, B, 16, 500, 0, .0, None)


def main():
    flt = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = 20
    if "--iters" in sys.argv:
        iters = int(sys.argv[sys.argv.index("--iters") + 1])
        flt = [f for f in flt if f != str(iters)]
    for name, (fn, fl, by) in ops.items():
        if flt and not any(f in name for f in flt):
            continue
        us = timeit(fn, iters)
        print("%-20s %9.1f us   %8.2f TFLOP/s   %8.1f GB/s" % (name, us, fl / us / 1e6, by / us / 1e3))


if __name__ == "__main__":
    main()
