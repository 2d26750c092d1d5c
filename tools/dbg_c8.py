"""Cycle stamps of the fp16-resident conv kernel (TN_C8_DBG=1): where a block's lifetime goes.
   WB/WC/WK/WH = images, channels, filters, map size; OP = fwd | fwdpool | dgrad | dgradpool"""
import ctypes, os, sys
os.environ["TN_C8_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from theanet_amd import _lib
from theanet_amd.device import get_context
N, C, K, H = [int(os.environ.get(k, d)) for k, d in (("WB", 128), ("WC", 64), ("WK", 64), ("WH", 64))]
op = os.environ.get("OP", "fwd")
ctx = get_context()
rng = np.random.default_rng(1)
C8, K8, Hp = (C + 7) // 8, K // 8, H // 2
r16 = lambda shape, s=1.0: ctx.array((rng.standard_normal(shape) * s).astype(np.float16).view(np.uint16))
x = r16((N, C8, H, H, 8)); W = ctx.array((rng.standard_normal((K, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
b = ctx.zeros((K,)); y = ctx.empty((N, K8, H, H, 8), np.uint16); yp = ctx.empty((N, K8, Hp, Hp, 8), np.uint16)
mk = ctx.empty((N, K8, Hp, Hp, 8), np.uint8); dz = r16((N, K8, H, H, 8), 1e-2); gp = r16((N, K8, Hp, Hp, 8), 1e-2)
dx = ctx.empty((N, C8, H, H, 8), np.uint16)
LK = _lib.TN_ACT_LEAKY
dW, db = ctx.empty((K, C, 3, 3)), ctx.empty((K,))
ctx.call("tn_set_matmul_dtype", 1, 4096.0)
for it in range(int(os.environ.get("ITERS", 3))):      # (ITERS=2000: stamps of a launch in a sustained run)
    if op == "fwd":
        ctx.call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, y.ptr, None, N, C, H, H, K, LK, .1, 0, None)
    elif op == "fwdpool":
        ctx.call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, yp.ptr, mk.ptr, N, C, H, H, K, LK, .1, 1, None)
    elif op == "dgrad":
        ctx.call("tn_c8_conv_dgrad", dz.ptr, W.ptr, dx.ptr, N, C, H, H, K, x.ptr, LK, .1, 0, None, None)
    elif op == "dgradpool":
        ctx.call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, yp.ptr, mk.ptr, N, C, H, H, K, LK, .1, 1, None)
        ctx.call("tn_c8_conv_dgrad", gp.ptr, W.ptr, dx.ptr, N, C, H, H, K, x.ptr, LK, .1, 1, mk.ptr, None)
    elif op == "wgrad":
        ctx.call("tn_c8_conv_wgrad", x.ptr, dz.ptr, dW.ptr, db.ptr, N, C, H, H, K, 0, None)
    else:
        ctx.call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, yp.ptr, mk.ptr, N, C, H, H, K, LK, .1, 1, None)
        ctx.call("tn_c8_conv_wgrad", x.ptr, gp.ptr, dW.ptr, db.ptr, N, C, H, H, K, 1, mk.ptr)
ctx.sync()
nb = int(os.environ.get("NB", 8192))
buf = np.zeros((nb, 8), np.uint64)
rc = ctx.lib.tn_c8_dbg_read(ctx.h, ctypes.c_void_p(buf.ctypes.data), ctypes.c_int(nb))
assert rc == 0, rc
if os.environ.get("WAVES", "0") == "1":           # sixteen-wave weight gradient: per compute wave (records 4096 + 16 block + wave)
    wv = buf[4096:4096 + 16 * 256].reshape(256, 16, 8).astype(np.int64)
    live = wv[:, 0, 0] > 0
    for w in range(12):
        print("wave %2d (ft %d ct %d u %d): matrix steps median %7d   barrier wait %7d" % (
            w, w & 1, (w >> 1) & 1, w >> 2, np.median(wv[live, w, 7]), np.median(wv[live, w, 6])))
    buf = buf[:4096]
buf = buf[buf[:, 0] > 0]
tot = (buf[:, 2] - buf[:, 0]).astype(np.int64)
print("%s N%d C%d K%d H%d: blocks stamped: %d" % (op, N, C, K, H, len(buf)))
wg = op.startswith("wgrad")
dW, db = ctx.empty((K, C, 3, 3)), ctx.empty((K,))
tr = wg and os.environ.get("TN_C8_WTR", "1") != "0" and C > 32 and K > 32      # sixteen-wave form: d[1] = loader issue time
for nm, v in (("block life", tot), ("loader: DMA issue + set-up" if tr else "prologue", (buf[:, 1] if tr else buf[:, 1] - buf[:, 0]).astype(np.int64)),
              ("DMA wait" if wg else "LDS stores (+ wait for loads)", buf[:, 3].astype(np.int64)),
              ("barriers" if wg else "epilogues", buf[:, 6].astype(np.int64)), ("matrix steps" if wg else "barriers", buf[:, 7].astype(np.int64))):
    print("%-30s cycles: median %8d  p10 %8d  p90 %8d   (%.0f %% of life)" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90),
                                                                              100.0 * np.median(v) / np.median(tot)))
w0 = buf[:, 4].astype(np.int64); w1 = buf[:, 5].astype(np.int64)
print("wall clock (100 MHz ticks): kernel span %d, block life median %d -> %.2f GHz" % (
    w1.max() - w0.min(), np.median(w1 - w0), np.median(tot) / np.median(w1 - w0) / 10.0))
