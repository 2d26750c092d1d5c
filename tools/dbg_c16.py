"""Cycle stamps of the fp16-operand LDS-tile conv kernel (TN_CT_DBG=1): where a block's lifetime goes.
   WB/WC/WK/WH = images, channels, filters, map size; OP = fwd | dgrad"""
import ctypes, os, sys
os.environ["TN_CT_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.gpu_util import ctx, dev, empty, call, act_code
N, C, K, H = [int(os.environ.get(k, d)) for k, d in (("WB", 128), ("WC", 64), ("WK", 64), ("WH", 64))]
op = os.environ.get("OP", "fwd")
rng = np.random.RandomState(1)
x = dev(rng.randn(N, C, H, H).astype(np.float32)); W = dev(rng.randn(K, C, 3, 3).astype(np.float32))
b = dev(rng.randn(K).astype(np.float32)); a = empty((N, K, H, H)); dx = empty((N, C, H, H))
kind, prm = act_code("relu10")
c_ = ctx(); c_.set_matmul_dtype("float16", 4096.)
for it in range(3):
    if op == "fwdpool":
        y = empty((N, K, H // 2, H // 2)); m = empty((N, K, H // 2, H // 2), np.uint8)
        call("tn_convpool_fwd_mask", x.ptr, W.ptr, b.ptr, y.ptr, m.ptr, N, C, H, H, K, 3, 1, H, H, 2, H // 2, H // 2,
             kind, prm)
    elif op == "fwd":
        call("tn_conv2d_fwd", x.ptr, W.ptr, b.ptr, a.ptr, N, C, H, H, K, 3, 1, 1, H, H, kind, prm)
    else:
        call("tn_conv2d_dgrad", a.ptr, W.ptr, dx.ptr, N, C, H, H, K, 3, 1, 1, H, H, None, 0, 0.0)
nb = int(os.environ.get("NB", 2048))
buf = np.zeros((nb, 8), np.uint64)
rc = c_.lib.tn_conv_tile16_dbg_read(c_.h, ctypes.c_void_p(buf.ctypes.data), ctypes.c_int(nb))
assert rc == 0, rc
buf = buf[buf[:, 0] > 0]
d = buf[:, 1:4].astype(np.int64) - buf[:, 0:3].astype(np.int64)
print("blocks stamped:", len(buf))
for nm, col in (("prologue", 0), ("main loop", 1), ("epilogue", 2)):
    v = d[:, col]
    print("%-10s cycles: median %8d  p10 %8d  p90 %8d" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
w0 = buf[:, 4].astype(np.int64); w1 = buf[:, 5].astype(np.int64)
print("wall clock (100 MHz ticks): kernel span %d, block life median %d" % (w1.max() - w0.min(), np.median(w1 - w0)))
st = np.sort(w0 - w0.min())
print("block start times (ticks) deciles:", [int(st[int(q * (len(st) - 1))]) for q in np.linspace(0, 1, 11)])
