"""Run one fp16-resident conv op a few times (for rocprofv3 passes): OP = fwd | fwdpool | dgrad | dgradpool | wgrad | wgradpool;
   WB/WC/WK/WH = images, channels, filters, map size; IT = launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from theanet_amd import _lib
from theanet_amd.device import get_context
N, C, K, H, IT = [int(os.environ.get(k, d)) for k, d in (("WB", 128), ("WC", 64), ("WK", 64), ("WH", 64), ("IT", 5))]
op = os.environ.get("OP", "wgrad")
ctx = get_context()
rng = np.random.default_rng(1)
C8, K8, Hp = (C + 7) // 8, K // 8, H // 2
r16 = lambda shape, s=1.0: ctx.array((rng.standard_normal(shape) * s).astype(np.float16).view(np.uint16))
x = r16((N, C8, H, H, 8)); W = ctx.array((rng.standard_normal((K, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
b = ctx.zeros((K,)); y = ctx.empty((N, K8, H, H, 8), np.uint16); yp = ctx.empty((N, K8, Hp, Hp, 8), np.uint16)
mk = ctx.array(rng.integers(0, 16, (N, K8, Hp, Hp, 8)).astype(np.uint8)); dz = r16((N, K8, H, H, 8), 1e-2); gp = r16((N, K8, Hp, Hp, 8), 1e-2)
dx = ctx.empty((N, C8, H, H, 8), np.uint16); dW, db = ctx.empty((K, C, 3, 3)), ctx.empty((K,))
LK = _lib.TN_ACT_LEAKY
ctx.call("tn_set_matmul_dtype", 1, 4096.0)
ctx.call("tn_defer_reductions", 1)
for it in range(IT):
    if op == "fwd":
        ctx.call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, y.ptr, None, N, C, H, H, K, LK, .1, 0, None)
    elif op == "fwdpool":
        ctx.call("tn_c8_conv_fwd", x.ptr, W.ptr, b.ptr, yp.ptr, mk.ptr, N, C, H, H, K, LK, .1, 1, None)
    elif op == "dgrad":
        ctx.call("tn_c8_conv_dgrad", dz.ptr, W.ptr, dx.ptr, N, C, H, H, K, x.ptr, LK, .1, 0, None, None)
    elif op == "dgradpool":
        ctx.call("tn_c8_conv_dgrad", gp.ptr, W.ptr, dx.ptr, N, C, H, H, K, x.ptr, LK, .1, 1, mk.ptr, None)
    elif op == "wgrad":
        ctx.call("tn_c8_conv_wgrad", x.ptr, dz.ptr, dW.ptr, db.ptr, N, C, H, H, K, 0, None)
    else:
        ctx.call("tn_c8_conv_wgrad", x.ptr, gp.ptr, dW.ptr, db.ptr, N, C, H, H, K, 1, mk.ptr)
    ctx.call("tn_defer_reductions", 0)
    ctx.call("tn_defer_reductions", 1)
ctx.sync()
