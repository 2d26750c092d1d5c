"""GPU check + timing of MATMUL 'bf16x3' (gemm_b3.hip) against float64 and the exact fp32 path (development aid)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from theanet_amd import _lib
from theanet_amd.device import get_context
ctx = get_context(); lib = ctx.lib
rng = np.random.RandomState(0)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    ctx.sync()
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    lib.tn_event_create(ctx.h, ctypes.byref(a)); lib.tn_event_create(ctx.h, ctypes.byref(b))
    lib.tn_event_record(ctx.h, a)
    for _ in range(iters):
        fn()
    lib.tn_event_record(ctx.h, b)
    ms = ctypes.c_float(); ctx.call("tn_event_elapsed_ms", a, b, ctypes.byref(ms))
    return ms.value * 1e3 / iters


ok_all = True
for (B, n_in, n_out) in [(4096, 720, 500), (512, 720, 500), (37, 100, 36), (200, 64, 132), (2048, 2048, 512), (128, 16384, 1024)]:
    x = rng.randn(B, n_in).astype(np.float32); W = (rng.randn(n_in, n_out) / np.sqrt(n_in)).astype(np.float32)
    b = (rng.randn(n_out) * .1).astype(np.float32); mask = (rng.rand(B, n_out) < .5).astype(np.uint8)
    dz = (rng.randn(B, n_out) * 1e-3).astype(np.float32); pa = rng.randn(B, n_in).astype(np.float32); pm = (rng.rand(B, n_in) < .7).astype(np.uint8)
    x64, W64 = x.astype(np.float64), W.astype(np.float64)
    z = x64 @ W64 + b; a_w = np.where(z > 0, z, .1 * z) * mask
    dW_w = x64.T @ dz.astype(np.float64); db_w = dz.astype(np.float64).sum(0)
    dx_w = (dz.astype(np.float64) @ W64.T) * np.where(pa > 0, 1., np.where(pa < 0, .1, 1.1)) * pm
    xd, Wd, bd, md, dzd, pad, pmd = [ctx.array(v) for v in (x, W, b, mask, dz, pa, pm)]
    a, dW, db, dx = ctx.empty((B, n_out)), ctx.empty((n_in, n_out)), ctx.empty((n_out,)), ctx.empty((B, n_in))
    ws = ctx.empty(((lib.tn_fc_wgrad_ws_bytes(B, n_in, n_out) + 3) // 4,))
    res = {}
    for mode in (0, 1):
        ctx.call("tn_set_fc_matmul", mode)
        f = lambda: ctx.call("tn_fc_fwd", xd.ptr, Wd.ptr, bd.ptr, a.ptr, B, n_in, n_out, _lib.TN_ACT_LEAKY, .1, md.ptr)
        g = lambda: ctx.call("tn_fc_bwd", xd.ptr, dzd.ptr, Wd.ptr, dW.ptr, db.ptr, dx.ptr, B, n_in, n_out, ws.ptr, pad.ptr, _lib.TN_ACT_LEAKY, .1, pmd.ptr)
        f(); g()
        e = [np.abs(a.get_value() - a_w).max() / np.abs(a_w).max(), np.abs(dW.get_value() - dW_w).max() / np.abs(dW_w).max(),
             np.abs(db.get_value() - db_w).max() / np.abs(db_w).max(), np.abs(dx.get_value() - dx_w).max() / np.abs(dx_w).max()]
        res[mode] = (e, timeit(f), timeit(g))
    ctx.call("tn_set_fc_matmul", 0)
    for mode in (0, 1):
        e, tf, tb = res[mode]
        print("B%d %d->%d  %s: fwd %.1e dW %.1e db %.1e dx %.1e   us: fwd %.1f bwd %.1f" % (
            B, n_in, n_out, "bf16x3" if mode else "fp32  ", e[0], e[1], e[2], e[3], tf, tb))
    ok_all &= all(v < 5e-6 for v in res[1][0])
print("ALL OK" if ok_all else "FAILURES")
