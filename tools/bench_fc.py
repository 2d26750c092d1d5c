#!/usr/bin/env python
"""Times the fully-connected ops at a given shape: python tools/bench_fc.py B n_in n_out"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theanet_amd.device import get_context
B, n_in, n_out = [int(v) for v in sys.argv[1:4]]
ctx = get_context(); lib = ctx.lib
rng = np.random.default_rng(0)
x = ctx.array(rng.standard_normal((B, n_in)).astype(np.float32)); W = ctx.array(rng.standard_normal((n_in, n_out)).astype(np.float32) * .01)
b = ctx.zeros((n_out,)); a = ctx.empty((B, n_out)); dz = ctx.array(rng.standard_normal((B, n_out)).astype(np.float32))
dW = ctx.empty((n_in, n_out)); db = ctx.empty((n_out,)); dx = ctx.empty((B, n_in))
ws = ctx.empty(((lib.tn_fc_wgrad_ws_bytes(B, n_in, n_out) + 3) // 4,))
def timeit(fn, iters=20):
    for _ in range(3): fn()
    ctx.sync()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.tn_event_create(ctx.h, ctypes.byref(e0)); lib.tn_event_create(ctx.h, ctypes.byref(e1))
    lib.tn_event_record(ctx.h, e0)
    for _ in range(iters): fn()
    lib.tn_event_record(ctx.h, e1)
    ms = ctypes.c_float(); ctx.call("tn_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    return ms.value * 1e3 / iters
fl = 2.0 * B * n_in * n_out
for name, fn in (("fwd", lambda: ctx.call("tn_fc_fwd", x.ptr, W.ptr, b.ptr, a.ptr, B, n_in, n_out, 1, 0.1, None)),
                 ("wgrad", lambda: ctx.call("tn_fc_wgrad", x.ptr, dz.ptr, dW.ptr, db.ptr, B, n_in, n_out, ws.ptr)),
                 ("dgrad", lambda: ctx.call("tn_fc_dgrad", dz.ptr, W.ptr, dx.ptr, B, n_in, n_out, None, 0, 0.0, None)),
                 ("bwd pair", lambda: ctx.call("tn_fc_bwd", x.ptr, dz.ptr, W.ptr, dW.ptr, db.ptr, dx.ptr, B, n_in, n_out, ws.ptr, None, 0, 0.0, None))):
    t = timeit(fn)
    print("%-9s %8.1f us  %6.1f TFLOP/s  (weight matrix %.0f MB)" % (name, t, fl * (2 if name == "bwd pair" else 1) / t / 1e6, n_in * n_out * 4 / 1e6))
