#!/usr/bin/env python
"""Cycle stamps of the LDS-DMA GEMM (TN_GEMM_DBG=1): where a wave's life goes.  python tools/dbg_gemm.py B n_in n_out [fwd|dgrad|wgrad]"""
import ctypes, os, sys
os.environ["TN_GEMM_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from theanet_amd.device import get_context
B, n_in, n_out = [int(v) for v in sys.argv[1:4]]
op = sys.argv[4] if len(sys.argv) > 4 else "fwd"
ctx = get_context(); lib = ctx.lib
rng = np.random.default_rng(0)
x = ctx.array(rng.standard_normal((B, n_in)).astype(np.float32)); W = ctx.array(rng.standard_normal((n_in, n_out)).astype(np.float32) * .01)
b = ctx.zeros((n_out,)); a = ctx.empty((B, n_out)); dz = ctx.array(rng.standard_normal((B, n_out)).astype(np.float32))
dW = ctx.empty((n_in, n_out)); db = ctx.empty((n_out,)); dx = ctx.empty((B, n_in))
ws = ctx.empty(((lib.tn_fc_wgrad_ws_bytes(B, n_in, n_out) + 3) // 4,))
for it in range(int(os.environ.get("ITERS", 3))):      # (ITERS=2000: stamps of a launch in a sustained run)
    if op == "fwd": ctx.call("tn_fc_fwd", x.ptr, W.ptr, b.ptr, a.ptr, B, n_in, n_out, 1, 0.1, None)
    elif op == "dgrad": ctx.call("tn_fc_dgrad", dz.ptr, W.ptr, dx.ptr, B, n_in, n_out, None, 0, 0.0, None)
    else: ctx.call("tn_fc_wgrad", x.ptr, dz.ptr, dW.ptr, db.ptr, B, n_in, n_out, ws.ptr)
ctx.sync()
nrec = 65536
buf = np.zeros((nrec, 8), np.uint64)
lib.tn_gemm_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
rc = lib.tn_gemm_dbg_read(ctx.h, ctypes.c_void_p(buf.ctypes.data), ctypes.c_int(nrec))
assert rc == 0, rc
buf = buf[buf[:, 0] > 0].astype(np.int64)
print("%s %d x %d x %d: waves stamped %d" % (op, B, n_in, n_out, len(buf)))
life = buf[:, 0]
for nm, v in (("wave life", life), ("tile loop", buf[:, 4]), ("DMA wait (vmcnt)", buf[:, 1]), ("barrier wait", buf[:, 2]), ("prologue", buf[:, 3])):
    print("%-38s cycles: median %8d  p10 %8d  p90 %8d  (%.0f %% of life)" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90), 100.0 * np.median(v) / np.median(life)))
w0 = buf[:, 6]; w1 = buf[:, 5]
print("wall clock (100 MHz ticks): kernel span %d = %.1f us, wave life median %d -> %.2f GHz; start spread p50 %d p90 %d max %d ticks" % (
    w1.max() - w0.min(), (w1.max() - w0.min()) / 100.0, np.median(w1 - w0), np.median(life) / np.median(w1 - w0) / 10.0,
    *np.percentile(w0 - w0.min(), [50, 90, 100])))
