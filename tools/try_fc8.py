"""Quick GPU check + timing of the fp16-resident FC products (theanet_amd/csrc/fc_c8.hip) against numpy (development aid)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from theanet_amd import _lib
from theanet_amd.device import get_context

ctx = get_context(); lib = ctx.lib
rng = np.random.RandomState(0)
r16 = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float64)


def timeit(fn, iters=20):
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


def rowmap(C, HW):
    C8 = (C + 7) // 8
    k = np.arange(C8 * HW * 8)
    cell, e = k >> 3, k & 7
    o, p = cell // HW, cell % HW
    ch = o * 8 + e
    return np.where(ch < C, ch * HW + p, -1)


ok_all = True
gs = 1024.
ctx.call("tn_set_matmul_dtype", 1, gs)
for (B, C, HW, N, big) in [(5, 16, 4, 32, 0), (37, 24, 16, 96, 0), (128, 40, 8, 160, 0), (200, 64, 1, 64, 0), (128, 256, 64, 1024, 1), (2048, 128, 16, 512, 1)]:
    rm = rowmap(C, HW); Kc = len(rm); n_in = C * HW
    x = np.zeros((B, Kc)); x[:, rm >= 0] = r16(rng.randn(B, n_in))[:, rm[rm >= 0]]
    W = (rng.randn(n_in, N) / np.sqrt(n_in)).astype(np.float32); b = (rng.randn(N) * .1).astype(np.float32)
    mask = (rng.rand(B, N) < .5).astype(np.uint8)
    Wp = np.zeros((Kc, N)); Wp[rm >= 0] = r16(W)[rm[rm >= 0]]
    z = x @ Wp + b
    a_w = np.where(z > 0, z, .1 * z) * mask
    dx_ = ctx.array(x.astype(np.float16).view(np.uint16)); dW_, db_, dm = ctx.array(W), ctx.array(b), ctx.array(mask)
    a_d = ctx.empty((B, N))
    f = lambda: ctx.call("tn_c8_fc_fwd", dx_.ptr, dW_.ptr, db_.ptr, a_d.ptr, B, C, HW, N, _lib.TN_ACT_LEAKY, .1, dm.ptr)
    f(); err_f = np.abs(a_d.get_value() - a_w).max() / np.abs(a_w).max()
    dz = (rng.randn(B, N) * 1e-3).astype(np.float32)
    dz16 = r16(gs * dz)
    y = r16(rng.randn(B, Kc))
    dxw = (dz16 @ Wp.T) * np.where(y > 0, 1., np.where(y < 0, .1, 1.1))
    ddz, dy = ctx.array(dz), ctx.array(y.astype(np.float16).view(np.uint16))
    dxo = ctx.empty((B, Kc), np.uint16)
    d = lambda: ctx.call("tn_c8_fc_dgrad", ddz.ptr, dW_.ptr, dxo.ptr, B, C, HW, N, dy.ptr, _lib.TN_ACT_LEAKY, .1)
    d(); got = dxo.get_value().view(np.float16).astype(np.float64)
    err_d = np.abs(got - r16(dxw))[:, rm >= 0].max() / np.abs(dxw).max()
    dWw = np.zeros((n_in, N)); dWw[rm[rm >= 0]] = (x.T @ dz16)[rm >= 0] / gs
    dbw = dz16.sum(0) / gs
    gW, gb = ctx.zeros((n_in, N)), ctx.zeros((N,))
    w = lambda: ctx.call("tn_c8_fc_wgrad", dx_.ptr, ddz.ptr, gW.ptr, gb.ptr, B, C, HW, N)
    w(); err_w = np.abs(gW.get_value() - dWw).max() / np.abs(dWw).max(); err_b = np.abs(gb.get_value() - dbw).max() / np.abs(dbw).max()
    line = "B%d C%d HW%d N%d: fwd %.2e dgrad %.2e wgrad %.2e db %.2e" % (B, C, HW, N, err_f, err_d, err_w, err_b)
    if big:
        line += "   us: fwd %.1f dgrad %.1f wgrad %.1f" % (timeit(f), timeit(d), timeit(w))
    print(line)
    ok_all &= err_f < 2e-5 and err_d < 1e-3 and err_w < 2e-5 and err_b < 2e-5
ctx.call("tn_set_matmul_dtype", 0, 1.0)
print("ALL OK" if ok_all else "FAILURES")
