"""Timing probe: one-step-at-a-time training step replayed as a HIP graph vs enqueued eagerly (numbers only: the
captured step freezes the elastic field's ping-pong buffers, so the replayed results are not meaningful)."""
import copy, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TN_PIPELINE"] = "0"
import numpy as np
from bench import load_prms, synthetic
from theanet_amd import NeuralNet
from theanet_amd.device import get_context

for B in (20, 128, 512, 2048):
    prms = load_prms("mnist.prms")
    prms["layers"][0][1]["img_sz"] = 28
    tr = prms["training_params"]; tr["SEED"] = 555555; tr["BATCH_SZ"] = B
    x, y = synthetic(8 * B, 1, 28)
    net = NeuralNet(copy.deepcopy(prms["layers"]), dict(tr))
    fn = net.get_trin_model(x, y)
    ctx = get_context()
    for i in range(10): fn.enqueue(i % 8)
    ctx.sync()
    def timed(step, n=400, sync_each=False):
        for i in range(20): step(i % 8)
        ctx.sync(); t0 = time.perf_counter()
        for i in range(n):
            step(i % 8)
            if sync_each: ctx.sync()
        ctx.sync()
        return (time.perf_counter() - t0) / n * 1e6
    e_async, e_sync = timed(fn.enqueue), timed(fn.enqueue, sync_each=True)
    # capture one step
    seq = fn._seq if hasattr(fn, "_seq") and fn._seq is not None else fn
    slot = net.x
    slot.bind(seq.x_data); slot.row0, slot.d_row0, slot.row_global0 = 0, net.d_row0, net.shard_lo
    g = ctypes.c_void_p()
    ctx.call("tn_graph_begin")
    try:
        net._train_step(seq.y_data, 0, net.d_row0)
    finally:
        ctx.call("tn_graph_end", ctypes.byref(g))
    slot.d_row0 = None
    def replay(i):
        ctx.call("tn_set_i64", net.d_row0.ptr, i * B)
        ctx.call("tn_graph_launch", g)
    g_async, g_sync = timed(replay), timed(replay, sync_each=True)
    print("B=%5d  eager %6.1f us/step (sync every step %6.1f)   graph %6.1f (sync every step %6.1f)" %
          (B, e_async, e_sync, g_async, g_sync))
