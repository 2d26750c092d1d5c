import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from gpu_util import ctx, dev, empty, call, act_code
N = int(os.environ.get("OPBENCH_B", 4096)); C, H, K, f = 4, 13, 20, 3
Ho = 11; Hp = 6
rng = np.random.RandomState(1)
x = dev(rng.randn(N, C, H, H).astype(np.float32)); W = dev(rng.randn(K, C, f, f).astype(np.float32))
b = dev(rng.randn(K).astype(np.float32)); g = dev(rng.randn(N, K, Hp, Hp).astype(np.float32))
kind, prm = act_code("relu05")
dx, dW, db = empty((N, C, H, H)), empty((K, C, f, f)), empty((K,))
geom = (N, C, H, H, K, f, 0, Ho, Ho, 2, Hp, Hp, kind, prm)
y = empty((N, K, Hp, Hp)); m = empty((N, K, Hp, Hp), np.uint8)
call("tn_convpool_fwd_mask", x.ptr, W.ptr, b.ptr, y.ptr, m.ptr, *geom)
for i in range(3):
    db.fill_bytes(0)
    if os.environ.get("MASK", "1") == "1":
        call("tn_convblock_bwd_mask", x.ptr, W.ptr, g.ptr, y.ptr, m.ptr, dx.ptr, dW.ptr, db.ptr, *geom)
    else:
        call("tn_convblock_bwd", x.ptr, W.ptr, b.ptr, g.ptr, dx.ptr, dW.ptr, db.ptr, *geom)
    t = db.get_value()
    print("stamps (cycles):", [int(v) for v in t if v > 0])
    print("deltas:", [int(v) for v in np.diff(np.concatenate([[0], t[t > 0]]))])
