import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from gpu_util import ctx, dev, empty, call, act_code
import oracle.theanet_oracle as O
N, C, H, K, f = 2, 4, 13, 20, 3
pad_lo, _, Ho = O.conv_geometry(H, f, 1, "valid")
rng = np.random.RandomState(1)
x = rng.randn(N, C, H, H).astype(np.float32)
W = (rng.randn(K, C, f, f)).astype(np.float32)
b = rng.randn(K).astype(np.float32)
Hp = (Ho + 1) // 2
g = rng.randn(N, K, Hp, Hp).astype(np.float32)
fa, dfa = O.activation("relu05")
x64, W64, b64 = x.astype(np.float64), W.astype(np.float64), b.astype(np.float64)
z = O.conv2d_fwd(x64, W64, b64, 1, "valid")
dz_w = O.pool_bwd(fa(z), g.astype(np.float64), 2, False) * dfa(z)
dx_w, dW_w, db_w = O.conv2d_bwd(x64, W64, dz_w, 1, "valid")
kind, prm = act_code("relu05")
dx, dW, db = empty(x.shape), empty(W.shape), empty((K,))
geom = (N, C, H, H, K, f, pad_lo, Ho, Ho, 2, Hp, Hp, kind, prm)
call("tn_convblock_bwd", dev(x).ptr, dev(W).ptr, dev(b).ptr, dev(g).ptr, dx.ptr, dW.ptr, db.ptr, *geom)
got = dx.get_value()
err = np.abs(got - dx_w)
np.set_printoptions(precision=3, suppress=True, linewidth=200)
print("dW err", np.abs(dW.get_value() - dW_w).max(), "db err", np.abs(db.get_value() - db_w).max())
print("err by channel", err.max(axis=(0, 2, 3)))
print("err img0 ch0\n", err[0, 0])
print("got img0 ch0\n", got[0, 0])
print("want img0 ch0\n", dx_w[0, 0])
