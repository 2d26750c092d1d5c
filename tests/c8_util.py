"""Host-side helpers of the fp16-resident ("c8") tests: the layout and a numpy statement of the mode's arithmetic
(operands are halfs, products exact, float64 sums standing in for the fp32 accumulation, one rounding on store)."""
import numpy as np


def r16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float64)


def to_c8(x):
    """(N, C, H, W) float -> raw uint16 image of the c8 tensor [N][C8][H][W][8]."""
    x = np.asarray(x, np.float32)
    N, C, H, W = x.shape
    C8 = (C + 7) // 8
    buf = np.zeros((N, C8 * 8, H, W), np.float16)
    buf[:, :C] = x.astype(np.float16)
    return np.ascontiguousarray(buf.reshape(N, C8, 8, H, W).transpose(0, 1, 3, 4, 2))


def from_c8(raw, C):
    """[N][C8][H][W][8] halfs -> (N, C, H, W) float32."""
    N, C8, H, W, _ = raw.shape
    return raw.transpose(0, 1, 4, 2, 3).reshape(N, C8 * 8, H, W)[:, :C].astype(np.float32)


def conv_same(x, W):
    """z[n,k,i,j] = sum x[n,c,i+u-1,j+v-1] W[k,c,2-u,2-v] (true convolution, 'same'), float64."""
    N, C, H, Wd = x.shape
    K = W.shape[0]
    xp = np.zeros((N, C, H + 2, Wd + 2))
    xp[:, :, 1:-1, 1:-1] = x
    z = np.zeros((N, K, H, Wd))
    for u in range(3):
        for v in range(3):
            z += np.einsum("nchw,kc->nkhw", xp[:, :, u:u + H, v:v + Wd], W[:, :, 2 - u, 2 - v])
    return z


def conv_same_dgrad(dz, W):
    """dx[n,c,y,x] = sum_k,u,v dz[n,k,y-u+1,x-v+1] W[k,c,2-u,2-v]."""
    N, K, H, Wd = dz.shape
    C = W.shape[1]
    dp = np.zeros((N, K, H + 2, Wd + 2))
    dp[:, :, 1:-1, 1:-1] = dz
    dx = np.zeros((N, C, H, Wd))
    for u in range(3):
        for v in range(3):
            # i = y - u + 1  ->  padded index y + 2 - u
            dx += np.einsum("nkhw,kc->nchw", dp[:, :, 2 - u:2 - u + H, 2 - v:2 - v + Wd], W[:, :, 2 - u, 2 - v])
    return dx


def conv_same_wgrad(x, dz):
    N, C, H, Wd = x.shape
    K = dz.shape[1]
    xp = np.zeros((N, C, H + 2, Wd + 2))
    xp[:, :, 1:-1, 1:-1] = x
    dW = np.zeros((K, C, 3, 3))
    for u in range(3):
        for v in range(3):
            dW[:, :, 2 - u, 2 - v] = np.einsum("nkhw,nchw->kc", dz, xp[:, :, u:u + H, v:v + Wd])
    return dW


def leaky(z, s):
    return np.maximum(0, z) + np.minimum(0, z) * s


def leaky_grad_from_out(a, s):
    return np.where(a > 0, 1.0, np.where(a < 0, s, 1.0 + s if s > 0 else 0.0))


def pool2(a):
    """2x2 max-pool + the mask byte of the c8 kernels."""
    N, K, H, W = a.shape
    w = a.reshape(N, K, H // 2, 2, W // 2, 2)
    m = w.max(axis=(3, 5))
    bits = np.zeros(m.shape, np.uint8)
    for di in range(2):
        for dj in range(2):
            bits |= ((w[:, :, :, di, :, dj] == m).astype(np.uint8) << (2 * di + dj))
    bits |= (m > 0).astype(np.uint8) << 4
    bits |= (m < 0).astype(np.uint8) << 5
    return m, bits


def unpool_dz(g, bits):
    """dz of a pooled block from the pooled gradient (which already carries act'(pooled output)) and the mask, as the
    kernels form it: every window element that attained the maximum receives the pooled gradient."""
    N, K, Hp, Wp = g.shape
    dz = np.zeros((N, K, Hp, 2, Wp, 2))
    for di in range(2):
        for dj in range(2):
            sel = (bits >> (2 * di + dj)) & 1
            dz[:, :, :, di, :, dj] = np.where(sel, g, 0.0)
    return dz.reshape(N, K, 2 * Hp, 2 * Wp)
