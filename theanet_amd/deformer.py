"""Per-image elastic deformation on the GPU -- counterpart of the reference's
extras/deformer.py.

``transform`` (:7-18): trans = indices + scale*U(-1,1); each coordinate plane
gaussian-filtered (sigma, truncate 2, edge replicated); bilinear resample with
constant ``cval`` outside.  ``Deformer`` (:30-79) deforms a database IN PLACE,
batch by batch, and iterating it yields finished batch ids; the reference fans
batches out to worker processes over shared memory, here each batch is one HIP
kernel launch (one workgroup per image, float64 math in LDS like scipy).
The uniform noise comes from an on-device Philox stream keyed by (seed, global
image index) unless ``noise`` is injected (parity tests).
"""
import numpy as np

from .device import get_context


def transform_flat_batch(imgs, shape, scale, sigma, cval=0, noise=None, seed=0, img0=0):
    """imgs: (N, H*W) -> deformed (N, H*W) float32 (deformer.py:20-25)."""
    ctx = get_context()
    imgs = np.ascontiguousarray(imgs, np.float32)
    n = imgs.shape[0]
    h, w = shape
    d_in = ctx.array(imgs.reshape(n, h, w))
    d_out = ctx.empty((n, h, w))
    d_noise = ctx.array(np.ascontiguousarray(noise, np.float32)) if noise is not None else None
    ctx.call("tn_deformer_transform", d_in.ptr, d_out.ptr, n, h, w, float(scale), float(sigma),
             float(cval), d_noise.ptr if d_noise is not None else None, int(seed), int(img0))
    return d_out.get_value().reshape(n, h * w)


def transform(img, scale, sigma, cval=0, noise=None, seed=0):
    """Transforms a single 2D image (deformer.py:7-18)."""
    img = np.asarray(img)
    out = transform_flat_batch(img.reshape(1, -1), img.shape, scale, sigma, cval,
                               None if noise is None else np.asarray(noise)[None], seed)
    return out.reshape(img.shape)


def transform_inplace(imgs, *args, **kwargs):
    imgs[:] = transform_flat_batch(imgs, *args, **kwargs)


class Deformer(object):
    """Deform a database of input images in place; iterate to get finished batch ids."""

    def __init__(self, data, batch_sz, img_shape, scale, sigma,
                 cval=0.0, ncpus=None, seed=0):
        self.data = data
        self.batch_sz = batch_sz
        self.nBatches = data.shape[0] // batch_sz
        self.img_shape = img_shape
        self.scale, self.sigma, self.cval = scale, sigma, cval
        self.ncpus = 0          # kept for signature compatibility: the GPU does the work
        self.seed = seed
        self.ndone = 0

    def __str__(self):
        return ('Deformer: Input Shape {} batch_sz {} '
                'WH {} #Batches {} device MI355X '
                'Scale {} Sigma {} Background {} ').format(
            self.data.shape, self.batch_sz, self.img_shape,
            self.nBatches, self.scale, self.sigma, self.cval)

    def __iter__(self):
        for b in range(self.nBatches):
            rows = slice(b * self.batch_sz, (b + 1) * self.batch_sz)
            transform_inplace(self.data[rows], self.img_shape, self.scale, self.sigma, self.cval,
                              seed=self.seed, img0=b * self.batch_sz)
            self.ndone += 1
            yield b
