"""Conv / Pool / Mean layers -- host mirror of theanet/layer/convpool.py.
Same constructor arguments, shape rules and ``representation`` strings; the
compute is enqueued on the HIP backend (include/theanet_hip.h)."""
import math
import os

import numpy as np

from .. import _lib
from .layer import Layer, activation_by_name
from .weights import init_wb


class ConvLayer(Layer):
    def __init__(self, inpt, wts, rand_gen,
                 batch_sz, num_prev_maps, in_sz,
                 num_maps, filter_sz, stride,
                 mode='valid',
                 actvn='relu50',
                 reg=()):
        assert (wts is not None or rand_gen is not None)
        assert mode in ("valid", "full", "same")
        if mode == "full":
            raise NotImplementedError(
                "ConvLayer mode 'full': the reference computes out_sz = in+f+1 "
                "(convpool.py:63-64), which no backend can honour")

        filter_shape = (num_maps, num_prev_maps, filter_sz, filter_sz)
        fan_in = num_prev_maps * filter_sz * filter_sz
        fan_out = num_maps * filter_sz * filter_sz
        self.W, self.b = init_wb(wts, rand_gen, filter_shape, (filter_shape[0], ),
                                 fan_in, fan_out, actvn, 'Conv')

        if mode == 'same':
            assert stride == 1, "For Same mode stride should be 1"
            shift = (filter_sz - 1) // 2
            self.pad_lo, pad_hi = filter_sz - 1 - shift, shift
            self.out_sz = in_sz
        else:
            self.pad_lo, pad_hi = 0, 0
            self.out_sz = in_sz - filter_sz + 1
        self.out_sz //= stride
        theano_out = (in_sz + self.pad_lo + pad_hi - filter_sz) // stride + 1
        assert self.out_sz == theano_out, (
            "stride {} must divide the stride-1 output size {}".format(
                stride, in_sz + self.pad_lo + pad_hi - filter_sz + 1))

        self.act = activation_by_name(actvn)
        assert self.act.kind is not None, "softmax is not a conv activation"
        self.ctx = self.W.ctx
        # DTYPE 'float16' (NeuralNet training param): fp16 operands / fp32 accumulation on the matrix
        # cores.  Every conv product of the net runs that way or construction fails -- no silent fp32.
        self.f16 = self.ctx.mm_dtype == "float16"
        if self.f16:
            bits = self.ctx.lib.tn_conv_f16_supported(batch_sz, num_prev_maps, in_sz, in_sz, num_maps, filter_sz,
                                                      stride, self.pad_lo, self.out_sz, self.out_sz)
            assert bits & 5 == 5, (
                "DTYPE float16 needs 3x3 stride-1 'same' conv layers on power-of-two maps of 8..64 pixels "
                "(got {}->{} maps, {}x{} {} filter {} stride {}: forward {}, weight gradient {})".format(
                    num_prev_maps, num_maps, in_sz, in_sz, mode, filter_sz, stride, bool(bits & 1), bool(bits & 4)))
            self._f16_dgrad = bool(bits & 2)
        self.inpt = inpt
        self.batch_sz, self.num_prev_maps, self.in_sz = batch_sz, num_prev_maps, in_sz
        self.filter_sz, self.stride = filter_sz, stride
        self.output = self.ctx.empty((batch_sz, num_maps, self.out_sz, self.out_sz))
        self.gin = None
        self.fused_pool = None     # set by NeuralNet: conv+act+pool run as ONE kernel
        self._tile_pool = False    # ... on the LDS-tile matrix-core kernels (wide layers)
        self._mask_block = None
        self.dz = None

        self.params = [self.W, self.b]
        self.num_maps = num_maps
        self.mode = mode
        self.n_out = num_maps * self.out_sz ** 2
        self.reg = {"L1": 0, "L2": 0,
                    "momentum": .95,
                    "rate": 1,
                    "maxnorm": 0, }
        self.reg.update(reg)

        self.args = (batch_sz, num_prev_maps, in_sz, num_maps, filter_sz,
                     stride, mode, actvn, reg)
        self.representation = (
            "Conv Maps:{:2d} Filter:{} Stride:{} Mode:{} Output:{:2d} "
            "Act:{}\n\t  L1:{L1} L2:{L2} Momentum:{momentum} Rate:{rate} Max Norm:{maxnorm}"
            "".format(num_maps, filter_sz, stride, mode, self.out_sz,
                      actvn, **self.reg))

    def TestVersion(self, inpt):
        return ConvLayer(inpt, (self.W, self.b), None, *self.args)

    def act_info(self):
        return self.output, self.act.kind, self.act.prm, None

    def _geom(self):
        return (self.batch_sz, self.num_prev_maps, self.in_sz, self.in_sz, self.num_maps,
                self.filter_sz, self.stride, self.pad_lo, self.out_sz, self.out_sz)

    def can_fuse_with(self, pool):
        """conv -> act -> 2x2 max-pool on small channel counts runs as one fused kernel pair
        (tn_convpool_fwd / tn_convpool_bwd): the conv activation never reaches HBM."""
        if self.f16:
            self._tile_pool = bool(self.ctx.lib.tn_convpool_f16_supported(
                self.batch_sz, self.num_prev_maps, self.in_sz, self.in_sz, self.num_maps, self.filter_sz,
                self.stride, self.pad_lo, self.out_sz, self.out_sz, pool.pool_sz, pool.out_sz, pool.out_sz))
            return self._tile_pool
        if self.stride == 1 and self.ctx.lib.tn_convpool_supported(
                self.num_prev_maps, self.filter_sz, self.stride, pool.pool_sz):
            return True
        # wide 3x3 'same' layers: the LDS-tile matrix-core kernels pool in their epilogue and run the
        # backward from the pooling mask (tn_convpool_fwd_mask / tn_convpool_bwd_mask_dx)
        self._tile_pool = bool(self.ctx.lib.tn_convpool_tile_supported(
            self.batch_sz, self.num_prev_maps, self.in_sz, self.in_sz, self.num_maps, self.filter_sz,
            self.stride, self.pad_lo, self.out_sz, self.out_sz, pool.pool_sz, pool.out_sz, pool.out_sz))
        return self._tile_pool

    def mask_backward_supported(self, pool):
        """True if the fused block's whole backward (dW, db and the input gradient) can run as
        the one-kernel matrix-core variant tn_convblock_bwd_mask."""
        if self._mask_block is None:
            self._mask_block = bool(self.stride == 1 and self.ctx.lib.tn_convblock_mask_supported(
                self.num_prev_maps, self.num_maps, self.filter_sz, self.stride, pool.pool_sz,
                self.in_sz, self.in_sz, self.pad_lo, self.out_sz, self.out_sz, pool.out_sz,
                pool.out_sz))
        return self._mask_block

    def _fused_geom(self):
        pool = self.fused_pool
        return (self.batch_sz, self.num_prev_maps, self.in_sz, self.in_sz, self.num_maps,
                self.filter_sz, self.pad_lo, self.out_sz, self.out_sz, pool.pool_sz,
                pool.out_sz, pool.out_sz, self.act.kind, self.act.prm)

    def forward(self, train=True):
        if self.fused_pool is not None:
            return                       # the pool layer launches the fused kernel
        self.ctx.call("tn_conv2d_fwd", self.inpt.ptr, self.W.ptr, self.b.ptr, self.output.ptr,
                      *self._geom(), self.act.kind, self.act.prm)

    def _backward_fused(self, gpool, need_gin, below):
        """gpool = d cost / d (pooled output).  One kernel recomputes the windows, routes the
        gradient through max-pool and activation and reduces dW/db; dz is only materialised
        when the layer below needs a gradient."""
        pool = self.fused_pool
        if self._tile_pool:
            # wide block: dW, db and the input gradient straight from the pooled gradient + mask
            assert pool.mask is not None
            b_out, b_act, b_prm, b_mask = below.act_info() if (need_gin and below is not None) \
                else (None, 0, 0., None)
            assert b_mask is None
            if need_gin and self.gin is None:
                self.gin = self.ctx.empty(self.inpt.shape)
            upd = self.has_updates()
            self.ctx.call("tn_convpool_bwd_mask_dx", self.inpt.ptr, self.W.ptr, gpool.ptr, pool.output.ptr,
                          pool.mask.ptr, self.gin.ptr if need_gin else None,
                          self.grads[0].ptr if upd else None, self.grads[1].ptr if upd else None,
                          *self._fused_geom(),
                          b_out.ptr if b_out is not None and b_act != _lib.TN_ACT_LINEAR else None,
                          b_act, b_prm)
            self._gin_done = True
            return self.gin if need_gin else None
        if pool.mask is not None:
            # the forward recorded where every pooled value came from: no conv recompute
            b_out, b_act, b_prm, b_mask = below.act_info() if below is not None else (None, 0, 0., None)
            fuse_below = need_gin and b_out is not None and b_act != _lib.TN_ACT_LINEAR
            if self.mask_backward_supported(pool) and not fuse_below:
                # small maps: weight and input gradients as matrix-core products over an
                # LDS-resident dz, one kernel
                if need_gin and self.gin is None:
                    self.gin = self.ctx.empty(self.inpt.shape)
                self.ctx.call("tn_convblock_bwd_mask", self.inpt.ptr, self.W.ptr, gpool.ptr,
                              pool.output.ptr, pool.mask.ptr, self.gin.ptr if need_gin else None,
                              self.grads[0].ptr, self.grads[1].ptr, *self._fused_geom())
                self._gin_done = True
                return self.gin if need_gin else None
            self._gin_done = False
            if need_gin and self.dz is None:
                self.dz = self.ctx.empty(self.output.shape)
            self.ctx.call("tn_convpool_bwd_mask", self.inpt.ptr, gpool.ptr, pool.output.ptr,
                          pool.mask.ptr, self.dz.ptr if need_gin else None, self.grads[0].ptr,
                          self.grads[1].ptr, *self._fused_geom())
            return self.dz if need_gin else None
        if self.ctx.lib.tn_convblock_supported(self.num_prev_maps, self.num_maps, self.filter_sz,
                                               self.stride, pool.pool_sz, self.out_sz, self.out_sz):
            # LDS-resident variant: dW/db AND the gradient w.r.t. the input in one kernel
            b_out, b_act, b_prm, b_mask = below.act_info() if below is not None else (None, 0, 0., None)
            if not (need_gin and b_out is not None and b_act != _lib.TN_ACT_LINEAR):
                if need_gin and self.gin is None:
                    self.gin = self.ctx.empty(self.inpt.shape)
                self.ctx.call("tn_convblock_bwd", self.inpt.ptr, self.W.ptr, self.b.ptr, gpool.ptr,
                              self.gin.ptr if need_gin else None, self.grads[0].ptr,
                              self.grads[1].ptr, *self._fused_geom())
                self._gin_done = True
                return self.gin if need_gin else None
        self._gin_done = False
        if need_gin and self.dz is None:
            self.dz = self.ctx.empty(self.output.shape)
        self.ctx.call("tn_convpool_bwd", self.inpt.ptr, self.W.ptr, self.b.ptr, gpool.ptr,
                      self.dz.ptr if need_gin else None, self.grads[0].ptr, self.grads[1].ptr,
                      *self._fused_geom())
        return self.dz if need_gin else None

    def backward(self, gout, need_gin, below):
        """gout = d cost / d z of this layer (activation gradient already fused in)."""
        if self.fused_pool is not None:
            gout = self._backward_fused(gout, need_gin, below)
            if not need_gin:
                return None
            if self._gin_done:
                return gout
        elif self.has_updates():
            self.ctx.call("tn_conv2d_wgrad", self.inpt.ptr, gout.ptr, self.grads[0].ptr,
                          self.grads[1].ptr, *self._geom())
        if not need_gin:
            return None
        if self.gin is None:
            self.gin = self.ctx.empty(self.inpt.shape)
        b_out, b_act, b_prm, b_mask = below.act_info()
        assert b_mask is None
        self.ctx.call("tn_conv2d_dgrad", gout.ptr, self.W.ptr, self.gin.ptr, *self._geom(),
                      b_out.ptr if b_out is not None and b_act != _lib.TN_ACT_LINEAR else None,
                      b_act, b_prm)
        return self.gin


class PoolLayer(Layer):
    def __init__(self, inpt, num_maps, in_sz, pool_sz, ignore_border=False):
        """Max-pool, stride = window.  ignore_border=False keeps the partial last
        window: (5,5) with pool 2 -> (3,3); True -> (2,2) (convpool.py:98-112)."""
        if ignore_border:
            self.out_sz = in_sz // pool_sz
        else:
            self.out_sz = math.ceil(in_sz / pool_sz)

        self.ctx = inpt.ctx
        self.params = []
        self.inpt = inpt
        self.num_maps = num_maps
        self.in_sz, self.pool_sz = in_sz, pool_sz
        self.ignore_border = ignore_border
        self.args = (num_maps, in_sz, pool_sz, ignore_border)
        self.n_out = num_maps * self.out_sz ** 2
        self.batch_sz = inpt.shape[0]
        self.output = self.ctx.empty((self.batch_sz, num_maps, self.out_sz, self.out_sz))
        self.gin = None
        self.fused_conv = None
        self.mask = None           # uint8 pooling mask of the fused forward (training graphs only)
        self.fused_elastic = None  # ElasticLayer whose resampling this block's forward performs
        self.representation = (
            "Pool Maps:{:2d} Pool_sz:{} Border:{} Output:{:2d}"
            "".format(num_maps, pool_sz,
                      "Ignore" if ignore_border else "Keep",
                      self.out_sz))

    def TestVersion(self, inpt):
        return PoolLayer(inpt, *self.args)

    def forward(self, train=True):
        conv = self.fused_conv
        if conv is not None:
            if train and self.mask is None and conv.filter_sz == 3 and self.pool_sz == 2 and \
                    (conv._tile_pool or os.environ.get("TN_POOL_MASK", "1") != "0"):
                self.mask = self.ctx.empty(self.output.shape, np.uint8)
            el = self.fused_elastic
            if el is not None and train and el._apply_args is not None:
                # ElasticLayer -> conv -> act -> pool in one launch (the resampled image is still
                # written to el.output for the backward pass)
                a = el._apply_args
                g = conv._fused_geom()          # (N, C, H, W, K, f, pad, Ho, Wo, p, Hp, Wp, act, prm)
                self.ctx.call("tn_elastic_convpool_fwd_mask", a[0], a[1], a[2], a[3], a[4], a[6], a[7],
                              *a[8:], conv.W.ptr, conv.b.ptr, self.output.ptr,
                              self.mask.ptr if self.mask is not None else None, g[4], g[5], g[6], g[7],
                              g[8], g[9], g[10], g[11], g[12], g[13])
                return
            self.ctx.call("tn_convpool_fwd_mask", conv.inpt.ptr, conv.W.ptr, conv.b.ptr,
                          self.output.ptr, self.mask.ptr if (train and self.mask is not None) else None,
                          *conv._fused_geom())
            return
        self.ctx.call("tn_pool_fwd", self.inpt.ptr, self.output.ptr,
                      self.batch_sz * self.num_maps, self.in_sz, self.in_sz, self.pool_sz,
                      self.out_sz, self.out_sz)

    def backward(self, gout, need_gin, below):
        if not need_gin:
            return None
        if self.fused_conv is not None:
            return gout               # the conv layer's fused backward consumes d cost / d y
        if self.gin is None:
            self.gin = self.ctx.empty(self.inpt.shape)
        b_out, b_act, b_prm, b_mask = below.act_info()
        assert b_mask is None
        # below.output IS self.inpt, so the activation gradient rides along for free
        self.ctx.call("tn_pool_bwd", self.inpt.ptr, self.output.ptr, gout.ptr, self.gin.ptr,
                      self.batch_sz * self.num_maps, self.in_sz, self.in_sz, self.pool_sz,
                      self.out_sz, self.out_sz, b_act, b_prm)
        return self.gin


class MeanLayer(Layer):
    def __init__(self, inpt, num_maps, in_sz):
        self.ctx = inpt.ctx
        self.params = []
        self.inpt = inpt
        self.num_maps = num_maps
        self.in_sz = in_sz
        self.out_sz = 1
        self.n_out = num_maps
        self.batch_sz = inpt.shape[0]
        self.output = self.ctx.empty((self.batch_sz, num_maps))
        self.gin = None
        self.representation = (
            "Mean Maps:{:2d} Output:{:2d}"
            "".format(num_maps, self.out_sz))

    def TestVersion(self, inpt):
        return MeanLayer(inpt, self.num_maps, self.in_sz)

    def forward(self, train=True):
        self.ctx.call("tn_mean_fwd", self.inpt.ptr, self.output.ptr,
                      self.batch_sz * self.num_maps, self.in_sz * self.in_sz)

    def backward(self, gout, need_gin, below):
        if not need_gin:
            return None
        if self.gin is None:
            self.gin = self.ctx.empty(self.inpt.shape)
        b_out, b_act, b_prm, b_mask = below.act_info()
        assert b_mask is None
        self.ctx.call("tn_mean_bwd", gout.ptr, self.gin.ptr, self.batch_sz * self.num_maps,
                      self.in_sz * self.in_sz,
                      b_out.ptr if b_out is not None and b_act != _lib.TN_ACT_LINEAR else None,
                      b_act, b_prm)
        return self.gin
