"""Conv / Pool / Mean layers -- host mirror of theanet/layer/convpool.py.
Same constructor arguments, shape rules and ``representation`` strings; the
compute is enqueued on the HIP backend (include/theanet_hip.h)."""
import math
import os

import numpy as np

from .. import _lib
from ..device import C8Array
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
        # DTYPE 'float16' (NeuralNet training param; BASELINE configs[4]): activations and gradients live in HBM as
        # halfs in the c8 layout (device.C8Array), fp16 MFMA operands / fp32 accumulation, fp32 master weights.  Every
        # conv layer of the net runs that way or construction fails -- no silent fp32 run of an unsupported shape.
        self.f16 = self.ctx.mm_dtype == "float16"
        self.inpt = inpt
        self.batch_sz, self.num_prev_maps, self.in_sz = batch_sz, num_prev_maps, in_sz
        self.filter_sz, self.stride = filter_sz, stride
        if self.f16:
            lib = self.ctx.lib
            ok = lib.tn_c8_conv_supported(batch_sz, num_prev_maps, in_sz, in_sz, num_maps, filter_sz, stride, self.pad_lo) and \
                lib.tn_c8_conv_wgrad_supported(batch_sz, num_prev_maps, in_sz, in_sz, num_maps)
            assert ok and mode == 'same', (
                "DTYPE float16 needs 3x3 stride-1 'same' conv layers with a multiple of 8 filters on maps of 8, 16, 32 "
                "or 64 pixels a side (got {}->{} maps, {}x{} {} filter {} stride {})".format(
                    num_prev_maps, num_maps, in_sz, in_sz, mode, filter_sz, stride))
            # the first conv layer of the net gets NCHW fp32 images: packed into a c8 tensor in front of the kernel
            self.x16 = None if getattr(inpt, "c8", None) else C8Array(self.ctx, batch_sz, num_prev_maps, in_sz, in_sz)
            self.output = C8Array(self.ctx, batch_sz, num_maps, self.out_sz, self.out_sz)
            # the weights as MFMA operand tiles (forward / input gradient): the net arranges every layer's in one launch
            # per step (NeuralNet._c8_arrange) and marks them valid until the next update; otherwise the ops do it per call
            self.wt_fwd = self.ctx.empty((lib.tn_c8_wt_elems(num_maps, num_prev_maps, 0),), np.uint16)
            self.wt_bwd = None
            self.wt_valid = False
        else:
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
            assert pool.pool_sz == 2 and self.out_sz % 2 == 0, "DTYPE float16: pooling layers are 2x2 on even maps"
            return True
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

    # -- DTYPE float16: the fp16-resident kernels (include/theanet_hip.h, tn_c8_*) --------------------------------
    _c8_prefilled = False

    def _c8_input(self, below=None):
        """The layer's input as a c8 tensor: the layer below's output, or -- first conv layer of the net -- the NCHW
        fp32 minibatch packed on the way in (straight from the dataset window when the layer below is an InputLayer)."""
        if self.x16 is None:
            return self.inpt
        if self._c8_prefilled:
            return self.x16             # written by the distortion stage below (tn_c8_elastic_apply)
        src, row0 = self.inpt, 0
        slot = getattr(self, "_pack_from", None)
        if slot is not None:
            src, row0 = slot.data, int(slot.row0)
        self.ctx.call("tn_c8_pack", src.ptr, row0, self.x16.ptr, self.batch_sz, self.num_prev_maps,
                      self.in_sz * self.in_sz, 1.0)
        return self.x16

    def _c8_forward(self, out, mask):
        x = self._c8_input()
        self.ctx.call("tn_c8_conv_fwd", x.ptr, self.W.ptr, self.b.ptr, out.ptr, mask.ptr if mask is not None else None,
                      self.batch_sz, self.num_prev_maps, self.in_sz, self.in_sz, self.num_maps, self.act.kind, self.act.prm,
                      1 if out is not self.output else 0, self.wt_fwd.ptr if self.wt_valid else None)

    def _c8_backward(self, gout, need_gin, below):
        """gout: d cost / d z of this layer as a c8 tensor carrying the gradient scale -- or, for a fused block, the
        gradient w.r.t. the POOLED output (act' already applied by its producer), dz being formed from it and the
        pooling mask inside the kernels."""
        pool = self.fused_pool
        pooled, mask = (1, pool.mask.ptr) if pool is not None else (0, None)
        x = self.x16 if self.x16 is not None else self.inpt
        geom = (self.batch_sz, self.num_prev_maps, self.in_sz, self.in_sz, self.num_maps)
        if self.has_updates():
            self.ctx.call("tn_c8_conv_wgrad", x.ptr, gout.ptr, self.grads[0].ptr, self.grads[1].ptr, *geom, pooled, mask)
        if not need_gin:
            return None
        assert self.x16 is None, "DTYPE float16: no trainable layer below the first conv layer"
        if self.gin is None:
            self.gin = C8Array(self.ctx, self.batch_sz, self.num_prev_maps, self.in_sz, self.in_sz)
        b_out, b_act, b_prm, b_mask = below.act_info()
        assert b_mask is None
        fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
        self.ctx.call("tn_c8_conv_dgrad", gout.ptr, self.W.ptr, self.gin.ptr, *geom, b_out.ptr if fuse else None,
                      b_act, b_prm, pooled, mask, self.wt_bwd.ptr if self.wt_valid and self.wt_bwd is not None else None)
        return self.gin

    def forward(self, train=True):
        if self.fused_pool is not None:
            return                       # the pool layer launches the fused kernel
        if self.f16:
            return self._c8_forward(self.output, None)
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
        if self.f16:
            return self._c8_backward(gout, need_gin, below)
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
        self.f16 = getattr(inpt, "c8", None) is not None      # DTYPE float16: pooled c8 tensor (only as a fused block)
        if self.f16:
            self.output = C8Array(self.ctx, self.batch_sz, num_maps, self.out_sz, self.out_sz)
        else:
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

    def act_info(self):
        """DTYPE float16: the gradient a pooled block receives carries act'(pooled output) (applied by whichever kernel
        produces it, from the block's stored output); fp32 blocks take the derivative from the pooling mask's sign
        bits inside their own backward kernels instead."""
        if self.f16:
            act = self.fused_conv.act
            return self.output, act.kind, act.prm, None
        return Layer.act_info(self)

    def forward(self, train=True):
        conv = self.fused_conv
        if self.f16:
            assert conv is not None, "DTYPE float16: a PoolLayer must directly follow a ConvLayer"
            if train and self.mask is None:
                self.mask = self.ctx.empty(self.output.shape, np.uint8)
            return conv._c8_forward(self.output, self.mask if train else None)
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
