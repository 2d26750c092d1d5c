"""Input layers -- host mirror of theanet/layer/inlayers.py.

``InputLayer`` passes the minibatch through; ``ElasticLayer`` is the in-graph
augmentation stage: invert -> one random coordinate field per CALL (translation,
gaussian-smoothed elastic displacement, zoom, rotation; shared by every image and
channel of the batch) -> clip -> nearest/bilinear resample -> pixel-flip noise
(:63-144).  The field is built on the device in float64 from a small vector of
random draws which either comes from the on-device Philox generator or is
injected by a parity test (``inject``).  The test version is identity + invert
(:157-163).
"""
import numpy as np

from .layer import Layer


class InputSlot:
    """Stand-in for the reference's symbolic ``x`` (neuralnet.py:79): a window of
    ``batch`` rows into a device-resident dataset, moved by the step functions."""

    def __init__(self, batch):
        self.batch = batch
        self.data = None        # DeviceArray (N, C, H, W)
        self.row0 = 0           # first row of the current minibatch (this rank's shard)
        self.row_global0 = 0    # same, as a global sample index (for sharding-proof RNG)
        self.d_row0 = None      # optional device-side offset (graph replay)

    def bind(self, data):
        self.data = data


class InputLayer(Layer):
    def __init__(self, inpt, img_sz, num_maps=1, rand_gen=None):
        self.params = []
        self.inpt = inpt
        self.out_sz = img_sz
        self.num_maps = num_maps
        self.n_out = self.num_maps * self.out_sz ** 2
        self.representation = \
            'Input Maps:{} Sizes Input:{:2d} Output:{:2d}'.format(num_maps,
                                                                  img_sz,
                                                                  img_sz)
        from ..device import get_context
        self.ctx = get_context()
        self.batch_sz = inpt.batch
        self.output = self.ctx.empty((self.batch_sz, num_maps, img_sz, img_sz))

    def TestVersion(self, inpt):
        return InputLayer(inpt, self.out_sz, self.num_maps)

    _packed_by_conv = False     # DTYPE float16: the first conv layer reads the dataset window itself (tn_c8_pack)

    def forward(self, train=True):
        s = self.inpt
        if self._packed_by_conv:
            assert s.d_row0 is None
            return
        self.ctx.call("tn_elastic_apply", s.data.ptr, int(s.row0),
                      s.d_row0.ptr if s.d_row0 is not None else None, self.output.ptr,
                      self.batch_sz, self.num_maps, self.out_sz, self.out_sz, 0, 1,
                      None, None, None, 0.0, None, 0, 0, None, int(s.row_global0))


class ElasticLayer(Layer):
    def __init__(self, inpt, img_sz,
                 num_maps=1,
                 translation=0,
                 zoom=1,
                 magnitude=0,
                 sigma=1,
                 pflip=0,
                 angle=0,
                 rand_gen=None,
                 invert_image=False,
                 nearest=False):
        self.inpt = inpt
        self.img_sz = img_sz
        self.translation = translation
        self.zoom = zoom
        self.magnitude = magnitude
        self.sigma = sigma
        self.pflip = pflip
        self.angle = angle
        self.invert = invert_image
        self.nearest = nearest

        self.out_sz = img_sz
        self.num_maps = num_maps
        self.n_out = self.num_maps * self.out_sz ** 2
        self.params = []
        self.representation = ('Elastic Maps:{:d} Size:{:2d} Translation:{:} '
                               'Zoom:{} Mag:{:d} Sig:{:d} Noise:{} '
                               'Angle:{} Invert:{} '
                               'Interpolation:{}'.format(
            self.num_maps, img_sz,
            translation, zoom, magnitude, sigma,
            pflip, angle, invert_image,
            'Nearest' if nearest else 'Linear'))

        from ..device import get_context
        self.ctx = ctx = get_context()
        self.batch_sz = inpt.batch if isinstance(inpt, InputSlot) else inpt.shape[0]
        self.output = ctx.empty((self.batch_sz, num_maps, img_sz, img_sz))

        assert zoom > 0
        self.active = bool(magnitude or translation or pflip or angle) or zoom != 1
        self.has_field = bool(magnitude or translation or angle) or zoom != 1
        self.d_step = None
        self._inj_draws = False
        self._inj_flip = None
        self.fused_conv = None          # set by NeuralNet: the first conv block resamples on the fly
        self._apply_args = None
        if not self.active:
            return

        # the stream seed consumes the seed chain exactly like inlayers.py:72-73
        self.seed = int(rand_gen.randint(1e6)) if rand_gen is not None \
            else int(np.random.randint(0, 1e6))
        h = w = img_sz
        n_draws = ctx.lib.tn_elastic_draws_count(h, w)
        self.draws = ctx.zeros((n_draws,))
        # two sample maps: while the minibatch of step t is resampled through one, the field of
        # step t+1 (it depends only on the step counter) is built into the other by a rider of the
        # backward pass / the update launch (NeuralNet._train_step)
        self._maps = [(ctx.empty((h * w,), np.int32), ctx.empty((h * w,)), ctx.empty((h * w,)),
                       ctx.empty((2, h, w), np.float64)) for _ in range(2)]
        self._cur = 0
        self._pre_valid = False
        self.map_idx, self.map_fy, self.map_fx, self.target = self._maps[0]

    def TestVersion(self, te_inpt):
        return ElasticLayer(te_inpt, self.img_sz,
                            num_maps=self.num_maps,
                            translation=0, zoom=1,
                            magnitude=0, sigma=1,
                            pflip=0, angle=0,
                            invert_image=self.invert,
                            nearest=self.nearest)

    # -- parity hooks -----------------------------------------------------------------
    def inject(self, transln=None, noise=None, origin_u=None, zoom_u=None, theta_u=None,
               flipmask=None):
        """Replace the device RNG by explicit draws (layout: include/theanet_hip.h).
        Call with no arguments to return to the generator."""
        if all(v is None for v in (transln, noise, origin_u, zoom_u, theta_u, flipmask)):
            self._inj_draws, self._inj_flip = False, None
            return
        h = w = self.img_sz
        d = np.zeros(self.draws.size, np.float32)
        if transln is not None:
            d[0:2] = np.asarray(transln, np.float32).reshape(2)
        if origin_u is not None:
            d[2:4] = np.asarray(origin_u, np.float32).reshape(2)
        if zoom_u is not None:
            d[4:6] = np.asarray(zoom_u, np.float32).reshape(2)
        if theta_u is not None:
            d[6] = np.float32(theta_u)
        if noise is not None:
            d[8:] = np.asarray(noise, np.float32).reshape(2 * h * w)
        self.draws.set_value(d)
        self._inj_draws = True
        if flipmask is not None:
            self._inj_flip = self.ctx.array(
                np.asarray(flipmask).reshape(self.output.shape).astype(np.uint8))

    _c8_consumer = None     # DTYPE float16: the first conv layer, whose c8 input this stage writes (NeuralNet._fuse)

    def forward(self, train=True):
        s = self.inpt
        if isinstance(s, InputSlot):
            x_ptr, row0, d_row0, rg0 = s.data.ptr, int(s.row0), s.d_row0, int(s.row_global0)
        else:       # mid-net elastic layer (neuralnet.py:132-142): plain device tensor
            x_ptr, row0, d_row0, rg0 = s.ptr, 0, None, 0
        h = w = self.img_sz
        d_row0_ptr = d_row0.ptr if d_row0 is not None else None
        if not self.active:
            self.ctx.call("tn_elastic_apply", x_ptr, row0, d_row0_ptr, self.output.ptr,
                          self.batch_sz, self.num_maps, h, w, int(self.invert), 1,
                          None, None, None, 0.0, None, 0, 0, None, rg0)
            return
        d_step_ptr = self.d_step.ptr if self.d_step is not None else None
        if self.has_field:
            self.map_idx, self.map_fy, self.map_fx, self.target = self._maps[self._cur]
            if self._pre_valid and not self._inj_draws:
                pass                # built by the previous step's closing launch (NeuralNet._train_step)
            else:
                if not self._inj_draws:      # draws generated inside the field launch
                    m = self._maps[self._cur]
                    self.ctx.call("tn_elastic_field_gen", self.draws.ptr, self.seed, 0, d_step_ptr,
                                  h, w, float(self.translation), float(self.zoom),
                                  float(self.magnitude), int(self.sigma), float(self.angle),
                                  int(self.nearest), m[0].ptr, m[1].ptr, m[2].ptr, m[3].ptr)
                else:
                    self._field(self._maps[self._cur])
            self._pre_valid = False
        # arguments of the resampling (tn_elastic_apply / the conv block it may be fused into)
        self._apply_args = (x_ptr, row0, d_row0_ptr, self.output.ptr, self.batch_sz, self.num_maps, h, w,
                            int(self.invert), int(self.nearest),
                            self.map_idx.ptr if self.has_field else None,
                            self.map_fy.ptr, self.map_fx.ptr,
                            float(self.pflip) if self._inj_flip is None else 0.0,
                            self._inj_flip.ptr if self._inj_flip is not None else None,
                            self.seed, 0, d_step_ptr, rg0)
        if self.fused_conv is not None and train:
            return                      # the conv block's forward resamples while it loads (PoolLayer)
        if self._c8_consumer is not None:
            # DTYPE float16: straight into the c8 tensor of the first conv layer (same values, rounded when stored)
            a = self._apply_args
            self.ctx.call("tn_c8_elastic_apply", *(a[:3] + (self._c8_consumer.x16.ptr,) + a[4:]))
            return
        self.ctx.call("tn_elastic_apply", *self._apply_args)

    def backward(self, gout, need_gin, below):
        """An ElasticLayer in the middle of a net (neuralnet.py:132-142): the gradient w.r.t. its input --
        the transposed gather, with the signs of the inversion and of the flip noise."""
        if not need_gin or isinstance(self.inpt, InputSlot):
            return None
        from .. import _lib
        b_out, b_act, b_prm, b_mask = below.act_info()
        if getattr(self, "gin", None) is None:
            self.gin = self.ctx.empty(self.inpt.shape)
        h = w = self.img_sz
        fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
        if self.active:
            a = self._apply_args
            self.ctx.call("tn_elastic_apply_bwd", gout.ptr, self.gin.ptr, self.batch_sz, self.num_maps, h, w,
                          a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18],
                          b_out.ptr if fuse else None, b_act, b_prm)
        else:
            self.ctx.call("tn_elastic_apply_bwd", gout.ptr, self.gin.ptr, self.batch_sz, self.num_maps, h, w,
                          int(self.invert), 1, None, None, None, 0.0, None, 0, 0, None, 0,
                          b_out.ptr if fuse else None, b_act, b_prm)
        if b_mask is not None:        # a Hidden layer with dropout right below: its mask, after the copy
            self.ctx.call("tn_scale_mask", self.gin.ptr, b_mask.ptr, 1.0, self.gin.ptr, self.gin.size,
                          None, _lib.TN_ACT_LINEAR, 0.0)
        return self.gin

    def _field(self, m):
        h = w = self.img_sz
        self.ctx.call("tn_elastic_field", self.draws.ptr, h, w, float(self.translation),
                      float(self.zoom), float(self.magnitude), int(self.sigma),
                      float(self.angle), int(self.nearest), m[0].ptr, m[1].ptr, m[2].ptr, m[3].ptr)

    @property
    def debugout(self):
        """[output, displacement field] like inlayers.py:145-147 (host copies)."""
        if not self.active or not self.has_field:
            return [self.output.get_value(), np.zeros(2)]
        h = w = self.img_sz
        return [self.output.get_value(), self.target.get_value() - np.indices((h, w))]


class ColorLayer(Layer):
    """theanet/layer/color.py:9-52 -- per-(image, channel) colour balance and gamma jitter:
    ``out = x/maxval * b -> clip(0,1) -> ** g1 -> 1 - (1 - .) ** g2 -> * maxval`` with
    ``b = exp(ln(balance) u0)``, ``g1 = exp(ln(gamma) u1)``, ``g2 = exp(ln(gamma) u2)``, three U(-1,1)
    random variables of shape (batch, num_maps).  balance == gamma == 1: identity (no stream is created,
    the seed chain is not consumed).  The test version is the identity (color.py:44-52).  Draws come from
    the device's Philox stream (keyed by the global image index) or are injected (``inject``)."""

    def __init__(self, inpt, img_sz,
                 num_maps=3,
                 rand_gen=None,
                 balance=1,
                 gamma=1,
                 maxval=1):
        self.params = []
        self.inpt = inpt
        self.out_sz = img_sz
        self.num_maps = num_maps
        self.n_out = self.num_maps * self.out_sz ** 2
        self.balance, self.gamma, self.maxval = balance, gamma, maxval
        self.representation = 'Color Maps:{} Size:{:2d} Balance:{:.2f} ' \
                              'Gamma:{:.2f} Maxval:{}'.format(
            num_maps, img_sz, balance, gamma, maxval)
        from ..device import get_context
        self.ctx = ctx = get_context()
        self.batch_sz = inpt.batch if isinstance(inpt, InputSlot) else inpt.shape[0]
        self.output = ctx.empty((self.batch_sz, num_maps, img_sz, img_sz))
        self.active = not (gamma == 1 and balance == 1)
        self.d_step = None
        self._inj = None
        self.gin = None
        if not self.active:
            return
        assert gamma > 0 and balance > 0
        # the stream seed consumes the seed chain like color.py:31-32
        self.seed = int(rand_gen.randint(1e6)) if rand_gen is not None else int(np.random.randint(0, 1e6))
        self.fac = ctx.empty((self.batch_sz * num_maps * 3,))

    def TestVersion(self, inpt):
        return ColorLayer(inpt,
                          self.out_sz,
                          num_maps=self.num_maps,
                          rand_gen=None,
                          balance=1,
                          gamma=1,
                          maxval=1)

    def inject(self, u=None):
        """Replace the device RNG by explicit uniforms, shape (3, batch, num_maps): the draws of the
        reference's three ``srs.uniform`` variables in creation order.  None: back to the generator."""
        self._inj = None if u is None else self.ctx.array(
            np.ascontiguousarray(np.asarray(u, np.float32).reshape(3, self.batch_sz, self.num_maps)))

    def _source(self):
        s = self.inpt
        if isinstance(s, InputSlot):
            return s.data.ptr, int(s.row0), int(s.row_global0)
        return s.ptr, 0, 0

    def forward(self, train=True):
        x_ptr, row0, rg0 = self._source()
        hw = self.out_sz * self.out_sz
        if not self.active:
            self.ctx.call("tn_elastic_apply", x_ptr, row0, None, self.output.ptr, self.batch_sz, self.num_maps,
                          self.out_sz, self.out_sz, 0, 1, None, None, None, 0.0, None, 0, 0, None, rg0)
            return
        self.ctx.call("tn_color_factors", self.fac.ptr, self.batch_sz, self.num_maps, float(self.balance),
                      float(self.gamma), self._inj.ptr if self._inj is not None else None, self.seed, 0,
                      self.d_step.ptr if self.d_step is not None else None, rg0)
        self.ctx.call("tn_color_apply", x_ptr, row0, self.fac.ptr, self.output.ptr, self.batch_sz,
                      self.num_maps, hw, float(self.maxval))

    def backward(self, gout, need_gin, below):
        if not need_gin or isinstance(self.inpt, InputSlot):
            return None
        from .. import _lib
        b_out, b_act, b_prm, b_mask = below.act_info()
        fuse = b_out is not None and b_act != _lib.TN_ACT_LINEAR
        if self.gin is None:
            self.gin = self.ctx.empty(self.inpt.shape)
        if not self.active:
            self.ctx.call("tn_scale_mask", gout.ptr, None, 1.0, self.gin.ptr, self.gin.size,
                          b_out.ptr if fuse else None, b_act, b_prm)
        else:
            self.ctx.call("tn_color_apply_bwd", self.inpt.ptr, 0, self.fac.ptr, gout.ptr, self.gin.ptr,
                          self.batch_sz, self.num_maps, self.out_sz * self.out_sz, float(self.maxval),
                          b_out.ptr if fuse else None, b_act, b_prm)
        if b_mask is not None:        # a Hidden layer with dropout right below: its mask, after the copy
            self.ctx.call("tn_scale_mask", self.gin.ptr, b_mask.ptr, 1.0, self.gin.ptr, self.gin.size,
                          None, _lib.TN_ACT_LINEAR, 0.0)
        return self.gin
