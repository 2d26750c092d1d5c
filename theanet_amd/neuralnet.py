"""NeuralNet -- the drop-in for theanet/neuralnet.py on MI355X.

Same constructor, same ``.prms`` layer-spec surface and the same methods as the
reference class (neuralnet.py:59-332), but instead of building twin Theano graphs
and compiling them, construction plans device buffers (weights, activations,
gradients all resident in HBM) and the step functions enqueue hand-written HIP
kernels through the C-ABI of libtheanet_hip.so.  Per training step only the
minibatch index crosses the host/device boundary.

Data-parallel: one process per GPU (RANK/WORLD_SIZE from torch.distributed.run);
rank r works on rows [i*B + r*B/R, i*B + (r+1)*B/R) of minibatch i and the flat
gradient buffer (+ the scalar cost) is sum-all-reduced once per step over RCCL.
"""
import os
import sys
import weakref
from functools import reduce
from operator import mul

import numpy as np

from . import _lib, comm, layer
from .plan import StepPlan
from .device import DeviceArray, get_context, share
from .layer import (AuxConcatLayer, SoftAuxLayer, CenteredOutLayer, ColorLayer, ConvLayer, DropOutLayer, ElasticLayer, ExpLossLayer, HiddenLayer,
                    HingeLayer, InputLayer, InputSlot, MeanLayer, OutputLayer, PoolLayer, SoftmaxLayer)

# ########################### Helper Functions #################################


def get_layers_info(layers):
    out = []
    for name, args in layers:
        out.append('\n{} : '.format(name))
        out.extend('\n\t{} : \t{}'.format(key, args[key]) for key in args)
    return ''.join(out)


def get_wts_info(wts, detailed=False):
    out, n_wts = [], 0
    for l, ww in enumerate(wts):
        out.append("\nLayer {}:".format(l))
        for w in ww:
            n_ww = reduce(mul, w.shape)
            n_wts += n_ww
            out.append('\n\t {} {} ❲{}❳'.format(w.shape, w.dtype, n_ww))
            if detailed:
                out.append(" ❲{:.2e}, {:.2e}, {:.2e}❳".format(w.min(), w.mean(), w.max()))
    out.append('\n\nTotal Number of Weights : {:,}'.format(n_wts))
    return ''.join(out)


def get_training_params_info(training_params):
    return "Training Parameters:" + ''.join(
        '\n\t{} : \t{}'.format(key, training_params[key]) for key in sorted(training_params))


_GRAD_ALIGN = 64      # floats: every tensor in the flat gradient buffer starts 256-byte aligned


# The "compiled functions" get_trin_model / get_test_model return live in trainfn.py
from .trainfn import _PipeTrainFn, _TestFn, _TrainFn  # noqa: E402,F401


# ###############################################################################
#                            The Neural Network
# ###############################################################################


def _fma32(m, v, c):
    """fl32(m * v + c) with ONE rounding, element-wise (the device's __fmaf_rn; numpy has no fma).  m * v is exact in
    float64 (24 + 24 bits); the float64 sum may round, and rounding that to float32 rounds twice -- wrong only when the
    float64 sum lands exactly on a float32 midpoint that the exact sum is not on.  The error of the float64 addition
    (TwoSum) says which side the exact sum lies on."""
    p = np.float64(m) * np.asarray(v, np.float64)
    c = np.asarray(c, np.float64)
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)                      # exact: p + c = s + err
    r = s.astype(np.float32)
    d = s - r.astype(np.float64)                         # exact (both on the float64 grid near r)
    up, dn = np.nextafter(r, np.float32(np.inf)), np.nextafter(r, np.float32(-np.inf))
    half_up = (up.astype(np.float64) - r.astype(np.float64)) / 2
    half_dn = (r.astype(np.float64) - dn.astype(np.float64)) / 2
    tie_up, tie_dn = (d == half_up) & (d > 0), (-d == half_dn) & (d < 0)
    # on a midpoint the float64 -> float32 step went to even; the exact sum is off the midpoint by err
    r = np.where(tie_up & (err > 0), up, r)              # exact sum above the midpoint: the upper neighbour
    r = np.where(tie_dn & (err < 0), dn, r)
    return r.astype(np.float32)


class NeuralNet():
    fuse_conv_pool = True     # class-level switch (tests run both the fused and unfused paths)
    # ... and the step-level fusions (slab sums and cost inside the update launch, the next minibatch's elastic field riding in a backward launch or beside the update, two steps in flight):
    # False = the generic schedule, one launch per piece of work -- what the fused schedules are tested against
    fused_step = True
    # the C-ABI ops that carry a net's conv products (bench.py brackets them for the conv roofline legs)
    CONV_FWD_OPS = ("tn_conv2d_fwd", "tn_convpool_fwd_mask", "tn_elastic_convpool_fwd_mask", "tn_c8_conv_fwd")
    CONV_BWD_OPS = ("tn_conv2d_wgrad", "tn_conv2d_dgrad", "tn_convpool_bwd_mask_dx", "tn_convpool_bwd_mask",
                    "tn_convblock_bwd_mask", "tn_convblock_bwd", "tn_convpool_bwd", "tn_c8_conv_wgrad", "tn_c8_conv_dgrad")

    def __init__(self, layers, training_params, allwts=None,
                 test_x=None):
        # Either a random seed or the weights of a previously trained net (neuralnet.py:63-68)
        if allwts is None:
            self.rand_gen = np.random.RandomState(training_params['SEED'])
        else:
            self.rand_gen = None

        self.ctx = get_context()             # raises without libtheanet_hip.so / a GPU
        # the SoftmaxLayer's training step as one kernel (tn_fc_softmax_train) or as its three ops: a choice of kernels
        # (results agree to rounding, not bit for bit), read once per net
        self._softmax_train = os.environ.get("TN_SOFTMAX_TRAIN", "1") != "0"
        # DTYPE: 'float32' (default = the reference's floatX, weights.py:8) or 'float16' = fp16 operands /
        # fp32 accumulation for the conv products, fp32 master weights; GRAD_SCALE: power of two applied
        # to dz before it is rounded to fp16 (results are scaled back: exact)
        self.dtype = training_params.get('DTYPE', 'float32')
        assert self.dtype in ('float32', 'float16'), "DTYPE must be 'float32' or 'float16'"
        self.grad_scale = float(training_params.get('GRAD_SCALE', 4096.))
        # MATMUL: 'float32' (default: exact fp32 MFMA) or 'bf16x3' -- the dense layers' products as six bf16 MFMA
        # products of exactly split operands (fp32-grade accuracy, not the same bits; gemm_b3.hip)
        self.matmul = training_params.get('MATMUL', 'float32')
        assert self.matmul in ('float32', 'bf16x3'), "MATMUL must be 'float32' or 'bf16x3'"
        self._apply_dtype()
        self.world = comm.get_world()
        self._dev_group = None

        self.tr_prms = training_params
        self.layers = layers
        self.allwts = allwts
        self.tr_layers = []
        self.te_layers = []
        self.batch_sz = training_params['BATCH_SZ']
        self.shard_lo, hi = comm.shard_rows(self.batch_sz, self.world.size, self.world.rank)
        self.local_bsz = hi - self.shard_lo
        self.num_layers = 0

        # "symbolic variables": windows into device-resident datasets
        self.x = InputSlot(self.local_bsz)
        self.y = None
        if test_x is None:
            self.test_x = InputSlot(self.local_bsz)
        else:
            self.test_x = InputSlot(self.local_bsz)
            self.test_x.bind(share(test_x))

        # device-side step state (read by the kernels, so a captured graph can be replayed)
        self.d_step = self.ctx.zeros((1,), np.uint32)       # RNG step counter
        self.d_row0 = self.ctx.zeros((1,), np.int64)        # first dataset row of the minibatch
        self.cur_learn_rate = self.ctx.zeros((1,), np.float32)
        # Input Layer
        input_layer_type = getattr(layer, layers[0][0])
        assert input_layer_type in (InputLayer, ElasticLayer, ColorLayer), \
            "First layer needs to be Input or Elastic or Color Layer"

        self.tr_layers.append(input_layer_type(self.x, rand_gen=self.rand_gen,
                                               **layers[0][1]))
        self.te_layers.append(self.tr_layers[0].TestVersion(self.test_x))
        self.num_layers += 1

        # Rest of the layers
        while self.num_layers < len(layers):
            self.append_next_layer()

        if self.fuse_conv_pool:
            self._fuse(self.tr_layers)
            self._fuse(self.te_layers)

        # Handle Auxiliary input (neuralnet.py:100-105)
        for tr_layer, te_layer in zip(self.tr_layers, self.te_layers):
            if type(tr_layer) in (AuxConcatLayer, SoftAuxLayer):
                assert not hasattr(self, 'aux_inpt_tr'), "Multiple Aux Inputs"
                self.aux_inpt_tr = tr_layer.aux_inpt
                self.aux_inpt_te = te_layer.aux_inpt
                tr_layer.aux.d_step = self.d_step

        assert isinstance(self.tr_layers[-1], OutputLayer), \
            "the last layer must be an output head (Softmax, ExpLoss, Hinge or CenteredOut layer)"

        # random streams read the device step counter; masks/noise are keyed by the
        # position inside the GLOBAL minibatch so that sharding does not change them
        for lyr in self.tr_layers:
            if isinstance(lyr, (ElasticLayer, ColorLayer)):
                lyr.d_step = self.d_step
            drop = getattr(lyr, "drop", None)
            if drop is not None:
                drop.d_step = self.d_step
                drop.elem0 = self.shard_lo * int(np.prod(drop.shape[1:]))
        out = self.tr_layers[-1]
        out.inv_batch = 1.0 / self.batch_sz          # global batch: grads sum to the mean
        self.te_layers[-1].inv_batch = 1.0 / self.batch_sz

        self._grads_ready = False

        # Set Epoch and learning rate
        if 'CUR_EPOCH' not in training_params:
            training_params['CUR_EPOCH'] = 0
        self.set_rate()

    def append_next_layer(self):
        layer_type, layer_args = self.layers[self.num_layers]
        prev_tr_layer = self.tr_layers[self.num_layers - 1]
        prev_te_layer = self.te_layers[self.num_layers - 1]
        wts = self.allwts[self.num_layers] if self.allwts else None

        tr_inpt = prev_tr_layer.output
        te_inpt = prev_te_layer.output
        curr_layer_type = getattr(layer, layer_type)

        if curr_layer_type in (ElasticLayer, ColorLayer, ConvLayer, PoolLayer, MeanLayer):
            if type(prev_tr_layer) is DropOutLayer:
                use_tr_layer = self.tr_layers[self.num_layers - 2]
            else:
                use_tr_layer = prev_tr_layer
            num_prev_maps = use_tr_layer.num_maps
            prev_out_sz = use_tr_layer.out_sz
            if getattr(tr_inpt, "c8", None) is not None:
                # DTYPE float16: fp16-resident tensors of the conv stack (device.C8Array) pass from layer to layer as they are
                assert curr_layer_type in (ConvLayer, PoolLayer), \
                    "DTYPE float16: only Conv / Pool layers take the conv stack's fp16-resident tensors (got {})".format(layer_type)
            elif tr_inpt.ndim != 4:
                tr_inpt = tr_inpt.reshape(self.local_bsz, num_prev_maps, prev_out_sz, prev_out_sz)
                te_inpt = te_inpt.reshape(self.local_bsz, num_prev_maps, prev_out_sz, prev_out_sz)

        if curr_layer_type in (ElasticLayer, ColorLayer):
            layer_args = dict(layer_args)
            layer_args.pop("num_maps", None)
            layer_args.pop("img_sz", None)
            curr_layer = curr_layer_type(tr_inpt,
                                         num_maps=num_prev_maps,
                                         img_sz=prev_out_sz,
                                         rand_gen=self.rand_gen,
                                         **layer_args)

        elif curr_layer_type is ConvLayer:
            curr_layer = ConvLayer(tr_inpt,
                                   wts,
                                   self.rand_gen,
                                   self.local_bsz,
                                   num_prev_maps,
                                   prev_out_sz,
                                   **layer_args)

        elif curr_layer_type in (PoolLayer, MeanLayer):
            curr_layer = curr_layer_type(tr_inpt,
                                         num_maps=num_prev_maps,
                                         in_sz=prev_out_sz,
                                         **layer_args)

        elif curr_layer_type is DropOutLayer:
            curr_layer = DropOutLayer(tr_inpt,
                                      self.rand_gen,
                                      prev_tr_layer.n_out,
                                      **layer_args)

        elif curr_layer_type is CenteredOutLayer:
            # Needs the hidden layer's weights (or a seed) and CENTERS (n_classes x n_features): neuralnet.py:175-194
            centers = None
            if wts:
                centers = wts[2] if len(wts) > 2 else None
                wts = wts[:2]
            te_inpt = te_inpt.flatten(2)
            curr_layer = CenteredOutLayer(tr_inpt.flatten(2),
                                          wts, centers,
                                          self.rand_gen,
                                          prev_tr_layer.n_out,
                                          **layer_args)

        elif curr_layer_type is HiddenLayer and getattr(tr_inpt, "c8", None) is not None:
            # DTYPE float16: the dense layer above the conv stack reads the fp16-resident tensor in ITS order (the
            # kernels walk W through the NCHW row map of flatten(2), neuralnet.py:168-173)
            curr_layer = HiddenLayer(tr_inpt, wts, self.rand_gen, prev_tr_layer.n_out, **layer_args)

        elif curr_layer_type in (AuxConcatLayer, HiddenLayer, SoftmaxLayer, SoftAuxLayer, HingeLayer, ExpLossLayer):
            assert getattr(tr_inpt, "c8", None) is None, \
                "DTYPE float16: a HiddenLayer must follow the conv stack (got {})".format(layer_type)
            te_inpt = te_inpt.flatten(2)
            curr_layer = curr_layer_type(tr_inpt.flatten(2),
                                         wts,
                                         self.rand_gen,
                                         prev_tr_layer.n_out,
                                         **layer_args)
        else:
            raise NotImplementedError("Unknown Layer Type" + layer_type)

        self.tr_layers.append(curr_layer)
        self.te_layers.append(curr_layer.TestVersion(te_inpt))
        if not getattr(self, "_is_twin", False):
            curr_layer._wts_hook = self.te_layers[-1]._wts_hook = self._sync_weights
        self.num_layers += 1

    @staticmethod
    def _fuse(lyrs):
        """Pair every ConvLayer that is directly followed by a 2x2 PoolLayer (and is small
        enough for the register-resident kernel) into one fused conv+act+pool launch."""
        for conv, pool in zip(lyrs[:-1], lyrs[1:]):
            if isinstance(conv, ConvLayer) and isinstance(pool, PoolLayer) \
                    and conv.can_fuse_with(pool):
                conv.fused_pool, pool.fused_conv = pool, conv
        for lyr in lyrs:
            assert not (isinstance(lyr, PoolLayer) and lyr.f16 and lyr.fused_conv is None), \
                "DTYPE float16: a PoolLayer must directly follow a ConvLayer"
        # DTYPE float16: the first conv layer packs its c8 input straight from the dataset window
        if len(lyrs) >= 2 and isinstance(lyrs[0], InputLayer) and isinstance(lyrs[1], ConvLayer) and lyrs[1].f16:
            lyrs[1]._pack_from, lyrs[0]._packed_by_conv = lyrs[0].inpt, True
        # ... or has it written by the distortion stage below it (tn_c8_elastic_apply): no fp32 image, no packing pass
        if len(lyrs) >= 2 and isinstance(lyrs[0], ElasticLayer) and lyrs[0].active and isinstance(lyrs[1], ConvLayer) \
                and lyrs[1].f16 and lyrs[1].x16 is not None and (lyrs[0].img_sz * lyrs[0].img_sz) % 4 == 0:
            lyrs[0]._c8_consumer, lyrs[1]._c8_prefilled = lyrs[1], True
        # an active single-channel ElasticLayer feeding such a block: the block's forward resamples
        # the raw images itself (tn_elastic_convpool_fwd_mask)
        if len(lyrs) >= 3 and isinstance(lyrs[0], ElasticLayer) and lyrs[0].active and \
                lyrs[0].num_maps == 1 and not getattr(lyrs[1], "f16", False) and isinstance(lyrs[1], ConvLayer) and \
                lyrs[1].fused_pool is lyrs[2]:
            el, conv, pool = lyrs[0], lyrs[1], lyrs[2]
            if conv.ctx.lib.tn_elastic_convpool_supported(
                    el.img_sz, el.img_sz, conv.num_maps, conv.filter_sz, conv.pad_lo, conv.out_sz,
                    conv.out_sz, pool.pool_sz, pool.out_sz, pool.out_sz) and not pool.ignore_border:
                el.fused_conv, pool.fused_elastic = conv, el

    _ctx_owner = None       # weakref to the net whose pipelined steps may have work parked in the context

    def _apply_dtype(self):
        """Called at the start of everything this net enqueues: sets the context's matmul dtype and takes
        over the context from whichever net used it last.  A pipelined training function leaves a step's
        slab sums and cost parked with its stream until the stream's next step; before another net's work
        goes onto those streams they are finished (the other net is alive: its buffers exist) or forgotten
        (it has been collected: the recorded outputs are dangling)."""
        self.ctx.set_matmul_dtype(self.dtype, self.grad_scale)
        self.ctx.set_fc_matmul(self.matmul)
        me = getattr(self, "_main", self)
        ref = NeuralNet._ctx_owner
        prev = ref() if ref is not None else None
        if prev is me:
            return
        if prev is not None and getattr(prev, "_pipe_fn", None) is not None:
            prev._pipe_fn._flush_parked()
        elif ref is not None:
            self.ctx.call("tn_defer_discard")
        NeuralNet._ctx_owner = weakref.ref(me)

    # ------------------------------------------------------------------------------
    def _group(self):
        if self._dev_group is None:
            self._dev_group = comm.DeviceGroup(self.ctx, self.world)
        return self._dev_group

    def _prepare_training(self):
        """Flat gradient buffer (all dW/db as views, + the scalar cost at the end so that
        ONE all-reduce moves everything) and the velocity buffers (layer.py:77-79)."""
        if self._grads_ready:
            return
        plist = [(lyr, p) for lyr in self.tr_layers for p in lyr.params]
        offs, total, self.n_flat = comm.flat_layout([p.size for _, p in plist], _GRAD_ALIGN)
        slots = [(lyr, p, off) for (lyr, p), off in zip(plist, offs)]
        self.flat_grads = self.ctx.zeros((total + _GRAD_ALIGN,))
        self.d_cost = self.flat_grads.view(total, (1,))
        for lyr in self.tr_layers:
            if lyr.params:
                lyr.grads = []
                lyr.accumulated_updates = []
        for lyr, p, off in slots:
            lyr.grads.append(self.flat_grads.view(off, p.shape))
            lyr.accumulated_updates.append(self.ctx.zeros(p.shape))
        self.tr_layers[-1].d_cost = self.d_cost
        segs = host = self._build_seg_table()
        # the minibatch cost can ride in the update launch unless something must be added to it
        # first (weight costs) or it has to travel through the all-reduce (data-parallel ranks)
        has_wtcost = any(getattr(l, 'reg', None) and l.params and (l.reg['L1'] or l.reg['L2'])
                         for l in self.tr_layers)
        # data-parallel step (TN_DP_FORCE=1: exercise it with a 1-rank communicator)
        self._dp = self.world.size > 1 or os.environ.get("TN_DP_FORCE") == "1"
        self._cost_rider = self.fused_step and (not has_wtcost) and not self._dp
        self._cost_rider_ok = self._cost_rider
        # which layers must propagate a gradient to their input
        self._need_gin = []
        seen = False
        for lyr in self.tr_layers:
            self._need_gin.append(seen)
            seen = seen or lyr.has_updates()
        if self._dp and not self.world.dry:
            self._group()
            if self.world.size > 1 and not getattr(self, "_is_twin", False):
                # replicas must start from identical weights (same SEED or same checkpoint on every rank)
                chk = float(sum(np.float64(w.astype(np.float64).sum()) for l in self.tr_layers for w in l.get_wts()))
                comm.agree(chk, "the initial weights (checksum)")
                # The ORDER in which partial sums are added is part of a gradient's bits, and replicas must stay
                # bit-identical: the process-static knobs that set slab counts / kernel forms and the CU count the slab
                # geometry is cut for must be the same on every rank (a rank with another TN_C8_WSLAB_DIV or another
                # device would round differently, silently)
                import zlib
                knobs = ("TN_C8_WSLAB_DIV", "TN_C8_ROLL", "TN_FC8_WSLABS", "TN_FC_WSPLIT", "TN_FC_DGRAD_SPLIT", "TN_GEMM_DEEP",
                         "TN_GEMM_DMA", "TN_GEMM_DMA_NS", "TN_PAIR_DMA", "TN_CONVPOOL_KS", "TN_SOFTMAX_TRAIN", "TN_FC_SKINNY",
                         "TN_CB_W44", "TN_POOL_MASK", "TN_ELASTIC_CONV", "TN_MN_FUSED", "TN_FC8_FIN", "TN_FC8_XCD")
                sig = "|".join("%s=%s" % (k, os.environ.get(k, "")) for k in knobs) + "|cus=%s" % self.ctx.info()[1]
                comm.agree(float(zlib.crc32(sig.encode())), "the kernel tunables / CU count (%s)" % sig)
        # Optional overlap of the gradient all-reduce with the backward pass (TN_DP_OVERLAP=1): the
        # fully-connected layers sit on top of the net and hold almost all parameters; their gradients
        # (the tail of the flat buffer, cost included) are reduced on the second stream while the conv
        # blocks below are still in their backward kernels.  _dp_split = first layer of that top group.
        # Off by default: on one GPU the second flush, the stream joins and the second collective cost
        # 18 us per step, about what a 1.5 MB all-reduce costs in the first place; the remaining
        # small all-reduce is latency-bound either way.
        self._dp_split, self._dp_off = None, 0
        self._dp_cand, self._dp_tune, self.dp_schedule = None, None, "plain"
        self._dp_delayed, self._dp_pending, self._dp_cur, self._dp_can_delay = False, False, 0, False
        self._dp_bound = 0
        self._dp_bucket = None
        if self._dp:
            j = len(self.tr_layers)
            while j > 0 and isinstance(self.tr_layers[j - 1], HiddenLayer):
                j -= 1
            top = [l for l in self.tr_layers[j:] if l.params]
            if 0 < j < len(self.tr_layers) and top and any(l.has_updates() for l in self.tr_layers[:j]):
                self._dp_cand = (j, (top[0].grads[0].ptr - self.flat_grads.ptr) // 4)
                # pipelined schedule: a bucket of its own for the dense group when what is left for the second
                # collective (the conv layers' gradients) is worth one -- mnist.prms: 780 floats, one all-reduce;
                # cifar_like: 93 k + 1.05 M, wide6: 1.15 M + 16.8 M floats, two.  TN_DP_BUCKETS=0/1 overrides.
                want = os.environ.get("TN_DP_BUCKETS", "auto")
                if want == "1" or (want == "auto" and self._dp_cand[1] * 4 >= (64 << 10)):
                    self._dp_bucket = self._dp_cand
            # "delayed" schedule: the all-reduce of step t runs under the whole of step t+1 (exact, see
            # _train_step).  It needs a second flat gradient buffer (g_{t+1} is produced while G_t is
            # in flight) and gradients that do not depend on the weights they are applied to (no L1/L2).
            self._dp_can_delay = len(segs) > 0 and not has_wtcost
            if self._dp_can_delay:
                self._flat_ab = [self.flat_grads, self.ctx.zeros((total + _GRAD_ALIGN,))]
                self._grads_ab, self._segs_ab = [], [self._d_segs]
                for buf in self._flat_ab:
                    self._grads_ab.append({id(lyr): [buf.view(off, p.shape) for l2, p, off in slots if l2 is lyr]
                                           for lyr in self.tr_layers if lyr.params})
                host_b = host.copy()
                host_b['g'] = host['g'] - self.flat_grads.ptr + self._flat_ab[1].ptr
                self._segs_ab.append(self.ctx.array(host_b.view(np.uint8)))
            # Which schedule is fastest depends on what the all-reduce costs on this node (RCCL latency
            # over xGMI vs. the extra launches and stream joins): with more than one rank it is
            # MEASURED -- TN_DP_OVERLAP=auto times a few steps of each schedule on the first calls,
            # the ranks agree on the result through an all-reduce(max) and keep the fastest one.
            mode = os.environ.get("TN_DP_OVERLAP", "auto" if self.world.size > 1 else "0")
            if self._dp_cand and mode == "1":
                self._dp_split, self._dp_off = self._dp_cand
                self.dp_schedule = "overlap"
            elif self._dp_can_delay and mode == "2":
                self._dp_delayed, self.dp_schedule = True, "delayed"
            elif mode == "auto":
                cands = ["plain"] + (["overlap"] if self._dp_cand else []) + \
                    (["delayed"] if self._dp_can_delay else [])
                if len(cands) > 1:
                    self._dp_tune = {"k": 0, "ev": {}, "cands": cands, "ms": []}
        self._grads_ready = True

    def _build_seg_table(self):
        """One multi-tensor momentum-SGD launch for every parameter tensor (layer.py:70-107): the table
        of (param, velocity, gradient, ...) segments, on the device and (for the lazy update) on the host.
        Rebuilt whenever a layer's velocity buffers are re-pointed (the twin of the pipelined schedule)."""
        seg_dt = np.dtype([('p', 'u8'), ('v', 'u8'), ('g', 'u8'), ('n', 'u8'),
                           ('momentum', 'f4'), ('rate', 'f4'), ('L1', 'f4'), ('L2', 'f4')])
        segs = []
        for lyr in self.tr_layers:
            if lyr.has_updates():
                for p, v, g in zip(lyr.params, lyr.accumulated_updates, lyr.grads):
                    segs.append((p.ptr, v.ptr, g.ptr, p.size, lyr.reg['momentum'], lyr.reg['rate'],
                                 lyr.reg['L1'], lyr.reg['L2']))
        self._n_segs = len(segs)
        self._max_seg = max([sg[3] for sg in segs] or [0])
        host = np.array(segs, dtype=seg_dt)
        if segs:
            self._d_segs = self.ctx.array(host.view(np.uint8))
            self._h_segs = host                       # kept alive: tn_sgd_update_net (TN_UPD_LAZY) reads it
        return host

    _DP_TUNE_PRE, _DP_TUNE_WARM, _DP_TUNE_STEPS = 32, 8, 24      # settle-in steps, per-leg warm-up, timed

    def _dp_bind(self, cur):
        """Point every layer's gradient views, the cost slot and the update's segment table at flat
        gradient buffer ``cur`` (the delayed schedule alternates between two)."""
        if not self._dp_can_delay or self._dp_bound == cur:
            return
        self._dp_bound = cur
        self.flat_grads = self._flat_ab[cur]
        for lyr in self.tr_layers:
            if lyr.params:
                lyr.grads = self._grads_ab[cur][id(lyr)]
        self.d_cost = self.flat_grads.view(self.n_flat - 1, (1,))
        self.tr_layers[-1].d_cost = self.d_cost
        self._d_segs = self._segs_ab[cur]

    def _dp_set_schedule(self, name):
        """Switch the data-parallel schedule between steps (all ranks at the same step index)."""
        if self._dp_delayed and name != "delayed" and self._dp_pending:
            # leaving the delayed schedule: the velocity is one gradient behind -- catch it up
            prev = 1 - self._dp_cur
            self.ctx.call("tn_stream_wait", 0, 1)
            self.ctx.call("tn_sgd_update_net", _lib.TN_UPD_DELAYED, self._segs_ab[prev].ptr, None, self._n_segs, self._max_seg,
                          self.cur_learn_rate.ptr, 1.0, None, 0, 3, None, 0, 0.0, None)
            self._dp_pending = False
        if name != "delayed":
            self._dp_cur = 0
        self._dp_delayed = name == "delayed"
        self._dp_split, self._dp_off = self._dp_cand if name == "overlap" else (None, 0)
        self.dp_schedule = name

    def _dp_tune_tick(self):
        """TN_DP_OVERLAP=auto: every candidate schedule (plain / overlapped all-reduce / delayed
        all-reduce) runs W warm-up + M timed steps under a pair of HIP events; then every rank takes
        the max over ranks of the times and keeps the fastest schedule.  All ranks switch at the same
        step index (the schedules issue different collectives).  The steps are ordinary training
        steps: nothing is thrown away, the weight trajectory is the same under every schedule."""
        import ctypes
        ctx, T = self.ctx, self._dp_tune
        W, M = self._DP_TUNE_WARM, self._DP_TUNE_STEPS
        k, cands = T["k"] - self._DP_TUNE_PRE, T["cands"]
        T["k"] += 1
        if k < 0:                                 # the first steps of a run are not representative
            return
        leg, pos = divmod(k, W + M)

        def mark(name):
            e = ctypes.c_void_p()
            ctx.call("tn_event_create", ctypes.byref(e))
            ctx.call("tn_event_record", e)
            T["ev"][name] = e

        if pos == 0:
            if leg > 0:
                mark("e%d" % (leg - 1))
            if leg < len(cands):
                self._dp_set_schedule(cands[leg])
        if pos == W and leg < len(cands):
            mark("s%d" % leg)
        if leg == len(cands) and pos == 0:
            ctx.sync()
            ms = []
            for q in range(len(cands)):
                v = ctypes.c_float()
                ctx.call("tn_event_elapsed_ms", T["ev"]["s%d" % q], T["ev"]["e%d" % q], ctypes.byref(v))
                ms.append(v.value)
            for e in T["ev"].values():
                ctx.lib.tn_event_destroy(ctx.h, e)
            t = ctx.array(np.asarray(ms, np.float32))
            self._group().allreduce_max(t)
            ms = [float(v) / M for v in t.get_value()]
            self.dp_tuned_ms = dict(zip(cands, ms))
            self._dp_tune = None
            self._dp_set_schedule(cands[int(np.argmin(ms))])
            if self.world.rank == 0:
                sys.stderr.write("theanet_amd: data-parallel schedule '%s' (%s)\n" % (
                    self.dp_schedule, ", ".join("%s %.1f us/step" % (c, 1e3 * m) for c, m in zip(cands, ms))))

    def _send_outputs(self, out, with_cost=False):
        """The step's features / logprob start travelling to page-locked host memory now (ordered behind the
        output layer's forward, on the context's copy stream): the copies run under the backward pass instead
        of after the step (the drop-in call fn(i) reads them every step, neuralnet.py:236-241)."""
        from .device import HostBuffer
        early = getattr(self, "_early", None)
        if early is None:
            early = self._early = {"live": False, "logprob": HostBuffer(self.ctx, out.logprob.shape),
                                   "cost": HostBuffer(self.ctx, (1,))}
            if out.features is not out.logprob:
                early["features"] = HostBuffer(self.ctx, out.features.shape)
        self.ctx.call("tn_d2h_early", early["logprob"].ptr, out.logprob.ptr, out.logprob.nbytes)
        if out.features is not out.logprob:
            self.ctx.call("tn_d2h_early", early["features"].ptr, out.features.ptr, out.features.nbytes)
        if with_cost:
            self.ctx.call("tn_d2h_early", early["cost"].ptr, self.d_cost.ptr, 4)
        early["cost_sent"] = with_cost
        early["live"] = True

    def _train_step(self, y, y_row0, d_row0=None, pipe_stride=0):
        """forward + backward + all-reduce + update for the minibatch the input slot
        currently points at.  Everything is enqueued; nothing is read back."""
        ctx = self.ctx
        self._apply_dtype()
        out = self.tr_layers[-1]
        first = self.tr_layers[0]
        if self._dp_tune is not None and not pipe_stride:
            self._dp_tune_tick()
        if self._dp_can_delay:
            self._dp_bind(self._dp_cur if self._dp_delayed else 0)
        if self.dtype == 'float16':
            self._c8_arrange(self.tr_layers, True)
        for lyr in self.tr_layers[:-1]:
            lyr.forward(True)
        # the weight-gradient ops only record their finishing slab sums; one launch does them all
        ctx.call("tn_defer_reductions", 1)
        n_lyr = len(self.tr_layers)
        fuse_out = self._softmax_train and n_lyr >= 2 and self._need_gin[n_lyr - 1] \
            and isinstance(out, SoftmaxLayer) and out.loss == "nll"
        try:
            out.forward(True, y=y, y_row0=y_row0, d_row0=d_row0,
                        below=self.tr_layers[-2] if fuse_out else None)
        except Exception:
            ctx.call("tn_defer_reductions", 0)
            raise
        want = getattr(self, "_want_outputs", False)
        cost_sent = False
        if want:
            if self._cost_rider_ok and not self._dp:
                # the caller reads [cost, features, logprob] of this step: the cost is summed now (the cost block
                # of the update launch on its own: same summation order, same bits) and leaves with the outputs
                self._guard_cost()
                ctx.call("tn_sgd_update_net", _lib.TN_UPD_PLAIN, None, None, 0, 0, self.cur_learn_rate.ptr, 1.0, None, 0, 0,
                         out.rowloss.ptr, self.local_bsz, 1.0 / self.batch_sz, self.d_cost.ptr)
                cost_sent = True
            self._send_outputs(out, cost_sent)
        # cost = -mean logprob[n, y_n] (this rank's share of the global mean).  Without weight
        # costs it rides in the update launch at the end of the step (tn_sgd_update_net);
        # with them it must exist before tn_wtcost accumulates onto it: a leaf reduction here.
        rider = self._cost_rider and not pipe_stride
        lazy_pipe = bool(pipe_stride) and getattr(self, "_pipe_lazy", False) and self._cost_rider_ok
        if cost_sent:
            self._cost_pending = False
        if lazy_pipe and not cost_sent:
            self._cost_pending = True             # summed by the launch that opens this stream's next step
        elif not rider and not cost_sent:
            self._guard_cost()
            if pipe_stride and self._cost_rider_ok:
                # the cost block of the update launch on its own: the same summation order as the
                # one-step-at-a-time schedule, so the reported cost is bit-identical too
                ctx.call("tn_sgd_update_net", _lib.TN_UPD_PLAIN, None, None, 0, 0, self.cur_learn_rate.ptr, 1.0, None, 0, 0,
                         out.rowloss.ptr, self.local_bsz, 1.0 / self.batch_sz, self.d_cost.ptr)
            else:
                ctx.call("tn_reduce_sum", out.rowloss.ptr, self.local_bsz, 1.0 / self.batch_sz,
                         self.d_cost.ptr, 0)
        g = out.dlogits
        # The elastic field of the NEXT minibatch only depends on the step counter.  It is left with
        # the context as a rider (offset +1: the counter advances at the end of the step) and
        # travels as extra blocks of the backward pass's paired GEMM launch; nets without such a
        # launch build it beside the update instead (tn_step_tail).
        ahead = (isinstance(first, ElasticLayer) and first.active and first.has_field and
                 not first._inj_draws and first.d_step is not None and (self._n_segs or rider) and
                 self.fused_step)
        if ahead:
            nxt = 1 - first._cur
            m = first._maps[nxt]
            hw = first.img_sz
            field_args = (hw, hw, float(first.translation), float(first.zoom), float(first.magnitude),
                          int(first.sigma), float(first.angle), int(first.nearest), m[0].ptr, m[1].ptr,
                          m[2].ptr, m[3].ptr)
            ctx.call("tn_rider_elastic_field", first.draws.ptr, first.seed, pipe_stride or 1,
                     self.d_step.ptr, *field_args)
        tail = False
        dp_async = False
        bucket_sent = False
        # single-GPU steps leave the finishing slab sums to the update launch (tn_sgd_update_net, TN_UPD_LAZY)
        lazy = False
        try:
            for idx in range(len(self.tr_layers) - 1, -1, -1):
                lyr = self.tr_layers[idx]
                below = self.tr_layers[idx - 1] if idx > 0 else None
                g = lyr.backward(g, self._need_gin[idx], below)
                if idx == self._dp_split and g is not None:
                    # the top (fully-connected) group is done: finish its slab sums and send its
                    # gradients through the all-reduce on the second stream, under the conv backward
                    ctx.call("tn_defer_reductions", 0)
                    ctx.call("tn_defer_reductions", 1)
                    ctx.call("tn_stream_wait", 1, 0)
                    ctx.call("tn_stream_select", 1)
                    self._group().allreduce_sum(self.flat_grads.view(self._dp_off, (self.n_flat - self._dp_off,)))
                    ctx.call("tn_stream_select", 0)
                    dp_async = True
                elif pipe_stride and self._dp and self._dp_bucket is not None and idx == self._dp_bucket[0] \
                        and g is not None:
                    # two steps in flight, bucketed (SURVEY 8e "bucket by layer"): the dense group on top of the net
                    # holds almost all parameters and its gradients exist NOW -- their slab sums are finished and the
                    # bucket [dense gradients | cost] leaves on the communication stream while this stream carries on
                    # with the conv blocks' backward kernels; the conv bucket follows at the end of the step
                    ctx.call("tn_defer_reductions", 0)
                    ctx.call("tn_defer_reductions", 1)
                    off = self._dp_bucket[1]
                    self._group().allreduce_sum_async(self.flat_grads.view(off, (self.n_flat - off,)), None, None)
                    bucket_sent = True
                if g is None:
                    break
            lazy = self.fused_step and not self._dp and 0 < self._n_segs <= 32
        finally:
            waiting = bool(ctx.lib.tn_rider_pending(ctx.h))
            if waiting:
                ctx.call("tn_rider_cancel")       # nobody carried it: it joins the update launch
            rode = ahead and not waiting
            tail = ahead and not rode
            if pipe_stride:                       # pipelined schedule: the update is not part of the step
                ahead, tail, lazy = rode, False, False
            lazy = lazy and not tail
            if tail:
                ctx.call("tn_defer_flush_step", self.d_step.ptr)      # the counter advances here
            elif not lazy and not lazy_pipe:
                ctx.call("tn_defer_reductions", 0)
        if pipe_stride:
            # two steps in flight (_PipeTrainFn): this stream's next step starts with the update.
            # Data-parallel: the all-reduce simply follows on this stream -- its latency is covered by
            # the other stream's step, and the update that needs it is a whole step away.
            if self._dp:
                # every collective of the pipelined schedule goes through the context's ONE communication stream
                # (tn_allreduce_sum_async): one order of collectives on the communicator whatever stream the step
                # ran on, and neither compute stream ever waits inside a collective.  The consumer -- the update
                # that opens this stream's next step -- waits for _ar_done_ev (_PipeTrainFn._update_for).
                n = self._dp_bucket[1] if bucket_sent else self.n_flat
                self._group().allreduce_sum_async(self.flat_grads, n, getattr(self, "_ar_done_ev", None))
            if ahead:
                first._cur, first._pre_valid = nxt, True
            if self.dtype == 'float16':
                self._c8_stale()
            return
        delayed = self._dp_delayed
        if delayed and tail:
            self._dp_set_schedule("plain")        # (configuration-determined: the same on every rank)
            delayed = False
        if delayed:
            # Delayed schedule.  layer.py:82-86 applies the OLD velocity, so p_{t+1} = p_t - s*v_t needs
            # the gradient of step t-1, not of this step: update with the REDUCED gradient of the
            # previous step (its all-reduce had this whole step to finish), then start this step's
            # all-reduce on the second stream, where it runs under the next step.  Bit-identical weights.
            cur = self._dp_cur
            if self._dp_pending:
                ctx.call("tn_stream_wait", 0, 1)
                ctx.call("tn_sgd_update_net", _lib.TN_UPD_DELAYED, self._segs_ab[1 - cur].ptr, None, self._n_segs,
                         self._max_seg, self.cur_learn_rate.ptr, 1.0, self.d_step.ptr, 1, 1, None, 0, 0.0, None)
            else:
                ctx.call("tn_sgd_update_net", _lib.TN_UPD_DELAYED, self._segs_ab[cur].ptr, None, self._n_segs,
                         self._max_seg, self.cur_learn_rate.ptr, 1.0, self.d_step.ptr, 1, 2, None, 0, 0.0, None)
            ctx.call("tn_stream_wait", 1, 0)
            ctx.call("tn_stream_select", 1)
            self._group().allreduce_sum(self._flat_ab[cur], self.n_flat)
            ctx.call("tn_stream_select", 0)
            self._dp_pending, self._dp_cur = True, 1 - cur
        elif self._dp:
            if dp_async:
                if self._dp_off:
                    self._group().allreduce_sum(self.flat_grads, self._dp_off)     # the conv head
                ctx.call("tn_stream_wait", 0, 1)                                  # join the tail
            else:
                self._group().allreduce_sum(self.flat_grads, self.n_flat)
        for lyr in self.tr_layers:
            lyr.get_wtcost(self.d_cost)
        if rider:
            self._guard_cost()
        mn_done = False
        if delayed:
            pass
        elif tail:
            ctx.call("tn_step_tail", self._d_segs.ptr if self._n_segs else None, self._n_segs,
                     self._max_seg, self.cur_learn_rate.ptr, 1.0,
                     out.rowloss.ptr if rider else None, self.local_bsz, 1.0 / self.batch_sz,
                     self.d_cost.ptr if rider else None, first.draws.ptr, first.seed, self.d_step.ptr,
                     *field_args)
        elif lazy:
            self._update_and_maxnorm(_lib.TN_UPD_LAZY, self._d_segs.ptr, self._h_segs.ctypes.data, self._n_segs,
                                     self._max_seg, self.cur_learn_rate.ptr, 1.0, self.d_step.ptr, 1, 0,
                                     out.rowloss.ptr if rider else None, self.local_bsz, 1.0 / self.batch_sz,
                                     self.d_cost.ptr if rider else None)
            mn_done = True
        elif self._n_segs or rider:               # also advances the RNG step counter
            ctx.call("tn_sgd_update_net", _lib.TN_UPD_PLAIN, self._d_segs.ptr if self._n_segs else None, None,
                     self._n_segs, self._max_seg, self.cur_learn_rate.ptr, 1.0, self.d_step.ptr, 1, 0,
                     out.rowloss.ptr if rider else None, self.local_bsz, 1.0 / self.batch_sz,
                     self.d_cost.ptr if rider else None)
        else:
            ctx.call("tn_add_u32", self.d_step.ptr, 1)
        if ahead:
            first._cur, first._pre_valid = nxt, True
        if not mn_done:
            self._apply_maxnorm_all()
        if self.dtype == 'float16':
            self._c8_stale()

    def _update_and_maxnorm(self, *args):
        """Layer.get_updates of every tensor (layer.py:70-107) as ONE call: tn_sgd_update_net + the max-norm projection,
        the column sums of the dense matrices left by the update launch itself (tn_sgd_update_net_maxnorm)."""
        tab = self._maxnorm_table()
        if 0 < len(tab) <= 32 and os.environ.get("TN_MN_FUSED", "1") != "0":
            self.ctx.call("tn_sgd_update_net_maxnorm", *args, tab.ctypes.data, len(tab))
        else:
            self.ctx.call("tn_sgd_update_net", *args)
            self._apply_maxnorm_all()

    def _guard_cost(self):
        """In front of a launch that writes ``d_cost``: a step_cost() loop may still owe the host the previous value (a
        4-byte copy on the copy stream, _CostRing.send) -- the stream waits for that copy's event."""
        ev = getattr(self, "_cost_guard_ev", None)
        if ev is not None:
            self.ctx.call("tn_event_wait", ev)
            self._cost_guard_ev = None

    def _injecting(self):
        """A parity test has injected random draws somewhere (dropout masks, elastic / color draws)."""
        for lyr in self.tr_layers:
            drop = getattr(lyr, "drop", None)
            if (drop is not None and drop.injected) or getattr(lyr, "_inj_draws", False) or \
                    getattr(lyr, "_inj", None) is not None or getattr(lyr, "_inj_flip", None) is not None:
                return True
        return False

    def _c8_arrange(self, lyrs, train):
        """DTYPE float16: the conv layers' weights as fp16 MFMA operand tiles, all products of the pass in ONE launch
        (tn_c8_arrange_multi); valid until the next update (_c8_stale)."""
        key = "_c8_tab_tr" if train else "_c8_tab_te"
        tab = getattr(self, key, None)
        if tab is None:
            dt = np.dtype([('W', 'u8'), ('wt', 'u8'), ('K', 'i4'), ('C', 'i4'), ('dgrad', 'i4'), ('pad', 'i4')])
            assert dt.itemsize == 32          # tn_c8_wt_seg
            rows, convs = [], []
            for idx, lyr in enumerate(lyrs):
                if isinstance(lyr, ConvLayer) and lyr.f16:
                    convs.append(lyr)
                    rows.append((lyr.W.ptr, lyr.wt_fwd.ptr, lyr.num_maps, lyr.num_prev_maps, 0, 0))
                    if train and self._need_gin[idx]:
                        if lyr.wt_bwd is None:
                            n = self.ctx.lib.tn_c8_wt_elems(lyr.num_maps, lyr.num_prev_maps, 1)
                            lyr.wt_bwd = self.ctx.empty((n,), np.uint16)
                        rows.append((lyr.W.ptr, lyr.wt_bwd.ptr, lyr.num_maps, lyr.num_prev_maps, 1, 0))
            tab = (np.array(rows, dtype=dt) if rows else np.zeros((0,), dt), convs)
            setattr(self, key, tab)
        segs, convs = tab
        for i in range(0, len(segs), 32):
            chunk = segs[i:i + 32]
            self.ctx.call("tn_c8_arrange_multi", chunk.ctypes.data, len(chunk))
        for lyr in convs:
            lyr.wt_valid = True

    def _c8_stale(self):
        """The weights are about to change: the arranged operand tiles of both graphs are no longer theirs."""
        for key in ("_c8_tab_tr", "_c8_tab_te"):
            tab = getattr(self, key, None)
            if tab is not None:
                for lyr in tab[1]:
                    lyr.wt_valid = False

    def _apply_maxnorm_all(self):
        """layer.py:88-103 for every parameter of the net in ONE call (tn_maxnorm_multi: the biases and conv kernels
        share a launch; Layer.apply_maxnorm is the per-layer form of the same projection)."""
        tab = self._maxnorm_table()
        for i in range(0, len(tab), 32):
            chunk = tab[i:i + 32]
            self.ctx.call("tn_maxnorm_multi", chunk.ctypes.data, len(chunk))

    def _maxnorm_table(self):
        tab = getattr(self, "_mn_tab", None)
        if tab is None:
            rows = []
            for lyr in self.tr_layers:
                if not lyr.has_updates() or not lyr.reg['maxnorm']:
                    continue
                for p in lyr.params:
                    if p.ndim in (1, 2, 4):
                        rows.append((p.ptr, p.ndim, p.shape[0],
                                     1 if p.ndim == 1 else int(np.prod(p.shape[1:])), float(lyr.reg['maxnorm'])))
            dt = np.dtype([('p', 'u8'), ('ndim', 'i4'), ('d0', 'i4'), ('rest', 'i4'), ('mx', 'f4')])
            assert dt.itemsize == 24          # tn_mn_seg
            tab = self._mn_tab = np.array(rows, dtype=dt) if rows else np.zeros((0,), dt)
        return tab

    # ------------------------------------------------------------------------------
    def _check_images(self, x_data, y_data=None):
        """The dataset must have the image shape the net was built for (Theano reports the mismatch when the compiled
        function first runs; here the kernels index the dataset with the net's shape, so it is checked up front) and at
        least one minibatch."""
        first = self.tr_layers[0]
        want = (first.num_maps, first.out_sz, first.out_sz)
        shape = tuple(x_data.shape)
        assert len(shape) == 4 and shape[1:] == want, \
            "image data of shape {} for a net built for (N, {}, {}, {}) images".format(shape, *want)
        assert shape[0] >= self.batch_sz, "{} images for minibatches of {}".format(shape[0], self.batch_sz)
        if y_data is not None:
            assert y_data.shape[0] == shape[0], "{} labels for {} images".format(y_data.shape[0], shape[0])
            # labels index the rows of logprob / the class centres (outlayers.py:50-51: logprob[arange, y] -- an IndexError
            # in the reference)
            last = self.tr_layers[-1]
            n_cls = last.centers.shape[0] if getattr(last, "centers", None) is not None else last.n_out
            yv = y_data.get_value() if hasattr(y_data, "get_value") else np.asarray(y_data)
            if yv.size and (int(yv.min()) < 0 or int(yv.max()) >= n_cls):
                raise IndexError("labels in [{}, {}] for an output layer of {} classes".format(
                    int(yv.min()), int(yv.max()), n_cls))

    def get_trin_model(self, x_data, y_data, aux_data=None,
                       take_index_list=False):
        print('Compiling training function...')
        self.tr_layers[-1].cost(None)            # validates the loss name (outlayers.py:12-36)
        self._check_images(x_data, y_data)
        if hasattr(self, 'aux_inpt_tr'):
            assert aux_data is not None, "Auxillary data not supplied"        # neuralnet.py:216-217
            aux_data = share(aux_data)
        else:
            aux_data = None
        self._prepare_training()
        if getattr(self, "_pipe_fn", None) is not None:
            self._pipe_fn._fall_back()           # an earlier training function: bring the net up to date
        if aux_data is None and self._pipe_ok(take_index_list):
            return _PipeTrainFn(self, share(x_data), share(y_data, np.int32))
        return _TrainFn(self, share(x_data), share(y_data, np.int32), take_index_list, aux_data)

    def _sync_weights(self):
        """With two steps in flight (_PipeTrainFn) the net's own weight buffers lag behind: catch up."""
        fn = getattr(self, "_pipe_fn", None)
        if fn is not None:
            fn.sync_weights()

    def _pipe_ok(self, take_index_list):
        """Two-steps-in-flight schedule (_PipeTrainFn): single GPU, plain momentum-SGD nets."""
        if getattr(self, "_is_twin", False) or not self.fused_step or os.environ.get("TN_PIPELINE", "1") == "0":
            return False
        has_wtcost = any(getattr(l, 'reg', None) and l.params and (l.reg['L1'] or l.reg['L2'])
                         for l in self.tr_layers)
        inject = any((getattr(l, "drop", None) is not None and l.drop.injected) or getattr(l, "_inj_draws", False)
                     or getattr(l, "_inj", None) is not None
                     or getattr(l, "_inj_flip", None) is not None for l in self.tr_layers)
        if self._dp and os.environ.get("TN_DP_PIPELINE", "1") == "0":
            return False
        return not take_index_list and self._n_segs > 0 and not has_wtcost and not inject

    def reset_accumulated_gradients(self):
        self._prepare_training()
        if getattr(self, "_pipe_fn", None) is not None:
            self._pipe_fn._fall_back()           # steps in flight: apply their gradients first
        if self._dp_delayed and self._dp_pending:
            # delayed all-reduce schedule: the reduced gradient of the last step is still to be folded
            # into the velocity -- the reference zeroes that contribution too: drop it
            self.ctx.call("tn_stream_wait", 0, 1)
            self._dp_pending = False
        for lyr in self.tr_layers:
            for au in (lyr.accumulated_updates or ()):
                au.fill_bytes(0)

    def get_test_model(self, x_data, y_data, aux_data=None, preds_feats=False):
        print('Compiling testing function... ')
        self._check_images(x_data, y_data)
        if hasattr(self, 'aux_inpt_te'):
            assert aux_data is not None, "Auxillary data not supplied"        # neuralnet.py:266-267
            aux_data = share(aux_data)
        else:
            aux_data = None
        return _TestFn(self, share(x_data), share(y_data, np.int32), preds_feats, aux_data)

    def takes_aux(self):
        return hasattr(self, 'aux_inpt_te')

    def get_data_test_model(self, get_output_of_layers=()):
        print('Compiling full test function...')
        if self.tr_prms['BATCH_SZ'] != 1:
            print("\n****WARNING****: BATCH SIZE IS NOT 1. "
                  "WILL BE EXPECTING A BATCH OF INPUT IMAGES AT A TIME.\n")
        first = self.te_layers[0]
        stage = self.ctx.empty((self.local_bsz, first.num_maps, first.out_sz, first.out_sz))

        for index in get_output_of_layers:          # a requested conv map must be materialised
            lyr = self.te_layers[index]
            if isinstance(lyr, ConvLayer) and lyr.fused_pool is not None:
                assert not lyr.f16, "DTYPE float16: the conv map of a fused conv + pool block is never materialised"
                lyr.fused_pool.fused_conv, lyr.fused_pool = None, None

        def fn(x, aux=None):
            self._sync_weights()
            self._apply_dtype()
            if self.dtype == 'float16':
                self._c8_arrange(self.te_layers, False)
            x = np.ascontiguousarray(x, np.float32).reshape(stage.shape)
            stage.set_value(x)
            if self.takes_aux():                           # neuralnet.py:289-290
                assert aux is not None, "Auxillary data not supplied"
                self.aux_inpt_te.bind(share(np.ascontiguousarray(aux, np.float32)))
                self.aux_inpt_te.row0 = 0
            slot = self.test_x
            slot.bind(stage)
            slot.row0 = slot.row_global0 = 0
            for lyr in self.te_layers[:-1]:
                lyr.forward(False)
            out = self.te_layers[-1]
            out.forward(False)
            res = [out.features.get_value(), out.y_preds.get_value().astype(np.int64)]
            for index in get_output_of_layers:
                res.append(self.te_layers[index].output.get_value())
            return res

        return fn

    def get_init_params(self, with_opt_state=None):
        """neuralnet.py:298-301: {"layers", "training_params", "allwts"} -- what the reference pickles.  With
        ``with_opt_state`` (default: training param SAVE_OPT_STATE, off) one more key, "opt_state": the velocities
        (layer.py:77-79; the reference drops them, so a resumed run restarts its momentum), the RNG step counter and the
        seeds of the dropout / distortion streams; ``load_opt_state`` puts them back.  Readers of the reference's pickles ignore
        the extra key."""
        out = {"layers": self.layers,
               "training_params": self.tr_prms,
               "allwts": [l.get_wts() for l in self.tr_layers]}
        if with_opt_state is None:
            with_opt_state = bool(self.tr_prms.get('SAVE_OPT_STATE', False))
        if with_opt_state:
            out["opt_state"] = self._opt_state()
        return out

    def _opt_state(self):
        self._prepare_training()
        fn = getattr(self, "_pipe_fn", None)
        pend, step = None, int(self.d_step.get_value()[0])
        if fn is not None and fn._seq is None and fn._twin is not None and fn.t > 0:
            # two steps in flight: the velocity on the device is one gradient behind (that of step t-1, still with the
            # stream that ran it, and folded in by that stream's next update): fold it in on the host, same expression
            # as the kernels (common.h tn_vel: fma(m, v, rn((1-m) g)))
            fn._flush_parked()
            self.ctx.sync()
            X = fn.nets[(fn.t - 1) & 1]
            pend, step = (lambda i, j: X.tr_layers[i].grads[j].get_value()), fn._base + fn.t
        elif self._dp_delayed and self._dp_pending:
            # delayed all-reduce (TN_DP_OVERLAP=2): the same situation -- the reduced gradient of the last step is still
            # travelling (second stream) and the update that folds it in belongs to the next step.  Read-only here: the
            # schedule is left alone (a checkpoint is written by one rank; a schedule change must happen on all)
            self.ctx.sync()
            prev = self._grads_ab[1 - self._dp_cur]
            pend = lambda i, j: prev[id(self.tr_layers[i])][j].get_value()
        vel = []
        for i, lyr in enumerate(self.tr_layers):
            row = []
            for j, v in enumerate(lyr.accumulated_updates or ()):
                a = v.get_value()
                if pend is not None and lyr.has_updates():
                    a = _fma32(np.float32(lyr.reg['momentum']), a, (np.float32(1) - np.float32(lyr.reg['momentum'])) * pend(i, j))
                row.append(a)
            vel.append(row)
        seeds = [(getattr(l, "seed", None), l.drop.seed if getattr(l, "drop", None) is not None else None)
                 for l in self.tr_layers]
        return {"velocities": vel, "rng_step": step, "stream_seeds": seeds}

    def load_opt_state(self, state):
        """Velocities and RNG step counter saved by get_init_params(with_opt_state=True); call before training."""
        self._prepare_training()
        if getattr(self, "_pipe_fn", None) is not None:
            self._pipe_fn._fall_back()
        for lyr, row in zip(self.tr_layers, state["velocities"]):
            for v, a in zip(lyr.accumulated_updates or (), row):
                v.set_value(np.asarray(a, np.float32))
        self.ctx.call("tn_set_u32", self.d_step.ptr, int(state["rng_step"]))
        # a net rebuilt from weights draws fresh stream seeds (neuralnet.py:63-68: no SEED chain): put the run's back
        for lyr, (seed, dseed) in zip(self.tr_layers, state.get("stream_seeds", ())):
            if seed is not None and hasattr(lyr, "seed"):
                lyr.seed = seed
            if dseed is not None and getattr(lyr, "drop", None) is not None:
                lyr.drop.seed = dseed

    def set_rate(self):
        self.cur_learn_rate.set_value(np.float32(
            self.tr_prms['INIT_LEARNING_RATE'] /
            (1 + self.tr_prms['CUR_EPOCH'] /
             self.tr_prms['EPOCHS_TO_HALF_RATE'])))

    def inc_epoch_set_rate(self):
        self.tr_prms['CUR_EPOCH'] += 1
        self.set_rate()

    def get_epoch(self):
        return self.tr_prms['CUR_EPOCH']

    def __str__(self):
        prmstr = '; '.join([', '.join([getattr(prm, "name", "param") for prm in lyr.params])
                            for lyr in self.tr_layers])
        return \
            '\nTrain Layers\n\t' + \
            '\n\t'.join([str(l) for l in self.tr_layers]) + \
            '\nTest Layers\n\t' + \
            '\n\t'.join([str(l) for l in self.te_layers]) + \
            '\nParams ' + prmstr

    def get_layers_info(self):
        return get_layers_info(self.layers)

    def get_wts_info(self, detailed=False):
        return get_wts_info((l.get_wts() for l in self.tr_layers), detailed)

    def get_training_params_info(self):
        return get_training_params_info(self.tr_prms)
