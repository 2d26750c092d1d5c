"""What ``get_trin_model`` / ``get_test_model`` return -- the counterparts of the reference's compiled Theano functions
(neuralnet.py:236-241, :270-277): ``fn(i) -> [cost, features, logprob]`` one step at a time (``_TrainFn``) or with two
steps in flight (``_PipeTrainFn``), ``fn(i) -> [sym_err, P(MLE)]`` (``_TestFn``).  ``enqueue(i)`` issues a step without
reading anything back; once the calls of a function have been seen to repeat, a step is ONE C call (plan.py,
tn_net_step)."""
import os

import numpy as np

from . import _lib
from .device import share  # noqa: F401
from .layer import ElasticLayer
from .plan import StepPlan


def _features_logprob(out):
    """[features, logprob] of the output head (neuralnet.py:236-241); for Softmax and Hinge heads features IS
    logprob (outlayers.py:92-93, :137-139)."""
    logprob = out.logprob.get_value()
    feats = logprob if out.features is out.logprob else out.features.get_value()
    return [feats, logprob]


def _step_outputs(net, out):
    """[cost, features, logprob] of the step ``net`` ran last, on the stream currently selected (the one that
    holds the launch summing the cost).  When the step sent features / logprob ahead (``NeuralNet._send_outputs``:
    copies into page-locked memory that ran under the backward pass) the cost follows them through the copy
    stream and one wait covers all three; otherwise three blocking copies."""
    early = getattr(net, "_early", None)
    if early is not None and early["live"]:
        early["live"] = False
        if not early["cost_sent"]:
            net.ctx.call("tn_d2h_early", early["cost"].ptr, net.d_cost.ptr, 4)
        net.ctx.call("tn_copy_sync")
        logprob = early["logprob"].array.copy()
        feats = logprob if out.features is out.logprob else early["features"].array.copy()
        return [np.float32(early["cost"].array[0]), feats, logprob]
    cost = net.d_cost.get_value()[0]
    return [cost] + _features_logprob(out)


class _CostRing:
    """The costs of enqueued steps, read a few steps LATER so that the loop of train.py (train.py:207-226: it sums the
    cost of every step and raises on NaN) never waits for the GPU: a step's cost (4 bytes) leaves for one of four
    page-locked slots as soon as it exists (tn_d2h_early_ev: copy stream, behind the launch that sums it, an event
    behind the copy) and the host picks it up ``lag`` calls later -- by then it has long arrived; if not, the host
    waits for that slot's event only."""
    R = 4

    def __init__(self, ctx, lag):
        import ctypes
        from .device import HostBuffer
        self.ctx, self.lag = ctx, lag
        self.buf = HostBuffer(ctx, (self.R,), np.float32)
        self.ev = []
        for _ in range(self.R):
            e = ctypes.c_void_p()
            ctx.call("tn_event_create", ctypes.byref(e))
            self.ev.append(e)
        self.next_take = 0          # first step number whose cost the caller has not been handed yet
        self.sent_upto = 0          # copies have been issued for the steps below
        self.strict = False         # True while a step_cost() call is enqueueing
        self.stale = False          # steps were enqueued outside step_cost() since the last one

    def send(self, step, d_cost, net=None):
        """The copy is ordered behind the launch that summed the cost, on the copy stream; nothing else holds the
        compute stream back, so the NEXT launch that writes ``d_cost`` must wait for the slot's event first:
        ``net._guard_cost()`` (NeuralNet) does, in front of every such launch -- a step or two later in the lazy
        schedules (the copy has long run), a few tens of microseconds later where the cost is summed mid-step
        (data-parallel pipelined steps, weight-cost nets)."""
        assert step == self.sent_upto, "cost ring out of step"
        if step - self.next_take >= self.R:
            # nobody is collecting: plain enqueue() calls while the ring is kept for the next step_cost() loop (drain_costs
            # leaves it in place) -- the oldest cost is dropped; inside a step_cost() loop this would be a bug
            assert not self.strict, "cost ring overrun"
            self.next_take = step - self.R + 1
        s = step % self.R
        self.ctx.call("tn_d2h_early_ev", self.buf.ptr + 4 * s, d_cost.ptr, 4, self.ev[s])
        if net is not None:
            net._cost_guard_ev = self.ev[s]
        self.sent_upto = step + 1

    def take(self, step):
        s = step % self.R
        self.ctx.lib.tn_event_sync(self.ctx.h, self.ev[s])
        self.next_take = step + 1
        return np.float32(self.buf.array[s])

    def __del__(self):
        try:
            for e in self.ev:
                self.ctx.lib.tn_event_destroy(self.ctx.h, e)
        except Exception:       # interpreter teardown
            pass


def _batch_in_range(i, n_rows, batch_sz):
    """Minibatch i must lie inside the dataset (the reference's ``data[i*B:(i+1)*B]`` would come out short and fail in the
    compiled function; the kernels here would read past the array)."""
    i = int(i)
    if i < 0 or (i + 1) * batch_sz > n_rows:
        raise IndexError("minibatch %d of %d rows each in a dataset of %d rows" % (i, batch_sz, n_rows))


class _TrainFn:
    """What ``get_trin_model`` returns: ``fn(i) -> [cost, features, logprob]``
    (neuralnet.py:236-241).  ``enqueue(i)`` issues the step without reading anything
    back (the GPU runs ahead of the host); ``fetch()`` copies the last step's outputs."""

    def __init__(self, net, x_data, y_data, take_index_list, aux_data=None):
        self.net, self.x_data, self.y_data = net, x_data, y_data
        self.aux_data = aux_data
        self.take_index_list = take_index_list
        ctx = net.ctx
        if take_index_list:
            row = int(np.prod(x_data.shape[1:]))
            self.x_stage = ctx.empty((net.local_bsz,) + tuple(x_data.shape[1:]))
            self.y_stage = ctx.empty((net.local_bsz,), np.int32)
            self.idx_dev = ctx.empty((net.local_bsz,), np.int32)
            self.row_bytes = row * 4
            if aux_data is not None:
                self.aux_stage = ctx.empty((net.local_bsz,) + tuple(aux_data.shape[1:]))
        # the step as one C call once its calls have been seen to repeat (plan.py); index-list batches upload per step
        self._plan = None if take_index_list else StepPlan(ctx, net.batch_sz, net.shard_lo)
        self._ring, self._n = None, 0          # step_cost(): costs read two calls late; steps enqueued so far
        self._sc_n, self._owed = 0, []

    def _plan_state(self):
        net = self.net
        first = net.tr_layers[0]
        return (getattr(first, "_cur", None), getattr(first, "_pre_valid", None), getattr(net, "_cost_pending", None),
                net._dp_cur, net._dp_pending, (self._n & 3) if self._ring is not None else -1)

    def _plan_set_state(self, st):
        net = self.net
        first = net.tr_layers[0]
        if st[0] is not None:
            first._cur, first._pre_valid = st[0], st[1]
        if st[2] is not None:
            net._cost_pending = st[2]
        net._dp_cur, net._dp_pending = st[3], st[4]
        if self._ring is not None:                # (the replayed step has sent its cost like an interpreted one)
            r = self._ring
            r.sent_upto = self._n + 1
            if not r.strict:
                r.stale = True
                r.next_take = max(r.next_take, self._n + 1 - r.R)      # (as send(): the last R steps stay)
            net._cost_guard_ev = r.ev[self._n % r.R]
        self._n += 1

    def _plannable(self):
        net = self.net
        return not getattr(net, "_want_outputs", False) and net._dp_tune is None and net.ctx.ev_hook is None and \
            not net._injecting()

    def enqueue(self, i):
        if self.take_index_list:
            idx = np.asarray(i)
            if idx.shape != (self.net.batch_sz,) or idx.min() < 0 or idx.max() >= self.x_data.shape[0]:
                raise IndexError("an index list must hold %d row numbers of a dataset of %d rows"
                                 % (self.net.batch_sz, self.x_data.shape[0]))
        else:
            _batch_in_range(i, self.x_data.shape[0], self.net.batch_sz)
        r = self._ring
        if r is not None and not r.strict and self._n == getattr(self, "_sc_hi", -1) and r.next_take < self._n:
            # a plain enqueue() / fn(i) in the middle of a step_cost() loop: the costs that loop is still owed are
            # collected NOW (the ring only keeps the last R steps) and handed out by the next step_cost() / drain_costs()
            # -- dropped, a NaN among them would slip past train.py's guard (train.py:225)
            self._owed += self._ring_rest(keep=True)
            r.stale = True                        # (the next step_cost() call starts its numbering behind the plain steps)
        pl = self._plan
        if pl is not None and not pl.off:
            ok = self._plannable()
            if pl.ready:
                if ok:                       # (the learning rate is a device scalar here: no argument changes with it)
                    self.net._apply_dtype()
                    st = pl.step(i, self._plan_state())
                    if st is not None:
                        return self._plan_set_state(st)
            if ok:
                pl.begin(i, self._plan_state())
            try:
                self._enqueue(i)
            except Exception:
                ok = False
                raise
            finally:
                pl.end(self._plan_state() if ok else None, ok)
            return
        self._enqueue(i)

    def _enqueue(self, i):
        net, ctx = self.net, self.net.ctx
        B, lo = net.batch_sz, net.shard_lo
        slot = net.x
        slot.d_row0 = None
        if self.take_index_list:
            idx = np.ascontiguousarray(np.asarray(i, np.int32)[lo:lo + net.local_bsz])
            self.idx_dev.set_value(idx)
            ctx.call("tn_gather_rows", self.x_data.ptr, self.idx_dev.ptr, self.x_stage.ptr,
                     net.local_bsz, self.row_bytes)
            ctx.call("tn_gather_rows", self.y_data.ptr, self.idx_dev.ptr, self.y_stage.ptr,
                     net.local_bsz, 4)
            slot.bind(self.x_stage)
            slot.row0, y, y_row0 = 0, self.y_stage, 0
            if self.aux_data is not None:                  # neuralnet.py:233-234
                ctx.call("tn_gather_rows", self.aux_data.ptr, self.idx_dev.ptr, self.aux_stage.ptr, net.local_bsz,
                         int(np.prod(self.aux_data.shape[1:])) * 4)
                net.aux_inpt_tr.bind(self.aux_stage)
                net.aux_inpt_tr.row0 = 0
        else:
            slot.bind(self.x_data)
            slot.row0 = int(i) * B + lo
            y, y_row0 = self.y_data, slot.row0
            if self.aux_data is not None:                  # neuralnet.py:225-226
                net.aux_inpt_tr.bind(self.aux_data)
                net.aux_inpt_tr.row0 = slot.row0
        slot.row_global0 = lo
        if self.aux_data is not None:
            net.aux_inpt_tr.row_global0 = int(i) * B + lo if not self.take_index_list else lo
        net._train_step(y, y_row0)
        if self._ring is not None:                # the step's cost exists behind its last launch: off it goes
            self._ring.send(self._n, net.d_cost, net)
            if not self._ring.strict:
                self._ring.stale = True
        self._n += 1

    # -- costs a few calls late (what train.py's loop needs of a step; see _CostRing) ----------------------------
    def step_cost(self, i):
        """Enqueue step i; return [(step number, cost), ...] of the steps whose cost has become due (step numbers count
        the step_cost calls since the last drain_costs()).  Do not mix with enqueue() / fn(i) before drain_costs()."""
        net = self.net
        pre, self._owed = self._owed, []
        if net._dp_delayed or net._dp_tune is not None or self.take_index_list or net._injecting():
            out = pre + self._ring_rest()             # (the cost travels on the second stream / per-step host work)
            out.append((self._sc_n, np.float32(self(i)[0])))
            self._sc_n += 1
            return out
        if self._ring is None:
            self._ring = _CostRing(net.ctx, 2)
            self._ring.sent_upto = self._ring.next_take = self._n
            self._ring_base = self._n - self._sc_n
            if self._plan is not None:
                self._plan.restart("cost ring on")
        r, out = self._ring, pre
        if r.stale:                               # plain enqueue() calls in between: their costs are nobody's
            r.next_take = r.sent_upto = self._n   # (what step_cost() calls before them were owed is in `pre`)
            self._ring_base = self._n - self._sc_n
            r.stale = False
        if self._n - r.next_take >= r.lag:
            out.append((r.next_take - self._ring_base, r.take(r.next_take)))
        r.strict = True
        try:
            self.enqueue(i)
        finally:
            r.strict = False
        self._sc_n += 1
        self._sc_hi = self._n                     # (a plain enqueue() right behind this step finds the loop's costs owed)
        return out

    def _ring_rest(self, keep=False):
        """Everything the ring still owes, in order.  ``keep``: the ring (and with it the recorded steps, whose baked
        slot and event pointers follow the step number modulo 4) stays for the next loop."""
        r, out = self._ring, []
        if r is None:
            return out
        if r.stale:                               # only plain enqueue() calls since the last loop: nothing is owed
            r.next_take = r.sent_upto
            r.stale = False
        while r.next_take < r.sent_upto:
            out.append((r.next_take - self._ring_base, r.take(r.next_take)))
        if keep:
            return out
        self._ring = None
        if self._plan is not None:
            self._plan.restart("cost ring off")
        return out

    def drain_costs(self):
        """The costs step_cost() has not handed out yet, in order; afterwards step numbers start from 0 again.  train.py
        calls this at the end of every epoch: ring and plan survive it (an epoch of mnist.prms at batch 4096 is 12
        steps -- fewer than it takes to watch and record a step)."""
        out, self._owed = self._owed, []
        out += self._ring_rest(keep=True)
        self._sc_n = 0
        if self._ring is not None:
            self._ring_base = self._n
        return out

    def fetch(self):
        net = self.net
        out = net.tr_layers[-1]
        if getattr(net, "_dp_pending", False):
            net.ctx.sync()                        # the cost travels with the all-reduce on the second stream
        return _step_outputs(net, out)

    def __call__(self, i):
        self.net._want_outputs = True       # features / logprob leave right after the forward pass
        try:
            self.enqueue(i)
        finally:
            self.net._want_outputs = False
        return self.fetch()


class _PipeTrainFn:
    """``get_trin_model``'s function for single-GPU training with TWO STEPS IN FLIGHT.

    The reference's update applies the OLD velocity (layer.py:82-86: ``v' = m v + (1-m) g``,
    ``p' = p - rate*lr*v``), so the weights of step t are ``p_{t-1} - s*v_{t-1}`` with ``v_{t-1}`` built
    from the gradient of step t-2: step t does not depend on the backward pass of step t-1.  Even steps
    run on the context's first stream with the net itself, odd steps on the second stream with a twin
    (own weights copy, activations and gradients; the velocities are shared).  Step t starts by waiting
    for the other stream's update, then ``v <- m v + (1-m) g_{t-2}`` (its own gradient of two steps ago)
    and ``p_own <- p_other - s*v`` in one launch (tn_sgd_update_net, TN_UPD_PIPE), then runs its forward and
    backward passes while the other stream is still busy with step t-1 -- the two fill each other's
    launch gaps and lock-step phases (-18 % per step on mnist.prms).  Same weights, costs and outputs as
    the sequential schedule, bit for bit (tests/test_gpu_net.py::test_pipelined_steps_equal_sequential);
    reading weights (get_wts, test functions, checkpoints) first brings the net up to date."""

    def __init__(self, net, x_data, y_data):
        import ctypes
        self.net, self.x_data, self.y_data = net, x_data, y_data
        self.take_index_list = False
        self.t = 0                   # steps enqueued so far
        self._updated = False        # the update for step self.t has already been applied (weights were read)
        self._seq = None             # sequential fallback (_TrainFn) once something rules pipelining out
        self._twin = None
        self._last = net
        net._pipe_fn = self
        self._ctypes = ctypes
        self._plan = StepPlan(net.ctx, net.batch_sz, net.shard_lo)
        self._ring = None            # step_cost(): costs read four calls late (_CostRing)
        self._sc_n, self._owed = 0, []

    # -- set-up of the twin on first use ---------------------------------------------------------
    def _build(self):
        net, ctx = self.net, self.net.ctx
        import copy
        from .neuralnet import NeuralNet
        twin = NeuralNet.__new__(NeuralNet)
        twin._is_twin = True
        twin._main = net
        tp = dict(net.tr_prms)
        tp.setdefault('SEED', 0)      # (a net loaded from a checkpoint has none; weights and stream seeds are copied below)
        twin.__init__(copy.deepcopy(net.layers), tp)
        if net._dp:
            twin._dev_group = net._group()                 # ONE communicator; the streams alternate on it
        twin._prepare_training()
        if net._dp:
            net._dp_set_schedule("plain")
            net._dp_tune = twin._dp_tune = None              # nothing to tune: the all-reduce rides in-stream
            net.dp_schedule = twin.dp_schedule = "pipelined"
        for a, b in zip(net.tr_layers, twin.tr_layers):
            if hasattr(a, "seed"):
                b.seed = a.seed
            if getattr(a, "drop", None) is not None:
                b.drop.seed = a.drop.seed
            for pa, pb in zip(a.params, b.params):
                ctx.call("tn_d2d", pb.ptr, pa.ptr, pa.size * 4)
            if hasattr(a, "centers") and not a.learn_centers:        # fixed class centers: the same on both streams
                ctx.call("tn_d2d", b.centers.ptr, a.centers.ptr, a.centers.size * 4)
            if a.params:
                b.accumulated_updates = a.accumulated_updates        # ONE velocity per tensor
        # the twin's own velocity buffers are gone with that: its update table must name the shared ones
        # (it is what _fall_back folds the last gradient through when the twin ran the last step)
        twin._build_seg_table()
        if getattr(twin, "_dp_can_delay", False):
            twin._segs_ab[0] = twin._d_segs
        self._twin = twin
        self.nets = (net, twin)
        seg_dt = np.dtype([('p', 'u8'), ('psrc', 'u8'), ('v', 'u8'), ('g', 'u8'), ('n', 'u8'),
                           ('momentum', 'f4'), ('rate', 'f4')])
        self._segs, self._hsegs, self._lr, self._lr_set = [], [], [], [None, None]
        for X, Y in ((net, twin), (twin, net)):
            rows = []
            for lx, ly in zip(X.tr_layers, Y.tr_layers):
                if lx.has_updates():
                    for p, ps, v, g in zip(lx.params, ly.params, lx.accumulated_updates, lx.grads):
                        rows.append((p.ptr, ps.ptr, v.ptr, g.ptr, p.size, lx.reg['momentum'], lx.reg['rate']))
            host = np.array(rows, dtype=seg_dt)
            self._hsegs.append(host)             # kept alive: the update matches pending slab sums against it
            self._segs.append(ctx.array(host.view(np.uint8)))
            X._cost_pending = False
            # single-GPU runs leave a step's slab sums and cost to the update that opens the stream's next
            # step (one launch instead of three); data-parallel steps need both before their all-reduce
            X._pipe_lazy = not net._dp
            self._lr.append(ctx.zeros((1,)))
            X._cost_rider = False
        self._nseg, self._max_seg = net._n_segs, net._max_seg
        self._ev, arev = [], []
        for _ in range(2):
            for lst in (self._ev, arev):
                e = self._ctypes.c_void_p()
                ctx.call("tn_event_create", self._ctypes.byref(e))
                lst.append(e)
        for k, X in enumerate(self.nets):                  # recorded behind step's last collective (communication stream)
            X._ar_done_ev = arev[k]
        base = int(net.d_step.get_value()[0])            # steps already taken (an earlier training function)
        self._base = base
        ctx.call("tn_set_u32", twin.d_step.ptr, base + 1)    # the twin takes every second step
        self._lr_prev = None

    def _lr_now(self):
        tp = self.net.tr_prms
        return float(np.float32(tp['INIT_LEARNING_RATE'] / (1 + tp['CUR_EPOCH'] / tp['EPOCHS_TO_HALF_RATE'])))

    def _blocked(self):
        return self.net._injecting() or self.net.ctx.ev_hook is not None

    # -- the step as one C call (plan.py) -----------------------------------------------------------
    def _plan_state(self):
        """What a step leaves behind on the host (restored after a replayed step of the same phase)."""
        per_net = []
        for X in self.nets:
            first = X.tr_layers[0]
            per_net.append((X._cost_pending, getattr(first, "_cur", None), getattr(first, "_pre_valid", None)))
        return (self.nets.index(self._last), tuple(per_net), (self.t & 3) if self._ring is not None else (self.t & 1))

    def _plan_set_state(self, st):
        self._last = self.nets[st[0]]
        for X, (cp, cur, pv) in zip(self.nets, st[1]):
            X._cost_pending = cp
            if cur is not None:
                first = X.tr_layers[0]
                first._cur, first._pre_valid = cur, pv
        self._updated = False
        r = self._ring
        if r is not None and self.t - 2 >= r.sent_upto:
            r.sent_upto = self.t - 1              # (the replayed step has sent the cost of step t - 2)
            if not r.strict:
                r.stale = True
                r.next_take = max(r.next_take, self.t - 1 - r.R)       # (as send(): the last R sent steps stay)
            self.nets[self.t & 1]._cost_guard_ev = r.ev[(self.t - 2) % r.R]
        self.t += 1

    def _plannable(self):
        lr = self._lr_now()
        return self._seq is None and self._twin is not None and not self._updated and not getattr(self, "_want", False) \
            and self.t >= 4 and lr == self._lr_prev and self._lr_set[0] == lr and self._lr_set[1] == lr \
            and not self._blocked()

    # -- the start-of-step update -----------------------------------------------------------------
    def _update_for(self, t):
        """weights (and velocity) for step t on the stream that will run it"""
        ctx, k = self.net.ctx, t & 1
        X = self.nets[k]
        self.net._apply_dtype()
        ctx.call("tn_stream_select", k)
        ctx.call("tn_event_wait", self._ev[1 - k])
        if X._dp and t >= 2:
            ctx.call("tn_event_wait", X._ar_done_ev)      # this stream's gradient of step t-2, back from the all-reduce
        if self._lr_set[k] != self._lr_prev:              # the rate step t-1 was enqueued under
            ctx.call("tn_set_f32", self._lr[k].ptr, self._lr_prev)
            self._lr_set[k] = self._lr_prev
        out = X.tr_layers[-1]
        rider = X._cost_pending
        if rider:
            X._guard_cost()
        X._update_and_maxnorm(_lib.TN_UPD_PIPE, self._segs[k].ptr, self._hsegs[k].ctypes.data, self._nseg,
                              self._max_seg, self._lr[k].ptr, 1.0, X.d_step.ptr, 2 if t >= 2 else 0, 1 if t >= 2 else 0,
                              out.rowloss.ptr if rider else None, X.local_bsz, 1.0 / X.batch_sz,
                              X.d_cost.ptr if rider else None)
        X._cost_pending = False
        ctx.call("tn_event_record", self._ev[k])

    def sync_weights(self):
        """Bring the net's own weights up to date (p_t after t steps) before anything reads them."""
        if self._seq is not None or self._twin is None or self.t == 0:
            return
        ctx, t = self.net.ctx, self.t
        if not self._updated:
            self._update_for(t)
            self._updated = True
        if t & 1:                                         # the twin holds p_t: copy into the net
            ctx.call("tn_stream_select", 0)
            ctx.call("tn_event_wait", self._ev[1])
            for a, b in zip(self.net.tr_layers, self._twin.tr_layers):
                for pa, pb in zip(a.params, b.params):
                    ctx.call("tn_d2d", pa.ptr, pb.ptr, pa.size * 4)
        ctx.call("tn_stream_select", 0)
        ctx.sync()

    def _flush_parked(self):
        """Finish the slab sums and costs the last steps left parked with their streams (the gradients
        become ordinary buffers; the next update simply reads them)."""
        if self._twin is None or self._seq is not None:
            return
        ctx = self.net.ctx
        for k, X in enumerate(self.nets):
            ctx.call("tn_stream_select", k)
            ctx.call("tn_defer_reductions", 0)
            self._finish_cost(X)
        ctx.call("tn_stream_select", 0)

    def _fall_back(self):
        """Leave the pipelined schedule for good: bring weights AND velocity to the sequential state."""
        net, ctx = self.net, self.net.ctx
        if self._ring is not None:                # costs still owed to a step_cost() loop: collect them first
            self._owed = self._owed + self._leave_ring()
        self._flush_parked()
        if self._twin is not None and self.t > 0:
            self.sync_weights()
            # the velocity is one gradient behind (that of step t-1, held by the stream that ran it)
            Y = self.nets[(self.t - 1) & 1]
            ctx.call("tn_stream_select", 0)
            ctx.call("tn_sgd_update_net", _lib.TN_UPD_DELAYED, Y._d_segs.ptr, None, Y._n_segs, Y._max_seg,
                     net.cur_learn_rate.ptr, 1.0, None, 0, 3, None, 0, 0.0, None)
            ctx.call("tn_set_u32", net.d_step.ptr, self._base + self.t)
            ctx.sync()
        has_wtcost = any(getattr(l, 'reg', None) and l.params and (l.reg['L1'] or l.reg['L2'])
                         for l in net.tr_layers)
        net._cost_rider = net.fused_step and (not has_wtcost) and not net._dp
        net._pipe_fn = None
        first = net.tr_layers[0]
        if isinstance(first, ElasticLayer):
            first._pre_valid = False              # a field built ahead was for this stream's step t+2
        self._seq = _TrainFn(net, self.x_data, self.y_data, False)

    # -- the step ---------------------------------------------------------------------------------
    def enqueue(self, i):
        _batch_in_range(i, self.x_data.shape[0], self.net.batch_sz)
        r = self._ring
        if r is not None and self._seq is None and not r.strict and self.t == getattr(self, "_sc_hi", -1) and r.next_take < self.t:
            # a plain enqueue() / fn(i) in the middle of a step_cost() loop: what that loop is still owed is collected now
            # (as at the end of an epoch) instead of being dropped -- see _TrainFn.enqueue
            self._owed = self._owed + self._leave_ring(keep=True)
            r.stale = True                        # (the next step_cost() call starts its numbering behind the plain steps)
        pl = self._plan
        if pl is None or pl.off:
            return self._enqueue(i)
        ok = self._plannable()
        if pl.ready and ok:
            self.net._apply_dtype()
            st = pl.step(i, self._plan_state())
            if st is not None:
                self._plan_set_state(st)
                return
        if ok:
            pl.begin(i, self._plan_state())
        try:
            self._enqueue(i)
        except Exception:
            ok = False
            raise
        finally:
            pl.end(self._plan_state() if ok and self._seq is None else None, ok and self._seq is None)

    def _enqueue(self, i):
        if self._seq is None and self._blocked():
            self._fall_back()
        if self._seq is not None:
            self.net._want_outputs = getattr(self, "_want", False)
            try:
                return self._seq.enqueue(i)
            finally:
                self.net._want_outputs = False
        if self._twin is None:
            self._build()
        net, ctx, t = self.net, self.net.ctx, self.t
        k = t & 1
        X = self.nets[k]
        if t >= 1 and not self._updated:
            self._update_for(t)
        elif t == 0:
            ctx.call("tn_stream_select", 0)
            ctx.call("tn_event_record", self._ev[0])
        else:
            ctx.call("tn_stream_select", k)
        r = self._ring
        if r is not None and t >= 2 and t - 2 >= r.sent_upto:
            # the update that opens step t has summed the cost of this stream's previous step (t - 2): off it goes
            # -- also when that update already ran because something read the weights in between (sync_weights)
            r.sent_upto = t - 2
            r.send(t - 2, X.d_cost, X)
            if not r.strict:
                r.stale = True
        self._updated = False
        self._lr_prev = self._lr_now()
        slot = X.x
        slot.d_row0 = None
        slot.bind(self.x_data)
        slot.row0 = int(i) * net.batch_sz + net.shard_lo
        slot.row_global0 = net.shard_lo
        X._want_outputs = getattr(self, "_want", False)
        try:
            X._train_step(self.y_data, slot.row0, pipe_stride=2)
        finally:
            X._want_outputs = False
            ctx.call("tn_stream_select", 0)
        self._last = X
        self.t = t + 1

    # -- costs a few calls late (what train.py's loop needs of a step; see _CostRing) ----------------------------
    def step_cost(self, i):
        """Enqueue step i; return [(step number, cost), ...] of the steps whose cost has become due (step numbers count
        the step_cost calls since the last drain_costs()).  With two steps in flight the cost of step t is summed by
        the launch that opens step t + 2 and handed out at call t + 4.  Do not mix with enqueue() / fn(i) before
        drain_costs()."""
        if self._seq is None and self._blocked():
            self._fall_back()                     # (collects what the ring owes) one step at a time from here on
        pre, self._owed = self._owed, []
        if self._seq is not None:
            off = self._sc_n - getattr(self._seq, "_sc_n", 0)
            return pre + [(k + off, c) for k, c in self._seq.step_cost(i)] + self._count()
        if self._ring is None:
            self._ring = _CostRing(self.net.ctx, 4)
            self._ring.sent_upto = self._ring.next_take = self.t
            self._ring_base = self.t - self._sc_n
            self._plan.restart("cost ring on")
        r, out = self._ring, []
        if r.stale:                               # plain enqueue() calls in between: their costs are nobody's
            r.next_take = r.sent_upto = self.t
            self._ring_base = self.t - self._sc_n
            r.stale = False
        if self.t - r.next_take >= r.lag:
            out.append((r.next_take - self._ring_base, r.take(r.next_take)))
        r.strict = True
        try:
            self.enqueue(i)
        finally:
            r.strict = False
        self._count()
        self._sc_hi = self.t                      # (a plain enqueue() right behind this step finds the loop's costs owed)
        return pre + out

    def _count(self):
        self._sc_n += 1
        return []

    def _leave_ring(self, keep=False):
        """Everything the ring still owes, in order: the copies already under way, then the last two steps' costs read
        directly (nothing would ever open the steps that sum them).  ``keep`` (drain_costs at the end of an epoch): ring
        and recorded steps stay -- the next steps go through the interpreter until both streams have a cost pending
        again (two steps), then the recorded phases match and replay resumes."""
        r, out = self._ring, []
        if r is None:
            return out
        assert self._seq is None
        if r.stale:                               # only plain enqueue() calls since the last loop: nothing is owed
            r.next_take = r.sent_upto = self.t
            r.stale = False
        if not keep:
            self._ring = None
        self._flush_parked()
        self.net.ctx.sync()
        while r.next_take < r.sent_upto:
            out.append((r.next_take - self._ring_base, r.take(r.next_take)))
        for t in range(r.next_take, self.t):
            out.append((t - self._ring_base, np.float32(self.nets[t & 1].d_cost.get_value()[0])))
        if keep:
            r.next_take = r.sent_upto = self.t
            self._ring_base = self.t
        else:
            self._plan.restart("cost ring off")
        return out

    def drain_costs(self):
        out, self._owed = self._owed, []
        if self._seq is None:
            out += self._leave_ring(keep=True)
        if self._seq is not None:
            off = self._sc_n - getattr(self._seq, "_sc_n", 0)
            out += [(k + off, c) for k, c in self._seq.drain_costs()]
        self._sc_n = 0
        return out

    def fetch(self):
        if self._seq is not None:
            return self._seq.fetch()
        X = self._last
        X.ctx.call("tn_stream_select", self.nets.index(X))
        if X._dp:
            X.ctx.call("tn_event_wait", X._ar_done_ev)    # the cost travels with the all-reduce (communication stream)
        try:
            early = getattr(X, "_early", None)
            sent = early is not None and early["live"]
            if not (sent and early["cost_sent"]):
                self._finish_cost(X)
            if not sent:
                X.ctx.sync()
            return _step_outputs(X, X.tr_layers[-1])
        finally:
            X.ctx.call("tn_stream_select", 0)

    @staticmethod
    def _finish_cost(X):
        """The cost of X's last step, if nothing has summed it yet (on the stream currently selected)."""
        if getattr(X, "_cost_pending", False):
            out = X.tr_layers[-1]
            X._guard_cost()
            X.ctx.call("tn_sgd_update_net", _lib.TN_UPD_PLAIN, None, None, 0, 0, X.cur_learn_rate.ptr, 1.0, None, 0, 0,
                       out.rowloss.ptr, X.local_bsz, 1.0 / X.batch_sz, X.d_cost.ptr)
            X._cost_pending = False

    def __call__(self, i):
        self._want = True                   # features / logprob leave right after the forward pass
        try:
            self.enqueue(i)
        finally:
            self._want = False
        return self.fetch()


class _TestFn:
    """``get_test_model``'s function: ``fn(i) -> [sym_err, P(MLE)](, features, y_preds)``."""

    def __init__(self, net, x_data, y_data, preds_feats, aux_data=None):
        self.net, self.x_data, self.y_data, self.preds_feats = net, x_data, y_data, preds_feats
        self.aux_data = aux_data

    def __call__(self, i):
        net, ctx = self.net, self.net.ctx
        _batch_in_range(i, self.x_data.shape[0], net.batch_sz)
        net._sync_weights()
        net._apply_dtype()
        if net.dtype == 'float16':
            net._c8_arrange(net.te_layers, False)
        slot = net.test_x
        slot.bind(self.x_data)
        slot.row0 = int(i) * net.batch_sz + net.shard_lo
        slot.row_global0 = net.shard_lo
        if self.aux_data is not None:                      # neuralnet.py:266-269
            net.aux_inpt_te.bind(self.aux_data)
            net.aux_inpt_te.row0 = slot.row0
        out = net.te_layers[-1]
        for lyr in net.te_layers[:-1]:
            lyr.forward(False)
        out.forward(False, y=self.y_data, y_row0=slot.row0)
        ctx.call("tn_error_stats", out.y_preds.ptr, self.y_data.ptr, slot.row0, out.rowp.ptr,
                 net.local_bsz, out.d_stats.ptr)
        if net.world.size > 1:
            net._group().allreduce_sum(out.d_stats)
        stats = out.d_stats.get_value() / net.world.size
        if net.world.size > 1:
            net._group().verify_order()      # the host has synchronised anyway: cheap point to compare
        res = [stats[0], stats[1]]
        if self.preds_feats:
            res += [out.features.get_value(), out.y_preds.get_value().astype(np.int64)]
        return res


