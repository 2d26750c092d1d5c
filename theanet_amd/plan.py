"""A training step as ONE C call (include/theanet_hip.h tn_net_plan_* / tn_net_step; SURVEY.md 8(b)'s coarse entry
point for launch-bound steps).

The reference compiles ``fn(i)`` once (``theano.function``, neuralnet.py:236-241) and every call afterwards runs without
the interpreter.  Here a step is ~14 C-ABI calls issued by Python: ~65 us of interpreter + ctypes per mnist.prms step
against ~1.4 us per kernel launch from C (tools/probe/launchrate.hip) -- more than the GPU needs for the 512-image
shard each rank of the 8-GPU run processes.  ``StepPlan`` is the compile step of this build: it WATCHES the calls of a
few ordinary steps of a training function (``Context.call`` records entry point + arguments while executing them),
finds the period after which the sequence repeats (1, 2 or 4 steps: the two streams of the pipelined schedule
alternate, the elastic stage's sample maps ping-pong), finds the integer arguments that follow the minibatch index
(``row0 = i * B + shard offset``), and hands one flat plan per phase to the library.  From then on ``enqueue(i)`` is
``tn_net_step(plan[phase], i)``; the function's host-side state (step counter, which stream is next, pending cost,
ping-pong indices) is advanced from the snapshots taken while recording, so any step can go back through the
interpreter -- one that returns outputs, follows a weight read-back, sees a new learning rate or an injected draw.
Replayed and interpreted steps issue the same calls with the same arguments: results are bit-identical
(tests/test_gpu_net.py::test_planned_steps_equal_interpreted_steps)."""
import ctypes
import os

import numpy as np

from . import _lib

_INT_TYPES = (ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_uint8, ctypes.c_uint32,
              ctypes.c_uint64, ctypes.c_void_p)


class PlanError(Exception):
    pass


def _encode(name, args):
    """(kinds, 64-bit values) of one recorded call, from the ctypes signature of the entry point."""
    argtypes = _lib.SIGNATURES[name][1][1:]            # without the context
    if len(argtypes) != len(args):
        raise PlanError("%s: %d arguments recorded, %d declared" % (name, len(args), len(argtypes)))
    kinds, vals = [], []
    for t, a in zip(argtypes, args):
        if t is ctypes.c_float:
            kinds.append(1)
            vals.append(int(np.float32(a).view(np.uint32)))
        elif t is ctypes.c_double:
            kinds.append(2)
            vals.append(int(np.float64(a).view(np.uint64)))
        elif t in _INT_TYPES:
            if a is None:
                v = 0
            elif isinstance(a, ctypes.c_void_p):
                v = a.value or 0
            elif isinstance(a, (int, np.integer)):
                v = int(a)
            else:
                raise PlanError("%s: argument of type %s cannot be replayed" % (name, type(a).__name__))
            kinds.append(0)
            vals.append(v & 0xFFFFFFFFFFFFFFFF)
        else:                                          # byref outputs, strings: the call talks to the host
            raise PlanError("%s: not a replayable entry point" % name)
    return kinds, vals


class StepPlan:
    """Watches, then replays, the steps of ONE training function.

    ``owner`` supplies ``_plan_state()`` / ``_plan_set_state(s)`` (the host-side state a step leaves behind) and the
    row arithmetic ``row0 = i * batch + lo``."""
    WARM, PERIODS = 6, (1, 2, 4)

    def __init__(self, ctx, batch, lo):
        self.ctx, self.batch, self.lo = ctx, int(batch), int(lo)
        self.n = 0                  # steps seen (interpreted or replayed) since the plan object exists
        self.steps = []             # recorded: (n, i, [(name, args)...], state after the step, state before it)
        self.plans = None           # per phase: (handle, state after, state before)
        self.period = 0
        self.off = os.environ.get("TN_NET_PLAN", "1") == "0"
        self.why = "TN_NET_PLAN=0" if self.off else ""
        self._cur = None

    @property
    def ready(self):
        return self.plans is not None

    # -- recording ----------------------------------------------------------------------------------------------
    def begin(self, i, pre=None):
        """Before an interpreted ordinary step; ``pre`` = the owner's host-side state the step starts from."""
        if self.off or self.ready or self.n < self.WARM:
            return
        self._cur = (self.n, int(i), [], pre)
        self.ctx.rec = self._cur[2]

    def end(self, state, ok=True):
        """After an interpreted step (ok False: it was not an ordinary step, or it raised -- recording starts
        over)."""
        self.n += 1
        if self.ctx.rec is not None and self._cur is not None and self.ctx.rec is self._cur[2]:
            self.ctx.rec = None
            tainted = self.ctx.rec_tainted
            self.ctx.rec_tainted = False
            if ok and not tainted:
                self.steps.append(self._cur[:3] + (state, self._cur[3]))
                if len(self.steps) >= 3 * max(self.PERIODS):
                    self._build()
            else:
                if tainted:
                    self._give_up("host-side work inside the step")
                self.steps = []
        elif not ok:
            self.steps = []
        self._cur = None

    def restart(self, why=""):
        """Something the recorded steps depend on has changed (learning rate, schedule): watch again."""
        self.drop()
        self.steps, self.n, self.why = [], 0, why

    def _give_up(self, why):
        self.off, self.why, self.steps = True, why, []

    def _row(self, i):
        return (i * self.batch + self.lo) & 0xFFFFFFFFFFFFFFFF

    def _build(self):
        try:
            enc = [[(name,) + _encode(name, args) for name, args in calls] for _, _, calls, _, _ in self.steps]
        except PlanError as e:
            return self._give_up(str(e))
        idx = [st[1] for st in self.steps]
        for P in self.PERIODS:
            rows = self._match(enc, idx, P)
            if rows is not None:
                break
        else:
            return self._give_up("the call sequence does not repeat within %d steps" % max(self.PERIODS))
        lib, h = self.ctx.lib, self.ctx.h
        plans = [None] * P
        for s in range(len(self.steps) - P, len(self.steps)):
            n, i, _, state, pre = self.steps[s]
            handle = ctypes.c_void_p()
            self.ctx.call("tn_net_plan_create", ctypes.byref(handle))
            for c, (name, kinds, vals) in enumerate(enc[s]):
                strides = [0] * len(vals)
                vals = list(vals)
                for k in rows[n % P].get(c, ()):
                    vals[k], strides[k] = self.lo, self.batch          # row0 = lo + i * batch
                K = (ctypes.c_uint8 * len(kinds))(*kinds)
                V = (ctypes.c_uint64 * len(vals))(*vals)
                S = (ctypes.c_int64 * len(strides))(*strides)
                rc = lib.tn_net_plan_add(h, handle, name.encode(), len(kinds), K, V, S)
                if rc:
                    _lib.check(h, rc, "tn_net_plan_add")
            plans[n % P] = (handle, state, pre)
        self.plans, self.period, self.steps = plans, P, []

    def _match(self, enc, idx, P):
        """The recorded steps repeat with period P: {phase: {call: [row arguments]}} or None."""
        rows = {}
        for ph in range(P):
            mine = [s for s in range(len(enc)) if self.steps[s][0] % P == ph]
            if len(mine) < 2:
                return None
            if any(self.steps[s][4] != self.steps[mine[0]][4] for s in mine):
                return None                             # a phase is identified by the state its steps start from
            ref = enc[mine[0]]
            cand = None
            for s in mine:
                if len(enc[s]) != len(ref) or any(a[0] != b[0] or a[1] != b[1] for a, b in zip(enc[s], ref)):
                    return None
            # arguments that differ between steps of the phase must equal that step's row0 everywhere
            cand = {}
            for c, (name, kinds, vals) in enumerate(ref):
                for k in range(len(vals)):
                    col = [enc[s][c][2][k] for s in mine]
                    if any(v != col[0] for v in col):
                        if kinds[k] != 0 or any(col[j] != self._row(idx[s]) for j, s in enumerate(mine)):
                            return None
                        cand.setdefault(c, []).append(k)
            if len(set(idx[s] for s in mine)) < 2:
                # every recorded step of the phase saw the same minibatch: a constant cannot be told from its row
                return None
            rows[ph] = cand
        return rows

    # -- replay -------------------------------------------------------------------------------------------------
    def step(self, i, cur=None):
        """Replay the phase whose recorded step STARTED from the owner's current host-side state ``cur`` and return
        the state it leaves behind -- or None when no recorded phase starts there (an interpreted step in between
        left the ping-pong buffers / stream parity elsewhere: the caller interprets this step too).  The baked
        pointers of a phase are only valid from that state, so the phase is never picked by counting steps."""
        k = self.n % self.period
        order = [k] + [j for j in range(self.period) if j != k]
        for j in order:
            handle, state, pre = self.plans[j]
            if pre == cur:
                break
        else:
            return None
        rc = self.ctx.lib.tn_net_step(self.ctx.h, handle, int(i))
        if rc:
            _lib.check(self.ctx.h, rc, "tn_net_step")
        self.n = j + 1
        return state

    def drop(self):
        if self.plans:
            for handle, _, _ in self.plans:
                self.ctx.lib.tn_net_plan_destroy(self.ctx.h, handle)
        self.plans = None

    def __del__(self):
        try:
            self.drop()
        except Exception:       # interpreter teardown
            pass
