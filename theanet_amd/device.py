"""Device context and device-resident arrays.

Counterpart of the reference's ``theano.shared`` variables (train.py:18-19,
theanet/layer/weights.py:18-22,73-79): datasets, weights, velocities and every
activation live in HBM for the life of the net; only the minibatch index crosses
the host/device boundary per step.
"""
import ctypes
import os

import numpy as np

from . import _lib

_context = None


class Context:
    """One tn_ctx: one GPU, one HIP stream.  One process drives one GPU
    (LOCAL_RANK picks it when launched by torch.distributed.run)."""

    def __init__(self, device=None):
        self.lib = _lib.get_lib()
        self.backend = _lib.backend()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        h = ctypes.c_void_p()
        rc = self.lib.tn_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            msg = self.lib.tn_last_error(None)
            raise _lib.BackendError(
                "theanet_amd needs an AMD GPU (MI355X/gfx950); tn_ctx_create(%d) failed: %s. "
                "There is no CPU fallback." % (device, msg.decode() if msg else "?"))
        self.h = h
        self.device = device
        self.mm_dtype = "float32"  # operand precision of the conv products (NeuralNet's DTYPE training param)
        self._mm_set = ("float32", 1.0)
        self.ev_hook = None       # (name, nth) -> HIP-event bracket around that C-ABI call
        self._fns = {}            # name -> bound ctypes function (the attribute lookup costs per call otherwise)
        self._ev_seen = 0
        self.ev_pairs = []
        self.rec = None           # list collecting (name, args) of every call while a StepPlan watches a step (plan.py)
        self.rec_tainted = False  # ... set by host-side work that a replay of the calls would miss

    def call(self, name, *args):
        if self.rec is not None:
            self.rec.append((name, args))
        hook = self.ev_hook
        if hook is not None and hook[0] == name:
            self._ev_seen += 1
            if self._ev_seen == hook[1] or hook[1] == 0:         # nth 0: every call
                return self._timed_call(name, args)
        fn = self._fns.get(name)
        if fn is None:
            fn = self._fns[name] = getattr(self.lib, name)
        rc = fn(self.h, *args)
        if rc != 0:
            _lib.check(self.h, rc, name)

    # -- per-kernel timing with HIP events on the compute stream (bench.py roofline leg) --
    def _timed_call(self, name, args):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        for e in (a, b):
            rc = self.lib.tn_event_create(self.h, ctypes.byref(e))
            _lib.check(self.h, rc, "tn_event_create")
        self.lib.tn_event_record(self.h, a)
        rc = getattr(self.lib, name)(self.h, *args)
        self.lib.tn_event_record(self.h, b)
        self.ev_pairs.append((a, b))
        if rc != 0:
            _lib.check(self.h, rc, name)

    def time_calls(self, name, nth=1):
        """Bracket the nth call of C-ABI function ``name`` after each ``new_step()``."""
        self.ev_hook = (name, nth)
        self._ev_seen = 0
        self.ev_pairs = []

    def new_step(self):
        self._ev_seen = 0

    def collect_times_ms(self):
        out = []
        for a, b in self.ev_pairs:
            ms = ctypes.c_float()
            self.call("tn_event_elapsed_ms", a, b, ctypes.byref(ms))
            out.append(ms.value)
            self.lib.tn_event_destroy(self.h, a)
            self.lib.tn_event_destroy(self.h, b)
        self.ev_pairs = []
        self.ev_hook = None
        return out

    def sync(self):
        self.call("tn_sync")

    def set_matmul_dtype(self, dtype, grad_scale=1.0):
        """'float32' (the reference's floatX) or 'float16' = fp16 operands / fp32 accumulation for the
        3x3 conv products (tn_set_matmul_dtype); a no-op when already in that mode."""
        want = (dtype, float(grad_scale) if dtype == "float16" else 1.0)
        if want != self._mm_set:
            self.call("tn_set_matmul_dtype", 1 if dtype == "float16" else 0, want[1])
            self._mm_set = want
        self.mm_dtype = dtype

    def set_fc_matmul(self, mode):
        """'float32' (exact fp32 MFMA) or 'bf16x3' (six bf16 MFMA products of exactly split operands: tn_set_fc_matmul)."""
        if mode != getattr(self, "_fc_mm", "float32"):
            self.call("tn_set_fc_matmul", 1 if mode == "bf16x3" else 0)
            self._fc_mm = mode

    def info(self):
        name = ctypes.create_string_buffer(128)
        cus = ctypes.c_int()
        mem = ctypes.c_size_t()
        self.call("tn_device_info", name, 128, ctypes.byref(cus), ctypes.byref(mem))
        return name.value.decode(), cus.value, mem.value

    # -- allocation ----------------------------------------------------------
    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        a = DeviceArray(self, shape, dtype)
        a.fill_bytes(0)
        return a

    def array(self, data, dtype=None):
        data = np.ascontiguousarray(data, dtype=dtype)
        a = DeviceArray(self, data.shape, data.dtype)
        a.set_value(data)
        return a


def get_context():
    """Process-wide context (created on first use; raises without a GPU)."""
    global _context
    if _context is None:
        _context = Context()
    return _context


class DeviceArray:
    """Typed view of HBM.  ``get_value``/``set_value`` mirror the reference's
    shared-variable accessors (weights.py:18-22)."""

    def __init__(self, ctx, shape, dtype=np.float32, ptr=None, base=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if hasattr(shape, "__len__") else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self.base = base            # keeps the owning allocation alive for views
        if ptr is None:
            p = ctypes.c_void_p()
            ctx.call("tn_alloc", self.nbytes, ctypes.byref(p))
            self.ptr = p.value
            self._owns = True
        else:
            self.ptr = int(ptr)
            self._owns = False

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    def view(self, offset_elems, shape, dtype=None):
        dtype = self.dtype if dtype is None else np.dtype(dtype)
        return DeviceArray(self.ctx, shape, dtype,
                           ptr=self.ptr + offset_elems * self.dtype.itemsize,
                           base=self.base or self)

    def reshape(self, *shape):
        if len(shape) == 1 and hasattr(shape[0], "__len__"):
            shape = tuple(shape[0])
        if -1 in shape:
            known = -int(np.prod(shape))
            shape = tuple(self.size // known if s == -1 else s for s in shape)
        assert int(np.prod(shape)) == self.size
        return DeviceArray(self.ctx, shape, self.dtype, ptr=self.ptr, base=self.base or self)

    def flatten(self, ndim=1):
        """``tensor.flatten(2)`` of the reference (neuralnet.py:168-169)."""
        if ndim == 2:
            return self.reshape(self.shape[0], -1)
        return self.reshape(self.size)

    def get_value(self, borrow=True):
        out = np.empty(self.shape, self.dtype)
        self.ctx.call("tn_d2h", out.ctypes.data, self.ptr, self.nbytes)
        return out

    def set_value(self, data):
        data = np.ascontiguousarray(data, dtype=self.dtype)
        assert data.size == self.size, (data.shape, self.shape)
        self.ctx.call("tn_h2d", self.ptr, data.ctypes.data, self.nbytes)

    def fill_bytes(self, byte):
        self.ctx.call("tn_memset", self.ptr, byte, self.nbytes)

    def __del__(self):
        if getattr(self, "_owns", False) and self.ptr:
            try:
                self.ctx.lib.tn_free(self.ctx.h, self.ptr)
            except Exception:   # interpreter teardown
                pass
            self.ptr = 0

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, ptr=0x%x)" % (self.shape, self.dtype, self.ptr)


class C8Array(DeviceArray):
    """An fp16-RESIDENT activation / gradient tensor (DTYPE 'float16'): logical (N, C, H, W), stored
    [N][ceil(C/8)][H][W][8] halfs -- one 16-byte cell = the 8 channels of an octet at one pixel, channels beyond C zero
    (theanet_amd/csrc/conv_c8.hip).  ``get_value`` returns the logical NCHW float32 array."""

    def __init__(self, ctx, n, c, h, w):
        self.c8 = (int(c), int(h), int(w))
        super().__init__(ctx, (n, (c + 7) // 8, h, w, 8), np.uint16)

    def get_value(self, borrow=True):
        raw = DeviceArray.get_value(self).view(np.float16)
        n, c8, h, w, _ = raw.shape
        return raw.transpose(0, 1, 4, 2, 3).reshape(n, c8 * 8, h, w)[:, :self.c8[0]].astype(np.float32)

    def set_value(self, data):
        data = np.asarray(data, np.float32)
        n, c, h, w = data.shape
        buf = np.zeros((n, self.shape[1] * 8, h, w), np.float16)
        buf[:, :c] = data.astype(np.float16)
        DeviceArray.set_value(self, np.ascontiguousarray(buf.reshape(n, self.shape[1], 8, h, w).transpose(0, 1, 3, 4, 2)).view(np.uint16))


class HostBuffer:
    """Page-locked host memory (tn_host_alloc) viewed as a numpy array: the target of tn_d2h_early copies."""

    def __init__(self, ctx, shape, dtype=np.float32):
        self.ctx = ctx
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        ctx.call("tn_host_alloc", self.nbytes, ctypes.byref(p))
        self.ptr = p.value
        self.array = np.frombuffer((ctypes.c_char * self.nbytes).from_address(self.ptr), dtype=dtype).reshape(shape)

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                self.array = None
                self.ctx.lib.tn_host_free(self.ctx.h, self.ptr)
            except Exception:   # interpreter teardown
                pass
            self.ptr = 0


def share(data, dtype=np.float32, borrow=True):
    """train.py:18-19 ``share()``: put a host array in HBM once."""
    if isinstance(data, DeviceArray):
        return data
    return get_context().array(np.asarray(data), dtype=dtype)
