"""ctypes binding of libtheanet_hip.so (C-ABI declared in include/theanet_hip.h).

The HIP library IS the compute path: there is no CPU fallback.  If the shared
object is missing this module raises with build instructions; if no MI355X is
visible, context creation raises with the library's own error string.

libtheanet_cpu.so (theanet_amd/csrc_cpu: C++/OpenMP, the same C-ABI) is a separate, explicitly
selected backend -- ``THEANET_BACKEND=cpu`` in the environment -- for GPU-less plumbing runs
(BASELINE configs[0]), the timed CPU baseline and host-logic tests.  It is never picked
automatically: without that variable a missing GPU is an error.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint8, c_uint32, c_uint64, c_void_p)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtheanet_hip.so")
if os.environ.get("TN_HIP_LIB"):          # developer A/B of two builds of the HIP library on one box (tools/ab.py)
    LIB_PATH = os.path.abspath(os.environ["TN_HIP_LIB"])
CPU_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtheanet_cpu.so")


def backend():
    """'hip' (default) or 'cpu' -- only ever from an explicit THEANET_BACKEND."""
    b = os.environ.get("THEANET_BACKEND", "hip").lower()
    if b not in ("hip", "cpu"):
        raise BackendError("THEANET_BACKEND must be 'hip' or 'cpu', not %r" % b)
    return b

TN_ACT_LINEAR, TN_ACT_LEAKY, TN_ACT_TANH, TN_ACT_SIGMOID, TN_ACT_SOFTPLUS, TN_ACT_SCALED_TANH = range(6)
TN_UNIQUE_ID_BYTES = 128
TN_UPD_PLAIN, TN_UPD_LAZY, TN_UPD_DELAYED, TN_UPD_PIPE = range(4)     # modes of tn_sgd_update_net

P = c_void_p          # device or host pointer passed as integer
CTX = c_void_p

# name -> (restype, argtypes).  Mirrors include/theanet_hip.h one to one
# (tests/test_abi.py parses the header and checks nothing is missing).
SIGNATURES = {
    "tn_version": (c_int, []),
    "tn_device_count": (c_int, [POINTER(c_int)]),
    "tn_ctx_create": (c_int, [c_int, POINTER(CTX)]),
    "tn_ctx_destroy": (c_int, [CTX]),
    "tn_last_error": (c_char_p, [CTX]),
    "tn_sync": (c_int, [CTX]),
    "tn_stream_select": (c_int, [CTX, c_int]),
    "tn_stream_wait": (c_int, [CTX, c_int, c_int]),
    "tn_device_info": (c_int, [CTX, c_char_p, c_int, POINTER(c_int), POINTER(c_size_t)]),
    "tn_alloc": (c_int, [CTX, c_size_t, POINTER(c_void_p)]),
    "tn_free": (c_int, [CTX, P]),
    "tn_h2d": (c_int, [CTX, P, P, c_size_t]),
    "tn_d2h": (c_int, [CTX, P, P, c_size_t]),
    "tn_host_alloc": (c_int, [CTX, c_size_t, ctypes.POINTER(P)]),
    "tn_host_free": (c_int, [CTX, P]),
    "tn_d2h_early": (c_int, [CTX, P, P, c_size_t]),
    "tn_copy_sync": (c_int, [CTX]),
    "tn_d2h_early_ev": (c_int, [CTX, P, P, c_size_t, P]),
    "tn_d2d": (c_int, [CTX, P, P, c_size_t]),
    "tn_memset": (c_int, [CTX, P, c_int, c_size_t]),
    "tn_set_u32": (c_int, [CTX, P, c_uint32]),
    "tn_set_i64": (c_int, [CTX, P, c_int64]),
    "tn_set_f32": (c_int, [CTX, P, c_float]),
    "tn_add_u32": (c_int, [CTX, P, c_uint32]),
    "tn_net_plan_create": (c_int, [CTX, POINTER(c_void_p)]),
    "tn_net_plan_add": (c_int, [CTX, P, c_char_p, c_int, P, P, P]),
    "tn_net_step": (c_int, [CTX, P, c_int64]),
    "tn_net_plan_size": (c_int, [CTX, P]),
    "tn_net_plan_destroy": (c_int, [CTX, P]),
    "tn_graph_begin": (c_int, [CTX]),
    "tn_graph_end": (c_int, [CTX, POINTER(c_void_p)]),
    "tn_graph_launch": (c_int, [CTX, P]),
    "tn_graph_destroy": (c_int, [CTX, P]),
    "tn_event_create": (c_int, [CTX, POINTER(c_void_p)]),
    "tn_event_record": (c_int, [CTX, P]),
    "tn_event_wait": (c_int, [CTX, P]),
    "tn_event_elapsed_ms": (c_int, [CTX, P, P, POINTER(c_float)]),
    "tn_event_destroy": (c_int, [CTX, P]),
    "tn_event_query": (c_int, [CTX, P, POINTER(c_int)]),
    "tn_event_sync": (c_int, [CTX, P]),
    "tn_conv2d_fwd": (c_int, [CTX, P, P, P, P] + [c_int] * 10 + [c_int, c_float]),
    "tn_conv2d_wgrad": (c_int, [CTX, P, P, P, P] + [c_int] * 10),
    "tn_conv2d_dgrad": (c_int, [CTX, P, P, P] + [c_int] * 10 + [P, c_int, c_float]),
    "tn_set_matmul_dtype": (c_int, [CTX, c_int, c_float]),
    "tn_get_matmul_dtype": (c_int, [CTX]),
    "tn_set_fc_matmul": (c_int, [CTX, c_int]),
    "tn_c8_conv_supported": (c_int, [c_int] * 8),
    "tn_c8_conv_wgrad_supported": (c_int, [c_int] * 5),
    "tn_c8_wt_elems": (c_size_t, [c_int] * 3),
    "tn_c8_arrange_multi": (c_int, [CTX, P, c_int]),
    "tn_c8_conv_fwd": (c_int, [CTX, P, P, P, P, P] + [c_int] * 6 + [c_float, c_int, P]),
    "tn_c8_conv_dgrad": (c_int, [CTX, P, P, P] + [c_int] * 5 + [P, c_int, c_float, c_int, P, P]),
    "tn_c8_conv_wgrad": (c_int, [CTX, P, P, P, P] + [c_int] * 5 + [c_int, P]),
    "tn_c8_fc_supported": (c_int, [c_int] * 4),
    "tn_c8_fc_fwd": (c_int, [CTX, P, P, P, P] + [c_int] * 5 + [c_float, P]),
    "tn_c8_fc_fwd_dropout": (c_int, [CTX, P, P, P, P] + [c_int] * 5 + [c_float, P, c_float, c_uint64, c_uint32, P, c_uint64]),
    "tn_c8_fc_dgrad": (c_int, [CTX, P, P, P] + [c_int] * 4 + [P, c_int, c_float]),
    "tn_c8_fc_wgrad": (c_int, [CTX, P, P, P, P] + [c_int] * 4),
    "tn_c8_pack": (c_int, [CTX, P, c_int64, P, c_int, c_int, c_int, c_float]),
    "tn_c8_unpack": (c_int, [CTX, P, P, c_int, c_int, c_int, c_float]),
    "tn_pool_fwd": (c_int, [CTX, P, P] + [c_int] * 6),
    "tn_pool_bwd": (c_int, [CTX, P, P, P, P] + [c_int] * 6 + [c_int, c_float]),
    "tn_mean_fwd": (c_int, [CTX, P, P, c_int, c_int]),
    "tn_mean_bwd": (c_int, [CTX, P, P, c_int, c_int, P, c_int, c_float]),
    "tn_fc_fwd": (c_int, [CTX, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "tn_fc_fwd_dropout": (c_int, [CTX, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P, c_float,
                                  c_uint64, c_uint32, P, c_uint64]),
    "tn_fc_bwd": (c_int, [CTX, P, P, P, P, P, P, c_int, c_int, c_int, P, P, c_int, c_float, P]),
    "tn_fc_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "tn_fc_wgrad": (c_int, [CTX, P, P, P, P, c_int, c_int, c_int, P]),
    "tn_fc_dgrad": (c_int, [CTX, P, P, P, c_int, c_int, c_int, P, c_int, c_float, P]),
    "tn_dropout_mask": (c_int, [CTX, P, c_size_t, c_float, c_uint64, c_uint32, P, c_uint64]),
    "tn_scale_mask": (c_int, [CTX, P, P, c_float, P, c_size_t, P, c_int, c_float]),
    "tn_fc_softmax_nll": (c_int, [CTX, P, P, P, P, c_int, c_int, c_int, P, c_int64, P, P, P, P, P, P,
                                  c_float]),
    "tn_fc_softmax_train": (c_int, [CTX, P, P, P, P, c_int, c_int, c_int, P, c_int64, P, P, P, P, P, P,
                                    c_float, P, P, P, P, P, c_int, c_float, P]),
    "tn_softmax_nll": (c_int, [CTX, P, P, c_int64, P, P, P, P, P, P, c_int, c_int, c_float]),
    "tn_head_rows": (c_int, [CTX, c_int, c_int, c_float, P, P, c_int, P, c_int64, P, P, P, P, P, P, P, P,
                             c_int, c_int, c_float, c_float, c_int, c_float]),
    "tn_reduce_sum": (c_int, [CTX, P, c_size_t, c_float, P, c_int]),
    "tn_wtcost": (c_int, [CTX, P, c_size_t, c_float, c_float, P, c_int]),
    "tn_error_stats": (c_int, [CTX, P, P, c_int64, P, c_int, P]),
    "tn_sgd_update": (c_int, [CTX, P, P, P, c_size_t, c_float, c_float, P, c_float, c_float, c_float]),
    "tn_maxnorm": (c_int, [CTX, P, c_int, c_int, c_int, c_float]),
    "tn_maxnorm_multi": (c_int, [CTX, P, c_int]),
    "tn_defer_reductions": (c_int, [CTX, c_int]),
    "tn_defer_flush_step": (c_int, [CTX, P]),
    "tn_defer_discard": (c_int, [CTX]),
    "tn_elastic_convpool_supported": (c_int, [c_int] * 10),
    "tn_elastic_convpool_fwd_mask": (c_int, [CTX, P, c_int64, P, P, c_int, c_int, c_int, c_int, c_int,
                                             P, P, P, c_float, P, c_uint64, c_uint32, P, c_int64,
                                             P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_int, c_int, c_float]),
    "tn_rider_elastic_field": (c_int, [CTX, P, c_uint64, c_uint32, P, c_int, c_int, c_double, c_double,
                                       c_double, c_int, c_double, c_int, P, P, P, P]),
    "tn_rider_pending": (c_int, [CTX]),
    "tn_rider_cancel": (c_int, [CTX]),
    "tn_step_tail": (c_int, [CTX, P, c_int, c_size_t, P, c_float, P, c_int, c_float, P,
                             P, c_uint64, P, c_int, c_int, c_double, c_double, c_double, c_int, c_double,
                             c_int, P, P, P, P]),
    "tn_sgd_update_net": (c_int, [CTX, c_int, P, P, c_int, c_size_t, P, c_float, P, c_uint32, c_int, P, c_int, c_float, P]),
    "tn_sgd_update_net_maxnorm": (c_int, [CTX, c_int, P, P, c_int, c_size_t, P, c_float, P, c_uint32, c_int, P, c_int, c_float, P,
                                          P, c_int]),
    "tn_softmax_cost_ws_bytes": (c_size_t, [c_int]),
    "tn_softmax_nll_cost": (c_int, [CTX, P, P, c_int64, P, P, P, P, P, P, c_int, c_int, c_float,
                                    c_float, P, P]),
    "tn_conv_mfma_supported": (c_int, [c_int] * 4),
    "tn_convpool_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "tn_convpool_fwd": (c_int, [CTX, P, P, P, P] + [c_int] * 12 + [c_int, c_float]),
    "tn_convpool_fwd_mask": (c_int, [CTX, P, P, P, P, P] + [c_int] * 12 + [c_int, c_float]),
    "tn_convpool_bwd_mask": (c_int, [CTX, P, P, P, P, P, P, P] + [c_int] * 12 + [c_int, c_float]),
    "tn_convblock_mask_supported": (c_int, [c_int] * 12),
    "tn_convpool_tile_supported": (c_int, [c_int] * 13),
    "tn_convpool_bwd_mask_dx": (c_int, [CTX, P, P, P, P, P, P, P, P] + [c_int] * 12 + [c_int, c_float,
                                                                                     P, c_int, c_float]),
    "tn_convblock_bwd_mask": (c_int, [CTX, P, P, P, P, P, P, P, P] + [c_int] * 12 + [c_int, c_float]),
    "tn_convblock_supported": (c_int, [c_int] * 7),
    "tn_convblock_bwd": (c_int, [CTX, P, P, P, P, P, P, P] + [c_int] * 12 + [c_int, c_float]),
    "tn_convpool_bwd": (c_int, [CTX, P, P, P, P, P, P, P] + [c_int] * 12 + [c_int, c_float]),
    "tn_elastic_draws_count": (c_size_t, [c_int, c_int]),
    "tn_elastic_draws": (c_int, [CTX, P, c_int, c_int, c_uint64, c_uint32, P]),
    "tn_elastic_field": (c_int, [CTX, P, c_int, c_int, c_double, c_double, c_double, c_int, c_double,
                                 c_int, P, P, P, P]),
    "tn_elastic_field_gen": (c_int, [CTX, P, c_uint64, c_uint32, P, c_int, c_int, c_double, c_double,
                                     c_double, c_int, c_double, c_int, P, P, P, P]),
    "tn_elastic_apply": (c_int, [CTX, P, c_int64, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                 P, P, P, c_float, P, c_uint64, c_uint32, P, c_int64]),
    "tn_c8_elastic_apply": (c_int, [CTX, P, c_int64, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                    P, P, P, c_float, P, c_uint64, c_uint32, P, c_int64]),
    "tn_elastic_apply_bwd": (c_int, [CTX, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_float, P,
                                     c_uint64, c_uint32, P, c_int64, P, c_int, c_float]),
    "tn_color_factors": (c_int, [CTX, P, c_int, c_int, c_double, c_double, P, c_uint64, c_uint32, P, c_int64]),
    "tn_color_apply": (c_int, [CTX, P, c_int64, P, P, c_int, c_int, c_int, c_float]),
    "tn_color_apply_bwd": (c_int, [CTX, P, c_int64, P, P, P, c_int, c_int, c_int, c_float, P, c_int, c_float]),
    "tn_aux_mix": (c_int, [CTX, P, c_int64, P, c_int, c_int, c_float, c_int, P, c_uint64, c_uint32, P, c_int64]),
    "tn_copy_cols": (c_int, [CTX, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, c_float]),
    "tn_deformer_transform": (c_int, [CTX, P, P, c_int, c_int, c_int, c_double, c_double, c_double,
                                      P, c_uint64, c_int64]),
    "tn_gather_rows": (c_int, [CTX, P, P, P, c_int, c_size_t]),
    "tn_comm_unique_id": (c_int, [CTX, P]),
    "tn_comm_init": (c_int, [CTX, P, c_int, c_int]),
    "tn_comm_destroy": (c_int, [CTX]),
    "tn_allreduce_sum": (c_int, [CTX, P, c_size_t]),
    "tn_allreduce_max": (c_int, [CTX, P, c_size_t]),
    "tn_allreduce_sum_async": (c_int, [CTX, P, c_size_t, P]),
    "tn_allreduce_sum_rsag": (c_int, [CTX, P, c_size_t, c_int, P]),
    "tn_axpby": (c_int, [CTX, P, P, c_size_t, c_float, c_float]),
}

_lib = None


class BackendError(RuntimeError):
    """Raised when libtheanet_hip.so is missing or a C-ABI call returns an error."""


def bind(path, mode=ctypes.RTLD_GLOBAL):
    """dlopen ``path`` and attach the prototypes of every declared entry point."""
    lib = ctypes.CDLL(path, mode=mode)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def get_lib():
    """Load the backend library (once) and attach the prototypes.  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if backend() == "cpu":
        if not os.path.isfile(CPU_LIB_PATH):
            raise BackendError("theanet_amd: THEANET_BACKEND=cpu but %s is not built "
                               "(`make -C theanet_amd/csrc_cpu`)." % CPU_LIB_PATH)
        _lib = bind(CPU_LIB_PATH, ctypes.RTLD_LOCAL)
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise BackendError(
            "theanet_amd: HIP backend %s not built. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C theanet_amd/csrc` (needs hipcc, targets gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    _lib = bind(LIB_PATH)
    return _lib


def check(ctx_handle, rc, what=""):
    if rc != 0:
        msg = get_lib().tn_last_error(ctx_handle)
        raise BackendError("%s failed (rc=%d): %s" % (what or "C-ABI call", rc,
                                                      msg.decode() if msg else "?"))
