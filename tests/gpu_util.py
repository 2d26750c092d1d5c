"""Helpers for the -m gpu parity tests: everything goes through the C-ABI."""
import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ctx():
    from theanet_amd.device import get_context
    return get_context()


_KEEP = []      # device arrays made by dev() stay alive until the test ends (conftest clears)


def dev(a, dtype=None):
    d = ctx().array(np.ascontiguousarray(a), dtype=dtype)
    _KEEP.append(d)
    return d


def empty(shape, dtype=np.float32):
    return ctx().empty(shape, dtype)


def call(name, *args):
    ctx().call(name, *args)


def load_prms(name, img_sz=None, seed=555555, batch=None):
    with open(os.path.join(ROOT, "params", name)) as fh:
        prms = ast.literal_eval(fh.read())
    if img_sz is not None:
        prms["layers"][0][1]["img_sz"] = img_sz
    prms["training_params"]["SEED"] = seed
    if batch:
        prms["training_params"]["BATCH_SZ"] = batch
    return prms


def act_code(name):
    from theanet_amd.layer.layer import activation_by_name
    a = activation_by_name(name)
    return a.kind, a.prm


def assert_close(got, want, rtol=1e-4, atol=1e-5, what=""):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = atol + rtol * np.abs(want.astype(np.float64))
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: %d/%d mismatches, worst at %s: got %r want %r" %
                             (what, bad.sum(), bad.size, i, got[i], want[i]))
