"""CPU-side checks of the drop-in boundary: the shared library loads, exports every
symbol include/theanet_hip.h declares, the ctypes table covers the header one to one,
and the product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "theanet_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|size_t|char\s*\*)\s+\*?\s*(tn_\w+)\s*\(", text, re.M)
    assert len(names) > 50
    return names


def test_header_symbols_are_exported_and_bound():
    from theanet_amd import _lib
    lib = _lib.get_lib()
    names = declared_functions()
    for n in names:
        assert hasattr(lib, n), "libtheanet_hip.so does not export " + n
        assert n in _lib.SIGNATURES, "ctypes table lacks " + n
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.tn_version() >= 100


def test_cpu_backend_library_exports_the_same_abi():
    """lib/libtheanet_cpu.so (C++/OpenMP, opt-in via THEANET_BACKEND=cpu) implements the same header."""
    from theanet_amd import _lib
    if not os.path.isfile(_lib.CPU_LIB_PATH):
        pytest.skip("libtheanet_cpu.so not built")
    lib = _lib.bind(_lib.CPU_LIB_PATH, ctypes.RTLD_LOCAL)
    for n in declared_functions():
        assert hasattr(lib, n), "libtheanet_cpu.so does not export " + n
    assert lib.tn_version() >= 100


def test_struct_layout_matches_header():
    import numpy as np
    seg = np.dtype([('p', 'u8'), ('v', 'u8'), ('g', 'u8'), ('n', 'u8'),
                    ('momentum', 'f4'), ('rate', 'f4'), ('L1', 'f4'), ('L2', 'f4')])
    assert seg.itemsize == 48           # tn_sgd_seg


def test_argument_count_matches_header():
    from theanet_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_no_gpu_means_loud_failure(monkeypatch):
    from theanet_amd import _lib, device
    lib = _lib.get_lib()
    n = ctypes.c_int(0)
    rc = lib.tn_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    monkeypatch.setattr(device, "_context", None)
    from theanet_amd import NeuralNet
    with pytest.raises(_lib.BackendError, match="no CPU fallback"):
        NeuralNet([("InputLayer", {"img_sz": 8}), ("SoftmaxLayer", {"n_out": 2})],
                  {"SEED": 1, "BATCH_SZ": 2, "INIT_LEARNING_RATE": .1, "EPOCHS_TO_HALF_RATE": 1})


def test_missing_library_message(monkeypatch, tmp_path):
    from theanet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.BackendError, match="not built"):
        _lib.get_lib()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under theanet_amd/ (nor train.py) may
    import or execute it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "theanet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) or "oracle." in src:
                    bad.append(os.path.join(base, f))
    src = open(os.path.join(ROOT, "train.py")).read()
    if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
        bad.append("train.py")
    assert not bad, bad


def test_train_py_helpers_match_the_reference_run_fixture():
    """train.py's dataset coercion and rolling test windows against tests/golden/train_helpers.npz, which holds what
    the REFERENCE's own fixdim (train.py:22-34) and get_test_indices (train.py:170-176) return (run from
    /root/reference by tests/golden/make_golden.py::make_train_helpers; only inputs / outputs are stored)."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("tn_train", os.path.join(ROOT, "train.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)                 # defines helpers only; run() needs a device
    ref = np.load(os.path.join(ROOT, "tests", "golden", "train_helpers.npz"))
    for k in range(5):
        got = tr.as_nchw(ref["fixdim_in%d" % k])
        assert got.shape == ref["fixdim_out%d" % k].shape and np.array_equal(got, ref["fixdim_out%d" % k])
    with pytest.raises(ValueError):
        tr.as_nchw(np.zeros((2, 2, 2, 2, 2), np.float32))
    with pytest.raises(AssertionError):
        tr.as_nchw(np.zeros((2, 50), np.float32))
    for k in range(5):
        tot, bsz, samp = (int(v) for v in ref["win_case%d" % k])
        w = tr.BatchWindows(tot, bsz, samp)
        seq = np.array([w.next() for _ in range(9)])
        assert np.array_equal(seq, ref["win_seq%d" % k]), (tot, bsz, samp)
    # the report strings of neuralnet.py:16-51, from the reference's own functions
    import ast
    from theanet_amd import neuralnet as nn
    from oracle import theanet_oracle as O
    with open(os.path.join(ROOT, "params", "mnist.prms")) as fh:
        prms = ast.literal_eval(fh.read())
    prms["layers"][0][1]["img_sz"] = 28
    prms["training_params"]["SEED"] = 555555
    wts = [[p for p in l.params] for l in O.OracleNet(prms["layers"], prms["training_params"]).L]   # adds CUR_EPOCH
    assert nn.get_layers_info(prms["layers"]) == str(ref["info_layers"])
    assert nn.get_training_params_info(prms["training_params"]) == str(ref["info_prms"])
    assert nn.get_wts_info(wts) == str(ref["info_wts"])
    assert nn.get_wts_info(wts, True) == str(ref["info_wts_detailed"])
