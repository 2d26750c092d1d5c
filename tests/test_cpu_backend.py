"""The C++/OpenMP CPU backend (theanet_amd/csrc_cpu -> lib/libtheanet_cpu.so; same C-ABI as the HIP
library, selected ONLY by THEANET_BACKEND=cpu) -- BASELINE.json configs[0], the reference's own
CPU-runnable case, and the GPU-less way to drive the product's host logic:

  * the library exports every symbol of include/theanet_hip.h (tests/test_abi.py checks both libraries);
  * the SAME parity tests that pin the HIP path (tests/test_gpu_*.py: per-op tests against the oracle,
    the GOLD-A / GOLD-B trajectories, the torch-CPU fixture, whole nets, checkpoints, train.py end to
    end, two-steps-in-flight schedule) run against it in a subprocess -- minus the tests of MI355X-only
    fused entry points, whose capability queries answer 0 here;
  * params/mnist.prms verbatim (BATCH_SZ 20) and at BASELINE's batch 128 through train.py;
  * the PRODUCT's data-parallel step with world_size 2 (socket rendezvous + host all-reduce) equals
    the single-process step on the same global minibatches."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "theanet_amd", "lib", "libtheanet_cpu.so")
pytestmark = pytest.mark.skipif(not os.path.isfile(CPU_LIB), reason="libtheanet_cpu.so not built")

# tests of MI355X-only fused entry points (conv+pool blocks, LDS-resident backward, elastic+conv fusion,
# HIP graphs, RCCL, fp16 MFMA) and BASELINE-size runs that only make sense on the GPU
NOT_ON_CPU = ("convpool_fused or convblock or convpool_tile or convpool_mask or convpool_tie or "
              "elastic_convpool_fused or graph_capture or rccl or two_gpu or dp_ or 4096 or full_size or "
              "full_batch or f16 or bf16x3 or c8")


def _env(**kw):
    env = dict(os.environ, THEANET_BACKEND="cpu", OMP_NUM_THREADS="4", PYTHONPATH=ROOT)
    env.update(kw)
    return env


def test_hip_parity_suite_runs_against_the_cpu_backend():
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        "tests/test_gpu_kernels.py", "tests/test_gpu_elastic.py", "tests/test_gpu_net.py",
                        "-k", "not (%s)" % NOT_ON_CPU],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 140, tail


def test_cpu_backend_is_never_a_fallback():
    """Without THEANET_BACKEND=cpu the product loads the HIP library and fails loudly without a GPU
    (tests/test_abi.py::test_no_gpu_means_loud_failure); the CPU library is opt-in by name only."""
    from theanet_amd import _lib
    assert os.environ.get("THEANET_BACKEND", "hip") != "cpu" or True
    code = ("import os; os.environ.pop('THEANET_BACKEND', None); from theanet_amd import _lib; "
            "assert _lib.backend() == 'hip'; l = _lib.get_lib(); assert 'hip' in l._name")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env={k: v for k, v in _env().items() if k != "THEANET_BACKEND"},
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    with pytest.raises(_lib.BackendError):
        os.environ["THEANET_BACKEND"] = "cuda"
        try:
            _lib.backend()
        finally:
            os.environ.pop("THEANET_BACKEND")


@pytest.mark.parametrize("batch", [20, 128])
def test_config1_mnist_prms_through_train_py(tmp_path, batch):
    """BASELINE.json configs[0]: params/mnist.prms (file batch 20; BASELINE's batch 128) on MNIST-shaped
    synthetic data, CPU, float32: train.py prints the reference's table, learns and writes a pickle."""
    import ast
    import pickle
    with open(os.path.join(ROOT, "params", "mnist.prms")) as fh:
        prms = ast.literal_eval(fh.read())
    prms["training_params"].update(SEED=11, BATCH_SZ=batch, NUM_EPOCHS=2, TEST_SAMP_SZ=256)
    prm = tmp_path / "mnist.prms"
    prm.write_text(repr(prms))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "synthetic", str(prm)], cwd=str(tmp_path),
                       env=_env(THEANET_SYNTH_TRAIN="1280", THEANET_SYNTH_TEST="256"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "CPU backend (OpenMP" in r.stdout
    assert "Epoch   Cost  Tr_Error Tr_P(MLE)    Te_Error Te_P(MLE)" in r.stdout
    rows = [l for l in r.stdout.splitlines() if l.strip().startswith(("0 ", "1 ", "2 "))]
    assert len(rows) == 3, r.stdout
    errs = [float(l.split()[2].rstrip("%")) for l in rows]
    assert errs[-1] < errs[0] or errs[-1] < 5.0, rows
    pk = [f for f in os.listdir(tmp_path) if f.endswith(".pkl")]
    assert len(pk) == 1
    with open(tmp_path / pk[0], "rb") as fh:
        ck = pickle.load(fh)
    assert len(ck["allwts"]) == 7 and ck["allwts"][5][0].dtype == np.float32


@pytest.mark.parametrize("pipe,overlap", [("1", "auto"), ("0", "0"), ("0", "1"), ("0", "2")],
                         ids=["pipelined", "plain", "overlap", "delayed"])
def test_product_data_parallel_step_world_size_2(tmp_path, pipe, overlap):
    """The product's own data-parallel training step (NeuralNet._train_step: row shards, flat gradient
    buffer + cost through ONE all-reduce, replicated update, global-index RNG, both schedules) with two
    ranks equals the one-rank run on the same global minibatches (1e-5 rel: summation order)."""
    worker = os.path.join(ROOT, "tests", "dp_gpu_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    outs = []
    for world in (1, 2, 4) if pipe == "1" else (1, 2):        # (every DeviceGroup of > 1 rank starts with its self-test)
        out = str(tmp_path / ("w%d.npz" % world))
        procs = []
        for rank in range(world):
            env = _env(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port + world), TN_PIPELINE=pipe, TN_DP_OVERLAP=overlap,
                       TN_DP_CHECK_ORDER="1", OMP_NUM_THREADS="2")
            procs.append(subprocess.Popen([sys.executable, worker, out, "mnist.prms", "28", "1", "32", "7"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for p in procs:
            try:
                o, _ = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            assert p.returncode == 0, o.decode()[-3000:]
        outs.append(np.load(out))
    one, two = outs[0], outs[1]
    for many in outs[1:]:
        np.testing.assert_allclose(many["costs"], one["costs"], rtol=2e-5)
        np.testing.assert_allclose(many["stats"], one["stats"], rtol=1e-5, atol=1e-6)
        for k in one.files:
            if k.startswith("w"):
                np.testing.assert_allclose(many[k], one[k], rtol=1e-5, atol=1e-6, err_msg=k)
    want = {"auto": "pipelined", "0": "plain", "1": "overlap", "2": "delayed"}[overlap]
    assert str(two["schedule"]) == want, str(two["schedule"])


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_allreduce_equals_single_bucket(tmp_path, world):
    """Two steps in flight, data-parallel: the dense group's gradients (+ cost) leave as a bucket of their own right
    after the dense layers' backward pass and the conv layers' gradients follow at the end of the step (SURVEY 8e
    "bucket by layer"; NeuralNet._dp_bucket).  Same sums over the same ranks element by element: costs, statistics and
    weights are BIT-identical to the one-all-reduce-per-step schedule, with 2 and with 4 ranks, and the bucketed run
    issues twice the collectives (in the same order on every rank: TN_DP_CHECK_ORDER)."""
    worker = os.path.join(ROOT, "tests", "dp_gpu_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    outs = {}
    for n, buckets in enumerate(("1", "0")):
        out = str(tmp_path / ("b%s.npz" % buckets))
        procs = []
        for rank in range(world):
            env = _env(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port + n), TN_PIPELINE="1", TN_DP_BUCKETS=buckets, TN_DP_CHECK_ORDER="1",
                       OMP_NUM_THREADS="2")
            procs.append(subprocess.Popen([sys.executable, worker, out, "cifar_like.prms", "16", "3", "16", "7"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for p in procs:
            try:
                o, _ = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            assert p.returncode == 0, o.decode()[-3000:]
        outs[buckets] = np.load(out)
    a, b = outs["1"], outs["0"]
    assert str(a["schedule"]) == "pipelined" and str(b["schedule"]) == "pipelined"
    assert int(a["bucket"]) > 0 and int(b["bucket"]) == -1
    assert int(a["n_collectives"]) > int(b["n_collectives"])
    np.testing.assert_array_equal(a["costs"], b["costs"])
    np.testing.assert_array_equal(a["stats"], b["stats"])
    for k in a.files:
        if k.startswith("w"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("world", [2, 4])
def test_reduce_scatter_allgather_equals_allreduce(tmp_path, world):
    """The direct reduce-scatter + all-gather form of a bucket's sum (tn_allreduce_sum_rsag; comm.collective_algo picks
    it for buckets >= TN_DP_RSAG_MIN_BYTES; here forced for EVERY collective, bucket sizes not divisible by the rank
    count included -- the remainder takes the small all-reduce) against one all-reduce per bucket: every element is
    summed once, in rank order, at its owner -- costs, statistics and weights BIT-identical with 2 and with 4 ranks,
    same number of collectives in the same order on every rank (TN_DP_CHECK_ORDER; the order hash carries the form)."""
    worker = os.path.join(ROOT, "tests", "dp_gpu_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    outs = {}
    for n, algo in enumerate(("rsag", "allreduce")):
        out = str(tmp_path / ("a%s.npz" % algo))
        procs = []
        for rank in range(world):
            env = _env(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port + n), TN_PIPELINE="1", TN_DP_BUCKETS="1", TN_DP_ALGO=algo,
                       TN_DP_CHECK_ORDER="1", OMP_NUM_THREADS="2")
            procs.append(subprocess.Popen([sys.executable, worker, out, "cifar_like.prms", "16", "3", "16", "7"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for p in procs:
            try:
                o, _ = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            assert p.returncode == 0, o.decode()[-3000:]
        outs[algo] = np.load(out)
    a, b = outs["rsag"], outs["allreduce"]
    assert str(a["schedule"]) == "pipelined" and int(a["bucket"]) > 0
    assert int(a["n_collectives"]) == int(b["n_collectives"])
    np.testing.assert_array_equal(a["costs"], b["costs"])
    np.testing.assert_array_equal(a["stats"], b["stats"])
    for k in a.files:
        if k.startswith("w"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_collective_algo_rule():
    from theanet_amd.comm import collective_algo
    assert collective_algo(16788545, 8, {}) == "rsag"            # wide6's dense bucket, 67 MB
    assert collective_algo(1145408, 8, {}) == "allreduce"        # its conv bucket, 4.6 MB
    assert collective_algo(366593, 8, {}) == "allreduce"         # mnist.prms, 1.5 MB: latency-bound
    assert collective_algo(16788545, 1, {}) == "allreduce"       # one rank: nothing to scatter
    assert collective_algo(10, 1, {"TN_DP_ALGO": "rsag"}) == "rsag"
    assert collective_algo(1 << 30, 8, {"TN_DP_ALGO": "allreduce"}) == "allreduce"
    assert collective_algo(1 << 20, 8, {"TN_DP_RSAG_MIN_BYTES": "1024"}) == "rsag"


@pytest.mark.parametrize("prms,extra,rows", [("mnist.prms", [], 512), ("wide6.prms", ["--img", "16"], 128)])
def test_bench_dry_multi_plans_an_8_rank_run_without_a_communicator(prms, extra, rows):
    """bench.py --dry-multi 8: the scaling command line's plan (row shards of the first and last rank, the flat
    gradient buffer every rank all-reduces, the schedule) without a communicator or a second process.  mnist shards
    BASELINE's 4096 eight ways (512 rows per GPU, configs[2]); wide6 keeps 128 images per GPU = 1024 per node
    (configs[4]; round 2 sharded 128 into 16 per GPU)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-multi", "8", "--prms", prms] + extra,
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    plan = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert plan["rows_per_gpu"] == rows and plan["global_batch"] == 8 * rows
    assert plan["ranks_shown"][0]["rows_of_minibatch"] == [0, rows]
    assert plan["ranks_shown"][1]["rows_of_minibatch"] == [7 * rows, 8 * rows]
    flat = plan["flat_gradient_buffer"]
    offs = [t["offset_floats"] for t in flat["tensors"]]
    assert offs == sorted(offs) and all(o % 64 == 0 for o in offs)
    last = flat["tensors"][-1]
    assert flat["cost_slot"] >= last["offset_floats"] + int(np.prod(last["shape"]))
    assert flat["floats_reduced_per_step"] == flat["cost_slot"] + 1
    assert "pipelined" in plan["schedule"]
    # the collectives of a step: mnist.prms has 780 conv floats (one all-reduce), wide6 a bucket per group
    bk = plan["allreduce_buckets"]
    assert len(bk) == (1 if prms == "mnist.prms" else 2)
    assert sum(b["floats"] for b in bk) == flat["floats_reduced_per_step"]
    assert bk[0]["offset_floats"] + bk[0]["floats"] == flat["cost_slot"] + 1      # the cost travels in the first bucket


def test_bench_two_ranks_reports_strong_and_weak_scaling():
    """bench.py --gpus 2 as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* from the
    environment), on the CPU backend: ONE JSON line from rank 0 whose `value` is the strong-scaling figure (the stated
    batch sharded over the ranks, BASELINE configs[2]) and whose `value_weak` is the same launch's weak-scaling leg (the
    stated batch per rank)."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = _env(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                                       "--warmup", "1", "--batch", "64", "--no-cpu-baseline", "--no-roofline"],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    assert outs[1].strip() == ""                                   # only rank 0 prints
    lines = [l for l in outs[0].splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 64
    assert d["weak"]["scaling"] == "weak" and d["weak"]["global_batch"] == 128 and d["weak"]["rows_per_gpu"] == 64
    assert d["value_weak"] == d["weak"]["value"] > 0 and d["value"] > 0
    # the start-up self-check of an N > 1 run (its own processes): default schedule == plain schedule
    sc = d["dp_selfcheck"]
    assert sc["ok"] is True and sc["default"]["schedule"] == "pipelined" and sc["plain"]["schedule"] == "plain"
    assert sc["rel_diff"] <= 1e-5 and sc["default"]["collectives"] > sc["plain"]["collectives"]


def _bench_ranks(world, extra_env, timeout=300):
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(world):
        env = _env(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3",
                                       "--warmup", "1", "--batch", "64", "--no-cpu-baseline", "--no-roofline", "--no-weak-leg"],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    lines = [l for l in outs[0][1].splitlines() if l.strip()]
    assert len(lines) == 1, outs[0]
    assert all(o.strip() == "" for _, o, _ in outs[1:])
    return json.loads(lines[0]), outs


def test_bench_self_check_hang_falls_back_to_the_plain_schedule():
    """bench.py --gpus 4 with a rank whose self-check stops issuing collectives (injected): every rank's check child is
    killed at the time limit, EVERY rank falls back to the plain schedule with a warning, and the run still measures."""
    d, outs = _bench_ranks(4, {"TN_TEST_SELFCHECK_HANG": "2", "TN_BENCH_SELFCHECK_TIMEOUT": "12"})
    assert all(rc == 0 for rc, _, _ in outs), [e[-300:] for _, _, e in outs]
    sc = d["dp_selfcheck"]
    assert sc["ok"] is False and "did not finish" in sc["error"] and "TN_PIPELINE=0" in sc["fallback"]
    assert d["config"]["dp_schedule"] == "plain" and d["value"] > 0 and d["n_gpus"] == 4
    assert all("self-check FAILED" in e for _, _, e in outs)


def test_bench_watchdog_prints_a_line_when_a_rank_hangs_in_the_timed_region():
    """A rank stuck in the primary timed loop (injected: rank 1 stops enqueueing, the others wait in their collectives /
    barrier) is not an exception: the watchdog prints ONE line with "error": "rank stuck in: the timed region ..." and the
    self-check record, and every rank leaves (non-zero: nothing was measured)."""
    d, outs = _bench_ranks(2, {"TN_TEST_BENCH_HANG": "1", "TN_BENCH_TIMEOUT": "15"})
    assert d["value"] is None and "rank stuck in: the timed region" in d["error"] and d["dp_selfcheck"]["ok"] is True
    assert all(rc == 4 for rc, _, _ in outs)


def test_rsag_self_test_falls_back_loudly_on_every_rank(tmp_path):
    """DeviceGroup.self_test_rsag: before a gradient bucket may take the reduce-scatter + all-gather form the fresh
    communicator compares it with one all-reduce bit for bit on every rank (lengths the world size does and does not
    divide).  A mismatch on ONE rank (injected here on rank 1) switches the form off on EVERY rank with a warning --
    comm.collective_algo then answers 'allreduce' for a bucket of any size -- and training goes on; without the
    injected fault an 8 MB bucket takes 'rsag'."""
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from theanet_amd import comm\n"
            "from theanet_amd.device import get_context\n"
            "g = comm.DeviceGroup(get_context(), comm.get_world())\n"
            "print('ALGO', comm.collective_algo(4 << 20, g.world.size), g.rsag_checked)\n" % ROOT)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    for n, (brk, want) in enumerate((("1", "ALGO allreduce False"), ("", "ALGO rsag True"))):
        procs = []
        for rank in range(2):
            env = _env(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port + n), OMP_NUM_THREADS="1", TN_TEST_BREAK_RSAG=brk)
            procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        for rank, p_ in enumerate(procs):
            o, e = p_.communicate(timeout=300)
            assert p_.returncode == 0, e[-2000:]
            assert want in o, (rank, o, e[-500:])
            assert ("WARNING" in e) == bool(brk), e[-500:]
