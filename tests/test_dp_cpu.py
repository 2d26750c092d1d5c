"""Data-parallel host logic on CPU: shard math, flat layout, and a real world_size-2 run
(two processes, socket rendezvous + gloo all-reduce) whose reduced gradient must equal the
single-process full-batch gradient."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from theanet_amd import comm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_and_row0():
    assert comm.shard_rows(4096, 1, 0) == (0, 4096)
    assert [comm.shard_rows(4096, 8, r) for r in (0, 7)] == [(0, 512), (3584, 4096)]
    with pytest.raises(ValueError):
        comm.shard_rows(20, 8, 0)
    assert comm.minibatch_row0(3, 4096, 8, 2) == 3 * 4096 + 1024
    # shards tile the minibatch exactly
    rows = sorted(r for k in range(4) for r in range(*comm.shard_rows(64, 4, k)))
    assert rows == list(range(64))


def test_flat_layout_alignment():
    offs, cost_off, n = comm.flat_layout([36, 4, 720, 20, 360000, 500, 5000, 10])
    assert offs[0] == 0 and all(o % 64 == 0 for o in offs)
    assert offs[1] == 64 and offs[2] == 128 and offs[3] == 128 + 768
    assert n == cost_off + 1 and cost_off % 64 == 0
    sizes = [36, 4, 720, 20, 360000, 500, 5000, 10]
    for (o, s), o2 in zip(zip(offs, sizes), offs[1:] + [cost_off]):
        assert o + s <= o2


def test_world_from_env():
    w = comm.World.from_env({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3",
                             "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1234"})
    assert (w.rank, w.size, w.local_rank, w.master_port) == (3, 8, 3, 1234)
    assert comm.World.from_env({}).size == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gradient_allreduce_equals_full_batch(tmp_path):
    port = _free_port()
    out = str(tmp_path / "dp.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, o.decode()[-2000:]
    r = np.load(out)
    assert r["err"] < 1e-12
    np.testing.assert_allclose(r["cost"], r["cost_full"], rtol=1e-12)


def test_delayed_update_recurrence_is_the_reference_recurrence():
    """The data-parallel 'delayed' schedule (NeuralNet._train_step, tn_sgd_update_net in TN_UPD_DELAYED mode): the
    reference's update applies the OLD velocity (layer.py:82-86), so p_{t+1} only needs the gradient of
    step t-1.  Updating at the end of step t with the reduced gradient of step t-1 (mode 2 on the first
    step, mode 1 afterwards, mode 3 when leaving) must reproduce the reference weights exactly."""
    rng = np.random.RandomState(0)
    A, p0 = rng.randn(6, 6), rng.randn(6)
    grad = lambda p: A @ p + np.sin(p)          # any function of the CURRENT weights
    m, s = 0.95, 0.1

    p, v = p0.copy(), np.zeros(6)
    ref_p, ref_v = [], []
    for t in range(8):                           # layer.py:82-86, simultaneous updates
        g = grad(p)
        p, v = p - s * v, m * v + (1 - m) * g
        ref_p.append(p.copy())
        ref_v.append(v.copy())

    p, v, pending = p0.copy(), np.zeros(6), None
    got = []
    for t in range(8):
        g = grad(p)                              # its all-reduce starts here, lands during step t+1
        if pending is None:
            p = p - s * v                        # mode 2: v is v_t already
        else:
            v = m * v + (1 - m) * pending        # mode 1: v_t from the gradient of step t-1
            p = p - s * v
        pending = g
        got.append(p.copy())
    np.testing.assert_array_equal(np.array(got), np.array(ref_p))
    v = m * v + (1 - m) * pending                # mode 3: leaving the schedule catches v up
    np.testing.assert_array_equal(v, ref_v[-1])


def test_communicator_self_test_raises_instead_of_hanging():
    """DeviceGroup.self_test (run at communicator creation with more than one rank): a rank whose collectives never
    finish makes the watchdog RAISE after its timeout, and a wrong sum is reported by element -- exercised with a
    stand-in context (no GPU, no communicator): only the host logic is under test here; the real thing runs in
    tests/test_cpu_backend.py (world_size 2 and 4) and tests/test_gpu_net.py."""
    import ctypes
    from theanet_amd import comm

    class Arr:
        def __init__(self, a):
            self.a = np.array(a, np.float32)
            self.size, self.ptr = self.a.size, 0

        def get_value(self):
            return self.a

    class Ctx:
        backend = "hip"

        def __init__(self, finishes, scale):
            self.finishes, self.scale, self.calls = finishes, scale, []
            self.h = None
            self.lib = type("L", (), {"tn_event_destroy": staticmethod(lambda h, e: 0)})()

        def array(self, a):
            self.last = Arr(a)
            self.bufs = getattr(self, "bufs", []) + [self.last]
            return self.last

        def call(self, name, *args):
            self.calls.append(name)
            if name == "tn_event_query":
                ctypes.cast(args[1], ctypes.POINTER(ctypes.c_int))[0] = 1 if self.finishes else 0

    def group(ctx, size=4, rank=1):
        g = comm.DeviceGroup.__new__(comm.DeviceGroup)
        g.ctx, g.world = ctx, comm.World(rank, size)
        g.n_issued, g.order_hash, g.check_every_call = 0, 0, False
        g.verify_order = lambda: None
        # stand-in collective: what a correct sum over ``size`` ranks leaves behind (first reduction of a buffer:
        # the rank stamps add up to size*(size+1)/2; a second one sums ``size`` equal copies), times ``scale``
        seen = {}

        def allreduce_sum(b, n):
            first = seen.setdefault(id(b), 0) == 0
            seen[id(b)] += 1
            b.a *= ctx.scale * ((size * (size + 1) / 2.0) / (rank + 1) if first else size)
        g.allreduce_sum = allreduce_sum
        return g

    hung = Ctx(finishes=False, scale=1.0)
    with pytest.raises(RuntimeError, match="self-test timed out"):
        group(hung).self_test(timeout=0.05)
    assert hung.calls.count("tn_event_query") > 1 and hung.calls[-1] == "tn_stream_select"
    bad = Ctx(finishes=True, scale=1.5)
    with pytest.raises(RuntimeError, match="self-test failed on rank 1 of 4: element 0"):
        group(bad).self_test(timeout=1.0)
