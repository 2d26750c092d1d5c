"""Worker of tests/test_dp_cpu.py: one rank of a world_size-2 CPU job.  It covers the
host-side data-parallel logic of theanet_amd.comm -- shard ranges, the socket rendezvous
that carries the RCCL unique id, the flat gradient layout and the reduction contract
(gloo stands in for RCCL; the oracle stands in for the HIP kernels, which is allowed in
tests only)."""
import ast
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import theanet_oracle as O  # noqa: E402
from theanet_amd import comm  # noqa: E402


def main(out_path):
    world = comm.World.from_env()
    # 1. socket rendezvous: a 128-byte blob from rank 0 (the RCCL unique id in production)
    rdzv = comm.Rendezvous(world, timeout=60)
    blob = bytes(range(128)) if world.rank == 0 else b""
    got = rdzv.broadcast(blob, 128)
    assert got == bytes(range(128))
    assert rdzv.gather_max(float(world.rank)) == world.size - 1
    rdzv.barrier()
    # 1b. what train.py / NeuralNet use to keep replicas identical: rank 0's SEED everywhere, and a
    # loud failure on EVERY rank when replicas disagree (weights checksum, order of collectives)
    comm._rdzv = rdzv
    assert comm.broadcast_int(1234 if world.rank == 0 else 99) == 1234
    comm.agree(3.25, "a value all ranks share")
    try:
        comm.agree(float(world.rank), "the rank")
    except RuntimeError as e:
        assert "disagree" in str(e)
    else:
        raise SystemExit("comm.agree did not notice that the ranks differ")

    # 2. equal shards of a global minibatch; per-rank oracle gradients; flat buffer; all-reduce
    with open(os.path.join(ROOT, "params", "mnist.prms")) as fh:
        prms = ast.literal_eval(fh.read())
    B = 16
    prms["layers"][0] = ("InputLayer", {"img_sz": 28})
    prms["layers"][5][1]["pdrop"] = 0
    prms["training_params"].update(SEED=7, BATCH_SZ=B)
    rng = np.random.RandomState(0)
    x = rng.rand(3 * B, 1, 28, 28)
    y = rng.randint(0, 10, 3 * B)
    net = O.OracleNet(copy.deepcopy(prms["layers"]), dict(prms["training_params"]), dtype=np.float64)
    i = 2
    lo, hi = comm.shard_rows(B, world.size, world.rank)
    row0 = comm.minibatch_row0(i, B, world.size, world.rank)
    assert row0 == i * B + lo
    xs, ys = x[row0:row0 + hi - lo], y[row0:row0 + hi - lo]
    cost, _, grads, _ = net.grads(xs, ys)
    params = [g for gl in grads if gl is not None for g in gl]
    offs, cost_off, n_red = comm.flat_layout([g.size for g in params])
    flat = np.zeros(n_red, np.float64)
    for g, o in zip(params, offs):
        # the product scales d(logits) by 1/B_global, i.e. local-mean gradients by 1/R
        flat[o:o + g.size] = g.reshape(-1) / world.size
    flat[cost_off] = cost / world.size
    group = comm.HostGroup(world)
    group.allreduce_sum(flat)
    group.barrier()
    if world.rank == 0:
        cost_full, _, grads_full, _ = net.grads(x[i * B:(i + 1) * B], y[i * B:(i + 1) * B])
        full = [g for gl in grads_full if gl is not None for g in gl]
        err = max(np.abs(flat[o:o + g.size] - g.reshape(-1)).max() for g, o in zip(full, offs))
        np.savez(out_path, err=err, cost=flat[cost_off], cost_full=cost_full)
    rdzv.close()


if __name__ == "__main__":
    main(sys.argv[1])
