"""Worker of tests/test_gpu_net.py::test_two_gpu_product_path: one rank of a REAL data-parallel job
(one process per GPU, RCCL all-reduce) running the product's own training step.  Rank 0 stores the
costs and the weights after the last step."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, prms_name, img, ch, B, steps):
    from tests.gpu_util import load_prms
    from theanet_amd import NeuralNet, comm
    prms = load_prms(prms_name, img, batch=B)
    rng = np.random.RandomState(3)
    x = rng.rand(4 * B, ch, img, img).astype(np.float32)
    y = rng.randint(0, 10, 4 * B).astype(np.int32)
    net = NeuralNet(copy.deepcopy(prms["layers"]), dict(prms["training_params"]))
    fn = net.get_trin_model(x, y)
    costs = []
    for s in range(steps):
        if s == steps // 2:
            net.inc_epoch_set_rate()
        costs.append(float(fn(s % 4)[0]))
    tfn = net.get_test_model(x, y)
    stats = tfn(1)
    wts = [w for l in net.tr_layers for w in l.get_wts()]
    world = comm.get_world()
    if world.size > 1:
        net._group().verify_order()
    if world.rank == 0:
        np.savez(out_path, costs=np.array(costs), stats=np.array(stats[:2]), schedule=str(getattr(net, "dp_schedule", "")),
                 n_collectives=net._group().n_issued if world.size > 1 else 0,
                 bucket=-1 if getattr(net, "_dp_bucket", None) is None else net._dp_bucket[1],
                 **{"w%d" % i: w for i, w in enumerate(wts)})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
