#! /usr/bin/python
# -*- coding: utf-8 -*-
"""Training harness: the counterpart of the reference's train.py for the MI355X backend.

    python3 train.py <dataset> <params_file(.prms|.pkl)> [1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...

Same command line, same stdout table (Epoch Cost Tr_Error Tr_P(MLE) Te_Error Te_P(MLE)) and the same
pickle checkpoints ({"layers","training_params","allwts"}; float32 ndarrays) as the reference's
train.py:58-245, so parameter files, logs and checkpoints move freely between the two.  The dataset,
weights and all activations stay in HBM; per step only the minibatch index crosses to the device.

Data-parallel launches (one process per GPU): rank 0 chooses the SEED and broadcasts it, so every
replica builds the same net; rank 0 alone prints the table and writes / rotates checkpoints; every rank
runs the training and test functions (they contain the collectives).
"""
import ast
import contextlib
import importlib
import os
import pickle
import socket
import sys
import time
from datetime import datetime

import numpy as np

import theanet_amd.neuralnet as nn
from theanet_amd import comm
from theanet_amd.device import get_context, share

USAGE = '''Usage: {} <dataset> <params_file(s)> [redirect=0]
    dataset:
        Name of a module in the data folder: "synthetic", "mnist", ...
    params_file(s) :
        Parameters for the NeuralNet
        - name.prms : contains the initialization code
        - name.pkl  : pickled file from a previous run (has wts too).
    redirect:
        1 - redirect stdout to a params_<SEED>.txt file
'''


def as_nchw(images):
    """Dataset image arrays come flat (N, side*side), as (N, H, W) or already (N, C, H, W)."""
    a = np.asarray(images)
    if a.ndim == 4:
        return a
    if a.ndim == 3:
        return a[:, None, :, :]
    if a.ndim != 2:
        raise ValueError("Image data arrays must have 2,3 or 4 dimensions only")
    side = int(round(a.shape[1] ** .5))
    assert side * side == a.shape[1], "Need a perfect square"
    return a.reshape(a.shape[0], 1, side, side)


def load_params(path):
    if path.endswith('.pkl'):
        with open(path, 'rb') as fh:
            return pickle.load(fh)
    with open(path, 'r') as fh:
        return ast.literal_eval(fh.read())


class Log:
    """Where the report goes: the console, a line-buffered <params>_<SEED>.txt (redirect = 1), or
    nowhere (data-parallel ranks other than 0)."""

    def __init__(self, path=None, mute=False):
        self.path, self.mute = path, mute
        self.fh = open(os.devnull, 'w') if mute else (open(path, 'w', 1) if path else None)

    @contextlib.contextmanager
    def capture(self):
        if self.fh is None:
            yield
        else:
            with contextlib.redirect_stdout(self.fh):
                yield

    def checkpoint(self):
        """Make what was printed so far durable (the reference closes and re-opens its file)."""
        if self.path and not self.mute:
            self.fh.flush()
            os.fsync(self.fh.fileno())
        else:
            sys.stdout.flush()


class BatchWindows:
    """Rolling windows of minibatch indices for the periodic tests: each call yields the next
    TEST_SAMP_SZ / BATCH_SZ batches of a corpus, wrapping around (train.py:170-176)."""

    def __init__(self, corpus_sz, batch_sz, samp_sz):
        self.n_all = corpus_sz // batch_sz
        self.n_each = max(1, samp_sz // batch_sz)
        self.pos = 0

    def next(self):
        idx = [(self.pos + k) % self.n_all for k in range(self.n_each)]
        self.pos = (self.pos + self.n_each) % self.n_all
        return idx


def error_rates(test_fn, batches):
    """Mean symbol error and mean second statistic (P(MLE) / bit error) over ``batches``, in percent."""
    stats = np.array([test_fn(i)[:2] for i in batches], dtype=np.float64)
    return 100 * stats[:, 0].mean(), 100 * stats[:, 1].mean()


class Checkpoints:
    """<head>_<test error>.pkl after every test, the previous one removed (train.py:195-200)."""

    def __init__(self, head, enabled):
        self.head, self.enabled, self.last = head, enabled, None

    def save(self, net, test_err):
        if not self.enabled:
            return
        name = '{}_{:02.0f}.pkl'.format(self.head, test_err)
        tmp = name + '.tmp'
        with open(tmp, 'wb') as fh:
            pickle.dump(net.get_init_params(), fh, -1)
        os.replace(tmp, name)
        if self.last and self.last != name and os.path.exists(self.last):
            os.remove(self.last)
        self.last = name


def run(dataset_name, prms_file_name, redirect):
    world = comm.get_world()
    lead = world.rank == 0
    params = load_params(prms_file_name)
    layers, tr_prms, allwts = params['layers'], params['training_params'], params.get('allwts')

    # every replica must build the same net: the seed is rank 0's
    seed = tr_prms.get('SEED')
    if seed is None:
        seed = int(np.random.randint(0, 1e6))
    if world.size > 1:
        seed = comm.broadcast_int(seed)
    tr_prms['SEED'] = seed

    stem = os.path.splitext(os.path.basename(prms_file_name))[0]
    head = '{}_{:06d}'.format(stem, seed)
    if redirect and lead:
        print("Printing output to {}.txt".format(head), file=sys.stderr)
    log = Log(head + '.txt' if redirect else None, mute=not lead)

    with log.capture():
        ctx = get_context()
        dev_name, cus, mem = ctx.info()
        print(' '.join(sys.argv))
        print('Time   :' + datetime.now().strftime('%Y-%m-%d %H:%M:%S'))
        print('Device : {} ({} CUs, {:.0f} GB) ({}){}'.format(
            dev_name, cus, mem / 2 ** 30, tr_prms.get('DTYPE', 'float32'),
            ' x {} data-parallel ranks'.format(world.size) if world.size > 1 else ''))
        print('Host   :', socket.gethostname())
        print(nn.get_layers_info(layers))
        print(nn.get_training_params_info(tr_prms))

        # ------------------------------------------------------------ data -> HBM (once)
        data = importlib.import_module("data." + dataset_name)
        tr_x, te_x = as_nchw(data.training_x), as_nchw(data.testing_x)
        n_maps, img_sz = tr_x.shape[1], tr_x.shape[2]
        layers[0][1]['img_sz'] = img_sz
        if n_maps != 1:
            layers[0][1].setdefault('num_maps', n_maps)
        trin_x, trin_y = share(tr_x), share(data.training_y, 'int32')
        test_x, test_y = share(te_x), share(data.testing_y, 'int32')
        # datasets for AuxConcatLayer / SoftAuxLayer nets carry per-sample side inputs (reference train.py:133-137)
        trin_aux = share(data.training_aux) if hasattr(data, 'training_aux') else None
        test_aux = share(data.testing_aux) if hasattr(data, 'testing_aux') else None

        print("\nInitializing the net ... ")
        net = nn.NeuralNet(layers, tr_prms, allwts)
        if params.get('opt_state') is not None:        # (this build's optional extension of the pickle: momentum + RNG counter)
            net.load_opt_state(params['opt_state'])
        print(net)
        print(net.get_wts_info(detailed=True).replace("\n\t", ""))

        print("\nCompiling ... ")
        training_fn = net.get_trin_model(trin_x, trin_y, trin_aux)
        test_fn_tr = net.get_test_model(trin_x, trin_y, trin_aux)
        test_fn_te = net.get_test_model(test_x, test_y, test_aux)

        batch_sz = tr_prms['BATCH_SZ']
        n_tr_batches, n_te_batches = len(tr_x) // batch_sz, len(te_x) // batch_sz
        te_windows = BatchWindows(len(te_x), batch_sz, tr_prms['TEST_SAMP_SZ'])
        tr_windows = BatchWindows(len(tr_x), batch_sz, tr_prms['TEST_SAMP_SZ'])
        ckpt = Checkpoints(head, lead and not os.environ.get("THEANET_NO_PICKLE"))
        aux = 'BitErr' if net.tr_layers[-1].kind == 'LOGIT' else 'P(MLE)'
        row = "{:5.2f}%  ({:5.2f}%)      {:5.2f}%  ({:5.2f}%)"
        exp_head = layers[-1][0][:3] == "Exp"

        np.set_printoptions(precision=2)
        print("Training ...")
        print("Epoch   Cost  Tr_Error Tr_{0}    Te_Error Te_{0}".format(aux))
        def nan_guard(epoch, ibatch):
            print("Epoch:{} Iteration:{}".format(epoch, ibatch))
            print(net.get_wts_info(detailed=True))
            raise ZeroDivisionError("Nan cost at Epoch:{} Iteration:{}".format(epoch, ibatch))

        for epoch in range(tr_prms['NUM_EPOCHS']):
            total_cost, t0 = 0, time.perf_counter()
            if not exp_head:
                # The reference's loop (train.py:211-226) needs ONE number of a step: its cost, for the epoch's total and
                # the NaN guard.  step_cost(i) enqueues the step and hands back the costs that have ARRIVED (a step's cost
                # travels to page-locked memory by itself and is picked up a few calls later), so the host keeps running
                # ahead of the GPU like bench.py's enqueue-only loop; the guard reports the iteration the NaN belongs to.
                for ibatch in range(n_tr_batches):
                    for k, cost in training_fn.step_cost(ibatch):
                        total_cost += cost
                        if np.isnan(total_cost):
                            nan_guard(epoch, k)
                for k, cost in training_fn.drain_costs():
                    total_cost += cost
                    if np.isnan(total_cost):
                        nan_guard(epoch, k)
            for ibatch in range(n_tr_batches if exp_head else 0):
                cost, features, _ = training_fn(ibatch)
                total_cost += cost
                # ExpLoss nets: report samples whose true-class feature runs away (train.py:216-222)
                labels = np.asarray(data.training_y[ibatch * batch_sz:(ibatch + 1) * batch_sz])
                lo = net.shard_lo                         # a data-parallel rank holds its own rows of the batch
                own = labels[lo:lo + len(features)]
                true_features = features[np.arange(len(own)), own]
                if np.min(true_features) < -6:
                    print("Epoch:{} Iteration:{}".format(epoch, ibatch))
                    print(own)
                    print(true_features)
                    print(net.get_wts_info(detailed=True))
                if np.isnan(total_cost):
                    nan_guard(epoch, ibatch)
            rate = n_tr_batches * batch_sz / (time.perf_counter() - t0)

            if epoch % tr_prms['EPOCHS_TO_TEST'] == 0:
                te_err, te_aux = error_rates(test_fn_te, te_windows.next())
                tr_err, tr_aux = error_rates(test_fn_tr, tr_windows.next())
                print("{:3d} {:>8.2f}".format(net.get_epoch(), total_cost), end='    ')
                print(row.format(tr_err, tr_aux, te_err, te_aux))
                log.checkpoint()
                ckpt.save(net, te_err)
                if lead:
                    print("        [{:,.0f} images/sec]".format(rate), file=sys.stderr)
                if total_cost > 1e6:
                    print(net.get_wts_info(detailed=True))

            net.inc_epoch_set_rate()

        # ------------------------------------------------------------ final error rates, whole corpora
        te_err, te_aux = error_rates(test_fn_te, range(n_te_batches))
        tr_err, tr_aux = error_rates(test_fn_tr, range(n_tr_batches))
        print("{:3d} {:>8.2f}".format(net.get_epoch(), 0), end='    ')
        print(row.format(tr_err, tr_aux, te_err, te_aux))
        log.checkpoint()
    return 0


def main(argv):
    if len(argv) < 3:
        print(USAGE.format(argv[0]))
        return 1
    return run(argv[1], argv[2], redirect=argv[-1] == '1' and len(argv) > 3)


if __name__ == '__main__':
    sys.exit(main(sys.argv))
