#! /usr/bin/python
# -*- coding: utf-8 -*-
"""Training harness: the counterpart of the reference's train.py for the MI355X backend.

    python3 train.py <dataset> <params_file(.prms|.pkl)> [1]

Same command line, same stdout table (Epoch Cost Tr_Error Tr_P(MLE) Te_Error Te_P(MLE)),
same pickle checkpoints ({"layers","training_params","allwts"}; float32 ndarrays) so
files move freely between the reference and this build.  The dataset, weights and all
activations stay in HBM; per step only the minibatch index crosses to the device.
"""
import ast
import importlib
import os
import pickle
import socket
import sys
import time
from datetime import datetime

import numpy as np

import theanet_amd.neuralnet as nn
from theanet_amd.device import get_context, share


def fixdim(arr):
    """2-D (N, side*side) / 3-D (N,H,W) / 4-D image arrays -> NCHW."""
    if arr.ndim == 2:
        side = int(arr.shape[-1] ** .5)
        assert side ** 2 == arr.shape[-1], "Need a perfect square"
        return arr.reshape((arr.shape[0], 1, side, side))
    if arr.ndim == 3:
        return np.expand_dims(arr, axis=1)
    if arr.ndim == 4:
        return arr
    raise ValueError("Image data arrays must have 2,3 or 4 dimensions only")


class WrapOut:
    """stdout, optionally teed into <params>_<SEED>.txt (line buffered)."""

    def __init__(self, use_file, name=''):
        self.name, self.use_file = name, use_file
        self.stream = open(name, 'w', 1) if use_file else sys.__stdout__

    def write(self, data):
        self.stream.write(data)

    def forceflush(self):
        if self.use_file:
            self.stream.close()
            self.stream = open(self.name, 'a', 1)
        else:
            self.stream.flush()

    def __getattr__(self, attr):
        return getattr(self.stream, attr)


def main(argv):
    if len(argv) < 3:
        print('Usage:', argv[0], ''' <dataset> <params_file(s)> [redirect=0]
    dataset:
        Name of a module in the data folder: "synthetic", "mnist", ...
    params_file(s) :
        Parameters for the NeuralNet
        - name.prms : contains the initialization code
        - name.pkl  : pickled file from a previous run (has wts too).
    redirect:
        1 - redirect stdout to a params_<SEED>.txt file
    ''')
        return 1

    dataset_name, prms_file_name = argv[1], argv[2]

    # ---------------------------------------------------------------- parameters
    if prms_file_name.endswith('.pkl'):
        with open(prms_file_name, 'rb') as f:
            params = pickle.load(f)
    else:
        with open(prms_file_name, 'r') as f:
            params = ast.literal_eval(f.read())

    layers = params['layers']
    tr_prms = params['training_params']
    allwts = params.get('allwts')

    if tr_prms.get('SEED') is None:
        tr_prms['SEED'] = int(np.random.randint(0, 1e6))

    out_file_head = os.path.basename(prms_file_name).replace(
        os.path.splitext(prms_file_name)[1], "_{:06d}".format(tr_prms['SEED']))

    if argv[-1] == '1':
        print("Printing output to {}.txt".format(out_file_head), file=sys.stderr)
        sys.stdout = WrapOut(True, out_file_head + '.txt')
    else:
        sys.stdout = WrapOut(False)

    # ---------------------------------------------------------------- banner
    ctx = get_context()
    dev_name, cus, mem = ctx.info()
    print(' '.join(argv), file=sys.stderr)
    print(' '.join(argv))
    print('Time   :' + datetime.now().strftime('%Y-%m-%d %H:%M:%S'))
    print('Device : {} ({} CUs, {:.0f} GB) (float32)'.format(dev_name, cus, mem / 2 ** 30))
    print('Host   :', socket.gethostname())
    print(nn.get_layers_info(layers))
    print(nn.get_training_params_info(tr_prms))

    # ---------------------------------------------------------------- data -> HBM
    data = importlib.import_module("data." + dataset_name)
    data.training_x = fixdim(np.asarray(data.training_x))
    data.testing_x = fixdim(np.asarray(data.testing_x))
    tr_corpus_sz, n_maps, _, layers[0][1]['img_sz'] = data.training_x.shape
    te_corpus_sz = data.testing_x.shape[0]
    if n_maps != 1:
        layers[0][1].setdefault('num_maps', n_maps)

    trin_x = share(data.training_x)
    test_x = share(data.testing_x)
    trin_y = share(data.training_y, 'int32')
    test_y = share(data.testing_y, 'int32')

    print("\nInitializing the net ... ")
    net = nn.NeuralNet(layers, tr_prms, allwts)
    print(net)
    print(net.get_wts_info(detailed=True).replace("\n\t", ""))

    print("\nCompiling ... ")
    training_fn = net.get_trin_model(trin_x, trin_y)
    test_fn_tr = net.get_test_model(trin_x, trin_y)
    test_fn_te = net.get_test_model(test_x, test_y)

    batch_sz = tr_prms['BATCH_SZ']
    n_epochs = tr_prms['NUM_EPOCHS']
    n_tr_batches = tr_corpus_sz // batch_sz
    n_te_batches = te_corpus_sz // batch_sz

    def test_wrapper(nylist):
        sym_err, bit_err, n = 0., 0., 0
        for symdiff, bitdiff in nylist:
            sym_err += symdiff
            bit_err += bitdiff
            n += 1
        return 100 * sym_err / n, 100 * bit_err / n

    aux_err_name = 'BitErr' if net.tr_layers[-1].kind == 'LOGIT' else 'P(MLE)'

    def get_test_indices(tot_samps, bth_samps=tr_prms['TEST_SAMP_SZ']):
        n_bths_each = max(1, int(bth_samps / batch_sz))
        n_bths_all = int(tot_samps / batch_sz)
        cur = 0
        while True:
            yield [i % n_bths_all for i in range(cur, cur + n_bths_each)]
            cur = (cur + n_bths_each) % n_bths_all

    test_indices = get_test_indices(te_corpus_sz)
    trin_indices = get_test_indices(tr_corpus_sz)
    pickle_file_name = out_file_head + '_{:02.0f}.pkl'
    saved = {"name": None}

    def do_test():
        test_err, aux_test_err = test_wrapper(test_fn_te(i) for i in next(test_indices))
        trin_err, aux_trin_err = test_wrapper(test_fn_tr(i) for i in next(trin_indices))
        print("{:5.2f}%  ({:5.2f}%)      {:5.2f}%  ({:5.2f}%)".format(
            trin_err, aux_trin_err, test_err, aux_test_err))
        sys.stdout.forceflush()
        if os.environ.get("THEANET_NO_PICKLE"):
            return
        if saved["name"]:
            os.remove(saved["name"])
        saved["name"] = pickle_file_name.format(test_err)
        with open(saved["name"], 'wb') as pkl_file:
            pickle.dump(net.get_init_params(), pkl_file, -1)

    # ---------------------------------------------------------------- training loop
    np.set_printoptions(precision=2)
    print("Training ...")
    print("Epoch   Cost  Tr_Error Tr_{0}    Te_Error Te_{0}".format(aux_err_name))
    for epoch in range(n_epochs):
        total_cost = 0
        t0 = time.perf_counter()
        for ibatch in range(n_tr_batches):
            cost, features, logprobs = training_fn(ibatch)
            total_cost += cost
            if np.isnan(total_cost):
                print("Epoch:{} Iteration:{}".format(epoch, ibatch))
                print(net.get_wts_info(detailed=True))
                raise ZeroDivisionError("Nan cost at Epoch:{} Iteration:{}"
                                        "".format(epoch, ibatch))
        dt = time.perf_counter() - t0

        if epoch % tr_prms['EPOCHS_TO_TEST'] == 0:
            print("{:3d} {:>8.2f}".format(net.get_epoch(), total_cost), end='    ')
            do_test()
            print("        [{:,.0f} images/sec]".format(n_tr_batches * batch_sz / dt),
                  file=sys.stderr)
            if total_cost > 1e6:
                print(net.get_wts_info(detailed=True))

        net.inc_epoch_set_rate()

    # ---------------------------------------------------------------- final error rates
    test_err, aux_test_err = test_wrapper(test_fn_te(i) for i in range(n_te_batches))
    trin_err, aux_trin_err = test_wrapper(test_fn_tr(i) for i in range(n_tr_batches))
    print("{:3d} {:>8.2f}".format(net.get_epoch(), 0), end='    ')
    print("{:5.2f}%  ({:5.2f}%)      {:5.2f}%  ({:5.2f}%)".format(
        trin_err, aux_trin_err, test_err, aux_test_err))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
