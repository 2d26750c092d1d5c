"""Regenerates the committed golden fixtures (run in the BUILD container only:
`python tests/golden/make_golden.py`).  Everything comes from the oracle except
init_ref.npz (the draw arithmetic of theanet/layer/weights.py:51-65, compiled from /root/reference) and
deformer.npz, whose expected outputs come from the reference's own
extras/deformer.py:7-18 executed from /root/reference (the module cannot be
imported under python3 -- it has py2 print statements from line 92 -- so only its
first 28 lines are exec'd; no reference source is stored in this repo)."""
import ast
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import theanet_oracle as O  # noqa: E402


def load_prms(name, img_sz, seed=555555, batch=None):
    with open(os.path.join(ROOT, "params", name)) as fh:
        prms = ast.literal_eval(fh.read())
    prms["layers"][0][1]["img_sz"] = img_sz
    prms["training_params"]["SEED"] = seed
    if batch:
        prms["training_params"]["BATCH_SZ"] = batch
    return prms


def make_kat():
    prms = load_prms("mnist.prms", 28)
    net = O.OracleNet(prms["layers"], prms["training_params"])
    out = {}
    for i, l in enumerate(net.L):
        for j, p in enumerate(l.params):
            out["init_sha_%d_%d" % (i, j)] = hashlib.sha256(
                np.ascontiguousarray(p).tobytes()).hexdigest()
    np.savez(os.path.join(HERE, "kat.npz"), **out)


def sub_index(size, n=4096):
    """Fixed pseudo-random subsample used for tensors too large to commit whole."""
    return np.random.RandomState(size % (2 ** 31)).choice(size, n, replace=False)


def put(out, name, arr, f64_small=False):
    arr = np.asarray(arr)
    if arr.size > 20000:
        out[name + "@sub"] = arr.reshape(-1)[sub_index(arr.size)]
        out[name + "@sum"] = np.asarray(arr.sum(dtype=np.float64))
        out[name + "@abs"] = np.asarray(np.abs(arr).sum(dtype=np.float64))
    else:
        out[name] = arr


def draws_to_dict(prefix, d):
    return {prefix + k: np.asarray(getattr(d, k)) for k in d.__slots__
            if getattr(d, k) is not None}


def perturbed_init(prms):
    """Seed-chain initial weights with the conv kernels de-symmetrised.  The reference's
    conv init is +-1/sqrt(fan_in) (weights.py:52-54); with nearest-neighbour zoom the
    elastic stage duplicates pixels, so different pooling-window members become
    MATHEMATICALLY equal sums whose floating-point tie depends on summation order.
    GOLD-B pins gradient routing, so it must not depend on such ties."""
    import copy
    net = O.OracleNet(copy.deepcopy(prms["layers"]), dict(prms["training_params"]))
    rng = np.random.RandomState(123)
    allwts = []
    for l in net.L:
        ws = [p.copy() for p in l.params]
        if l.kind == "Conv":
            ws[0] = (ws[0] * (1 + .1 * rng.standard_normal(ws[0].shape))).astype(np.float32)
        allwts.append(ws)
    return allwts


def make_gold_net(fname, elastic_on, steps=3, B=8):
    """GOLD-A / GOLD-B: mnist.prms, B=8, every random draw recorded so the HIP
    path can replay them; float32 net with a float64 twin."""
    out = {}
    rng = np.random.default_rng(0)
    x = rng.random((steps * B, 1, 28, 28), dtype=np.float32)
    y = np.random.default_rng(1).integers(0, 10, steps * B).astype(np.int32)
    out["x"], out["y"] = x, y
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        prms = load_prms("mnist.prms", 28, batch=B)
        if not elastic_on:
            prms["layers"][0] = ("ElasticLayer", {"img_sz": 28, "invert_image": True})
        allwts = perturbed_init(prms) if elastic_on else None
        net = O.OracleNet(prms["layers"], prms["training_params"], allwts, dtype=dt)
        if elastic_on:      # with weights given the reference seeds its streams from OS entropy
            st = net.L[0].stage
            st.__init__(rand_gen=np.random.RandomState(99), **{k: v for k, v in
                        prms["layers"][0][1].items()})
            from oracle.randomstreams import RandomStreams
            net.L[5].mask_rv = RandomStreams(98).binomial(None, n=1, p=.5)
        if tag == "f32":
            for i, l in enumerate(net.L):
                for j, p in enumerate(l.params):
                    put(out, "init_%d_%d" % (i, j), p.copy())
        for s in range(steps):
            xb, yb = x[s * B:(s + 1) * B], y[s * B:(s + 1) * B]
            draws = {}
            if elastic_on:
                draws[0] = net.L[0].stage.draw(xb.shape)
            draws[5] = net.L[5].mask_rv.draw((B, 500))
            if tag == "f32":
                if elastic_on:
                    out.update(draws_to_dict("s%d_el_" % s, draws[0]))
                out["s%d_mask5" % s] = draws[5].astype(np.uint8)
            cost, logprob, grads, cache = net.grads(xb, yb, draws)
            out["%s_s%d_cost" % (tag, s)] = np.asarray(cost)
            out["%s_s%d_logprob" % (tag, s)] = logprob
            if s == 0:
                for i, c in enumerate(cache):
                    if tag == "f32" or i >= 5:
                        out["%s_s0_act%d" % (tag, i)] = c["out"]
                if elastic_on:
                    out["%s_s0_target" % tag] = cache[0]["target"]
            for i, g in enumerate(grads):
                if g is not None:
                    for j, gg in enumerate(g):
                        put(out, "%s_s%d_grad_%d_%d" % (tag, s, i, j), gg)
            # apply the update exactly as train_step does
            for l, g in zip(net.L, grads):
                if not l.params or not l.reg["rate"]:
                    continue
                if l.vel is None:
                    l.vel = [np.zeros_like(p) for p in l.params]
                for j in range(len(l.params)):
                    l.params[j], l.vel[j] = O.sgd_update(l.params[j], l.vel[j], g[j],
                                                         net.cur_learn_rate, l.reg)
            if s == steps - 1:
                for i, l in enumerate(net.L):
                    for j, p in enumerate(l.params):
                        put(out, "%s_w_%d_%d" % (tag, i, j), p.copy())
        sym, pm, lp, preds = net.test(x[:B], y[:B])
        out["%s_test_logprob" % tag] = lp
        out["%s_test_preds" % tag] = preds
        out["%s_test_stats" % tag] = np.array([sym, pm])
    np.savez_compressed(os.path.join(HERE, fname), **out)


def make_deformer():
    """GOLD-C: expected outputs from the REFERENCE's transform()."""
    ref = "/root/reference/extras/deformer.py"
    with open(ref) as fh:
        head = "".join(fh.readlines()[:28])
    ns = {}
    exec(compile(head, ref, "exec"), ns)     # numpy + scipy only
    rng = np.random.RandomState(11)
    imgs = rng.rand(4, 28, 28)
    imgs[1] = (imgs[1] > .7) * 1.0
    out = {"imgs": imgs}
    cases = [(3.0, 2.0, 0.0), (8.0, 4.0, 0.0), (5.0, 3.0, 1.0), (1.5, 1.0, 0.25)]
    for k, (img, (scale, sigma, cval)) in enumerate(zip(imgs, cases)):
        np.random.seed(100 + k)
        noise = np.random.uniform(-1, 1, (2,) + img.shape)   # same draw transform() makes
        np.random.seed(100 + k)
        ret, trans = ns["transform"](img.copy(), scale, sigma, cval=cval, ret_trans=True)
        out["noise%d" % k], out["out%d" % k], out["trans%d" % k] = noise, ret, trans
        out["prm%d" % k] = np.array([scale, sigma, cval])
    np.savez_compressed(os.path.join(HERE, "deformer.npz"), **out)


def ref_init_draws():
    """The reference's own draw arithmetic: theanet/layer/weights.py:51-65 (the body of `if wb is None:` in init_wb,
    pure numpy) compiled from /root/reference where it lies -- the module itself cannot be imported (it imports theano
    at :2) and nothing of Theano is stubbed: the lines that wrap the arrays in theano.shared (:73-79) are not run."""
    import textwrap
    ref = "/root/reference/theanet/layer/weights.py"
    with open(ref) as fh:
        lines = fh.readlines()
    body = textwrap.dedent("".join(lines[50:65]))
    assert body.lstrip().startswith("if len(size_w) == 4:") and "b_values += .5" in body, body
    code = compile(body, ref, "exec")

    def draw(rand_gen, size_w, size_b, fan_in, fan_out, actvn):
        ns = dict(np=np, float_x="float32", rand_gen=rand_gen, size_w=size_w, size_b=size_b, fan_in=fan_in,
                  fan_out=fan_out, actvn=actvn)
        exec(code, ns)
        return ns["w_values"], ns["b_values"]
    return draw


def make_init_ref():
    """INIT-REF: initial weights of mnist.prms (SEED 555555) and of single layers with the other activation
    rules, drawn by the REFERENCE's lines through the seed chain of neuralnet.py:65-68 / inlayers.py:72 /
    dropout.py:10 (those three `randint(1e6)` / RandomState calls are one-liners restated here)."""
    draw = ref_init_draws()
    prms = load_prms("mnist.prms", 28)
    out = {}
    rg = np.random.RandomState(prms["training_params"]["SEED"])          # neuralnet.py:66
    maps, sz, n_prev = 1, 28, None
    for i, (kind, a) in enumerate(prms["layers"]):
        if kind == "ElasticLayer":
            if any(a.get(k) for k in ("translation", "magnitude", "pflip", "angle")) or a.get("zoom", 1) != 1:
                rg.randint(1e6)                                          # inlayers.py:72-73
        elif kind == "ConvLayer":
            f, k = a["filter_sz"], a["num_maps"]
            w, b = draw(rg, (k, maps, f, f), (k,), maps * f * f, None, a.get("actvn", "relu50"))   # convpool.py:43-50
            out["mnist_%d_W" % i], out["mnist_%d_b" % i] = w, b
            sz, maps = (sz if a.get("mode", "valid") == "same" else sz - f + 1) // a.get("stride", 1), k
            n_prev = maps * sz * sz
        elif kind == "PoolLayer":
            sz = -(-sz // a["pool_sz"])
            n_prev = maps * sz * sz
        elif kind in ("HiddenLayer", "SoftmaxLayer"):
            n_out = a["n_out"]
            act = a.get("actvn", "relu01") if kind == "HiddenLayer" else "Softmax"
            fan = n_prev + n_out                                         # hidden.py:21-27: n_in + n_out, twice
            w, b = draw(rg, (n_prev, n_out), (n_out,), fan, fan, act)   # SoftmaxLayer is a HiddenLayer (outlayers.py:87)
            out["mnist_%d_W" % i], out["mnist_%d_b" % i] = w, b
            if kind == "HiddenLayer" and a.get("pdrop"):
                rg.randint(1e6)                                          # dropout.py:10
            n_prev = n_out
    for j, (act, size_w) in enumerate([("sigmoid", (7, 5)), ("softplus", (3, 2, 3, 3)), ("relu", (6, 4)),
                                       ("relu10", (2, 3, 5, 5)), ("relu05", (9, 2)), ("tanh", (4, 1, 3, 3))]):
        rg = np.random.RandomState(1000 + j)
        fan_in = int(np.prod(size_w[1:])) if len(size_w) == 4 else size_w[0]
        w, b = draw(rg, size_w, (size_w[0] if len(size_w) == 4 else size_w[1],), fan_in, size_w[-1], act)
        out["case%d_W" % j], out["case%d_b" % j] = w, b
        out["case%d_act" % j] = np.array(act)
    for k in [k for k, v in out.items() if v.size > 20000]:          # big tensors: digest + every 37th value
        v = out.pop(k)
        out[k + "_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest())
        out[k + "_every37"] = v.ravel()[::37].copy()
    np.savez_compressed(os.path.join(HERE, "init_ref.npz"), **out)


def make_train_helpers():
    """TRAIN-REF: the caller's two pure-python helpers run from the reference text -- fixdim (train.py:22-34: dataset
    arrays -> NCHW) and get_test_indices (train.py:170-176: rolling windows of minibatch indices for the periodic
    tests) -- on a few inputs; only inputs / outputs are stored."""
    import textwrap
    ref = "/root/reference/train.py"
    with open(ref) as fh:
        lines = fh.readlines()
    ns = {"np": np}
    exec(compile("".join(lines[21:35]), ref, "exec"), ns)
    assert "fixdim" in ns
    out = {}
    rng = np.random.RandomState(5)
    for k, shape in enumerate([(6, 49), (5, 7, 7), (4, 3, 6, 6), (3, 1, 5, 5), (2, 784)]):
        a = rng.rand(*shape).astype(np.float32)
        out["fixdim_in%d" % k], out["fixdim_out%d" % k] = a, ns["fixdim"](a)
    cases = [(60000, 20, 5000), (10000, 20, 5000), (70, 20, 60), (65536, 4096, 16384), (1000, 128, 500)]
    body = textwrap.dedent("".join(lines[169:177]))
    assert body.startswith("def get_test_indices(") and "yield" in body, body
    for k, (tot, bsz, samp) in enumerate(cases):
        g = {"tr_prms": {"TEST_SAMP_SZ": samp}, "batch_sz": bsz}
        exec(compile(body, ref, "exec"), g)
        it = g["get_test_indices"](tot)
        out["win_case%d" % k] = np.array([tot, bsz, samp])
        out["win_seq%d" % k] = np.array([next(it) for _ in range(9)])
    # the report strings train.py prints (neuralnet.py:16-51: get_layers_info / get_wts_info / get_training_params_info)
    nref = "/root/reference/theanet/neuralnet.py"
    with open(nref) as fh:
        nlines = fh.readlines()
    g = {}
    exec(compile("".join(nlines[15:51]), nref, "exec"), g)
    prms = load_prms("mnist.prms", 28)
    wts = [[p for p in l.params] for l in O.OracleNet(prms["layers"], prms["training_params"]).L]
    out["info_layers"] = np.array(g["get_layers_info"](prms["layers"]))
    out["info_prms"] = np.array(g["get_training_params_info"](prms["training_params"]))
    out["info_wts"] = np.array(g["get_wts_info"](wts))
    out["info_wts_detailed"] = np.array(g["get_wts_info"](wts, True))
    # the elastic stage's gaussian (inlayers.py:87-91, pure numpy), from the reference text
    iref = "/root/reference/theanet/layer/inlayers.py"
    with open(iref) as fh:
        ilines = fh.readlines()
    fbody = textwrap.dedent("".join(ilines[86:91]))
    assert fbody.startswith("var = sigma ** 2") and "filt /= 2 * np.pi * var" in fbody, fbody
    fcode = compile(fbody, iref, "exec")
    for sigma in (1, 2, 3, 4, 8):
        g2 = {"np": np, "sigma": sigma, "float_x": "float32"}
        exec(fcode, g2)
        out["elastic_filt_sigma%d" % sigma] = g2["filt"]
    np.savez_compressed(os.path.join(HERE, "train_helpers.npz"), **out)


if __name__ == "__main__":
    make_kat()
    make_init_ref()
    make_train_helpers()
    make_gold_net("gold_a.npz", elastic_on=False)
    make_gold_net("gold_b.npz", elastic_on=True)
    make_deformer()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
