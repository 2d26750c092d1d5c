#!/bin/bash
# Runs ON THE GPU BOX: train.py end to end (two epochs on the synthetic dataset) for every parameter file, fp32 and fp16-resident
cd /tmp && rm -rf e2e && mkdir e2e && cd e2e
for p in mnist cifar_like wide6; do
  python - <<PY
import ast, re
s = open("$GRAFT_REPO_ROOT/params/$p.prms").read()
d = ast.literal_eval(re.sub(r"^\s*#.*$", "", s, flags=re.M))
d["training_params"].update(NUM_EPOCHS=2, EPOCHS_TO_TEST=1, TEST_SAMP_SZ=d["training_params"]["BATCH_SZ"] * 2)
open("$p.prms", "w").write(repr(d))
if "$p" != "mnist":
    d["training_params"]["DTYPE"] = "float16"
    open("${p}_f16.prms", "w").write(repr(d))
PY
done
run() { echo "== $1 ($2 channels, $3 x $3)"; THEANET_SYNTH_CHANNELS=$2 THEANET_SYNTH_SIZE=$3 timeout 300 python $GRAFT_REPO_ROOT/train.py synthetic $1 2>&1 | tail -3; }
run mnist.prms 1 28
run cifar_like.prms 3 32
run cifar_like_f16.prms 3 32
run wide6.prms 3 64
run wide6_f16.prms 3 64
echo "== wrong dataset for the net (must be an assertion, not a fault)"
THEANET_SYNTH_CHANNELS=1 THEANET_SYNTH_SIZE=28 timeout 300 python $GRAFT_REPO_ROOT/train.py synthetic cifar_like.prms 2>&1 | tail -2
