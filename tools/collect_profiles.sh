#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats, the HBM-traffic PMC passes and the matrix-core
# utilisation counters for the BASELINE workloads, written under gpurun_out/<tag>/ (tools/make_traffic_json.py
# then copies the summaries into profiles/).
#   FETCH_SIZE / WRITE_SIZE are collected in their own passes (TCC has 4 slots: 3 + 2 do not fit
#   together) and PMC passes never carry a trace domain other than --kernel-trace, exactly as
#   /opt/skills/guides/MI355X_MICROARCH.md prescribes.
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs"
run() {   # name, rocprof args..., -- bench args
    local name=$1; shift
    timeout 600 rocprofv3 "$@" > $OUT/$name.log 2>&1 || echo "FAILED $name (see $name.log)"
}
stats() { run $1 --kernel-trace --stats --output-format csv -d $OUT/$1 -o s -- $B "${@:2}"; }
pmc()   { run $1 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -o p -- $B --no-roofline "${@:3}"; }
MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"
# ---- mnist.prms, B = 4096 (the headline): one step at a time (the roofline leg's schedule), default schedule
stats mnist_seq --sequential --steps 50 --warmup 5
stats mnist_pipe --steps 200 --warmup 20 --no-roofline
pmc mnist_fetch FETCH_SIZE --sequential --steps 5 --warmup 2
pmc mnist_write WRITE_SIZE --sequential --steps 5 --warmup 2
pmc mnist_mfma "$MFMA" --sequential --steps 5 --warmup 2
# ---- what every rank of the 8-GPU strong-scaling run does: 512 images per step
stats mnist512_seq --batch 512 --sequential --steps 200 --warmup 20 --no-roofline
stats mnist512_pipe --batch 512 --steps 200 --warmup 20 --no-roofline
# ---- the wider configurations (BASELINE.json configs 4 and 5), fp32 and fp16 operands
for c in cifar_like wide6; do
    for d in f32 f16; do
        # one step at a time, like the roofline leg of bench.py: with two steps in flight the launches of the two
        # streams share the GPU and a kernel's duration in the trace is not its own
        stats ${c}_${d} --prms $c.prms --dtype $d --sequential --steps 10 --warmup 3
        stats ${c}_${d}_pipe --prms $c.prms --dtype $d --steps 10 --warmup 3 --no-roofline
        pmc ${c}_${d}_fetch FETCH_SIZE --prms $c.prms --dtype $d --sequential --steps 3 --warmup 1
        pmc ${c}_${d}_write WRITE_SIZE --prms $c.prms --dtype $d --sequential --steps 3 --warmup 1
        pmc ${c}_${d}_mfma "$MFMA" --prms $c.prms --dtype $d --sequential --steps 3 --warmup 1
    done
done
python $GRAFT_REPO_ROOT/tools/condense_profiles.py $OUT
du -sh $OUT
grep -l FAILED $OUT/*.log 2>/dev/null | head
