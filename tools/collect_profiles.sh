#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats and the HBM-traffic PMC passes for the
# headline workload, written under gpurun_out/ (copy the summaries into profiles/ afterwards).
#   FETCH_SIZE / WRITE_SIZE are collected in their own passes (TCC has 4 slots: 3 + 2 do not
#   fit together), exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel quality: one step at a time (the schedule bench.py's roofline leg measures in)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o mnist -- \
    python $GRAFT_REPO_ROOT/bench.py --sequential --steps 50 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1
# the default schedule: two steps in flight (kernel durations overlap)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipe -o mnist_pipe -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > $OUT/stats_pipe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --sequential --steps 5 --warmup 2 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --sequential --steps 5 --warmup 2 --no-cpu-baseline > $OUT/write.log 2>&1
# the two wider configurations (BASELINE.json configs 4 and 5): kernel-trace stats only
for c in cifar_like wide6; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -o $c -- \
        python $GRAFT_REPO_ROOT/bench.py --prms $c.prms --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats_$c.log 2>&1
done
tail -1 $OUT/stats.log | cut -c1-200
