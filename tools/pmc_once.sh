#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass (counters in $2) of bench.py with the arguments after --, condensed to
# gpurun_out/<tag>/counters.csv.   tools/pmc_once.sh <tag> "<counters>" -- <bench args>
TAG=$1; CNT=$2; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/$TAG -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --no-roofline "$@" > $OUT/$TAG.log 2>&1 || echo "FAILED (see $TAG.log)"
python $GRAFT_REPO_ROOT/tools/condense_profiles.py $OUT
cat $OUT/$TAG/counters.csv
