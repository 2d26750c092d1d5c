#!/bin/bash
# Runs ON THE GPU BOX: kernel start/end times of the default (two steps in flight) schedule -> gpurun_out/timeline.txt
# (tools/timeline.py: how much of the time 0 / 1 / 2 kernels are on the GPU, one step pair printed as a timeline).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --no-roofline --steps 200 --warmup 20 ${BENCH_ARGS} > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log | cut -c1-300
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/timeline.py "$f" | tee $GRAFT_REPO_ROOT/gpurun_out/timeline.txt
