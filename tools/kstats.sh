#!/bin/bash
# Runs ON THE GPU BOX: per-kernel durations (rocprofv3 --kernel-trace --stats) of one bench.py configuration.
#   bash tools/kstats.sh <name> <bench args...>     -> prints the top kernels, keeps gpurun_out/ks_<name>.csv
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o s -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline "$@" > /tmp/ks_$name.log 2>&1
f=$(find /tmp/ks_$name -name "*kernel_stats.csv" | head -1)
cp $f $out/ks_$name.csv
grep '^{' /tmp/ks_$name.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'sustained', (d.get('sustained') or {}).get('ms_per_step'))"
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    n = re.sub(r"^void ", "", r["Name"]).split("(")[0][:60]
    print("%-60s calls %6s  avg %8.1f us  %6s %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
