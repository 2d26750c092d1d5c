#!/bin/bash
# Runs ON THE GPU BOX: kernel durations (rocprofv3 --kernel-trace --stats) of one fp16-resident conv op (tools/one_c8.py; OP, WB, WC, WK, WH, IT).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_c8
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c8 -o s -- python $GRAFT_REPO_ROOT/tools/one_c8.py > /tmp/kt_c8.log 2>&1
f=$(find /tmp/kt_c8 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    n = re.sub(r"^void ", "", r["Name"]).split("(")[0][:60]
    print("%-60s calls %6s  avg %8.1f us  %6s %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
