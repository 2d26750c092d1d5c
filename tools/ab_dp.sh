#!/bin/bash
# same-box A/B of a config with and without the forced 1-rank data-parallel step
for r in 1 2; do for f in 0 1; do
  echo -n "TN_DP_FORCE=$f: "; TN_DP_FORCE=$f python bench.py --prms ${1:-wide6.prms} --steps ${2:-20} --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
