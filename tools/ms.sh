#!/bin/bash
# usage: tools/ms.sh [bench args] -> prints ms_per_step and images/sec
python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step=%.4f  img/s=%.0f' % (d['ms_per_step'], d['value']))"
