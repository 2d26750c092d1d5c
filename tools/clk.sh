#!/bin/bash
# sample clocks/power while bench runs
run() {
  python bench.py --no-roofline --no-cpu-baseline --steps 30000 --warmup 20 $1 > /tmp/b.log 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.4; done
  wait $pid
  tail -1 /tmp/b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
}
echo "== pipelined"; run ""
echo "== sequential"; run "--sequential"
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo
