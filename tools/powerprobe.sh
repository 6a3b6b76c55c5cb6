#!/bin/bash
# Samples rocm-smi power/clock while the default bench workload runs for a few seconds.  Diagnostic only.
LIB=$1; shift
(VHP_LIB=$LIB python3 bench.py --steps 6000 --warmup 3 --no-cpu-baseline "$@" > /tmp/pp_bench.json 2>/dev/null) &
BP=$!
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.5
done
wait $BP
python3 -c "import json; d=json.load(open('/tmp/pp_bench.json')); print('bench', d['value'], d['roofline']['kernel_ms'])"
