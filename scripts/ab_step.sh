#!/bin/bash
# Alternates bench.py of scripts/ab_base (a snapshot made by snapshot_base.sh) and of the working tree, N times each.
N=${1:-3}
for i in $(seq $N); do
  for which in base new; do
    if [ $which = base ]; then dir=scripts/ab_base; else dir=.; fi
    out=$(cd $dir && timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-roofline-extra --steps 50 --warmup 8 2>/dev/null | tail -1) || exit 1
    echo "$which $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
