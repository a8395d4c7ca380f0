#!/bin/bash
# alternate bench.py between library variants (same python tree)
for i in 1 2 3; do
  for f in scripts/variants/libavsr_*.so; do
    out=$(AVSR_HIP_LIB=$PWD/$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-roofline-extra --steps 50 --warmup 8 2>/dev/null | tail -1) || exit 1
    echo "$(basename $f) $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
