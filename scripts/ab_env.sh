#!/bin/bash
# Alternates bench.py of the working tree with and without an environment setting, N times each:
#   scripts/ab_env.sh "AVSR_GRAM_BN3=0" 3     (prints "<setting|default> ms_per_step ms_per_step_steady clips/s")
SETTING=$1
N=${2:-3}
for i in $(seq $N); do
  for which in base new; do
    if [ $which = base ]; then
      out=$(env $SETTING timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-roofline-extra --steps 50 --warmup 8 2>/dev/null | tail -1) || exit 1
      tag="$SETTING"
    else
      out=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-roofline-extra --steps 50 --warmup 8 2>/dev/null | tail -1) || exit 1
      tag="default"
    fi
    echo "$tag $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_steady"], d["value"])')"
  done
done
