#!/bin/bash
# The parity-test configurations of BASELINE.json timed with the benchmark loop (not bench lines): one JSON summary per line.
F="--no-cpu-baseline --no-roofline --steps 10 --warmup 4"
for flags in "--mode eval" "--whisper tiny" "--whisper medium" "--whisper large-v2 --seconds 60" "--whisper large-v2 --seconds 60 --attention fp8"; do
  echo "$flags: $(timeout -k 10 300 python bench.py $F $flags 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "clips/s", d["ms_per_step"], "ms/step")')"
done
