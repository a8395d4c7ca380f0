#!/bin/bash
# Per-step kernel table of bench.py under an environment setting:  bash scripts/ps_compare.sh TAG [VAR=VALUE]
# (run on the GPU box from the repo root; writes gpurun_out/TAG_per_step.json)
set -e
TAG=$1
SET=$2
export TMPDIR=/tmp
[ -n "$SET" ] && export $SET
O=$PWD/gpurun_out
B="python3 bench.py --no-cpu-baseline --no-roofline --no-roofline-extra"
for n in 3 13; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ps_$n -- $B --steps $n --warmup 1 > /dev/null 2> $O/${TAG}_ps_$n.err
done
python3 scripts/per_step_kernels.py $(ls $O/${TAG}_ps_3/*/*_kernel_stats.csv) 3 $(ls $O/${TAG}_ps_13/*/*_kernel_stats.csv) 13 > $O/${TAG}_per_step.json
rm -rf $O/${TAG}_ps_3 $O/${TAG}_ps_13
