#!/bin/bash
# Collects the judged profiling evidence of one state into gpurun_out/ (run on the GPU box from the repo root):
#   bash scripts/collect_profiles.sh TAG
# bench kernel stats, GEMM traffic passes, cross-attention counter passes at three shapes.
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
O=$PWD/gpurun_out
B="python3 bench.py --no-cpu-baseline --no-roofline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- $B --steps 3 --warmup 1 > $O/${TAG}_stats_bench.json 2> $O/${TAG}_stats.err
echo stats done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- $B --steps 2 --warmup 1 --no-graph --no-pipeline > /dev/null 2> $O/${TAG}_pmc_fetch.err
echo fetch done
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- $B --steps 2 --warmup 1 --no-graph --no-pipeline > /dev/null 2> $O/${TAG}_pmc_write.err
echo write done
python3 scripts/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write > $O/${TAG}_pmc_traffic.json
for S in 8,75 8,400 64,400; do
  N=${S/,/x}
  timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_xattn_$N -- python3 scripts/bench_xattn_block.py $S,fused --once > /dev/null 2> $O/${TAG}_xattn_$N.err
  python3 scripts/pmc_xattn.py $O/${TAG}_xattn_$N ${S/,/ } > $O/${TAG}_xattn_pmc_$N.json
  echo xattn $S done
done
timeout -k 10 200 python3 scripts/bench_xattn_block.py > $O/${TAG}_xattn_block_times.txt 2>&1
echo all done
