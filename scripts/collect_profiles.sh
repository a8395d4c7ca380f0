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
# per-launch counter table of the visual trunk's convolutions
timeout -k 10 120 python3 scripts/visual_only.py 2 --eager --dump-shapes $O/${TAG}_vis_shapes.json > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=$O/${TAG}_vis_$(echo $C | cut -c1-5)
  timeout -s KILL 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 scripts/visual_only.py 2 --eager > /dev/null 2> $D.err
done
python3 scripts/pmc_convs.py $O/${TAG}_vis_FETCH $O/${TAG}_vis_WRITE $O/${TAG}_vis_SQ_VA $O/${TAG}_vis_shapes.json > $O/${TAG}_conv_pmc.json
echo convs done
# launches per step, by kernel
for n in 3 13; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ps_$n -- $B --steps $n --warmup 1 > /dev/null 2> $O/${TAG}_ps_$n.err
done
python3 scripts/per_step_kernels.py $(ls $O/${TAG}_ps_3/*/*_kernel_stats.csv) 3 $(ls $O/${TAG}_ps_13/*/*_kernel_stats.csv) 13 > $O/${TAG}_per_step_kernels.json
echo all done
