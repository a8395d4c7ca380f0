#!/bin/bash
# Snapshot a commit (package + bench.py, library built) under scripts/ab_base/ for same-box A/B runs of the whole
# step against the working tree:   scripts/snapshot_base.sh <commit>  &&  gpurun -- 'bash scripts/ab_step.sh 3'
set -e
cd "$(dirname "$0")/.."
D=${2:-scripts/ab_base}
rm -rf $D && mkdir -p $D
git archive "$1" mocov2_whisper_flamingo_amd bench.py include oracle BASELINE.json profiles/r01_pmc_traffic.json | tar -x -C $D
make -C $D/mocov2_whisper_flamingo_amd/csrc -j8 >/dev/null
echo "snapshot of $(git rev-parse --short "$1") in $D"
