#!/bin/bash
# Build A/B variants of the GEMM kernel as separate shared libraries: scripts/build_variants.sh name "-DFLAGS" ...
set -e
cd "$(dirname "$0")/../mocov2_whisper_flamingo_amd/csrc"
make -j8 >/dev/null
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c gemm.hip -o /tmp/gemm_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/variants/libavsr_$name.so /tmp/gemm_$name.o attention.o norm.o visual.o loss.o elementwise.o logmel.o
  echo built $name "$flags"
done
