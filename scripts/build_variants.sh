#!/bin/bash
# Build A/B variants of one kernel file as separate shared libraries:
#   [SRC=gemm] scripts/build_variants.sh name "-DFLAGS" [name2 "-DFLAGS2" ...]
set -e
SRC=${SRC:-gemm}
cd "$(dirname "$0")/../mocov2_whisper_flamingo_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../scripts/variants
OTHERS=$(ls *.o | grep -v "^$SRC.o$" | tr '\n' ' ')
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c $SRC.hip -o /tmp/${SRC}_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/variants/libavsr_$name.so /tmp/${SRC}_$name.o $OTHERS
  echo built $name "$flags"
done
