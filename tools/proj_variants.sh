#!/bin/bash
# variant builds of csrc/projection.hip for A/B runs (run HERE: the .so files travel with gpurun; loaded through MMSSL_LIB):
#   bash tools/proj_variants.sh "name1:-DFLAG=1 -DOTHER=2" "name2:..."
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_dbg
OBJS=$(ls mmssl_amd/_obj/*.o | grep -v projection.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $flags -c mmssl_amd/csrc/projection.hip -o /tmp/proj_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libmmssl_$name.so $OBJS /tmp/proj_$name.o
done
ls tools/_dbg
