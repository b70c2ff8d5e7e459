#!/bin/bash
# variant builds of ONE csrc source for A/B runs (run HERE: the .so files travel with gpurun; loaded through MMSSL_LIB):
#   bash tools/variants.sh infonce "old:-DMMSSL_INFONCE_LDS_TILES=0" "new:"
set -e
cd "$(dirname "$0")/.."
src=$1; shift
mkdir -p tools/_dbg
OBJS=$(ls mmssl_amd/_obj/*.o | grep -v "/$src.o")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -Immssl_amd/csrc $flags -c mmssl_amd/csrc/$src.hip -o /tmp/${src}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libmmssl_$name.so $OBJS /tmp/${src}_$name.o
done
ls tools/_dbg
