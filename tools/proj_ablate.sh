#!/bin/bash
# decomposition builds of csrc/projection.hip (run HERE, the .so files travel with gpurun): MMSSL_PROJ_DBG bit 0 = no DMA
# in the steady loop, bit 1 = no MFMAs, bit 2 = no fragment reads
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_dbg
OBJS=$(ls mmssl_amd/_obj/*.o | grep -v projection.o)
for d in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DMMSSL_PROJ_DBG=$d -c mmssl_amd/csrc/projection.hip -o /tmp/proj_dbg$d.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libmmssl_dbg$d.so $OBJS /tmp/proj_dbg$d.o
done
ls -la tools/_dbg
