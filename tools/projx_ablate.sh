#!/bin/bash
# decomposition builds of the split-precision projection (run HERE; the .so files travel with gpurun in tools/_dbg):
# MMSSL_PROJX_DBG bit 0 = no A loads in the steady loop, bit 1 = no MFMAs, bit 2 = no cut into planes, bit 3 = no B DMA
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_dbg
OBJS=$(ls mmssl_amd/_obj/*.o | grep -v projection.o)
for d in 1 2 4 6 8 9 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -DMMSSL_PROJX_DBG=$d -c mmssl_amd/csrc/projection.hip -o /tmp/projx_dbg$d.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/libmmssl_xdbg$d.so $OBJS /tmp/projx_dbg$d.o
done
ls tools/_dbg
