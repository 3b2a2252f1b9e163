#!/bin/bash
# usage: bash tools/build_variant.sh <name> "<extra hipcc flags>" [file.hip ...]   (default file: dpr_tiled.hip)
# Builds diffpointrasterisation.jl_amd/variants_libdpr_<name>.so: the listed sources recompiled
# with the extra flags, the other objects taken from the regular build.  Load it with
# DPR_LIB_OVERRIDE=<path> (diffpointrasterisation.jl_amd/_lib.py).
set -e
NAME=$1; FLAGS=$2; shift 2
FILES=${@:-dpr_tiled.hip}
cd "$(dirname "$0")/../diffpointrasterisation.jl_amd/csrc"
make -s
OBJS=""
for o in dpr_api dpr_tiled dpr_chunked dpr_chunkown dpr_sort dpr_comm; do
  if echo " $FILES " | grep -q " $o.hip "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function $FLAGS -c $o.hip -o /tmp/${o}_$NAME.o
    OBJS="$OBJS /tmp/${o}_$NAME.o"
  else
    OBJS="$OBJS $o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants_libdpr_$NAME.so $OBJS -ldl
ls -la ../variants_libdpr_$NAME.so
