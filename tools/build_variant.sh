#!/bin/bash
# A/B builds of a kernel file: compiles csrc/level1.hip and small.hip (which shares level1_select.h) -- or the files named in FILES,
# e.g. FILES=query_fused -- with extra flags and links them with the shipped objects into
# pgr-tk_amd/lib/variants/libpgrhip_<name>.so (used through PGR_HIP_LIB=<path>).
#   [FILES="query_fused"] tools/build_variant.sh <name> [extra hipcc flags ...]
set -e
cd "$(dirname "$0")/../pgr-tk_amd"
name=$1; shift
mkdir -p build/var_$name lib/variants
FLAGS="-O3 -Wall -Wno-unused-function -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
for f in ${FILES:-level1 small}; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c csrc/$f.hip -o build/var_$name/$f.o &
done
wait
objs=""
for o in build/*.o; do
  b=$(basename $o)
  if [ -f build/var_$name/$b ]; then objs="$objs build/var_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o lib/variants/libpgrhip_$name.so $objs -ldl
echo lib/variants/libpgrhip_$name.so
