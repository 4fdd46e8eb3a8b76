#!/bin/bash
# A/B builds of libpgrhip.so with extra compile flags (kernel experiments):
#   tools/build_variant.sh <name> "<extra flags>"   -> pgr-tk_amd/lib/variants/libpgrhip_<name>.so
# Run a variant with  PGR_HIP_LIB=pgr-tk_amd/lib/variants/libpgrhip_<name>.so python bench.py ...
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)/pgr-tk_amd
B=$R/build_$NAME; mkdir -p $B $R/lib/variants
for f in level1 level2 pack scan ctx api index mapgraph exchange shard small query_fused; do
  /opt/rocm/bin/hipcc -O3 -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $FLAGS -c $R/csrc/$f.hip -o $B/$f.o &
done
g++ -O3 -std=c++17 -fPIC -pthread -c $R/csrc/hostpack.cpp -o $B/hostpack.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $R/lib/variants/libpgrhip_$NAME.so $B/*.o -ldl
rm -rf $B
ls -la $R/lib/variants/libpgrhip_$NAME.so
