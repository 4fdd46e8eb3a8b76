#!/bin/bash
# A/B timing of tile-kernel build variants on the GPU box: tools/ab_variants.sh <outfile> <variant> [<variant> ...]  ("base" = the shipped library)
out=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/pgr-tk_amd/lib/variants/libpgrhip_$v.so"; fi
  for rep in 1 2; do
    PGR_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --queries 0 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l)
        print('%-12s step %.3f ms  tile %.3f ms  list %.3f ms  value %.1f Gbp/s' % ('$v', d['ms_per_step'], d['stage_ms']['level1_tile'], d['stage_ms']['level2'], d['value']))
" >> $out
  done
done
cat $out
