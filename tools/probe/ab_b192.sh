for rep in 1 2; do for v in "" b192; do L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/pgr-tk_amd/lib/variants/libpgrhip_$v.so; echo "== [$v] rep $rep"; PGR_HIP_LIB=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --queries 0 --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['ms_per_step'], d['stage_ms'])"; done; done
PGR_HIP_LIB=$GRAFT_REPO_ROOT/pgr-tk_amd/lib/variants/libpgrhip_b192.so python -m pytest tests/test_gpu_shmmr.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
