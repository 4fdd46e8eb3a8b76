#!/bin/bash
# The 1 -> 8 GPU sweep of bench.py on ONE node, weak (1000 x 10 Mbp per GPU) and strong (--strong: ONE set of 8000 contigs cut by
# exchange.shard_contigs), into one JSON: per N the line's value, ms_per_step, exchange_ms, merge_ms, value_overlapped (the merge of
# step i beside the tiles of step i + 1), the transport that ran and how many ranks the library's own RCCL communicator had.
#   tools/scale_sweep.sh [out.json] [steps] [warmup]          (run it where the GPUs are; 127.0.0.1 rendezvous, one rank per GPU)
#   DRY=1 tools/scale_sweep.sh [out.json] [steps] [warmup]    the dry run of a box with ONE GPU: N = 1 as above (through the process-group
#         path: --force-dist, the library's RCCL communicator with one rank) and N = 2 as two ranks on that one device over gloo
#         (--backend gloo --single-device; RCCL refuses two ranks on one device), 200 contigs per rank: every field of the sweep's
#         JSON is produced by the code path the 8-GPU node will run, only the transport differs
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/scale_sweep.json}
STEPS=${2:-10}
WARM=${3:-3}
NG=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
mkdir -p "$(dirname "$OUT")"
TMP=$(mktemp -d)
DRY=${DRY:-0}
for mode in weak strong; do
  for n in 1 2 4 8; do
    if [ "$DRY" = 1 ]; then [ "$n" -gt 2 ] && continue; else [ "$n" -gt "$NG" ] && continue; fi
    extra=""
    [ "$mode" = strong ] && extra="--strong --contigs 8000"
    if [ "$DRY" = 1 ]; then
      extra="--contigs 200 --no-cpu-baseline"
      [ "$mode" = strong ] && extra="--strong --contigs 400 --no-cpu-baseline"
      [ "$n" -eq 1 ] && extra="$extra --force-dist"
      [ "$n" -eq 2 ] && extra="$extra --backend gloo --single-device"
    fi
    off=0
    [ "$mode" = strong ] && off=10
    port=$((29600 + n + off))
    if [ "$n" -eq 1 ]; then
      MASTER_ADDR=127.0.0.1 MASTER_PORT=$port python "$R/bench.py" --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-extras --queries 0 $extra > "$TMP/$mode.$n.json" 2> "$TMP/$mode.$n.err"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
        "$R/bench.py" --gpus "$n" --steps "$STEPS" --warmup "$WARM" --queries 0 $extra > "$TMP/$mode.$n.json" 2> "$TMP/$mode.$n.err"
    fi
    echo "$mode N=$n rc=$?" >&2
  done
done
python - "$TMP" "$OUT" <<'PY'
import glob, json, os, sys
tmp, out = sys.argv[1], sys.argv[2]
rows = []
for f in sorted(glob.glob(os.path.join(tmp, "*.json"))):
    mode, n = os.path.basename(f).split(".")[:2]
    line = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if not line:
        rows.append({"mode": mode, "n_gpus": int(n), "error": open(f.replace(".json", ".err")).read()[-500:]})
        continue
    d = json.loads(line[-1])
    x = d.get("exchange") or {}
    rows.append({"mode": mode, "n_gpus": d["n_gpus"], "value_Gbp_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                 "exchange_ms": d.get("exchange_ms"), "merge_ms": d.get("merge_ms"), "value_overlapped": d.get("value_overlapped"),
                 "overlapped": d.get("overlapped"), "transport": x.get("transport"),
                 "rccl_ranks_in_the_librarys_communicator": x.get("rccl_ranks_in_the_librarys_communicator"),
                 "exchange_content_match": x.get("content_match"), "cpu_content_match_all_ranks": (d.get("cpu_baseline") or {}).get("content_match_all_ranks"),
                 "roofline_frac": (d.get("roofline") or {}).get("frac")})
rows.sort(key=lambda r: (r["mode"], r["n_gpus"]))
json.dump({"what": "bench.py --gpus N on one node, weak and --strong", "rows": rows}, open(out, "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
