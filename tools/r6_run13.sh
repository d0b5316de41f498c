#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run13; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python bench.py --gpus 8 --single-process --share-gpu --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_n8_single_process.json" 2> "$out/bench_n8_single_process.err" || { tail -8 "$out/bench_n8_single_process.err"; exit 1; }
python - "$out" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench_n8_single_process.json").read().strip().splitlines()[-1])
print(j["value"], j["config"]["transport"], j.get("check"), j.get("collectives"))
PY
