#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/final_r06; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
bash tools/final_profile.sh gpurun_out/final_r06 2>&1 | tail -12
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_default_20_steps.json" 2> "$out/bench_default_20_steps.err" || { tail -5 "$out/bench_default_20_steps.err"; exit 1; }
python - "$out/bench_default_20_steps.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["step_ms"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["valu"]["frac"], j["roofline"]["valu"]["issue_frac"], {k: v["ms"] for k, v in j["kernels"].items()})
print(j["box"]["identity"], j["box"]["smi_under_load"])
print({k: v["value"] for k, v in j["other_configs"].items()}, j["io_inclusive_s"], j["io_inclusive_fused_order_s"], j["marshal_s"], j["cpu_baseline"]["value"])
PY
