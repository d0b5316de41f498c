#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run9; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
rocm-smi -d 0 --showclocks --showpower --showtemp > "$out/smi_raw.txt" 2>&1
rocm-smi -d 0 --showbus --showuniqueid --showdriverversion --showserial >> "$out/smi_raw.txt" 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/gputest.log" 2>&1; rc=$?; tail -4 "$out/gputest.log"; [ $rc = 0 ] || exit $rc
timeout -k 10 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" || { tail -5 "$out/bench_default.err"; exit 1; }
python - "$out/bench_default.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["step_ms"], j["roofline"]["frac"], {k: v["ms"] for k, v in j["kernels"].items()})
print(json.dumps(j["box"])[:1500])
print({k: v["value"] for k, v in j["other_configs"].items()})
PY
