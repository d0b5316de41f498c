#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6_box
python bench.py --config 2048x1024 --steps 3 --no-cpu-baseline --no-io --no-other-configs > gpurun_out/r6_box/plain.json 2> gpurun_out/r6_box/plain.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r6_box/prof" -- python3 $GRAFT_REPO_ROOT/bench.py --config 2048x1024 --steps 2 --no-cpu-baseline --no-io --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/r6_box/prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r6_box/prof.err"
cd "$GRAFT_REPO_ROOT"; rm -rf gpurun_out/r6_box/prof
python - <<'PY'
import json
for f in ("plain", "prof"):
    j = json.loads(open(f"gpurun_out/r6_box/{f}.json").read().strip().splitlines()[-1])
    print(f, j["value"], json.dumps(j["box"])[:700])
PY
