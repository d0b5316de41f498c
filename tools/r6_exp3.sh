#!/bin/bash
# round 6, experiment 3: the limb-major product with / without placement selection, process by process; then the full GPU suite
out=$GRAFT_REPO_ROOT/gpurun_out/r6_exp3; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
set -o pipefail
for r in 1 2 3 4; do for sel in 0 6; do
  LUMEN_KS_PLACEMENT=$sel LUMEN_DEBUG=1 timeout -k 10 300 python tools/ks_mac_placement.py --insitu --cands 0 --tag sel${sel}_$r >> "$out/insitu.jsonl" 2>> "$out/insitu.err" || { tail -5 "$out/insitu.err"; exit 1; }
done; done
grep "placement" "$out/insitu.err" | tail -8
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1] + "/insitu.jsonl"):
    j = json.loads(l)
    print(j["tag"], j.get("s_per_step"), j.get("insitu_ms_per_step"), "probe", j["probe_product_blocks_ms"])
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/gputest.log" 2>&1; rc=$?; tail -4 "$out/gputest.log"; exit $rc
