#!/bin/bash
# Runs bench.py (headline config, no CPU baseline) once per environment setting given as arguments and
# prints one line per run: the setting, seconds per step and the per-kernel milliseconds.
#   bash tools/exp_env.sh "" "LUMEN_KS_LANES=2" "LUMEN_KS_BATCH=32"
cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
  out=$(env $e python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 ${BENCH_ARGS} 2>/dev/null | tail -1)
  python3 - "$e" "$out" <<'PY'
import json, sys
e, line = sys.argv[1], sys.argv[2]
try:
    j = json.loads(line)
    k = j.get("kernels") or {}
    print(f"[{e or 'default'}] {j['value']:.4f} s/step | " + " ".join(f"{n}={v['ms']:.0f}" for n, v in k.items()))
except Exception as ex:
    print(f"[{e}] FAILED: {ex}: {line[:300]}")
PY
done
