#!/bin/bash
# round 6, experiment 2: is bandwidth a function of the block streamed (ubench_place), and how is the gadget product's time
# distributed over MANY candidate placements (40 u blocks, 16 ext blocks) in one process
out=$GRAFT_REPO_ROOT/gpurun_out/r6_exp2; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
set -o pipefail
for r in 1 2; do timeout -k 10 120 ./tools/ubench_place 12 > "$out/ubench_place_$r.txt" 2>&1 || exit 1; done
cat "$out/ubench_place_1.txt"
for r in 1 2; do timeout -k 10 300 python tools/ks_mac_placement.py --cands 30 --reps 60 --tag many$r >> "$out/many.jsonl" 2>> "$out/many.err" || exit 1; done
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1] + "/many.jsonl"):
    j = json.loads(l)
    print(j["tag"], "product blocks", j["probe_product_blocks_ms"])
    for k in ("vary_ext", "vary_u"):
        print("  ", k, " ".join(f"{m:.4f}" for _, m in j[k]))
PY
