#!/bin/bash
# round 6, experiment 5: variants of the gadget product (columns per thread x prefetch depth) on identical allocation sequences
out=$GRAFT_REPO_ROOT/gpurun_out/r6_exp5b; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/lumenos_amd/csrc/variants
for v in c4pf2 c2pf4; do
LUMEN_HIP_LIB=$V/$v/liblumenos_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inner_sum or key_switch or matrix" > "$out/parity_$v.log" 2>&1 || { tail -20 "$out/parity_$v.log"; exit 1; }
tail -1 "$out/parity_$v.log"
done
for r in 1 2 3; do for v in c4pf1 c4pf2 c4pf3 c2pf2 c2pf3 c2pf4 c2pf5 c1pf5; do
  export LUMEN_HIP_LIB=$V/$v/liblumenos_hip.so
  LUMEN_KS_PLACEMENT=0 timeout -k 10 200 python tools/ks_mac_placement.py --cands 0 --reps 400 --tag $v >> "$out/variants.jsonl" 2>> "$out/variants.err" || exit 1
done; done
python - "$out" <<'PY'
import json, sys, collections
acc = collections.defaultdict(list)
for l in open(sys.argv[1] + "/variants.jsonl"):
    j = json.loads(l)
    acc[j["tag"]].append(j["probe_product_blocks_ms"][1:] + [j["probe_product_blocks_again_ms"]])
for k, v in acc.items():
    print(f"{k:14s}", " | ".join(" ".join(f"{x:.4f}" for x in r) for r in v))
PY
