#!/bin/bash
# the gadget product alone on the library's own (unselected) blocks, 8 processes per build: does the cache policy of its streams change the level or the spread?
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run22; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/lumenos_amd/csrc/variants
for r in 1 2 3 4 5 6 7 8; do for v in product mac_ldnormal mac_ntstore; do
  if [ $v = product ]; then unset LUMEN_HIP_LIB; else export LUMEN_HIP_LIB=$V/$v/liblumenos_hip.so; fi
  LUMEN_KS_PLACEMENT=0 timeout -k 10 200 python tools/ks_mac_placement.py --cands 0 --reps 300 --tag $v >> "$out/variants.jsonl" 2>> "$out/variants.err" || exit 1
done; done
python - "$out" <<'PY'
import json, sys, collections
acc = collections.defaultdict(list)
for l in open(sys.argv[1] + "/variants.jsonl"):
    j = json.loads(l)
    acc[j["tag"]].append(j["probe_product_blocks_ms"][1])
for k, v in acc.items():
    print(f"{k:14s}", " ".join(f"{x:.4f}" for x in v), f"| mean {sum(v)/len(v):.4f} min {min(v):.4f} max {max(v):.4f}")
PY
