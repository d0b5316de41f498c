#!/bin/bash
# round 6, experiment 1: ks_mac placement + layout / occupancy variants of the gadget product (one MI355X)
out=$GRAFT_REPO_ROOT/gpurun_out/r6_exp1; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/lumenos_amd/csrc/variants
set -o pipefail
echo "== parity of the limb-major variant" | tee "$out/log.txt"
LUMEN_HIP_LIB=$V/limbmajor/liblumenos_hip.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_reference_shapes.py -m gpu -x -q -k "inner_sum or key_switch or matrix or config" > "$out/parity_limbmajor.log" 2>&1 || { tail -20 "$out/parity_limbmajor.log"; exit 1; }
tail -2 "$out/parity_limbmajor.log" | tee -a "$out/log.txt"
echo "== placement, in situ, 3 processes" | tee -a "$out/log.txt"
for r in 1 2 3; do timeout -k 10 300 python tools/ks_mac_placement.py --insitu --sweep --tag insitu$r >> "$out/placement.jsonl" 2>> "$out/placement.err" || exit 1; echo "insitu $r done"; done
echo "== placement, probe only, 5 processes" | tee -a "$out/log.txt"
for r in 1 2 3 4 5; do timeout -k 10 200 python tools/ks_mac_placement.py --tag probe$r >> "$out/placement.jsonl" 2>> "$out/placement.err" || exit 1; done
echo "== variants, probe only, 2 rounds" | tee -a "$out/log.txt"
for r in 1 2; do for v in product limbmajor cols2 cols2vec2; do
  if [ $v = product ]; then unset LUMEN_HIP_LIB; else export LUMEN_HIP_LIB=$V/$v/liblumenos_hip.so; fi
  timeout -k 10 200 python tools/ks_mac_placement.py --cands 2 --tag $v$r >> "$out/variants.jsonl" 2>> "$out/variants.err" || exit 1
done; done
unset LUMEN_HIP_LIB
python - "$out" <<'PY' | tee -a "$out/log.txt"
import json, sys
for f in ("placement.jsonl", "variants.jsonl"):
    for l in open(sys.argv[1] + "/" + f):
        j = json.loads(l)
        print(j["tag"], j.get("s_per_step"), j.get("insitu_ms_per_step", {}).get("ks_mac"), j.get("insitu_ks_mac_ms_per_launch"), "probe", j["probe_product_blocks_ms"],
              "ext", [x[1] for x in j["vary_ext"]], "u", [x[1] for x in j["vary_u"]], "key", [x[1] for x in j["vary_key"]])
PY
echo "== full step, product vs limbmajor, 2 rounds" | tee -a "$out/log.txt"
bash tools/ab_variants.sh gpurun_out/r6_exp1/ab 2 product limbmajor 2>&1 | tee -a "$out/log.txt"
