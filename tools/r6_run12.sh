#!/bin/bash
# (NOTE: the LUMEN_KS_PLACEMENT_SPACER variable set below was never read by the library in this run -- the variant was not built then: twelve identical processes.
#  The spacer experiment proper is tools/exp_spacer.patch + tools/exp_spacer.py + tools/r6_run18.sh.)
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run12; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
for r in 1 2 3 4; do for sp in 0 4096 16384; do
  LUMEN_KS_PLACEMENT_SPACER=$sp LUMEN_DEBUG=1 timeout -k 10 300 python tools/ks_mac_placement.py --insitu --cands 0 --tag sp${sp}_$r >> "$out/spread.jsonl" 2>> "$out/spread.err" || { tail -5 "$out/spread.err"; exit 1; }
done; done
python - "$out" <<'PY' | tee -a "$out/log.txt"
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1] + "/spread.jsonl")]
for j in rows:
    print(j["tag"], j["s_per_step"], j["insitu_ms_per_step"], "probe", j["probe_product_blocks_ms"][1])
PY
grep placement "$out/spread.err" | tee -a "$out/log.txt"
