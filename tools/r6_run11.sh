#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run11; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
echo "== 6 processes, placement incl. the group accumulator" | tee "$out/log.txt"
for r in 1 2 3 4 5 6; do
  LUMEN_DEBUG=1 timeout -k 10 300 python tools/ks_mac_placement.py --insitu --cands 0 --tag p$r >> "$out/spread.jsonl" 2>> "$out/spread.err" || { tail -5 "$out/spread.err"; exit 1; }
done
grep placement "$out/spread.err" | tail -6 | tee -a "$out/log.txt"
python - "$out" <<'PY' | tee -a "$out/log.txt"
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1] + "/spread.jsonl")]
for j in rows:
    print(j["tag"], j["s_per_step"], j["insitu_ms_per_step"], "probe", j["probe_product_blocks_ms"][1])
s = [j["s_per_step"] for j in rows]; m = [j["insitu_ms_per_step"]["ks_mac"] for j in rows]
print(f"step {min(s):.4f} .. {max(s):.4f} s ({(max(s) / min(s) - 1) * 100:.2f} %), ks_mac {min(m):.1f} .. {max(m):.1f} ms ({(max(m) / min(m) - 1) * 100:.2f} %)")
PY
timeout -k 10 600 python tools/ab_interleaved.py --switch LUMEN_KS_P_LAST --values 0 1 --rounds 4 --steps 6 > "$out/p_last.txt" 2>&1 || { tail -5 "$out/p_last.txt"; exit 1; }
grep "^# LUMEN" "$out/p_last.txt" | tee -a "$out/log.txt"
timeout -k 10 600 python tools/ab_interleaved.py --switch LUMEN_MODDOWN_TGROUP --values 2 1 3 --rounds 3 --steps 6 > "$out/moddown_tgroup.txt" 2>&1 || exit 1
grep "^# LUMEN" "$out/moddown_tgroup.txt" | tee -a "$out/log.txt"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_reference_shapes.py -m gpu -x -q -k "inner_sum or key_switch or matrix or config or lazy" > "$out/parity.log" 2>&1; tail -2 "$out/parity.log" | tee -a "$out/log.txt"
