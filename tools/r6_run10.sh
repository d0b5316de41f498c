#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run10; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
echo "== 8 processes, final build: step and key-switch kernels" | tee "$out/log.txt"
for r in 1 2 3 4 5 6 7 8; do
  LUMEN_DEBUG=1 timeout -k 10 300 python tools/ks_mac_placement.py --insitu --cands 0 --tag p$r >> "$out/spread.jsonl" 2>> "$out/spread.err" || { tail -5 "$out/spread.err"; exit 1; }
done
python - "$out" <<'PY' | tee -a "$out/log.txt"
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1] + "/spread.jsonl")]
for j in rows:
    print(j["tag"], j["s_per_step"], j["insitu_ms_per_step"], "probe", j["probe_product_blocks_ms"][1])
s = [j["s_per_step"] for j in rows]; m = [j["insitu_ms_per_step"]["ks_mac"] for j in rows]
print(f"step {min(s):.4f} .. {max(s):.4f} s ({(max(s) / min(s) - 1) * 100:.2f} %), ks_mac {min(m):.1f} .. {max(m):.1f} ms ({(max(m) / min(m) - 1) * 100:.2f} %)")
PY
echo "== rehearsals of the N > 1 paths on the one GPU" | tee -a "$out/log.txt"
timeout -k 10 300 python bench.py --gpus 4 --single-process --share-gpu --config 2048x1024 --steps 3 --no-cpu-baseline > "$out/bench_n4_single_process.json" 2> "$out/bench_n4_single_process.err" || { tail -5 "$out/bench_n4_single_process.err"; exit 1; }
timeout -k 10 400 python bench.py --gpus 2 --share-gpu --dist-backend gloo --config 2048x1024 --steps 2 --no-cpu-baseline > "$out/bench_n2_gloo.json" 2> "$out/bench_n2_gloo.err" || { tail -5 "$out/bench_n2_gloo.err"; exit 1; }
python - "$out" <<'PY' | tee -a "$out/log.txt"
import json, sys
for f in ("bench_n4_single_process.json", "bench_n2_gloo.json"):
    j = json.loads(open(sys.argv[1] + "/" + f).read().strip().splitlines()[-1])
    print(f, j["value"], j["config"]["transport"], (j.get("check") or {}).get("ok"))
PY
echo "== fuzz" | tee -a "$out/log.txt"
timeout -k 10 900 python tests/dev/fuzz_gpu.py 1500 606 > "$out/fuzz_1500.txt" 2>&1 || { tail -5 "$out/fuzz_1500.txt"; exit 1; }
tail -1 "$out/fuzz_1500.txt" | tee -a "$out/log.txt"
FUZZ_LOGN=13,14 timeout -k 10 600 python tests/dev/fuzz_gpu.py 80 79 > "$out/fuzz_logn13_14.txt" 2>&1 || { tail -5 "$out/fuzz_logn13_14.txt"; exit 1; }
tail -1 "$out/fuzz_logn13_14.txt" | tee -a "$out/log.txt"
