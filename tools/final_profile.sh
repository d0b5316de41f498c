#!/bin/bash
# Round-end evidence in one GPU call: the bench line of every BASELINE configuration, the I/O-inclusive
# run, and the rocprofv3 kernel stats + PMC passes of the headline command (tools/profile_bench.sh).
#   usage (GPU box, repo root): bash tools/final_profile.sh gpurun_out/final_rNN
out=$1; mkdir -p "$out"
cd "$GRAFT_REPO_ROOT"
bash tools/profile_bench.sh "$out/prof" 16384x4096 > "$out/profile.log" 2>&1
cp "$out/prof/pmc_traffic_16384x4096.json" profiles/pmc_traffic_16384x4096.json   # so that the runs below quote it
for cfg in 2048x1024 4096x2048 8192x4096; do
  python3 bench.py --config $cfg --steps 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_$cfg.json"
done
python3 bench.py --config 16384x4096 --steps 3 --ring-switch-logn 10 --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_16384x4096_ringswitch.json"
python3 bench.py --include-io --steps 3 2>/dev/null | tail -1 > "$out/bench_16384x4096.json"
python3 tools/ntt_only.py 14 512 1500 > "$out/ntt_only_n14.txt" 2>&1
./tools/ubench_bfly > "$out/ubench_bfly.txt" 2>&1
for f in "$out"/bench_*.json; do python3 -c "
import json,sys
j=json.load(open('$f')); print('$f'.split('/')[-1], j['value'], j.get('io_inclusive_s'), (j.get('roofline') or {}).get('frac'), (j.get('roofline') or {}).get('valu_insts_per_butterfly'))"; done
