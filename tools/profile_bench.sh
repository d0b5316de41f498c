#!/bin/bash
# Collects the evidence bench.py's roofline object refers to, for one config:
#   1. rocprofv3 --kernel-trace --stats of the bench command      -> OUT/stats
#   2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes        -> OUT/fetch, OUT/write
#   3. tools/collect_pmc.py                                           -> OUT/pmc_traffic_<cfg>.json
# usage (on the GPU box, from the repo root): bash tools/profile_bench.sh gpurun_out/prof 16384x4096
set -e
out=$GRAFT_REPO_ROOT/$1; cfg=${2:-16384x4096}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# (the default run's extra legs -- io, other configurations -- launch other kernels or the same ones on other
# rings; the summaries are per kernel NAME, so they are left out of the profiled command to keep it short)
B="$GRAFT_REPO_ROOT/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-io --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $B > "$out/stats.log" 2>&1
B1="$GRAFT_REPO_ROOT/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-io --no-other-configs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 $B1 > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 $B1 > "$out/write.log" 2>&1
# SQ counters of the same command (instruction counts and VALU-busy cycles), two more PMC-only passes
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/sq0" -- python3 $B1 > "$out/sq0.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/sq1" -- python3 $B1 > "$out/sq1.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 tools/collect_pmc.py "$out/fetch" "$out/write" "$out/pmc_traffic_$cfg.json" "$out/sq0" "$out/sq1"
f=$(ls $out/stats/*/*kernel_stats.csv | head -1)
cp "$f" "$out/kernel_stats_$cfg.csv"
# the raw per-dispatch traces are large: keep only the summaries
rm -rf "$out/fetch" "$out/write" "$out/stats" "$out/sq0" "$out/sq1"
head -20 "$out/kernel_stats_$cfg.csv"
