#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run16; mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-io --no-other-configs > "$out/bench.json" 2> "$out/bench.err"
cd "$GRAFT_REPO_ROOT"; python tools/kernel_gaps.py "$out/trace" | tee "$out/gaps.txt"; head -2 $(ls $out/trace/*/*kernel_trace.csv | head -1) | cut -c1-400; rm -rf "$out/trace"
