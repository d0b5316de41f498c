#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run18; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
for r in 1 2 3 4 5 6; do timeout -k 10 300 python tools/exp_spacer.py >> "$out/spacer.jsonl" 2>> "$out/spacer.err" || { tail -5 "$out/spacer.err"; exit 1; }; done
cat "$out/spacer.jsonl"; grep placement "$out/spacer.err" | cut -c1-260 | tail -40
