#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_fuzz; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python tests/dev/fuzz_gpu.py ${1:-3000} ${2:-808} > "$out/fuzz_$2.txt" 2>&1 || { tail -5 "$out/fuzz_$2.txt"; exit 1; }
tail -1 "$out/fuzz_$2.txt"
