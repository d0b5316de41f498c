#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/final_r06; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/gputest.log" 2>&1; rc=$?; tail -4 "$out/gputest.log"; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
