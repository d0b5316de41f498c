#!/bin/bash
# round 6, experiment 8: work-list group sizes and batch size again, now that the key switch's streams are limb-major
out=$GRAFT_REPO_ROOT/gpurun_out/r6_exp8; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python tools/ab_interleaved.py --switch LUMEN_MODUP_TGROUP --values 4 2 7 --rounds 3 --steps 6 > "$out/modup_tgroup.txt" 2>&1 || exit 1
grep "^# LUMEN" "$out/modup_tgroup.txt"
timeout -k 10 600 python tools/ab_interleaved.py --switch LUMEN_MODDOWN_TGROUP --values 4 2 6 12 --rounds 3 --steps 6 > "$out/moddown_tgroup.txt" 2>&1 || exit 1
grep "^# LUMEN" "$out/moddown_tgroup.txt"
timeout -k 10 600 python tools/ab_interleaved.py --switch LUMEN_KS_BATCH --values 64 128 32 --rounds 3 --steps 6 > "$out/ks_batch.txt" 2>&1 || exit 1
grep "^# LUMEN" "$out/ks_batch.txt"
