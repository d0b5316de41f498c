#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_exp7; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/lumenos_amd/csrc/variants
for r in 1 2 3; do for v in product inv_tw8; do
  if [ $v = product ]; then unset LUMEN_HIP_LIB; else export LUMEN_HIP_LIB=$V/$v/liblumenos_hip.so; fi
  echo "round $r $v: $(timeout -k 10 200 python tools/ntt_only.py 14 512 1500 | tr '\n' ' ')" | tee -a "$out/inv_tw8.txt"
done; done
