#!/bin/bash
# A/B of the inverse transform's builds: tools/r6_exp7.sh VARIANT (product alternated with lumenos_amd/csrc/variants/VARIANT), tools/ntt_only.py 14 512 1500
v=${1:-inv_tw8}
out=$GRAFT_REPO_ROOT/gpurun_out/r6_exp7; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/lumenos_amd/csrc/variants
for r in 1 2 3; do for b in product $v; do
  if [ $b = product ]; then unset LUMEN_HIP_LIB; else export LUMEN_HIP_LIB=$V/$b/liblumenos_hip.so; fi
  echo "round $r $b: $(timeout -k 10 200 python tools/ntt_only.py 14 512 1500 | tr '\n' ' ')" | tee -a "$out/$v.txt"
done; done
