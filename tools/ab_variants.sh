#!/bin/bash
# A/B of library BUILDS on one box, alternated process by process: tools/ab_variants.sh OUT ROUNDS NAME [NAME ...]
#   NAME = "product" (lumenos_amd/csrc/liblumenos_hip.so) or a variant of tools/build_variant.sh
# Every visit = bench.py --steps 8 --warmup 2 on the headline configuration; prints step time and the key-switch kernels.
# (Run-time switches are alternated inside ONE process by tools/ab_interleaved.py; a different build needs a process.)
out=$GRAFT_REPO_ROOT/$1; rounds=$2; shift; shift
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT"
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = product ]; then unset LUMEN_HIP_LIB; else export LUMEN_HIP_LIB=$GRAFT_REPO_ROOT/lumenos_amd/csrc/variants/$v/liblumenos_hip.so; fi
    python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-io --no-other-configs > "$out/bench_${v}_$r.json" 2> "$out/bench_${v}_$r.err"
    python3 - "$out/bench_${v}_$r.json" "$v" "$r" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = j["kernels"]
print(f"round {sys.argv[3]} {sys.argv[2]:12s} step {j['value']:.4f} s | " + " ".join(f"{n}={k[n]['ms']:.1f}" for n in ("ks_moddown_ntt", "ks_modup_ntt", "ks_mac", "ks_intt_c1")), flush=True)
PY
  done
done
