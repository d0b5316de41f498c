#!/bin/bash
# ModDown's surplus fetch (round 4): for the product build and each variant library given, the per-step time of
# k_moddown_ntt (bench.py's HIP-event table) and its FETCH_SIZE / WRITE_SIZE per launch (separate PMC passes).
#   usage (GPU box, repo root): bash tools/exp_moddown.sh gpurun_out/moddown product tg2 tg6 tg12 ...
# "product" = lumenos_amd/csrc/liblumenos_hip.so, NAME = lumenos_amd/csrc/variants/NAME/liblumenos_hip.so
set -e
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-io --no-other-configs"
B1="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-io --no-other-configs"
for v in "$@"; do
  if [ "$v" = product ]; then unset LUMEN_HIP_LIB; else export LUMEN_HIP_LIB=$GRAFT_REPO_ROOT/lumenos_amd/csrc/variants/$v/liblumenos_hip.so; fi
  python3 $B > "$out/bench_$v.json" 2> "$out/bench_$v.err"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch_$v" -- python3 $B1 > "$out/fetch_$v.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write_$v" -- python3 $B1 > "$out/write_$v.log" 2>&1
  (cd "$GRAFT_REPO_ROOT" && python3 tools/collect_pmc.py "$out/fetch_$v" "$out/write_$v" "$out/pmc_$v.json" > /dev/null)
  rm -rf "$out/fetch_$v" "$out/write_$v"
  python3 - "$out" "$v" <<'PY'
import json, sys
out, v = sys.argv[1:3]
b = json.load(open(f"{out}/bench_{v}.json"))
p = json.load(open(f"{out}/pmc_{v}.json"))
def pm(name):
    for k, e in p.items():
        if isinstance(e, dict) and k.startswith(name):
            return e
    return {}
line = [f"{v:10s} step {b['value']:.4f} s"]
for kern, pk in (("ks_moddown_ntt", "k_moddown_ntt"), ("ks_modup_ntt", "k_modup_ntt"), ("ks_mac", "k_ks_mac")):
    e, m = b["kernels"][kern], pm(pk)
    line.append(f"{kern} {e['ms']:.1f} ms fetch {m.get('fetch_bytes_per_launch', 0) / 1e6:.0f} write {m.get('write_bytes_per_launch', 0) / 1e6:.0f} MB")
print(" | ".join(line), flush=True)
open(f"{out}/summary.txt", "a").write(" | ".join(line) + "\n")
PY
done
