#!/bin/bash
# Where does a transform wave's time go?  SQ counters of the plain limb transform (tools/ntt_only.py), one --pmc pass per
# group (PMC-only passes: --kernel-trace, no other tracing).  usage (GPU box, repo root): bash tools/pmc_wave_probe.sh OUT [logN]
out=$GRAFT_REPO_ROOT/$1; logn=${2:-14}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$out/avail.txt" 2>&1
grep -o "SQ_[A-Z0-9_]*" "$out/avail.txt" | sort -u > "$out/sq_counters.txt"
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python3 $GRAFT_REPO_ROOT/tools/ntt_only.py $logn 64 3 > "$out/$name.log" 2>&1 || echo "pass $name failed" >> "$out/failed.txt"
}
run lds   SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run wait  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
run act   SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run busy  SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU
run dep   SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
cd "$GRAFT_REPO_ROOT"
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_limb_ntt" not in k:
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k, cs in sorted(acc.items()):
        print(k, file=fo); print(k)
        for c, (v, n) in sorted(cs.items()):
            line = f"   {c:24s} {v / n:16.1f} per launch ({n} launches)"
            print(line, file=fo); print(line)
PY
rm -rf "$out"/lds "$out"/wait "$out"/act "$out"/busy "$out"/dep
