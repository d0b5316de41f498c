#!/bin/bash
# Collects SQ counters for one command, one small counter group per rocprofv3 run (PMC only,
# with --kernel-trace), and prints per-kernel averages.   usage: pmc_sweep.sh OUTDIR -- python3 ...
set -e
out=$1; shift; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/g$i" -- "$@" > "$GRAFT_REPO_ROOT/$out/g$i.log" 2>&1 || echo "group $i failed: $g"
  i=$((i+1))
done
cd "$GRAFT_REPO_ROOT"
python3 - "$out" <<'PY'
import sys, glob, csv, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v, n = acc[k][c]
        print(f"    {c:28s} {v / n:16.1f}  (avg of {n})")
PY
