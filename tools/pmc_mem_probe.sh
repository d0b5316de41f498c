#!/bin/bash
# Memory-side request statistics per kernel of one prover step (bench.py, headline config): how many L1->L2 and L2->fabric
# requests each kernel makes per byte it moves -- the uncoalesced store phase of k_modup_ntt (profiles/r05_exp_linear_store.txt)
# would have shown here as four times the write requests per byte.  One --pmc pass per group (PMC-only: --kernel-trace).
#   usage (GPU box, repo root): bash tools/pmc_mem_probe.sh OUT
out=$GRAFT_REPO_ROOT/$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B1="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-io --no-other-configs"
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python3 $B1 > "$out/$name.log" 2>&1 || echo "pass $name failed" >> "$out/failed.txt"; }
run l1   TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
run l2   TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
run ea   TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum
run wr   TCC_WRITE_sum WRITE_SIZE GRBM_GUI_ACTIVE
run fe   FETCH_SIZE
cd "$GRAFT_REPO_ROOT"
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
keep = ("k_modup_ntt", "k_moddown_ntt", "k_ks_mac", "k_intt_pack", "k_limb_ntt", "k_pack_v", "k_ct_blocks", "k_rescale_coef", "k_mul_plain")
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k, cs in sorted(acc.items()):
        if not k.startswith(keep):
            continue
        v = {c: x[0] / x[1] for c, x in cs.items()}
        rd, wr = 2.0 * v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024  # gfx950: FETCH_SIZE x2 (tools/collect_pmc.py)
        line = (f"{k:28s} fetch {rd / 1e6:8.1f} MB  write {wr / 1e6:8.1f} MB | L1->L2 rd req {v.get('TCP_TCC_READ_REQ_sum', 0) / 1e6:8.2f} M "
                f"wr req {v.get('TCP_TCC_WRITE_REQ_sum', 0) / 1e6:8.2f} M | bytes written per L1->L2 write req "
                f"{wr / max(v.get('TCP_TCC_WRITE_REQ_sum', 0), 1):6.1f} | L2 hit {v.get('TCC_HIT_sum', 0) / max(v.get('TCC_REQ_sum', 1), 1):5.2f} "
                f"| EA wr req {v.get('TCC_EA0_WRREQ_sum', 0) / 1e6:7.2f} M of which 64B {v.get('TCC_EA0_WRREQ_64B_sum', 0) / 1e6:7.2f} M "
                f"| EA rd req {v.get('TCC_EA0_RDREQ_sum', 0) / 1e6:7.2f} M | TA busy {v.get('TA_BUSY_avr', 0):10.0f} | tag stall {v.get('TCC_TAG_STALL_sum', 0) / 1e6:7.2f} M")
        print(line, file=fo); print(line)
PY
rm -rf "$out"/l1 "$out"/l2 "$out"/ea "$out"/wr "$out"/fe
