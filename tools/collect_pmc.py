#!/usr/bin/env python3
"""Aggregates rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected in separate runs as
MI355X_MICROARCH.md prescribes) into per-launch HBM traffic per kernel.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -- python3 bench.py ...
    python tools/collect_pmc.py out/fetch out/write profiles/r01_pmc_traffic.json

gfx950 corrections applied (guide, section HBM): FETCH_SIZE reports half the bytes of a wide
coalesced read -> doubled; WRITE_SIZE is exact for 16-byte streaming stores.  Units: KiB.
"""
import collections
import csv
import glob
import json
import os
import sys


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[name]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        res[k] = {
            "launches": max(nf, nw),
            "fetch_bytes_per_launch": (2.0 * f * 1024 / nf) if nf else None,  # x2: gfx950 FETCH_SIZE correction
            "write_bytes_per_launch": (w * 1024 / nw) if nw else None,
        }
        if nf and nw:
            res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        if v.get("hbm_bytes_per_launch"):
            print(f"{k:40s} launches={v['launches']:6d}  HBM/launch = {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
