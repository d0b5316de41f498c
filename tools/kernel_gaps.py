"""Idle time between consecutive kernels of one stream, from a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp in
ns): how much of a prover step is kernel boundaries.  usage: python tools/kernel_gaps.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys
import collections


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                         r.get("Queue_Id", "0")))
    rows.sort()
    by_pair = collections.defaultdict(lambda: [0, 0])
    busy = gap = 0
    prev = None
    span0, span1 = rows[0][0], max(r[1] for r in rows)
    for s_, e_, name, q in rows:
        busy += e_ - s_
        if prev is not None:
            g = s_ - prev[1]
            if 0 <= g < 200_000:  # (longer: host-side pauses between phases, not kernel boundaries)
                gap += g
                k = (prev[2][:24], name[:24])
                by_pair[k][0] += g
                by_pair[k][1] += 1
        prev = (s_, e_, name)
    print(f"{len(rows)} dispatches, span {(span1 - span0) / 1e6:.1f} ms, kernels busy {busy / 1e6:.1f} ms, boundary gaps (< 0.2 ms each) {gap / 1e6:.1f} ms "
          f"= {100.0 * gap / (span1 - span0):.2f} % of the span, {gap / max(1, len(rows) - 1):.0f} ns per boundary")
    for k, (g, n) in sorted(by_pair.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  {k[0]:26s} -> {k[1]:26s} {n:6d} boundaries, {g / n:7.0f} ns each, {g / 1e6:7.2f} ms")


if __name__ == "__main__":
    main()
