#!/usr/bin/env python3
"""Static VALU instruction count per butterfly of tools/ubench_fold.hip's three kernels (the hot loop = the basic
block with the most VALU instructions; 32 butterflies per iteration).
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I lumenos_amd/csrc --cuda-device-only -S tools/ubench_fold.hip -o /tmp/ubench_fold.s
    python tools/ubench_fold_isa.py /tmp/ubench_fold.s"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
names = ["shoup-asm", "shoup-c", "fold-c", "shoup-asm2", "shoup-asm9", "shoup-asm29", "shoup-asm4"]
for m in re.finditer(r"^_Z6k_bflyILi(\d)EEvPyPK4tw_tyji:.*?\n(.*?)\n\s*s_endpgm", txt, re.S | re.M):
    mode, body = int(m.group(1)), m.group(2)
    blocks = re.split(r"^\.LBB\d+_\d+:.*$", body, flags=re.M)
    best = max(blocks, key=lambda b: len(re.findall(r"^\s+v_", b, re.M)))
    ins = re.findall(r"^\s+(v_\w+)", best, re.M)
    c = collections.Counter(ins)
    print(f"{names[mode]:10s} hot loop: {len(ins)} VALU instructions = {len(ins) / 32:.1f} per butterfly, "
          f"v_mad_u64_u32 {c['v_mad_u64_u32'] / 32:.1f} per butterfly; "
          + ", ".join(f"{k} {v}" for k, v in c.most_common(7)))
