#!/usr/bin/env python
"""Basic-block census of one kernel's gfx950 assembly (hipcc -S): per block the VALU / SALU /
VMEM / LDS / scratch instruction counts, and the natural loops (backward branches) with the
instruction totals of the blocks they span.  Usage: isa_blocks.py file.s [min_loop_valu]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
blocks, order, cur = {}, [], None
for ln in lines:
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    if m or cur is None:
        cur = m.group(1) if m else "entry"
        blocks[cur] = {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "scratch": 0, "trans": 0, "br": []}
        order.append(cur)
        if m:
            continue
    s = ln.strip()
    if not s or s.startswith((";", ".", "//")):
        continue
    op = s.split()[0]
    b = blocks[cur]
    if op.startswith("v_"):
        b["valu"] += 1
        if re.match(r"v_(rcp|sqrt|rsq|div_|exp|log|sin|cos)", op):
            b["trans"] += 1
    elif op.startswith("scratch_"):
        b["scratch"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")):
        b["vmem"] += 1
    elif op.startswith("ds_"):
        b["lds"] += 1
    elif op.startswith("s_"):
        b["salu"] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            b["br"].append(s.split()[-1])
idx = {n: i for i, n in enumerate(order)}
loops = []
for n in order:
    for t in blocks[n]["br"]:
        if t in idx and idx[t] <= idx[n]:
            loops.append((idx[t], idx[n]))
loops.sort(key=lambda p: (p[0], -p[1]))
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for a, z in loops:
    tot = {k: sum(blocks[order[i]][k] for i in range(a, z + 1)) for k in ("valu", "salu", "vmem", "lds", "scratch", "trans")}
    if tot["valu"] >= thr:
        print(f"loop {order[a]}..{order[z]} ({z - a + 1} blocks): {tot}")
