#!/usr/bin/env python3
"""Which source lines own each scratch slot of one kernel?  (assembly from hipcc -S -gline-tables-only)

    python tools/spill_slots.py file.s <mangled-name-prefix>

Per scratch offset: the source lines of its spill stores and of its reloads (static instructions, not executions)."""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
prefix = sys.argv[2]
files = {}
for l in text:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
inside, cur = False, None
slots = collections.defaultdict(lambda: {"st": [], "ld": []})
for l in text:
    if l.startswith(prefix) and ":" in l.split(";")[0]:
        inside = True
    if inside and l.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = "%s:%s" % (files.get(int(m.group(1)), m.group(1)), m.group(2))
    m = re.match(r"\s*scratch_(store|load)_dword", l)
    if m:
        off = re.search(r"offset:(\d+)", l)
        slots[int(off.group(1)) if off else 0]["st" if m.group(1) == "store" else "ld"].append(cur)
for off in sorted(slots):
    s = slots[off]
    print(off, "ST", dict(collections.Counter(s["st"])), "LD", dict(collections.Counter(s["ld"])))
