#!/usr/bin/env python
"""Writes forge3d_amd/data/spa_terms.json: the periodic-term tables of the NREL Solar Position Algorithm
(Reda & Andreas 2003, NREL/TP-560-34302, Appendix A: Earth heliocentric L / B / R terms, nutation arguments Y and
coefficients, mean-obliquity polynomial).  These are published constants; the numbers are read here from the
reference checkout's table file (src/geo/solar_coefficients.rs) because the report itself is not available offline.
Run in the build container only (needs /root/reference):  python tools/make_spa_tables.py
"""
import json
import re
import sys
from pathlib import Path

SRC = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/geo/solar_coefficients.rs")
OUT = Path(__file__).resolve().parent.parent / "forge3d_amd" / "data" / "spa_terms.json"


def block(text, name):
    start = text.index(f"const {name}:")
    start = text.index("=", start)
    depth, i = 0, text.index("&[", start)
    begin = i
    while True:
        if text[i] == "[":
            depth += 1
        elif text[i] == "]":
            depth -= 1
            if depth == 0:
                break
        i += 1
    body = re.sub(r"//[^\n]*", "", text[begin:i + 1]).replace("&", "")
    body = re.sub(r"\bPI\b", repr(3.141592653589793), body)
    body = re.sub(r"(-?\d+\.\d+)\s*/\s*(\d+\.\d+)", lambda m: repr(float(m.group(1)) / float(m.group(2))), body)
    body = re.sub(r",\s*\]", "]", body)
    return json.loads(body)


text = SRC.read_text()
tables = {name: block(text, name) for name in ("TERMS_L", "TERMS_B", "TERMS_R", "NUTATION_COEFFS", "TERMS_Y", "TERMS_PE", "OBLIQUITY_COEFFS")}
assert [len(t) for t in tables["TERMS_L"]] == [64, 34, 20, 7, 3, 1], [len(t) for t in tables["TERMS_L"]]
assert len(tables["TERMS_Y"]) == len(tables["TERMS_PE"]) == 63 and len(tables["OBLIQUITY_COEFFS"]) == 11
OUT.parent.mkdir(exist_ok=True)
OUT.write_text(json.dumps(tables, separators=(",", ":")))
print(OUT, OUT.stat().st_size, "bytes", {k: (len(v), len(v[0]) if isinstance(v[0], list) else "") for k, v in tables.items()})
