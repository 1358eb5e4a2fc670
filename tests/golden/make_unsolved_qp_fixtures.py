#!/usr/bin/env python3
"""Convert the QP data headers of the reference (test/unsolved_QPs/*.hpp: C array initialisers in the
qpOASES CSC layout -- lb, ub, lbA, ubA, g, A_jc, A_ir, A_val, H_jc, H_ir, H_val) into a JSON data fixture,
tests/golden/unsolved_qps.json. VALUES ONLY: array name -> list of numbers ("inf" / "-inf" kept as strings
because JSON has no infinity). Two file names carry the only solver verdict the reference records anywhere:
hs035_unbounded.hpp and hs067_unbounded.hpp -> "expected_status": 23 (QPERROR_UNBOUNDED, Types.hpp:51-73).

Run in the build container (needs /root/reference); tests only read the JSON."""
import glob
import json
import os
import re
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/test/unsolved_QPs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "unsolved_qps.json")
INT_ARRAYS = {"A_jc", "A_ir", "H_jc", "H_ir"}


def parse(path):
    text = open(path).read()
    out = {}
    for m in re.finditer(r"\b(\w+)\s*\[\s*\]\s*=\s*\{([^}]*)\}", text, flags=re.S):
        name, body = m.group(1), m.group(2)
        toks = [t.strip() for t in body.replace("\n", " ").split(",") if t.strip()]
        if name in INT_ARRAYS:
            out[name] = [int(t) for t in toks]
        else:
            out[name] = [t if "inf" in t else float(t) for t in toks]
    return out


if __name__ == "__main__":
    fixtures = {}
    for path in sorted(glob.glob(os.path.join(SRC, "*.hpp"))):
        name = os.path.splitext(os.path.basename(path))[0]
        d = parse(path)
        d["nV"] = len(d["g"]); d["nC"] = len(d["lbA"])
        if name.endswith("_unbounded"):
            d["expected_status"] = 23
        fixtures[name] = d
    json.dump({"source": "test/unsolved_QPs/*.hpp (values only)", "qps": fixtures}, open(OUT, "w"))
    print("%d QPs -> %s (%d bytes)" % (len(fixtures), OUT, os.path.getsize(OUT)))
