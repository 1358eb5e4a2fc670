#!/usr/bin/env python3
"""Extract the CONTRACT of the plug-in interface from the reference header -- the names of the pure
virtual methods of `QPSolverInterface` (include/sqphot/QPsolverInterface.hpp:43-194) with their number
of parameters and constness -- into tests/golden/qpsolver_interface_pure_virtuals.json.

Run in the build container (needs /root/reference); the GPU box only reads the JSON. Data only: no
reference text is copied, just (name, nparams, const) triples."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/include/sqphot/QPsolverInterface.hpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qpsolver_interface_pure_virtuals.json")


def pure_virtuals(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = []
    for m in re.finditer(r"virtual\s+([^;{}]*?)\b(\w+)\s*\(([^;{}]*?)\)\s*(const)?\s*=\s*0\s*;", text, flags=re.S):
        params = m.group(3).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        out.append({"name": m.group(2), "nparams": n, "const": bool(m.group(4))})
    return out


if __name__ == "__main__":
    pv = pure_virtuals(open(REF).read())
    json.dump({"source": "include/sqphot/QPsolverInterface.hpp:43-194", "pure_virtuals": pv}, open(OUT, "w"), indent=1)
    print("%d pure virtuals -> %s" % (len(pv), OUT))
