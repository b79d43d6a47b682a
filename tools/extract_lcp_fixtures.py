#!/usr/bin/env python3
"""Extract the literal LCP fixtures (A, x, lo, hi, b, fIndex) from the reference's own unit test
unittests/unit/test_LCPUtils.cpp into tests/golden/lcp_fixtures.json.  Build-container only
(reads /root/reference); the JSON is committed.  Only numeric test data is transcribed."""
import json
import os
import re

REF = os.environ.get("NIMBLE_REFERENCE", "/root/reference")
src = open(os.path.join(REF, "unittests/unit/test_LCPUtils.cpp")).read()
out = {}
for m in re.finditer(r"TEST\(LCP_UTILS, (\w+)\)\s*\{(.*?)\n\}\n", src, re.S):
    name, body = m.group(1), m.group(2)
    fx = {}
    for var in ("A", "x", "lo", "hi", "b", "fIndex"):
        mm = re.search(r"\n\s*%s\s*<<\s*(.*?);" % var, body, re.S)
        if not mm:
            continue
        txt = re.sub(r"//.*", "", mm.group(1))
        txt = txt.replace("std::numeric_limits<s_t>::infinity()", "inf")
        try:
            vals = [float(t) for t in re.split(r"[,\s]+", txt.strip()) if t]
        except ValueError:
            continue  # a `std::cout << ...` line, not a matrix literal
        fx[var] = vals
    if "A" in fx and "b" in fx and len(fx["A"]) == len(fx["b"]) ** 2:
        out[name] = fx
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "lcp_fixtures.json")
json.dump(out, open(path, "w"), indent=1)
print({k: len(v["b"]) for k, v in out.items()})
