#!/usr/bin/env python3
"""Transcribes a StateCheck reference-value file of the reference (test/Ocean/refvals/*.jl: the
`varr` rows [array, field, max?, ...] and the precision rows `parr`) into a JSON fixture: data only,
by a regex over the file's text.  Usage (in the build container, where /root/reference exists):

    python scripts/make_golden_refvals.py /root/reference/test/Ocean/refvals/simple_box_2dt_refvals.jl \
        tests/golden/ocean_simple_box_2dt_refvals.json
"""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
text = open(src, encoding="utf-8").read()
row = re.compile(r'\[\s*"([^"]+)",\s*"([^"]+)",\s*([^\]]+?)\s*\]')


def rows(block):
    out = []
    for a, f, rest in row.findall(block):
        vals = [v.strip() for v in rest.split(",") if v.strip()]
        out.append([a.strip(), f.strip()] + [float(v) if ("." in v or "e" in v.lower()) else int(v) for v in vals])
    return out


varr = text[text.index("varr = ["):text.index("parr = [")]
parr = text[text.index("parr = ["):text.index("# END SCPRINT")]
json.dump({"source": src.replace("/root/reference/", ""),
           "columns": ["array", "field", "min", "max", "mean", "std"],
           "varr": rows(varr), "parr": rows(parr)}, open(dst, "w"), ensure_ascii=False, indent=1)
print(len(rows(varr)), "rows")
