#!/usr/bin/env python3
"""tests/golden/bench_checksums.json from a bench.py line (the integer-exact outputs' CRCs and byte sums).
Usage: python tools/make_bench_checksums.py <file holding bench.py's JSON line>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
out = {"made_by": "tools/make_bench_checksums.py from a 1-GPU `python bench.py` line on MI355X",
       "c2": {k: d["checksum"][k] for k in ("crc32", "crc32_frame0", "byte_sum")},
       "c3": {k: d["c3"]["checksum"][k] for k in ("crc32_frames", "byte_sum")},
       "c4": {name: {k: v["checksum"][k] for k in ("crc32_bands", "byte_sum")} for name, v in d["c4"]["per_filter"].items()}}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "bench_checksums.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
