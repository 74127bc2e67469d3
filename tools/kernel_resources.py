#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stdin) as one line per kernel."""
import re
import sys

cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+([^:]+):\s+(\S+)\s+\[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"n": v}
    elif cur is not None:
        cur[k] = v
        if k.startswith("LDS Size"):
            print(f"{cur['n'][:64]:64s} vgpr {cur.get('VGPRs'):>4s} agpr {cur.get('AGPRs'):>3s} "
                  f"sgpr {cur.get('TotalSGPRs'):>3s} scratch {cur.get('ScratchSize [bytes/lane]'):>5s} "
                  f"occ {cur.get('Occupancy [waves/SIMD]')} lds {v}")
