#!/bin/bash
# VALU-issue roofline beside the HBM one: per kernel, the share of the launch during which the vector ALUs
# were issuing (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles)) -- a kernel at 90 % cannot get more than
# 1.1x faster without executing fewer vector instructions, whatever its HBM fraction says -- and the vector
# instructions per pixel.  One rocprofv3 --pmc pass over tools/run_zoo.py, one --kernel-trace pass for the times.
# Usage (GPU box, repo root): bash tools/valu_roofline.sh <tag>   -> gpurun_out/<tag>_valu_roofline.md
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/${tag}_valu
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d ${out}_pmc -- python3 $root/tools/run_zoo.py > ${out}_pmc.log 2>&1 || { echo "pmc pass failed"; tail -5 ${out}_pmc.log; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -- python3 $root/tools/run_zoo.py > ${out}_stats.log 2>&1 || { echo "stats pass failed"; tail -5 ${out}_stats.log; exit 1; }
python3 - ${out}_pmc ${out}_stats > $root/gpurun_out/${tag}_valu_roofline.md <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
times = {}
for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        times[row["Name"]] = float(row["MinNs"]) * 1e-6          # ms (min: the first launches are cold)
px = 16 * 3840 * 2160
print("| kernel | ms (16 x 3840x2160) | HBM roofline (8 B/px / 8 TB/s) | VALU busy | time at 100 % VALU issue (ms) | vector instructions per pixel | waiting share of wave cycles |")
print("|---|---|---|---|---|---|---|")
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if not m.get("SQ_WAVES") or "mi::" not in k or "synth" in k:
        continue
    gui = m["GRBM_GUI_ACTIVE"] / 8.0
    busy = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / max(gui, 1)
    ms = times.get(k)
    name = re.sub(r"\(.*", "", k).replace("void mi::", "")
    print(f"| `{name}` | {ms:.3f} | {100 * px * 8 / (ms * 1e-3) / 8e12:.0f} % | {100 * busy:.0f} % | {ms * busy:.3f} | "
          f"{m['SQ_INSTS_VALU'] * 64 / px:.0f} | {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.0f} % |")
PY
cat $root/gpurun_out/${tag}_valu_roofline.md
