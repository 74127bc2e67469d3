#!/bin/bash
# One rocprofv3 --pmc pass with SQ occupancy/issue counters for the kernels of bench.py.
# Usage: bash tools/pmc_sq.sh <tag> [ENV=VAL ...]   (PROG="tools/run_filter.py gauss15" profiles that instead)
set -u
tag=$1; shift
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc_${tag}_SQ
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 $root/${PROG:-bench.py --steps 2 --warmup 1 --no-cpu-baseline} > $out.log 2>&1 || { echo "pass failed"; tail -5 $out.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "SQ_WAVES" not in m or m["SQ_WAVES"] == 0: continue
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0   # summed over 8 XCDs
    print(f"{k[:44]:44s} waves={m['SQ_WAVES']:.0f} valu/wave={m['SQ_INSTS_VALU']/m['SQ_WAVES']:.0f} salu/wave={m['SQ_INSTS_SALU']/m['SQ_WAVES']:.0f} "
          f"gui_cycles={gui:.0f} valu_busy={100*m['SQ_ACTIVE_INST_VALU']*4/1024/max(gui,1):.1f}% "
          f"active_any/wave_cyc={m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f} wait_any={m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f} wait_inst={m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}")
PY
