#!/usr/bin/env python3
"""Copies one round's profile evidence from gpurun_out/ (tools/profile_round.sh, tools/valu_roofline.sh,
tools/bench_all.py) into profiles/ under the round's tag, and refreshes the two JSON files bench.py reads.
Usage: python tools/collect_profiles.py r03"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = {f"{tag}_kernel_stats.csv": f"{tag}_bench_kernel_stats.csv",
         f"{tag}_bench_line_profiled.json": f"{tag}_bench_line_profiled.json",
         f"{tag}_legs_kernel_stats.csv": f"{tag}_bench_legs_kernel_stats.csv",
         f"{tag}_bench_legs_line_profiled.json": f"{tag}_bench_legs_line_profiled.json",
         f"{tag}_traffic.txt": f"{tag}_pmc_traffic_bench.txt",
         f"{tag}_fft_kernel_stats.csv": f"{tag}_fft_kernel_stats.csv",
         f"{tag}_fft_traffic.txt": f"{tag}_fft_pmc_traffic.txt",
         f"{tag}_valu_roofline.md": f"{tag}_valu_roofline.md",
         f"{tag}_all_kernels.md": f"{tag}_all_kernels.md"}
for src, dst in pairs.items():
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("missing", src)
t = os.path.join(G, f"{tag}_traffic.json")
if os.path.exists(t):
    shutil.copyfile(t, os.path.join(P, "traffic_bytes_per_launch.json"))
v = os.path.join(G, f"{tag}_valu_roofline.md")
if os.path.exists(v):
    kernels = {}
    for line in open(v):
        m = re.match(r"\| `([^`]+)` \| ([\d.]+) \| \d+ % \| (\d+) % \| ([\d.]+) \| (\d+) \|", line)
        if not m:
            continue
        name = m.group(1)
        key = {"chain_kernel<1, 2>": "chain_kernel", "gauss_sep_kernel<2, 1>": "gauss_sep_kernel<2>", "gauss_wide_kernel<7, 1>": "gauss_wide_kernel<7>", "gauss_mfma_kernel<0>": "gauss_mfma_kernel", "gauss_mfma_kernel<1>": "gauss_mfma_kernel<1>", "gauss_mfma_kernel<0, 8>": "gauss_mfma_kernel", "gauss_mfma_kernel<1, 8>": "gauss_mfma_kernel<1>", "gauss_mfma_wide_kernel<0>": "gauss_mfma_wide_kernel<0>", "gauss_mfma_wide_kernel<1>": "gauss_mfma_wide_kernel<1>",
               "gray_kernel<3, 1>": "gray_kernel", "median3_kernel<1>": "median3_kernel", "median5_kernel<1>": "median5_kernel",
               "sobel_kernel<1>": "sobel_kernel"}.get(name)
        if key:
            kernels[key] = {"valu_busy": int(m.group(3)) / 100.0, "ms_at_full_issue": float(m.group(4)), "vector_instructions_per_pixel": int(m.group(5))}
    json.dump({"source": f"profiles/{tag}_valu_roofline.md (rocprofv3 --pmc SQ_ACTIVE_INST_VALU ... over tools/run_zoo.py, 16 x 3840x2160)",
               "kernels": kernels}, open(os.path.join(P, "valu_busy.json"), "w"), indent=1, sort_keys=True)
    print("valu_busy.json:", sorted(kernels))
