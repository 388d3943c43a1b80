#!/usr/bin/env python3
"""Summarise a counter_collection.csv of tools/pmc_valu.sh: per-kernel means, VALU per pixel, VALU busy."""
import collections
import csv
import sys

px = float(sys.argv[2]) if len(sys.argv) > 2 else 192e6
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "ss::" in k:
        agg[k.split("(")[0].replace("void ss::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    line = f"{k}: " + ", ".join(f"{n} {v:.4g}" for n, v in sorted(m.items()))
    if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        line += f" | VALU/px {m['SQ_INSTS_VALU'] / (px / 64):.1f}, VALU busy {m['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / (m['GRBM_GUI_ACTIVE'] / 8):.3f}"
    print(line)
