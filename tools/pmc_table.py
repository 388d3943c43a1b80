#!/usr/bin/env python3
"""Per-kernel summary of the passes tools/pmc_kernels.sh collected:

  tools/pmc_table.py gpurun_out/TAG PX_PER_LAUNCH [ALGORITHMIC_BYTES_PER_PX or -] [OUT.json]

For every kernel of the library (namespace ss) in the passes: launches, average duration, clock, VALU instructions per
pixel, VALU busy, the share of wave time spent waiting (s_waitcnt / barriers, and for an issue slot), LDS bank-conflict
cycles against LDS active cycles, vector memory instructions per pixel, and HBM bytes per pixel (FETCH_SIZE in
kilobytes, doubled: gfx950 counts 64-byte units of 128-byte requests; WRITE_SIZE in kilobytes)."""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name).replace("void ss::", "").replace("ss::", "")
    return name


def means(path):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    if not os.path.exists(path):
        return {}, {}
    for r in csv.DictReader(open(path)):
        if "ss::" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return ({k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in vals.items()},
            {k: (sum(d.values()) / len(d), len(d)) for k, d in dur.items()})


def main():
    root, px = sys.argv[1], float(sys.argv[2])
    algo = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] not in ("-", "") else None
    sq, dur = means(os.path.join(root, "sq", "a_counter_collection.csv"))
    lds, _ = means(os.path.join(root, "lds", "a_counter_collection.csv"))
    fetch, _ = means(os.path.join(root, "fetch", "a_counter_collection.csv"))
    write, _ = means(os.path.join(root, "write", "a_counter_collection.csv"))
    out = {}
    for k, c in sq.items():
        ns, n = dur[k]
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        e = {"launches_per_pass": n, "avg_us_under_counters": round(ns / 1e3, 1), "clock_ghz": round(cyc / ns, 3),
             "valu_per_px": round(c["SQ_INSTS_VALU"] * 64 / px, 2), "trans_per_px": round(c["SQ_INSTS_VALU_TRANS_F32"] * 64 / px, 2),
             "salu_per_px": round(c["SQ_INSTS_SALU"] * 64 / px, 2),
             "valu_busy": round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 3),
             "wait_any_share": round(c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3),
             "wait_issue_share": round(c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3)}
        if k in lds:
            m = lds[k]
            e.update({"lds_instr_per_px": round(m["SQ_INSTS_LDS"] * 64 / px, 2),
                      "lds_bank_conflict_share": round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), 3),
                      "lds_busy": round(m["SQ_ACTIVE_INST_LDS"] * 4 / 1024 / cyc, 3),
                      "vmem_rd_per_px": round(m["SQ_INSTS_VMEM_RD"] * 64 / px, 3), "vmem_wr_per_px": round(m["SQ_INSTS_VMEM_WR"] * 64 / px, 3),
                      "vmem_busy": round(m["SQ_ACTIVE_INST_VMEM"] * 4 / 1024 / cyc, 3)})
        if k in fetch and k in write:
            f = fetch[k]["FETCH_SIZE"] * 1024 * 2 / px
            w = write[k]["WRITE_SIZE"] * 1024 / px
            e.update({"fetch_bytes_per_px": round(f, 2), "write_bytes_per_px": round(w, 2)})
            if algo:
                e["traffic_over_algorithmic"] = round((f + w) / algo, 3)
            e["hbm_gbs_under_counters"] = round((f + w) * px / ns, 0)
        out[k] = e
    for k, e in out.items():
        print(k)
        print("   " + ", ".join(f"{a} {b}" for a, b in e.items()))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
