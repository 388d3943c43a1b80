#!/usr/bin/env python3
"""Counter passes of one kernel -> the JSON summary bench.py quotes.

  tools/pmc_summary.py OUT.json PX_PER_LAUNCH KERNEL_SUBSTRING SQ.csv FETCH.csv WRITE.csv [MIX.csv]

SQ.csv    : GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 ... (tools/pmc_bench.sh)
FETCH.csv : TCC FETCH_SIZE (kilobytes; gfx950 counts 64-byte units of what are 128-byte requests: doubled here)
WRITE.csv : TCC WRITE_SIZE (kilobytes)
MIX.csv   : SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _INT32, _INT64, _CVT, SQ_INSTS_VALU_FLOPS_FP32: the dynamic instruction
            mix by class, per pixel ("valu_mix"); a packed instruction counts once and gives twice the flops, so
            flops - (add + mul + 2 fma), all per wave, is the number of packed add / mul plus twice the packed fma
The clock is GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration from the same rows' timestamps.
The rows of the library's own kernels are copied beside OUT.json (OUT_sq.csv, OUT_fetch.csv, OUT_write.csv): the
summary names those copies as its sources.
"""
import collections
import csv
import json
import os
import sys


def keep_own_rows(src, dst):
    """Copy the rows of the library's kernels (namespace ss) and drop the framework's multi-kilobyte kernel names."""
    with open(src, newline="") as f, open(dst, "w", newline="") as g:
        rd = csv.DictReader(f)
        wr = csv.DictWriter(g, fieldnames=rd.fieldnames, quoting=csv.QUOTE_NONNUMERIC)
        wr.writeheader()
        for r in rd:
            if "ss::" in r["Kernel_Name"]:
                wr.writerow(r)
    return dst


def means(path, needle):
    vals = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(path)):
        if needle not in r["Kernel_Name"]:
            continue
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not vals:
        raise SystemExit(f"{path}: no dispatch of a kernel named *{needle}*")
    n = len(next(iter(vals.values())))
    return {k: sum(v) / len(v) for k, v in vals.items()}, sum(dur) / len(dur), n


def main():
    out, px, needle, sq_csv, fetch_csv, write_csv = sys.argv[1:7]
    mix_csv = sys.argv[7] if len(sys.argv) > 7 else None
    px = float(px)
    sq, ns, n_sq = means(sq_csv, needle)
    fetch, _, n_f = means(fetch_csv, needle)
    write, _, n_w = means(write_csv, needle)
    clock_ghz = sq["GRBM_GUI_ACTIVE"] / 8.0 / ns
    res = {
        "kernel": needle, "px_per_launch": px, "dispatches": {"sq": n_sq, "fetch": n_f, "write": n_w},
        "kernel_ns_under_counters": round(ns),
        "clock_ghz": round(clock_ghz, 4),
        "valu_per_px": round(sq["SQ_INSTS_VALU"] * 64.0 / px, 2),
        "trans_per_px": round(sq.get("SQ_INSTS_VALU_TRANS_F32", 0.0) * 64.0 / px, 2),
        "salu_per_px": round(sq.get("SQ_INSTS_SALU", 0.0) * 64.0 / px, 2),
        "valu_busy": round(sq["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (sq["GRBM_GUI_ACTIVE"] / 8.0), 4),
        "fetch_bytes_per_px": round(fetch["FETCH_SIZE"] * 1024.0 * 2.0 / px, 3),
        "write_bytes_per_px": round(write["WRITE_SIZE"] * 1024.0 / px, 3),
    }
    res["wait_any_share"] = round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_ANY" in sq else None
    res["wait_issue_share"] = round(sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_INST_ANY" in sq else None
    passes = [(sq_csv, "sq"), (fetch_csv, "fetch"), (write_csv, "write")]
    if mix_csv:
        mix, _, n_m = means(mix_csv, needle)
        per_px = lambda k: mix.get(k, 0.0) * 64.0 / px  # noqa: E731
        m = {"add_f32": per_px("SQ_INSTS_VALU_ADD_F32"), "mul_f32": per_px("SQ_INSTS_VALU_MUL_F32"),
             "fma_f32": per_px("SQ_INSTS_VALU_FMA_F32"), "trans_f32": per_px("SQ_INSTS_VALU_TRANS_F32"),
             "int32": per_px("SQ_INSTS_VALU_INT32"), "int64": per_px("SQ_INSTS_VALU_INT64"), "cvt": per_px("SQ_INSTS_VALU_CVT")}
        m["other"] = res["valu_per_px"] - sum(m.values())  # compares, selects, min / max / med3, moves, lane reads
        if "SQ_INSTS_VALU_FLOPS_FP32" in mix:
            # the counter adds an instruction's flops per lane (1, 2, or twice that when packed) once per wave
            m["flops_f32"] = per_px("SQ_INSTS_VALU_FLOPS_FP32")
            m["packed_extra_flops"] = m["flops_f32"] - (m["add_f32"] + m["mul_f32"] + 2.0 * m["fma_f32"])
        res["valu_mix"] = {k: round(v, 2) for k, v in m.items()}
        res["dispatches"]["mix"] = n_m
        passes.append((mix_csv, "mix"))
    stem = os.path.splitext(out)[0]
    res["sources"] = [os.path.basename(keep_own_rows(src, f"{stem}_{tag}.csv")) for src, tag in passes]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
