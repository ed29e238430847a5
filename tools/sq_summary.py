"""Summarise a rocprofv3 --pmc SQ_* pass per kernel:  python tools/sq_summary.py <dir> <out.json>

Per kernel name: launches, mean duration, the raw counter sums and the derived fractions
  mfma_busy_pct   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)   (gfx94x MfmaUtil formula; the guide:
                    MFMA_BUSY counts shader cycles, 32 per v_mfma_f32_32x32x16_bf16 = the back-to-back issue rate of a SIMD)
  wait_any / wait_inst / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, quad-cycles)
"""
import collections
import csv
import glob
import json
import os
import sys

CUS, SIMDS, XCDS = 256, 4, 8   # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (checked: gui / 8 / duration = the shader clock)


def main(d, out):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print("no counter_collection.csv under", d)
        return 1
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:64]
            a = agg[k]
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            did = r["Dispatch_Id"]
            if did not in seen[k]:
                seen[k].add(did)
                a["_n"] += 1
                if "Start_Timestamp" in r and r["Start_Timestamp"]:
                    a["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    res = {}
    for k, a in agg.items():
        n = a["_n"]
        gui = a.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        e = {"launches": int(n), "avg_us": a["_ns"] / n / 1e3 if n else None,
             "gui_active_cycles_per_launch": gui / n,
             "mfma_busy_cycles_per_launch": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n,
             "mfma_busy_pct": 100.0 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * CUS * SIMDS) if gui else None,
             "sq_busy_over_gui": a.get("SQ_BUSY_CYCLES", 0.0) / gui if gui else None,
             "wave_cycles_per_launch": wc / n,
             "wait_any_frac": a.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
             "wait_inst_any_frac": a.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
             "wait_inst_lds_frac": a.get("SQ_WAIT_INST_LDS", 0.0) / wc if wc else None,
             "active_inst_frac": a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else None,
             "eff_clock_mhz": gui / a["_ns"] * 1e3 if a["_ns"] else None}
        res[k] = e
    json.dump(res, open(out, "w"), indent=1)
    tot = sum(v["avg_us"] * v["launches"] for v in res.values() if v["avg_us"])
    print(f"{'kernel':64s} {'n':>6s} {'avg us':>8s} {'%time':>6s} {'MFMA%':>6s} {'wait':>5s} {'stall':>5s} {'lds':>5s} {'act':>5s} {'MHz':>5s}")
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["avg_us"] or 0) * kv[1]["launches"])[:32]:
        f = lambda x: f"{x:5.2f}" if x is not None else "  n/a"
        print(f"{k:64s} {v['launches']:6d} {v['avg_us']:8.1f} {100 * v['avg_us'] * v['launches'] / tot:6.1f} "
              f"{(v['mfma_busy_pct'] or 0):6.1f} {f(v['wait_any_frac'])} {f(v['wait_inst_any_frac'])} {f(v['wait_inst_lds_frac'])} {f(v['active_inst_frac'])} {(v['eff_clock_mhz'] or 0):5.0f}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
