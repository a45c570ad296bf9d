"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (tools/pmc_mfma.sh).

    python tools/pmc_mfma_summary.py <bench_dir> <peak_dir> <out.json>

busy = SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel (sums over its dispatches).  The counter's
normalisation (per SIMD / CU / XCD) is not documented for gfx950, so it is calibrated: the same ratio of
tools/mfma_peak.hip's back-to-back 16x16x32 bf16 loop (every SIMD issuing an MFMA every 16 cycles) is
the 100 % point; mfma_util = busy / busy_peak.  Also reported: bf16 MFMA "MOPS" per dispatch and the
wave-cycle split (SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall,
SQ_ACTIVE_INST_ANY = issuing; MI355X_MICROARCH.md, PMC slots).
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short_name


def collect(d):
    out = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        for row in csv.DictReader(open(f)):
            k = out.setdefault(short_name(row["Kernel_Name"]), {"_disp": set()})
            k["_disp"].add(row.get("Dispatch_Id"))
            k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return out


def main():
    bench, peak = collect(sys.argv[1]), collect(sys.argv[2])
    ratios = {k: v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / v["GRBM_GUI_ACTIVE"]
              for k, v in peak.items() if v.get("GRBM_GUI_ACTIVE")}
    busy_peak = max(ratios.values())
    kernels = {}
    for k, v in bench.items():
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        if not gui or not v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            continue
        n = len(v["_disp"])
        wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        kernels[k] = {"dispatches": n, "gui_active_cycles_per_dispatch": round(gui / n),
                      "mfma_busy_ratio": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / gui, 2),
                      "mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / gui / busy_peak, 3),
                      "mfma_mops_bf16_per_dispatch": round(v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / n),
                      "wave_cycles_wait_any": round(v.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                      "wave_cycles_wait_inst": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                      "wave_cycles_active_inst": round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)}
    json.dump({"source": "rocprofv3 --pmc (one pass, tools/pmc_mfma.sh) over `python3 bench.py --steps 1 --warmup 1 "
                         "--no-cpu-baseline --no-kernel-timing`; mfma_util = (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) "
                         "/ the same ratio of tools/mfma_peak.hip (back-to-back MFMAs on every SIMD)",
               "calibration": {"busy_ratio_at_peak": round(busy_peak, 2), "per_kernel": {k: round(r, 2) for k, r in ratios.items()}},
               "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_dispatch"] * kv[1]["dispatches"])[:16]:
        print(f"{k[:64]:64s} n={v['dispatches']:4d} mfma_util {v['mfma_util']:.3f}  wait {v['wave_cycles_wait_any']:.2f} "
              f"stall {v['wave_cycles_wait_inst']:.2f} active {v['wave_cycles_active_inst']:.2f}")


if __name__ == "__main__":
    main()
