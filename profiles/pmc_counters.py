#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC passes (each pass its own run with --kernel-trace only, MI355X_MICROARCH.md).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 tools/prof_step.py 2
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/sq -- python3 tools/prof_step.py 2
  python profiles/pmc_counters.py $OUT profiles/r02_pmc_counters.json

Output: {"<kernel symbol>|grid=<n>": {"launches": n, "<counter>": mean per launch, ..., "mfma_busy": f}}.
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x elapsed cycles), elapsed cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the
8 XCDs): the fraction of the matrix pipes' cycles that held an MFMA, at the clock the chip actually ran -- unlike
roofline.frac, which prices against the 2.4 GHz peak.  sclk_ghz = elapsed cycles / kernel duration (kernel trace of the same pass)."""
import collections
import csv
import glob
import json
import sys

N_SIMD = 4 * 256


def main(src, dst):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{src}/*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            key = f"{r['Kernel_Name'].split('(')[0].replace('void ', '')}|grid={r['Grid_Size']}"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{src}/*/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)) if "Grid_Size_X" in r else int(r["Grid_Size"])
            key = f"{r['Kernel_Name'].split('(')[0].replace('void ', '')}|grid={gx}"
            dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for key, cs in acc.items():
        e = {"launches": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            e[c] = sum(v) / len(v)
        if key in dur:
            e["duration_us"] = sum(dur[key]) / len(dur[key]) / 1e3
        if "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc)
            if "duration_us" in e:
                e["sclk_ghz"] = cyc / (e["duration_us"] * 1e3)
        out[key] = e
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    big = sorted(((k, v) for k, v in out.items() if "mfma_busy" in v and v.get("duration_us", 0) > 100), key=lambda kv: -kv[1]["duration_us"])
    for k, v in big[:16]:
        print(f"{k[:96]:96s} {v['duration_us']:8.1f} us  mfma_busy {v['mfma_busy']:.3f}  sclk {v.get('sclk_ghz', 0):.2f} GHz")
    print(f"wrote {dst}: {len(out)} (kernel, grid) entries")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
