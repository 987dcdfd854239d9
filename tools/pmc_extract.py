#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv files over the dispatches of one kernel (name substring).
usage: tools/pmc_extract.py <dir with pmc_*/> <kernel-substring> [build-id]
  ->  JSON {counter: mean per dispatch, "_dispatches": n, "build_id": cadm_build_id() of the library that was measured}"""
import csv
import glob
import json
import os
import sys


def main():
    root, sub = sys.argv[1], sys.argv[2]
    acc, cnt = {}, {}
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                kn = row.get("Kernel_Name") or row.get("kernel_name") or ""
                if sub not in kn:
                    continue
                name = row.get("Counter_Name") or row.get("counter_name")
                val = float(row.get("Counter_Value") or row.get("counter_value") or 0)
                acc[name] = acc.get(name, 0.0) + val
                cnt[name] = cnt.get(name, 0) + 1
    out = {k: acc[k] / cnt[k] for k in sorted(acc)}
    out["_dispatches"] = max(cnt.values()) if cnt else 0
    if len(sys.argv) > 3:
        out["build_id"] = sys.argv[3]      # bench.py quotes these counters only for the build they were measured on
    # share of SIMD-cycles of a launch in which the matrix pipe was busy: SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs;
    # GRBM_GUI_ACTIVE is the launch's duration in GPU clocks summed over the 8 XCDs (both per dispatch); 4 SIMDs x 256 CUs
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and out.get("GRBM_GUI_ACTIVE"):
        out["launch_cycles"] = out["GRBM_GUI_ACTIVE"] / 8.0
        out["mfma_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * 256.0 * out["launch_cycles"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
