#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv files over the dispatches of one kernel (name substring).
usage: tools/pmc_extract.py <dir with pmc_*/> <kernel-substring>  ->  JSON {counter: mean per dispatch, "_dispatches": n}"""
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
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
