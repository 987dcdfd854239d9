#!/usr/bin/env python3
"""Measured precision of the split-f16 rollout kernel vs the fp32-MFMA comparison kernel vs the fp32 oracle, all against the
fp64 oracle (tests/precision.py): python tools/precision_report.py [out.md] [out.json].  Needs a GPU and the developer library."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import precision


def main():
    eng = {"xdl": precision.product_engine, "f32mfma": precision.f32_engine}
    runs = [precision.measure(eng), precision.measure(eng, cfgname="cfg2", hidden=256, n_traj=12),
            precision.measure(eng, cfgname="cfg4", hidden=200, n_traj=12), precision.measure(eng, cfgname="cfg4", hidden=256, n=200, n_traj=12)]
    md = "\n".join(precision.markdown(r) for r in runs)
    print(md)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(md)
    if len(sys.argv) > 2:
        json.dump(runs, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
