#!/usr/bin/env python3
"""Checkpoint interop: reference `params_epoch_k` joblib lists <-> named .npz archives.

  python tools/ckpt_convert.py inspect  <params_epoch_k>                 architecture + named shapes
  python tools/ckpt_convert.py to-npz   <params_epoch_k> <out.npz>       (+ <params>_norm_stats -> norm_stats/<key>_{mean,std})
  python tools/ckpt_convert.py from-npz <in.npz> <params_epoch_k>        back to the reference's positional layout

`MLPEnsembleCEMDynamicsModel.load` reads the reference layout directly; this tool is for inspecting, editing and re-packing
checkpoints trained elsewhere with the TF1.15 reference (the only route to ever compare the HIP planner with real TF outputs).
"""
import os
import sys
from collections import OrderedDict

import joblib
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadm_amd import checkpoint as ck  # noqa: E402


def main(argv):
    if len(argv) < 2 or argv[0] not in ("inspect", "to-npz", "from-npz"):
        print(__doc__)
        return 2
    cmd = argv[0]
    if cmd in ("inspect", "to-npz"):
        arrays = joblib.load(argv[1])
        info = ck.describe(arrays)
        if cmd == "inspect":
            print({k: v for k, v in info.items() if k != "nets"})
            for k, v in ck.to_named(arrays).items():
                print("  %-40s %s %s" % (k, tuple(v.shape), v.dtype))
            return 0
        out = OrderedDict(ck.to_named(arrays))
        if os.path.exists(argv[1] + "_norm_stats"):
            for key, (mean, std) in joblib.load(argv[1] + "_norm_stats").items():
                out["norm_stats/%s_mean" % key], out["norm_stats/%s_std" % key] = np.asarray(mean), np.asarray(std)
        np.savez(argv[2], **out)
        print("wrote %s (%d arrays)" % (argv[2], len(out)))
        return 0
    z = np.load(argv[1])
    named = OrderedDict((k, z[k]) for k in z.files if not k.startswith("norm_stats/"))
    joblib.dump(ck.from_named(named), argv[2])
    stats = OrderedDict()
    for k in z.files:
        if k.startswith("norm_stats/") and k.endswith("_mean"):
            key = k[len("norm_stats/"):-len("_mean")]
            stats[key] = (z[k], z["norm_stats/%s_std" % key])
    if stats:
        joblib.dump(stats, argv[2] + "_norm_stats")
    print("wrote %s (+ _norm_stats: %s)" % (argv[2], bool(stats)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
