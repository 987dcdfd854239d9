#!/usr/bin/env python3
"""Per-phase cycle breakdown of the rollout kernel (workgroup 0) using the CADM_PHASE_TIMING build.
   make -C cadm_amd/csrc timing && python tools/phase_timing.py [cfg] [f32]      (loads cadm_amd/libcadm_hip_timing.so)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as ct

import numpy as np
import torch

from cadm_amd import _lib, synth
from cadm_amd.synth import make_engine

XDL_NAMES = ["state+noise", "bar", "L0 rest", "bar", "hidden 1 rest", "hidden bars", "head rest", "bar", "hidden 2 rest", "hidden 3+ rest",
             "sweeps: to chunk 1", "sweeps: chunks", "sweeps: last epilogue"]
NAMES = ["assembly", "bar0", "L0 mfma", "L0 epi", "bar1", "hid rebuild", "hid mfma", "hid epi", "hid bar",
         "out noise", "out rebuild", "out mfma", "out part wr", "bar5", "head epi", "bar6"]


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    cfg = synth.CONFIGS[cfgname]
    prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=cfg["E"], m=1, H=cfg["H"], seed=0)
    f32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
    libname = sys.argv[3] if len(sys.argv) > 3 else "libcadm_hip_timing.so"      # (a CADM_PHASE_TIMING build of an experiment variant)
    eng = make_engine(prob, p=cfg["p"], deterministic=cfg["deterministic"], lib=_lib.load_dev(os.path.join(ROOT, "cadm_amd", libname)))
    if f32:
        eng.dev_set_rollout("f32")
    NW = 4 if f32 else 8
    tbuf = torch.zeros(NW * 24, dtype=torch.int64, device=eng.device)
    eng._check(eng.lib.cadm_dev_set_timing_buffer(eng._ctx, ct.c_void_p(tbuf.data_ptr())))
    n = cfg["n"]
    mean, var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"]) if cfg["context"] else None
    acts = eng.sample_actions(mean, var, n, seed=1, call=1, it=0)
    for _ in range(3):
        eng.rollout_returns(prob["obs"], ctx, acts, seed=1, call=1)
    torch.cuda.synchronize()
    names = NAMES if f32 else XDL_NAMES
    t = tbuf.cpu().numpy().reshape(NW, 24)[:, :len(names)].astype(np.float64) / cfg["H"]
    print("cycles per step (s_memtime ticks), workgroup 0, per wave:")
    fmt = "%-16s" + " %8s" * NW
    print(fmt % ("phase", *["wave%d" % w for w in range(NW)]))
    fmt = "%-16s" + " %8.0f" * NW
    for i, nm in enumerate(names):
        print(fmt % (nm, *t[:, i]))
    print(fmt % ("TOTAL", *t.sum(1)))


if __name__ == "__main__":
    main()
