#!/usr/bin/env python3
"""Per-stage clock of the training chain kernels (workgroup (0, y, 0)): python tools/chain_timing.py"""
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from cadm_amd import _lib, synth
from cadm_amd.synth import make_engine


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    hid = int(os.environ.get("CHAIN_HID", "200"))
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, with_back=True, seed=0, hidden_sizes=(hid,) * 4)
    eng = make_engine(prob, p=20, lib=_lib.load_dev(os.environ.get("CHAIN_LIB")))      # CHAIN_LIB: a build with -DCADM_DW_TIMING for the dw_adam report
    batch = {k: eng._t(v) for k, v in synth.make_train_batch(prob, B=B, seed=1).items()}
    tbuf = torch.zeros(8192, dtype=torch.int64, device=eng.device)
    eng._check(eng.lib.cadm_dev_set_timing_buffer(eng._ctx, ct.c_void_p(tbuf.data_ptr())))
    eng.train_configure(1e-3, (0.000025, 0.00005, 0.000075, 0.000075, 0.0001), (0.000025, 0.00005, 0.000075), 1.0, 0.5, max_batch=B)
    for _ in range(3):
        eng.train_step(batch, train=True)
    torch.cuda.synchronize()
    raw = tbuf.cpu().numpy()[:768].reshape(3, 256)
    for li, lname in enumerate(["forward", "backward"]):
        row = raw[li, :64]
        n = int((row > 0).sum())
        if n < 2:
            continue
        d = np.diff(row[:n]).astype(np.float64)
        pro = raw[li, 200:203]
        print("   prologue: entry -> loads issued %d, -> committed %d, -> first barrier passed %d ticks" % (pro[1] - pro[0], pro[2] - pro[1], row[0] - pro[2]))
        print("%-20s total %8.0f ticks | " % (lname, row[n - 1] - row[0]) + " ".join("%5.0f" % x for x in d))
        fine = raw[li, 64:64 + 4 * (n - 1)].reshape(n - 1, 4)
        print("   wave 0 per GEMM stage: k loop / epilogue   " + "  ".join("%d/%d" % (r[1] - r[0], r[2] - r[1]) for r in fine if r[0]))
        ep = raw[2, 128 * li:128 * li + 6 * (n - 1)].reshape(n - 1, 6)
        print("   its epilogue: act math (+ z stores) | LDS tile (+ next-group request) | h stores | next group + operands | - | ring prologue")
        for r, f in zip(fine, ep):
            if r[0]:
                print("      %5d %5d %5d %5d %5d %5d" % (f[0] - r[1], f[1] - f[0], f[2] - f[1], f[3] - f[2], f[4] - f[3], r[2] - f[4]))
    dw_report(tbuf)


def dw_report(tbuf):
    """dw_adam_kernel: per job, the workgroups' slab-loop and epilogue times (the launch ends with its slowest workgroup)."""
    raw = tbuf.cpu().numpy()
    se = raw[1024:1024 + 2 * 940].reshape(-1, 2).astype(np.float64)
    jv, mid = raw[4096:4096 + 940], raw[5200:5200 + 940].astype(np.float64)
    ok = se[:, 1] > 0
    if not ok.any():
        return
    jv, se, mid = jv[ok], se[ok], mid[ok]
    t0 = se[:, 0].min()
    us = lambda x: x / 100.0                                  # s_memrealtime: 100 MHz
    for j in sorted(set(int(x) // 2 for x in jv)):
        sel = (jv // 2) == j
        print("   job %2d (%s): %3d workgroups, slab loop median %5.1f us, epilogue median %5.1f us, lifetime median %5.1f max %5.1f us, last end %5.1f us"
              % (j, "16-byte loads" if int(jv[sel][0]) & 1 else "scalar loads / no data", int(sel.sum()), float(np.median(us(mid[sel] - se[sel, 0]))),
                 float(np.median(us(se[sel, 1] - mid[sel]))), float(np.median(us(se[sel, 1] - se[sel, 0]))), float(us(se[sel, 1] - se[sel, 0]).max()),
                 float(us(se[sel, 1] - t0).max())))
    print("dw_adam workgroups: %d; start (us) percentiles 0/25/50/75/100: %s; end: %s; lifetime median %.1f us"
          % (len(se), np.percentile(us(se[:, 0] - t0), [0, 25, 50, 75, 100]).round(1), np.percentile(us(se[:, 1] - t0), [0, 25, 50, 75, 100]).round(1),
             float(np.median(us(se[:, 1] - se[:, 0])))))


if __name__ == "__main__":
    main()
