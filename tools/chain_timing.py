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
    eng = make_engine(prob, p=20, lib=_lib.load_dev())
    batch = {k: eng._t(v) for k, v in synth.make_train_batch(prob, B=B, seed=1).items()}
    tbuf = torch.zeros(4096, dtype=torch.int64, device=eng.device)
    eng._check(eng.lib.cadm_dev_set_timing_buffer(eng._ctx, ct.c_void_p(tbuf.data_ptr())))
    eng.train_configure(1e-3, (0.000025, 0.00005, 0.000075, 0.000075, 0.0001), (0.000025, 0.00005, 0.000075), 1.0, 0.5, max_batch=B)
    for _ in range(3):
        eng.train_step(batch, train=True)
    torch.cuda.synchronize()
    raw = tbuf.cpu().numpy()[:768].reshape(3, 256)
    for li, lname in enumerate(["forward", "backward", "backward (context)"]):
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


if __name__ == "__main__":
    main()
