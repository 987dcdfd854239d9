#!/usr/bin/env python3
"""Micro-benchmark of the fused training step (B=256, E=5, back_coeff=0.5, probabilistic).
Prints ms/step of cadm_train_step with a device-resident batch, and of fit()'s indexed step on a windowed dataset."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from cadm_amd import synth
from cadm_amd.synth import make_engine

WD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)
CWD = (0.000025, 0.00005, 0.000075)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, with_back=True, seed=0)
    eng = make_engine(prob, p=20)
    eng.train_configure(1e-3, WD, CWD, 1.0, 0.5, max_batch=B)
    batch = {k: eng._t(v) for k, v in synth.make_train_batch(prob, B=B, seed=1).items()}
    for _ in range(5):
        eng.train_step(batch, train=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 200
    for _ in range(N):
        eng.train_step(batch, train=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    flops = 3 * 2 * 5 * B * (2 * 134000 + 103040)
    print("train_step B=%d: %.3f ms/step  (%.1f GFLOP/step -> %.1f TFLOP/s)" % (B, dt * 1e3, flops / 1e9, flops / dt / 1e12))
    # fit() flavour: rows of a windowed device dataset addressed through bootstrap indices (cadm_train_step_rows)
    Nw, F = 2000, 10
    r = np.random.default_rng(0)
    dims = {k: v.shape[2] for k, v in batch.items()}
    dev = {k: eng._t(r.standard_normal((Nw, (1 if k.startswith("cp_") else F) * d))) for k, d in dims.items()}
    row_w = torch.arange(Nw, device=eng.device).repeat_interleave(F)
    row_f = torch.arange(F, device=eng.device).repeat(Nw)
    idx = torch.randint(0, Nw * F, (5, 20 * B), device=eng.device)
    for _ in range(5):
        eng.train_step_rows(dev, F, row_w, row_f, idx[:, :B], train=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        j = (i % 20) * B
        eng.train_step_rows(dev, F, row_w, row_f, idx[:, j:j + B], train=True)
    torch.cuda.synchronize()
    print("fit-loop step (rows gathered inside the kernels): %.3f ms/step" % ((time.perf_counter() - t0) / N * 1e3))


if __name__ == "__main__":
    main()
