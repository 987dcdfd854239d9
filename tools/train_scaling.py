#!/usr/bin/env python3
"""Training throughput beyond B = 256: the fused forward / backward / Adam step of the 5-member CaDM ensemble
(+ backward model) at B in {256, 1024, 4096} rows per member, and one `fit` epoch of the drop-in class on a 200 k-row synthetic
dataset (20 000 windows x 10 future steps; batch_size 256, the reference's default, run_cadm_pets.py:125).
  python tools/train_scaling.py [out.json]      (needs a GPU)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
from cadm_amd.envs import make_env_spec

WD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)
CWD = (0.000025, 0.00005, 0.000075)


def fit_epochs(N=20000, F=10, B=256, epochs=3):
    D, A, Hh = 18, 6, 10
    rng = np.random.default_rng(0)
    obs = rng.standard_normal((N, F * D))
    data = dict(obs=obs, act=rng.uniform(-1, 1, (N, F * A)), obs_next=obs + 0.05 * rng.standard_normal((N, F * D)),
                cp_obs=0.1 * rng.standard_normal((N, D * Hh)), cp_act=rng.uniform(-1, 1, (N, A * Hh)), future_bool=np.ones((N, F)))
    model = MLPEnsembleCEMDynamicsModel("dyn", make_env_spec("halfcheetah"), hidden_nonlinearity="swish", batch_size=B, n_forwards=30,
                                        n_candidates=200, ensemble_size=5, n_particles=20, use_cem=True, weight_decays=WD, weight_decay_coeff=1.0,
                                        context_weight_decays=CWD, state_diff=1, back_coeff=0.5, normalize_input=True, valid_split_ratio=0.1,
                                        future_length=F)
    model.fit(epochs=1, **data)                       # upload + first epoch (allocation, packs): untimed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit(epochs=epochs, obs=data["obs"][:0], act=data["act"][:0], obs_next=data["obs_next"][:0], cp_obs=data["cp_obs"][:0],
              cp_act=data["cp_act"][:0], future_bool=data["future_bool"][:0])          # no new samples: `epochs` more epochs on the same dataset
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = int(N * F * 0.9)
    steps = -(-rows // B)
    return {"dataset_rows": N * F, "train_rows_per_member": rows, "members": 5, "batch": B, "epochs_timed": epochs, "s_per_epoch": dt / epochs,
            "member_rows_per_s": 5 * rows * epochs / dt, "ms_per_step_incl_host_loop_and_validation": dt / epochs / steps * 1e3,
            "note": "wall time of MLPEnsembleCEMDynamicsModel.fit (host re-upload of the dataset, normalisation statistics, device epoch "
                    "shuffle, training steps, validation pass, early-stop bookkeeping) / epochs"}


def main():
    out = {"train_step": {}}
    for B in (256, 1024, 4096):
        r = bench.train_step_bench("cuda:0", steps=100 if B <= 1024 else 40, warmup=5, B=B)
        out["train_step"]["B=%d" % B] = r
        print("B=%5d  %.4f ms/step  %.2f M member-rows/s  %.1f TFLOP/s (%.0f %% of the fp32 matrix peak)"
              % (B, r["ms_per_step"], r["rows_per_s"] / 1e6, r["tflops"], 100 * r["tflops"] / bench.FP32_MFMA_PEAK_TFLOPS))
    out["fit"] = fit_epochs()
    print(json.dumps(out["fit"], indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
