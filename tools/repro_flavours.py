#!/usr/bin/env python3
"""One rollout problem through every flavour of the production kernel on a given build (developer tool; a faulting kernel takes the
process down, so run one build per process):  python tools/repro_flavours.py LIB env ctx hid E p m n H det inject"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cadm_amd import _lib, synth

lib, env, ctx, hid, E, p, m, n, H, det, inject = sys.argv[1:12]
ctx, hid, E, p, m, n, H, det, inject = (int(x) for x in (ctx, hid, E, p, m, n, H, det, inject))
prob = synth.make_problem(env=env, context=bool(ctx), E=E, m=m, H=H, hidden_sizes=(hid,) * 4, trained_like=True, seed=1234)
rng = np.random.default_rng(5)
acts = rng.uniform(-1, 1, (m, n, H, prob["A"])).astype(np.float32)
eps_np = rng.standard_normal((H, m, n, p, prob["D"])).astype(np.float32) if inject else None
L = _lib.load_dev(os.path.join(ROOT, "cadm_amd", lib)) if lib != "-" else _lib.load_dev()
eng = synth.make_engine(prob, p=p, deterministic=bool(det), lib=L)
c = eng.context_forward(prob["cp_obs"], prob["cp_act"]) if ctx else None
out = {}
for fl in sys.argv[12:] or ["1", "2", "3", "4", "0"]:
    eng.dev_set_rollout("xdl", row_tiles=int(fl))
    print("flavour", fl, flush=True)
    rows, traj = eng.rollout_returns(prob["obs"], c, eng._t(acts), eps=None if eps_np is None else eng._t(eps_np), want_traj=True, seed=7, call=3)
    torch.cuda.synchronize()
    out[fl] = (rows.cpu().numpy(), traj.cpu().numpy())
    print("   ok", float(np.nanmax(np.abs(out[fl][1]))), flush=True)
ks = list(out)
for k in ks[1:]:
    print(k, "== ", ks[0], all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[ks[0]], out[k])))
if len(ks) > 1 and os.environ.get("CADM_REPRO_DIFF"):
    a, b = out[ks[0]][1], out[ks[1]][1]        # traj [H, m, n, p, D]
    d = np.abs(a - b)
    print("max |diff| per step:", np.nanmax(d.reshape(d.shape[0], -1), axis=1))
    print("max |diff| per dim :", np.nanmax(d.reshape(-1, d.shape[-1]), axis=0))
    print("max |diff| per env :", np.nanmax(d.transpose(1, 0, 2, 3, 4).reshape(d.shape[1], -1), axis=1))
    print("max |diff| per particle:", np.nanmax(d.transpose(3, 0, 1, 2, 4).reshape(d.shape[3], -1), axis=1))
    bad = np.argwhere(d[0].max(-1) > 1e-3)
    print("first bad (m, n, p) at step 0:", bad[:12].tolist(), "count", len(bad), "of", d[0][..., 0].size)
    r = np.abs(out[ks[0]][0] - out[ks[1]][0])
    print("returns max diff", np.nanmax(r))
