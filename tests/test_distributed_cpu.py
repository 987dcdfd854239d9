"""CPU, world_size 2, gloo: the candidate-sharded planner (one all-gather of per-candidate returns per
CEM iteration) equals the unsharded planner exactly; exercised through cadm_amd.planner with the
oracle-backed engine stand-in (the HIP engine needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cadm_amd import planner as hplanner
    from cadm_amd import synth
    from helpers import oracle_problem
    from oracle_engine import OracleEngine
    torch.set_num_threads(1)
    E, p, m, n, H = 5, 5, 2, 64, 4
    prob = synth.make_problem(env="halfcheetah", E=E, m=m, H=H, seed=7)
    eng = OracleEngine(oracle_problem(prob, np.float32), prob, p, H, num_elites=16)
    shard = hplanner.Shard.from_group(n, dist.group.WORLD)
    assert (shard.world, shard.rank, shard.n_local, shard.offset) == (world, rank, n // world, rank * (n // world))
    plan = hplanner.cem_plan(eng, prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], n,
                             seed=11, call=3, shard=shard)
    first, cand = hplanner.rs_plan(eng, prob["obs"], prob["cp_obs"], prob["cp_act"], n, seed=11, call=4, shard=shard)
    np.save(os.path.join(out_dir, "plan_%d.npy" % rank), plan.numpy())
    np.save(os.path.join(out_dir, "rs_%d.npy" % rank), first.numpy())
    np.save(os.path.join(out_dir, "cand_%d.npy" % rank), cand.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_cem_equals_single_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # single-rank reference in this process
    sys.path.insert(0, HERE)
    from cadm_amd import planner as hplanner
    from cadm_amd import synth
    from helpers import oracle_problem
    from oracle_engine import OracleEngine
    E, p, m, n, H = 5, 5, 2, 64, 4
    prob = synth.make_problem(env="halfcheetah", E=E, m=m, H=H, seed=7)
    eng = OracleEngine(oracle_problem(prob, np.float32), prob, p, H, num_elites=16)
    ref = hplanner.cem_plan(eng, prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], n,
                            seed=11, call=3).numpy()
    rs_ref, cand_ref = hplanner.rs_plan(eng, prob["obs"], prob["cp_obs"], prob["cp_act"], n, seed=11, call=4)
    p0, p1 = np.load(tmp_path / "plan_0.npy"), np.load(tmp_path / "plan_1.npy")
    np.testing.assert_array_equal(p0, p1)            # every rank ends with the identical plan
    np.testing.assert_array_equal(p0, ref)           # and it equals the unsharded planner bit for bit
    np.testing.assert_array_equal(np.load(tmp_path / "rs_0.npy"), rs_ref.numpy())
    np.testing.assert_array_equal(np.load(tmp_path / "rs_1.npy"), rs_ref.numpy())
    c0 = np.load(tmp_path / "cand_0.npy")            # gathered layout [G, m, n_local]
    assert c0.shape == (2, m, n // 2)
    np.testing.assert_array_equal(np.concatenate([c0[0], c0[1]], axis=1), cand_ref.numpy()[0])


def test_shard_validation():
    from cadm_amd.planner import Shard
    with pytest.raises(ValueError):
        Shard(10, 0, 4)
    s = Shard(12, 2, 4)
    assert s.n_local == 3 and s.offset == 6


def _repl_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cadm_amd import planner as hplanner
    shard = hplanner.Shard.from_group(8, dist.group.WORLD)
    res = []
    same = [torch.arange(6.0).reshape(2, 3), torch.tensor([1.0, float("nan"), 3.0])]      # NaN at the same place on every rank: equal
    cases = {"same": same,
             "differs": [same[0] + (0.5 if rank == 1 else 0.0), same[1]],
             "permuted": [same[0].flip(0) if rank == 1 else same[0], same[1]],              # same plain sum, different order
             "nan_moved": [same[0], torch.tensor([1.0, 3.0, float("nan")]) if rank == 1 else same[1]]}
    for name, ts in cases.items():
        try:
            hplanner.check_replicated(ts, shard)
            res.append((name, "ok"))
        except RuntimeError:
            res.append((name, "raised"))
    np.save(os.path.join(out_dir, "repl_%d.npy" % rank), np.array(res))
    dist.barrier()
    dist.destroy_process_group()


def test_check_replicated_is_nan_aware_and_order_sensitive(tmp_path):
    """the replicated-input guard of sharded planning -- ONE all-reduce, NaN-aware (a NaN observation on every rank is
    'equal'), sensitive to permutations (a plain sum is not)."""
    mp.spawn(_repl_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = dict(np.load(tmp_path / ("repl_%d.npy" % r)))
        assert got == {"same": "ok", "differs": "raised", "permuted": "raised", "nan_moved": "raised"}, got


def test_replication_check_schedule():
    """The sharded get_action's guard: the first `check_replicated_calls` calls AND every `check_replicated_every`-th
    one after them verify the replicated inputs; 0 switches the periodic part off.  (The method only reads three counters, so it
    is exercised here on a stand-in object: the class itself needs a GPU.)"""
    import types
    from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel as M
    me = types.SimpleNamespace(_check_replicated_left=2, _check_replicated_every=5, _sharded_calls=0)
    due = []
    for call in range(1, 13):
        d = M._replication_check_due(me)
        assert M._replication_check_due(me, peek=True) == d        # asking again for the same call neither counts nor changes the answer
        if d and me._check_replicated_left > 0:
            me._check_replicated_left -= 1                         # (what get_action does when it runs the check)
        due.append(d)
    assert due == [True, True, False, False, True, False, False, False, False, True, False, False]
    off = types.SimpleNamespace(_check_replicated_left=0, _check_replicated_every=0, _sharded_calls=0)
    assert not any(M._replication_check_due(off) for _ in range(600))


def test_replication_guard_fires_on_the_unfused_sharded_path(monkeypatch):
    """with the torch.distributed fallback (no in-library communicator: `fused` False) the fast-signature branch of
    get_action used to mark the call as counted without counting it, so `_sharded_calls` stayed at 1 and the periodic check never
    fired -- in exactly the degraded mode where ranks are most likely to drift apart.  get_action on a stand-in object (the class
    itself needs a GPU): the check must run on calls 1, 2 (check_replicated_calls) and then on every 4th."""
    import types
    import torch
    from cadm_amd.dynamics import mlp_cadm_ensemble_cem_dynamics as mod
    M = mod.MLPEnsembleCEMDynamicsModel
    checked, planned = [], []
    monkeypatch.setattr(mod._planner, "check_replicated", lambda tensors, shard: checked.append(len(planned) + 1))
    monkeypatch.setattr(mod._planner, "cem_plan", lambda eng, *a, **k: (planned.append(1), torch.zeros(1, 3, 2))[1])
    eng = types.SimpleNamespace(stage=lambda xs: tuple(None if x is None else torch.as_tensor(x) for x in xs), _t=lambda x: x)
    me = types.SimpleNamespace(_stats_dirty=False, _checked_sig=None, _call=0, _check_replicated_left=2, _check_replicated_every=4,
                               _sharded_calls=0, engine=eng, n_candidates=8, seed=0, discrete=False, n_forwards=3, action_space_dims=2,
                               _sharding=lambda: (types.SimpleNamespace(world=2), False))
    me._next_call = lambda: M._next_call(me)
    me._replication_check_due = lambda peek=False: M._replication_check_due(me, peek=peek)
    me._check_planner_inputs = lambda *a: None
    obs, mean, var = np.zeros((1, 4), np.float32), np.zeros((1, 3, 2), np.float32), np.ones((1, 3, 2), np.float32)
    for _ in range(12):
        M.get_action(me, obs, None, None, mean, var)
    assert len(planned) == 12 and me._sharded_calls == 12
    assert checked == [1, 2, 4, 8, 12], checked
