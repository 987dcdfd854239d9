"""CPU, world_size 2, gloo: the host side of candidate-sharded planning.  The sharded planner is ONE implementation -- the library's
loop (`cadm_cem_plan`), on every rank -- whose collective is a plug; what runs without a GPU is that plug
(`cadm_amd.planner.ExternalAllGather`, driven through its C function pointer exactly as the library drives it), the shard arithmetic
and the replicated-input guard.  The loop itself runs with two LIVE ranks in tests/test_gpu_multi.py (two processes on one GPU over
gloo, and two GPUs over RCCL where a box has them)."""
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
    """The host-supplied all-gather of the sharded planner (cadm_amd.planner.ExternalAllGather), called the way the library calls it:
    through its C function pointer, with raw pointers into a "workspace" -- here host memory, so the SAME product code runs under gloo
    without a GPU.  The payload is the planner's: m * n/G candidate means + one checksum word per rank."""
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as ct
    from cadm_amd import _lib
    from cadm_amd import planner as hplanner
    torch.set_num_threads(1)
    m, n = 2, 64
    shard = hplanner.Shard.from_group(n, dist.group.WORLD)
    assert (shard.world, shard.rank, shard.n_local, shard.offset) == (world, rank, n // world, rank * (n // world))
    count = m * shard.n_local + 1
    ws = torch.zeros(4096, dtype=torch.uint8)                       # the "workspace": send at byte 256, recv at byte 1024
    base = ws.data_ptr()

    def resolve(ptr, nbytes):
        off = int(ptr) - base
        assert 0 <= off and off + nbytes <= ws.numel()
        return ws[off:off + nbytes].view(torch.float32)
    ext = hplanner.ExternalAllGather(dist.group.WORLD, resolve)
    assert (ext.world, ext.rank, ext.backend) == (world, rank, "gloo")
    fn = ct.cast(ext.cfunc, _lib.ALLGATHER_FN)                       # the pointer the library would hold
    send = resolve(base + 256, 4 * count)
    for it in range(3):                                              # one call per CEM iteration
        send[:count - 1] = torch.arange(count - 1, dtype=torch.float32) + 1000.0 * rank + 0.5 * it
        send[count - 1:] = torch.tensor([0x5EED0001 + it], dtype=torch.int32).view(torch.float32)      # the checksum word: a bit pattern
        assert fn(None, base + 256, base + 1024, count, None) == 0
        np.save(os.path.join(out_dir, "gath_%d_%d.npy" % (rank, it)), resolve(base + 1024, 4 * count * world).view(torch.int32).numpy().copy())
    assert ext.calls == 3 and ext.error is None
    # an exception inside the callback never crosses the C frames: a non-zero code, the exception kept for HipEngine._check to re-raise
    assert fn(None, base + 4000, base + 1024, count, None) == 1
    assert isinstance(ext.error, AssertionError)
    dist.barrier()
    dist.destroy_process_group()


def test_external_allgather_world_size_2_gloo(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    m, n, world = 2, 64, 2
    count = m * (n // world) + 1
    for it in range(3):
        g0, g1 = (np.load(tmp_path / ("gath_%d_%d.npy" % (r, it))) for r in range(2))
        np.testing.assert_array_equal(g0, g1)                        # every rank holds the same gathered buffer
        g = g0.view(np.float32).reshape(world, count)                # rank-major: [G][m * n_local + 1]
        for r in range(world):
            np.testing.assert_array_equal(g[r, :-1], np.arange(count - 1, dtype=np.float32) + 1000.0 * r + 0.5 * it)
            assert g0.reshape(world, count)[r, -1] == 0x5EED0001 + it       # the checksum word survives bit for bit


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


def test_replication_guard_fires_on_the_sharded_path(monkeypatch):
    """the fast-signature branch of get_action used to mark a sharded call as counted without counting it, so
    `_sharded_calls` stayed at 1 and the periodic check never fired.  get_action on a stand-in object (the class itself needs a GPU),
    sharded over 2 ranks: the check must run on calls 1, 2 (check_replicated_calls) and then on every 4th, and every call is planned
    by the engine's one planner (`cem_plan` when the check has staged the inputs, `cem_plan_host` otherwise)."""
    import types
    import torch
    from cadm_amd.dynamics import mlp_cadm_ensemble_cem_dynamics as mod
    M = mod.MLPEnsembleCEMDynamicsModel
    checked, planned = [], []
    monkeypatch.setattr(mod._planner, "check_replicated", lambda tensors, shard: checked.append(len(planned) + 1))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: types.SimpleNamespace(synchronize=lambda: None))

    def cem_plan(*a, out=None, **k):
        planned.append("device")
        out.zero_()
    host = torch.zeros(1, 3, 2)
    eng = types.SimpleNamespace(stage=lambda xs: tuple(None if x is None else torch.as_tensor(x) for x in xs), _t=lambda x: x, device="cpu",
                                cem_plan=cem_plan, host_out=lambda shape: host, dist_mismatch=lambda: False, MISMATCH_MSG="mismatch",
                                cem_plan_host=lambda *a, **k: (planned.append("host"), np.zeros((1, 3, 2), np.float32))[1])
    me = types.SimpleNamespace(_stats_dirty=False, _checked_sig=None, _call=0, _check_replicated_left=2, _check_replicated_every=4,
                               _sharded_calls=0, engine=eng, n_candidates=8, seed=0, discrete=False, n_forwards=3, action_space_dims=2,
                               _sharding=lambda: (types.SimpleNamespace(world=2), True))
    me._next_call = lambda: M._next_call(me)
    me._replication_check_due = lambda peek=False: M._replication_check_due(me, peek=peek)
    me._check_planner_inputs = lambda *a: None
    obs, mean, var = np.zeros((1, 4), np.float32), np.zeros((1, 3, 2), np.float32), np.ones((1, 3, 2), np.float32)
    for _ in range(12):
        M.get_action(me, obs, None, None, mean, var)
    assert len(planned) == 12 and me._sharded_calls == 12
    assert checked == [1, 2, 4, 8, 12], checked
    assert [i + 1 for i, k in enumerate(planned) if k == "device"] == checked      # a checked call stages its inputs; the others are one host call
