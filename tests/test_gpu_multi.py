"""Two GPUs, two processes, the in-library RCCL communicator (csrc/dist.hip): cadm_dist_unique_id / init / info, one
ncclAllGather per CEM iteration issued from inside cadm_cem_plan.  The sharded plan of BOTH ranks must equal the unsharded
plan bit for bit (candidates' Philox streams are keyed by the GLOBAL candidate id).  Needs >= 2 visible GPUs: skipped on the
1-GPU boxes of this pool, runs unchanged the moment a box has two (the world-size-2 logic itself is covered
on CPU by tests/test_distributed_cpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from cadm_amd import synth
    prob = synth.make_problem(env="halfcheetah", m=2, H=8, seed=17, trained_like=True)
    eng = synth.make_engine(prob, p=10, H=8, device="cuda:%d" % rank)
    args = (prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], 128)
    plain = eng.cem_plan(*args, seed=3, call=9).cpu().numpy()            # unsharded, before the communicator exists
    rs_plain = eng.rs_plan(prob["obs"], prob["cp_obs"], prob["cp_act"], 128, seed=3, call=10).cpu().numpy()
    eng.dist_init()
    assert eng.dist_info() == (world, rank)
    shard = eng.cem_plan(*args, seed=3, call=9).cpu().numpy()             # 64 candidates per rank + 5 all-gathers
    rs_shard = eng.rs_plan(prob["obs"], prob["cp_obs"], prob["cp_act"], 128, seed=3, call=10).cpu().numpy()
    np.save(os.path.join(out_dir, "plain_%d.npy" % rank), plain)
    np.save(os.path.join(out_dir, "shard_%d.npy" % rank), shard)
    np.save(os.path.join(out_dir, "rs_%d.npy" % rank), np.stack([rs_plain, rs_shard]))
    eng.dist_destroy()
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_gpu_sharded_plan_equals_unsharded(gpu, tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (this box has %d)" % torch.cuda.device_count())
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    plain = np.load(tmp_path / "plain_0.npy")
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / ("plain_%d.npy" % r)), plain)
        np.testing.assert_array_equal(np.load(tmp_path / ("shard_%d.npy" % r)), plain)
        rs = np.load(tmp_path / ("rs_%d.npy" % r))
        np.testing.assert_array_equal(rs[0], rs[1])


# ------------------------------------------------------------------------------------------------------------------------------
# Two LIVE ranks on ONE GPU: the same sharded loop inside cadm_cem_plan / cadm_rs_plan, its collective supplied by the host
# (cadm_dist_init_external + cadm_amd.planner.ExternalAllGather over gloo -- RCCL refuses two ranks on one device).  Everything of the
# sharded path but the ncclAllGather call itself runs here on every box of the pool: shard offsets, per-shard draws keyed by the global
# candidate id, the payload with its checksum word, the regenerating refit, the mismatch flag.
# ------------------------------------------------------------------------------------------------------------------------------
def _worker_one_gpu(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cadm_amd import synth
    from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
    from cadm_amd.envs import make_env_spec
    res = {}
    prob = synth.make_problem(env="halfcheetah", m=2, H=8, seed=17, trained_like=True)
    eng = synth.make_engine(prob, p=10, H=8, device="cuda:0")
    n = 128
    args = (prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], n)
    host_args = tuple(np.asarray(a, np.float32) for a in args[:5])
    res["plain"] = eng.cem_plan(*args, seed=3, call=9).cpu().numpy()                  # unsharded, before any communicator exists
    res["plain_host"] = eng.cem_plan_host(host_args, n, seed=3, call=9)
    res["rs_plain"] = eng.rs_plan(prob["obs"], prob["cp_obs"], prob["cp_act"], n, seed=3, call=10).cpu().numpy()
    nan_obs = host_args[0].copy()
    nan_obs[1, 3] = np.nan
    res["nan_plain"] = eng.cem_plan_host((nan_obs,) + host_args[1:], n, seed=3, call=11)
    eng.dist_init_external(dist.group.WORLD)
    assert eng.dist_info() == (world, rank) and eng.dist_world == world
    res["shard"] = eng.cem_plan(*args, seed=3, call=9).cpu().numpy()                  # 64 candidates per rank + 5 all-gathers over gloo
    assert eng._ext.calls == eng.num_cem_iters and not eng.dist_mismatch()
    res["shard_host"] = eng.cem_plan_host(host_args, n, seed=3, call=9)               # the class's one-call path, completion flags + mismatch words
    res["rs_shard"] = eng.rs_plan(prob["obs"], prob["cp_obs"], prob["cp_act"], n, seed=3, call=10).cpu().numpy()
    # numerically equal inputs with different bit patterns are EQUAL: -0.0 on one rank, +0.0 on the other
    z = host_args[0].copy()
    z[0, 2] = -0.0 if rank == 1 else 0.0
    res["zero"] = eng.cem_plan_host((z,) + host_args[1:], n, seed=3, call=12)
    # a non-finite observation on EVERY rank: handled exactly like the unsharded call -- not an error
    res["nan_shard"] = eng.cem_plan_host((nan_obs,) + host_args[1:], n, seed=3, call=11)
    # different observations on the two ranks: both raise, on the same call
    bad = host_args[0].copy()
    bad[0, 0] += 0.25 * rank
    try:
        eng.cem_plan_host((bad,) + host_args[1:], n, seed=3, call=13)
        res["mismatch_host"] = np.array(0)
    except RuntimeError as exc:
        res["mismatch_host"] = np.array(1 if "different obs" in str(exc) else -1)
    dev_plan = eng.cem_plan(bad, *args[1:], seed=3, call=14).cpu().numpy()
    res["mismatch_dev"] = np.array([int(eng.dist_mismatch()), int(np.isnan(dev_plan).all()), int(eng.dist_mismatch())])      # raised, NaN plan, reset by the read
    res["after"] = eng.cem_plan_host(host_args, n, seed=3, call=9)                    # and the next good call is good again
    eng.dist_destroy()
    eng.close()
    # the class: process_group=... negotiates the collective (gloo: the host-supplied one), get_action runs the same loop
    model = MLPEnsembleCEMDynamicsModel("dyn_model", make_env_spec("halfcheetah"), hidden_nonlinearity="swish", n_forwards=8, n_candidates=n,
                                        ensemble_size=5, n_particles=10, use_cem=True, state_diff=1, normalize_input=False, seed=5,
                                        process_group=dist.group.WORLD, device="cuda:0")
    plans = [model.get_action(prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"]) for _ in range(4)]
    assert model.engine.dist_world == world and model.engine._ext is not None and model.engine._ext.calls == 4 * model.engine.num_cem_iters
    res["class"] = np.stack(plans)
    np.savez(os.path.join(out_dir, "one_gpu_%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_external_allgather(gpu, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker_one_gpu, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / ("one_gpu_%d.npz" % r)) for r in range(2))
    for r in (r0, r1):
        np.testing.assert_array_equal(r["shard"], r0["plain"])             # the sharded plan of BOTH ranks = the unsharded plan, bit for bit
        np.testing.assert_array_equal(r["shard_host"], r0["plain_host"])
        np.testing.assert_array_equal(r["rs_shard"], r0["rs_plain"])
        np.testing.assert_array_equal(r["zero"], r0["zero"])
        assert np.isfinite(r["zero"]).all()
        # a non-finite observation (every row of env 1 returns NaN, its elites are whatever the NaN keys rank as) is NOT an error, and the
        # sharded call does with it exactly what the unsharded call does
        np.testing.assert_array_equal(r["nan_shard"], r["nan_plain"])
        assert int(r["mismatch_host"]) == 1
        assert r["mismatch_dev"].tolist() == [1, 1, 0]
        np.testing.assert_array_equal(r["after"], r0["plain_host"])
    np.testing.assert_array_equal(r0["plain"], r0["plain_host"])
    np.testing.assert_array_equal(r0["class"], r1["class"])
    assert np.isfinite(r0["class"]).all() and np.abs(r0["class"]).max() <= 1.0
