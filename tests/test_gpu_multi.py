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
