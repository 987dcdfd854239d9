"""`python bench.py --gpus N` must launch its own N ranks when it is not started by torch.distributed.run (the
driver's multi-GPU run is `python bench.py --gpus 8`-shaped on some paths and torch.distributed.run-shaped on others; both
have to end in ONE JSON line from rank 0).  --dry-run swaps the planner for a placeholder and RCCL for gloo so the launcher,
the rank plumbing, the barrier / max-over-ranks timing and the JSON contract are exercised here, on CPU, every round."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n", [1, 2])
def test_bench_self_launches_its_ranks(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout           # rank 0 only
    out = lines[0]
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["data"].startswith("dry-run")
    assert out["config"]["rccl_nranks"] == n and out["config"]["global_candidates"] == 200 * n
    assert abs(out["per_gpu_value"] * n - out["value"]) < 1e-6 * out["value"]
    if n > 1:
        assert out["config"]["allgathers_timed"] == max(3, 60) * 5 and out["config"]["allgather_us"] is not None
        assert set(out["legs"]) == {"cfg5"} and out["legs"]["cfg5"]["rccl_nranks"] == n
        assert "cpu_baseline" not in out          # rank 0 at N = 1 only


def test_bench_under_torch_distributed_run():
    """The driver's own N > 1 form: torch.distributed.run starts the ranks, bench.py must NOT launch again."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-run", "--legs", "none"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["legs"] == {}


def test_gpus_mismatch_is_refused():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
