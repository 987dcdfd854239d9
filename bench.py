"""bench.py -- CEM rollout row-steps/s of the PE-TS+CaDM planner on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE get_action on BASELINE.json's configs[1] (halfcheetah PE-TS+CaDM, ens=5,
part=20, cand=200 per GPU, H=30, m=1): context encoder + 5 CEM iterations x (sample, 30-step
fused rollout of cand*part rows, refit).  Inputs (obs, history, weights, init mean/var) are
resident in HBM before the timed region.  Metric: row-steps/s = m*n*p*H*5 / wall(get_action)
(a row = one (candidate, particle) pair evaluated by exactly one member).
N > 1 is weak scaling: 200 candidates per GPU, one RCCL all-gather of per-candidate returns per
CEM iteration.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


def flops_per_row_step(K0, hid, D, n_hidden):
    """Algorithmic FLOPs of one row-step: the six matmuls only."""
    return 2 * (K0 * hid + (n_hidden - 1) * hid * hid + hid * 2 * D)


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the host's cores, which oversubscribes a quota-limited container)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per)))
        except Exception:
            pass
    return n


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(prob, n, p, budget_s=15.0):
    """Op-for-op torch-CPU restatement of the TF1.15 graph on this node's host cores (baseline only)."""
    from oracle import torch_baseline as tb
    cores = min(usable_cores(), 64)   # [5,800,200] batched matmuls stop scaling long before 64 threads
    tp = tb.prepare(prob)
    torch.manual_seed(0)
    rs = prob["m"] * n * p * prob["H"] * 5
    torch.set_num_threads(1)
    t0 = time.time()
    tb.cem_get_action(tp, n, p)               # doubles as warm-up; single-thread figure
    single = time.time() - t0
    torch.set_num_threads(cores)
    tb.cem_get_action(tp, n, p)
    times = []
    t_start = time.time()
    while len(times) < 30 and (time.time() - t_start) < budget_s:
        t0 = time.time()
        tb.cem_get_action(tp, n, p)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return dict(value=rs / med, unit="row-steps/s", cores=cores, kind="port",
                single_thread_value=rs / single, cpu_model=_cpu_model(),
                sample="%d full %s get_action calls (m=1,n=%d,p=%d,H=%d, 5 CEM iters = %d row-steps each), median %.3f s "
                       "on %d threads (+1 single-thread call: %.2f s); torch-CPU fp32 restatement of the TF1.15 graph with "
                       "materialised tile/transpose/reshape, torch %s" % (len(times), prob["env"], n, p, prob["H"], rs, med, cores,
                                                                          single, torch.__version__))


def train_step_bench(device, steps=100, warmup=10, B=256):
    """Second leg of the path: one fused forward/backward/Adam step of the 5-member CaDM ensemble
    (+ backward model) on a device-resident [E, B, .] bootstrap batch.  Reported beside the headline, never as it."""
    from cadm_amd import synth
    from helpers import make_engine
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, with_back=True, seed=0)
    eng = make_engine(prob, p=20, device=device)
    eng.train_configure(1e-3, (0.000025, 0.00005, 0.000075, 0.000075, 0.0001), (0.000025, 0.00005, 0.000075), 1.0, 0.5,
                        max_batch=B)
    batch = {k: eng._t(v) for k, v in synth.make_train_batch(prob, B=B, seed=1).items()}
    for _ in range(warmup):
        eng.train_step(batch, train=True)
    torch.cuda.synchronize(eng.device)
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = eng.train_step(batch, train=True)
    torch.cuda.synchronize(eng.device)
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(losses).all()
    flops = 3 * 2 * 5 * B * (2 * 134000 + 103040)      # fwd + dX + dW of the ff, backward and context nets
    return {"ms_per_step": dt * 1e3, "batch": B, "members": 5, "rows_per_s": 5 * B / dt, "tflops": flops / dt / 1e12,
            "workload": "halfcheetah CaDM ensemble + backward model, fwd/bwd/TF1-Adam, fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", help="cfg2 | cfg3 | cfg4 (BASELINE.json configs)")
    ap.add_argument("--cand-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from cadm_amd import planner as hplanner
    from cadm_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = dict(synth.CONFIGS[args.config])
    n_per_gpu = args.cand_per_gpu or cfg["n"]
    n = n_per_gpu * world
    p, E, H = cfg["p"], cfg["E"], cfg["H"]
    prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=E, m=1, H=H, seed=0)
    eng = make_engine(prob, p=p, deterministic=cfg["deterministic"], device="cuda:%d" % local_rank)
    dev = eng.device
    obs, cp_obs, cp_act = eng._t(prob["obs"]), eng._t(prob["cp_obs"]), eng._t(prob["cp_act"])
    init_mean, init_var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
    shard = hplanner.Shard(n, rank, world)
    collective = "none"
    if world > 1:
        try:        # RCCL communicator inside libcadm_hip.so: one ncclAllGather per CEM iteration, all on-stream
            eng.dist_init()
            collective = "rccl all-gather in libcadm_hip.so"
        except Exception as exc:
            collective = "torch.distributed all_gather (in-library RCCL init failed: %s)" % exc
        flag = torch.tensor([1.0 if eng.dist_world == world else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # all ranks must agree on the path
        if flag.item() < 1.0 and eng.dist_world == world:
            collective = "torch.distributed all_gather (a peer failed in-library RCCL init)"
            eng.dist_world = 1
    fused = world == 1 or eng.dist_world == world

    def step(call):
        if fused:
            return eng.cem_plan(obs, cp_obs, cp_act, init_mean, init_var, n, seed=0, call=call)
        return hplanner.cem_plan(eng, obs, cp_obs, cp_act, init_mean, init_var, n, seed=0, call=call, shard=shard)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for w in range(args.warmup):
        step(w)
    eng.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        plan = step(args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_launches = eng.profile_read()
    eng.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(plan).all()

    row_steps_per_call = 1 * n * p * H * eng.num_cem_iters
    value = row_steps_per_call * args.steps / elapsed
    fl = flops_per_row_step(prob["K0"], 200, prob["D"], 4)
    # dominant kernel: rollout_kernel, one launch = n_local*p rows x H steps
    rows_per_launch = shard.n_local * p * H
    kern_avg_s = kern_ms / 1e3 / max(kern_launches, 1)
    achieved = rows_per_launch * fl / kern_avg_s / 1e12
    # HBM/fabric traffic of the dominant kernel: PMC counters cannot be read from inside this process; they are
    # collected by separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of THIS command and committed
    # under profiles/ (KB per launch).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide
    # coalesced reads at half their bytes -> doubled.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.config)
    if world == 1 and args.cand_per_gpu is None and os.path.exists(tpath):
        raw = json.load(open(tpath))
        traffic = (2.0 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"]) * 1024.0
    out = {
        "metric": "CEM rollout row-steps/s (cand x part x horizon x 5 CEM iters per get_action; ens=%d members)" % E,
        "value": value, "unit": "row-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "get_action_latency_ms": elapsed / args.steps * 1e3,
        "plans_per_s": args.steps / elapsed, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %s PE-TS+CaDM get_action, ens=%d part=%d cand=%d (%d/GPU) H=%d m=1, random-init weights"
                               % (args.config, cfg["env"], E, p, n, n_per_gpu, H),
                   "global_candidates": n, "parallelism": "candidate-shard x%d" % world, "collective": collective},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_unit": "bytes/launch (2*FETCH_SIZE + WRITE_SIZE from profiles/pmc_traffic_%s.json)" % args.config,
                     "kernel": "rollout_kernel", "avg_launch_ms": kern_avg_s * 1e3, "launches": kern_launches,
                     "flops_per_row_step": fl, "row_steps_per_launch": rows_per_launch},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, n, p)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if world == 1:
            out["train_step"] = train_step_bench("cuda:%d" % local_rank)
        print(json.dumps(out))
    torch.cuda.synchronize(dev)
    eng.close()                 # destroys the in-library RCCL communicator before the process group goes away
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
