"""bench.py -- CEM rollout row-steps/s of the PE-TS+CaDM planner on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE get_action on BASELINE.json's configs[1] (halfcheetah PE-TS+CaDM, ens=5, part=20, cand=200 per GPU,
H=30, m=1): context encoder + 5 CEM iterations x (sample, 30-step fused rollout of cand*part rows, refit).  Inputs
(obs, history, weights, init mean/var) are resident in HBM before the timed region.  Metric: row-steps/s =
m*n*p*H*5 / wall(get_action) (a row = one (candidate, particle) pair evaluated by exactly one member).
N > 1 is weak scaling of the headline (200 candidates per GPU, one RCCL all-gather of per-candidate returns per CEM
iteration, issued from inside libcadm_hip.so); the JSON line also carries `legs`: the same protocol on the other
BASELINE configs -- among them `cfg5`, north_star's 8-GPU point (1000 candidates per GPU; at N = 1 it is the per-GPU
share, the comparator of the ">= 6x at 8 GPUs" claim) -- and on the reference's launch shape m = 10.
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

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"; SURVEY.md 8d's denominator
F16_MFMA_PEAK_TFLOPS = 2500.0   # same guide, dense f16/bf16 MFMA


def flops_per_row_step(K0, hid, D, n_hidden):
    """Algorithmic FLOPs of one row-step: the six matmuls only."""
    return 2 * (K0 * hid + (n_hidden - 1) * hid * hid + hid * 2 * D)


def executed_mfma_flops_per_row_step(K0, hid, D, n_hidden):
    """f16 MFMA FLOPs the xdl kernel issues per row-step: 16x16x32 blocks over the padded tiles, 3 split products each
    (4 for widths > 256), rows padded to 16 being the caller's business (cadm_amd/csrc/xdl_geo.h)."""
    nt, nto = -(-hid // 16), -(-D // 8)
    nc0, nch = -(-K0 // 32), -(-nt // 2)
    blocks = nt * nc0 + (n_hidden - 1) * nt * nch + nto * nch
    return blocks * (16 * 32 * 2) * (4 if hid > 256 else 3)


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
    """Op-for-op torch-CPU restatement of the TF1.15 graph on this node's host cores (baseline only).
    The ONLY place bench.py touches oracle/ (imported here, on rank 0 at N = 1)."""
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
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, with_back=True, seed=0)
    eng = synth.make_engine(prob, p=20, device=device)
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
    eng.close()
    return {"ms_per_step": dt * 1e3, "batch": B, "members": 5, "rows_per_s": 5 * B / dt, "tflops": flops / dt / 1e12,
            "workload": "halfcheetah CaDM ensemble + backward model, fwd/bwd/TF1-Adam, fp32"}


class Planner:
    """One planner problem resident on this rank's GPU (+ the in-library RCCL communicator when world > 1)."""

    def __init__(self, cfg, m, n_per_gpu, world, rank, local_rank, dist):
        from cadm_amd import synth
        self.cfg, self.m, self.world, self.dist = cfg, m, world, dist
        self.n_per_gpu, self.n = n_per_gpu, n_per_gpu * world
        self.p, self.E, self.H = cfg["p"], cfg["E"], cfg["H"]
        self.prob = prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=self.E, m=m, H=self.H, seed=0)
        self.eng = eng = synth.make_engine(prob, p=self.p, deterministic=cfg["deterministic"], device="cuda:%d" % local_rank)
        self.obs, self.cp_obs, self.cp_act = eng._t(prob["obs"]), eng._t(prob["cp_obs"]), eng._t(prob["cp_act"])
        self.init_mean, self.init_var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
        self.collective, self.rccl_nranks = "none", 1
        if world > 1:
            # RCCL communicator inside libcadm_hip.so: one ncclAllGather per CEM iteration, all on-stream.  No fallback:
            # a SCALE run that silently used another path would not measure north_star's design -- fail loudly instead.
            eng.dist_init()
            self.rccl_nranks, rccl_rank = eng.dist_info()
            if self.rccl_nranks != world or rccl_rank != rank:
                raise RuntimeError("in-library RCCL communicator reports nranks=%d rank=%d, expected %d / %d"
                                   % (self.rccl_nranks, rccl_rank, world, rank))
            self.collective = "rccl all-gather in libcadm_hip.so"

    def step(self, call):
        return self.eng.cem_plan(self.obs, self.cp_obs, self.cp_act, self.init_mean, self.init_var, self.n, seed=0, call=call)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize(self.eng.device)

    def run(self, steps, warmup):
        """-> dict(elapsed [s, max over ranks], kern_ms, kern_launches)."""
        eng = self.eng
        for w in range(warmup):
            self.step(w)
        eng.profile_enable(True)
        self.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            plan = self.step(warmup + k)
        self.barrier()
        elapsed = time.perf_counter() - t0
        kern_ms, kern_launches = eng.profile_read()
        eng.profile_enable(False)
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert torch.isfinite(plan).all()
        return dict(elapsed=elapsed, kern_ms=kern_ms, kern_launches=kern_launches)

    def summary(self, r, steps):
        row_steps = self.m * self.n * self.p * self.H * self.eng.num_cem_iters
        prob = self.prob
        fl = flops_per_row_step(prob["K0"], 200, prob["D"], 4)
        xfl = executed_mfma_flops_per_row_step(prob["K0"], 200, prob["D"], 4)
        rows_per_launch = self.m * self.n_per_gpu * self.p * self.H
        kavg = r["kern_ms"] / 1e3 / max(r["kern_launches"], 1)
        return dict(value=row_steps * steps / r["elapsed"], ms_per_get_action=r["elapsed"] / steps * 1e3,
                    kernel_avg_launch_ms=kavg * 1e3, launches=r["kern_launches"], row_steps_per_launch=rows_per_launch,
                    achieved_tflops=rows_per_launch * fl / kavg / 1e12, executed_f16_mfma_tflops=rows_per_launch * xfl / kavg / 1e12,
                    flops_per_row_step=fl)

    def close(self):
        torch.cuda.synchronize(self.eng.device)
        self.eng.close()


LEGS = {   # name: (config, m, candidates per GPU)
    "cfg3": ("cfg3", 1, 2000), "cfg4": ("cfg4", 1, 1000), "cfg5": ("cfg5", 1, 1000), "m10": ("cfg2", 10, 200),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", help="headline workload: cfg2 (BASELINE metric) | cfg3 | cfg4 | cfg5 | m10")
    ap.add_argument("--cand-per-gpu", type=int, default=None)
    ap.add_argument("--legs", default=None, help="comma list of extra legs (default: cfg3,cfg4,cfg5,m10 at N=1; cfg5 at N>1; 'none')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from cadm_amd import synth

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

    def make(name, cand=None):
        cfgname, m, n_per_gpu = LEGS[name] if name in LEGS else (name, 1, synth.CONFIGS[name]["n"])
        return Planner(dict(synth.CONFIGS[cfgname]), m, cand or n_per_gpu, world, rank, local_rank, dist), cfgname, m

    head, cfgname, m = make(args.config, args.cand_per_gpu)
    r = head.run(args.steps, args.warmup)
    s = head.summary(r, args.steps)
    cfg = head.cfg
    # HBM/fabric traffic of the dominant kernel: PMC counters cannot be read from inside this process; they come from
    # separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of THIS command, committed under profiles/
    # (KB per launch).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide coalesced reads at half
    # their bytes -> doubled.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r2_pmc_traffic_%s.json" % args.config)
    if world == 1 and args.cand_per_gpu is None and os.path.exists(tpath):
        raw = json.load(open(tpath))
        traffic = (2.0 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"]) * 1024.0
    out = {
        "metric": "CEM rollout row-steps/s (cand x part x horizon x 5 CEM iters per get_action; ens=%d members)" % cfg["E"],
        "value": s["value"], "unit": "row-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": s["ms_per_get_action"], "get_action_latency_ms": s["ms_per_get_action"],
        "plans_per_s": 1e3 / s["ms_per_get_action"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (each fp32 product as 3 f16 MFMA products of 2-way split operands, fp32 accumulate; "
                                      "error <= 2^-22 per product, parity 1e-5 vs the fp32 oracle)",
        "data": "synthetic",
        "config": {"workload": "%s: %s PE-TS+CaDM get_action, ens=%d part=%d cand=%d (%d/GPU) H=%d m=%d, random-init weights"
                               % (args.config, cfg["env"], cfg["E"], cfg["p"], head.n, head.n_per_gpu, cfg["H"], m),
                   "global_candidates": head.n, "parallelism": "candidate-shard x%d" % world, "collective": head.collective,
                   "rccl_nranks": head.rccl_nranks},
        "roofline": {"bound": "mfma", "achieved": s["achieved_tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": s["achieved_tflops"] / FP32_MFMA_PEAK_TFLOPS,
                     "peak_note": "fp32 matrix peak = the roofline of fp32 arithmetic on this part (SURVEY.md 8d); the kernel "
                                  "computes the same fp32 result on the f16 matrix pipe (3 split products), so frac may exceed "
                                  "what v_mfma_f32_16x16x4_f32 could ever reach",
                     "pipe": "v_mfma_f32_16x16x32_f16", "pipe_peak": F16_MFMA_PEAK_TFLOPS,
                     "pipe_executed": s["executed_f16_mfma_tflops"], "pipe_frac": s["executed_f16_mfma_tflops"] / F16_MFMA_PEAK_TFLOPS,
                     "traffic": traffic,
                     "traffic_note": "bytes/launch = 2*FETCH_SIZE + WRITE_SIZE from profiles/r2_pmc_traffic_%s.json (a separate "
                                     "rocprofv3 --pmc pass of this command, NOT measured in this run)" % args.config,
                     "kernel": "rollout_xdl_kernel", "avg_launch_ms": s["kernel_avg_launch_ms"], "launches": s["launches"],
                     "launch_note": "one 'launch' = one rollout of all rows over the horizon, bracketed by hipEvents inside libcadm_hip.so "
                                    "(a single kernel at cfg2; at >= 2 row tiles per workgroup slot the launcher issues a tile-pair kernel "
                                    "plus a single-tile kernel for the remainder)",
                     "flops_per_row_step": s["flops_per_row_step"], "row_steps_per_launch": s["row_steps_per_launch"]},
    }
    prob_head, n_head, p_head = head.prob, head.n, head.p
    head.close()

    # extra legs, same protocol (barrier + sync bracketed, max over ranks)
    legs = args.legs if args.legs is not None else ("cfg3,cfg4,cfg5,m10" if world == 1 else "cfg5")
    out["legs"] = {}
    for name in [x for x in legs.split(",") if x and x != "none" and x != args.config]:
        pl, lcfg, lm = make(name)
        lsteps = max(5, min(args.steps, 20))
        ls = pl.summary(pl.run(lsteps, 2), lsteps)
        out["legs"][name] = {"workload": "%s env=%s m=%d cand=%d (%d/GPU)" % (lcfg, pl.cfg["env"], lm, pl.n, pl.n_per_gpu),
                             "value": ls["value"], "unit": "row-steps/s", "ms_per_get_action": ls["ms_per_get_action"],
                             "kernel_avg_launch_ms": ls["kernel_avg_launch_ms"], "achieved_tflops": ls["achieved_tflops"],
                             "frac_of_fp32_mfma_peak": ls["achieved_tflops"] / FP32_MFMA_PEAK_TFLOPS, "steps": lsteps,
                             "collective": pl.collective, "rccl_nranks": pl.rccl_nranks}
        pl.close()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob_head, n_head, p_head)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if world == 1:
            out["train_step"] = train_step_bench("cuda:%d" % local_rank)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
