"""bench.py -- CEM rollout row-steps/s of the PE-TS+CaDM planner on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without WORLD_SIZE in the environment: bench.py launches its own N ranks (re-executes itself under
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`); launched BY
  torch.distributed.run (WORLD_SIZE / RANK / LOCAL_RANK set) it is one of the ranks.  Rank 0 prints the one JSON line.

A "step" is ONE `MLPEnsembleCEMDynamicsModel.get_action` on BASELINE.json's configs[1] (halfcheetah PE-TS+CaDM, ens=5,
part=20, cand=200 per GPU, H=30, m=1): context encoder + 5 CEM iterations x (sample, 30-step fused rollout of cand*part rows,
refit).  Metric: row-steps/s = m*n*p*H*5 / wall_time(get_action), a row = one (candidate, particle) pair
evaluated by exactly one member.

`value` is timed THROUGH THE CLASS API the reference exposes (numpy in -> numpy out, the samplers' warm-start shift between
calls, cadm/samplers/sampler.py:109-120), i.e. it includes the staging copy of the five tiny inputs and the return of the plan;
`device_resident` is the same planner driven with inputs already in HBM (`cadm_cem_plan` only) -- the two differ by the
per-call host overhead (`api_overhead_us`).  N > 1 is weak scaling of the headline (200 candidates per GPU, one RCCL all-gather
of per-candidate returns per CEM iteration, issued from inside libcadm_hip.so); `legs` repeat the device-resident protocol on
the other BASELINE configs -- among them `cfg5`, north_star's 8-GPU point (1000 candidates per GPU) -- and on m = 10.

`roofline`: the dominant kernel (`rollout_xdl_kernel`) computes fp32 results on the f16 matrix pipe (3 f16 MFMA products of
2-way split operands per fp32 product).  `achieved` = algorithmic fp32-equivalent FLOPs (268 000 per row-step x
row-steps per launch) / launch time measured with hipEvents on the launch stream inside the library; `peak` = that pipe's ceiling
for this arithmetic, 2500 TFLOP/s dense f16 / 3 products = 833 TFLOP/s; `pipe_frac` = f16 MFMA FLOPs actually issued (incl. tile
padding) / 2500.  `mfma_busy` and `traffic` come from the committed rocprofv3 PMC passes of this command (profiles/).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

F16_MFMA_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: dense f16 / bf16 MFMA
FP32_MFMA_PEAK_TFLOPS = 157.3   # same guide, "Peak FP32 (matrix)": what v_mfma_f32_*_f32 could reach (reported for context only)
PMC_FILE = "r6_pmc_%s.json"     # profiles/: mean counters per rollout dispatch from separate rocprofv3 --pmc passes of this command


def flops_per_row_step(K0, hid, D, n_hidden):
    """Algorithmic FLOPs of one row-step: the six matmuls only."""
    return 2 * (K0 * hid + (n_hidden - 1) * hid * hid + hid * 2 * D)


def executed_mfma_flops_per_row_step(K0, hid, D, n_hidden):
    """f16 MFMA FLOPs the kernel issues per row-step: 16x16x32 blocks over the padded tiles, 3 split products each
    (4 for widths > 256) (cadm_amd/csrc/xdl_geo.h)."""
    nt, nto = -(-hid // 16), -(-D // 8)
    nc0, nch = -(-K0 // 32), -(-nt // 2)
    blocks = nt * nc0 + (n_hidden - 1) * nt * nch + nto * nch
    return blocks * (16 * 32 * 2) * (4 if hid > 256 else 3)


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
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
    """Op-for-op torch-CPU restatement of the TF1.15 graph on this node's host cores (baseline only).  This leg (rank 0 at
    N = 1) is the ONLY place bench.py touches oracle/: the timing here, and `parity_block` below which uses it as the checker."""
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


def parity_block():
    """Measured in THIS run (part of the cpu_baseline leg: the oracle is the checker): error of the production kernel against the
    fp64 numpy oracle at cfg2's full size with trained-like weights, next to the fp32 numpy oracle's own error on the same inputs
    (tests/precision.py; the asserting twin is tests/test_gpu_precision.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import precision
    res = precision.measure({"xdl": precision.product_engine})
    x, o = res["xdl"], res["fp32_oracle"]
    return {"config": "cfg2 full size (4000 rows, one teacher-forced step each; 24 candidates x 20 particles x 30 steps), trained-like weights",
            "reference": "fp64 numpy oracle", "tolerance_north_star": 1e-5,
            "one_step_max_rel": x["one_step_obs"]["max_rel"], "one_step_pure_rel_big": x["one_step_obs"]["pure_rel_big"],
            "one_step_vs_fp32_oracle_pure_rel_big": x["one_step_obs_vs_fp32_oracle"]["pure_rel_big"],
            "traj30_rel_vs_fp64": x["traj_obs"]["max_rel"], "returns_rel_vs_fp64": x["returns"]["max_rel"],
            "fp32_oracle_vs_fp64": {"one_step_max_rel": o["one_step_obs"]["max_rel"], "traj30_rel": o["traj_obs"]["max_rel"],
                                    "returns_rel": o["returns"]["max_rel"]},
            "metrics": "max_rel = max|d|/max|ref|; pure_rel_big = max elementwise |d|/|ref| over |ref| >= 0.25 rms (no floor)"}


def train_step_bench(device, lib=None, steps=100, warmup=10, B=256):
    """Second leg of the path: one fused forward/backward/Adam step of the 5-member CaDM ensemble
    (+ backward model) on a device-resident [E, B, .] bootstrap batch.  Reported beside the headline, never as it."""
    from cadm_amd import synth
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, with_back=True, seed=0)
    eng = synth.make_engine(prob, p=20, device=device, lib=lib)
    eng.train_configure(1e-3, (0.000025, 0.00005, 0.000075, 0.000075, 0.0001), (0.000025, 0.00005, 0.000075), 1.0, 0.5,
                        max_batch=B)
    batch = {k: eng._t(v) for k, v in synth.make_train_batch(prob, B=B, seed=1).items()}
    for _ in range(1500):             # untimed clock ramp (~0.2 s of continuous work), then the warmup steps
        eng.train_step(batch, train=True)
    torch.cuda.synchronize(eng.device)
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


def context_bench(device, lib=None, m=2048, steps=50, warmup=5):
    """batched context inference -- `get_context_pred` on m histories per call (the PPO consumer's scale,
    ppo_cadm.py:155-162) through `context_batched_kernel` (fp32 MFMA; weights reused across the rows of a member).  A row =
    one (member, history) pair through the 240-256-128-64-10 encoder: 2 x 103 040 FLOP."""
    from cadm_amd import synth
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, seed=0)
    eng = synth.make_engine(prob, p=5, device=device, lib=lib)
    rng = np.random.default_rng(1)
    cp_obs = eng._t(0.1 * rng.standard_normal((m, prob["D"] * prob["Hh"])))
    cp_act = eng._t(rng.uniform(-1, 1, (m, prob["A"] * prob["Hh"])))
    for _ in range(warmup + 200):            # (+ clock ramp)
        out = eng.context_forward(cp_obs, cp_act)
    torch.cuda.synchronize(eng.device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = eng.context_forward(cp_obs, cp_act)
    torch.cuda.synchronize(eng.device)
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out).all()
    rows = 5 * m
    flops = rows * 2 * (240 * 256 + 256 * 128 + 128 * 64 + 64 * 10)
    eng.close()
    return {"context_rows_per_s": rows / dt, "ms_per_call": dt * 1e3, "histories_per_call": m, "members": 5, "tflops": flops / dt / 1e12,
            "frac_of_fp32_matrix_peak": flops / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "workload": "get_context_pred-shaped: %d histories x 5 members, encoder 240-256-128-64-10, fp32 (v_mfma_f32_16x16x4_f32)" % m}


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: run the N ranks ourselves, one process per GPU, as the
    driver's own multi-GPU form does (torch.distributed.run, rendezvous on 127.0.0.1).  Rank 0's JSON line passes through on
    stdout; the launcher's exit code is the job's."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool's hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


class DryRunPlanner:
    """--dry-run: the launcher / rank / JSON plumbing of this file on CPU over gloo, with NO planner behind it (tests/
    test_bench_launcher.py: the N > 1 path must not rot between the rare runs on a multi-GPU node).  Every number it produces
    is a placeholder and the line says so (`data: "dry-run"`)."""
    RAMP_PLANS = 0

    def __init__(self, cfg, m, n_per_gpu, world, rank, dist):
        self.cfg, self.m, self.world, self.rank, self.dist = cfg, m, world, rank, dist
        self.n_per_gpu, self.n, self.p, self.E, self.H = n_per_gpu, n_per_gpu * world, cfg["p"], cfg["E"], cfg["H"]
        self.prob = {"K0": 34, "D": 18, "env": cfg["env"], "m": m, "H": self.H}
        self.collective, self.rccl_nranks = ("gloo all-gather (dry run)", world) if world > 1 else ("none", 1)
        self.degraded = False
        self.iters = 5

    def _step(self):
        time.sleep(1e-3)
        if self.dist is not None:
            mine = torch.full((self.m, self.n_per_gpu), float(self.rank))
            for _ in range(self.iters):                       # the path's one all-gather per CEM iteration
                out = [torch.empty_like(mine) for _ in range(self.world)]
                self.dist.all_gather(out, mine)
            assert [float(o[0, 0]) for o in out] == [float(r) for r in range(self.world)]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _timed(self, steps, warmup):
        for _ in range(warmup):
            self._step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self._step()
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def run_api(self, steps, warmup):
        return self._timed(steps, warmup)

    def run_device(self, steps, warmup, profile=True):
        e = self._timed(steps, warmup)
        return dict(elapsed=e, kern_ms=e * 1e3 * 0.8, kern_launches=steps * self.iters,
                    ag_ms=0.01 * steps * self.iters if self.world > 1 else 0.0, ag_calls=steps * self.iters if self.world > 1 else 0)

    summary = None   # bound below (same arithmetic as Planner.summary)

    def close(self):
        pass


class Planner:
    """One planner problem on this rank's GPU, built through the DROP-IN CLASS (cadm_amd.dynamics) with synthetic weights and
    statistics; candidate-sharded over the ranks of the default process group when world > 1 (in-library RCCL)."""

    def __init__(self, cfg, m, n_per_gpu, world, rank, local_rank, dist, lib=None):
        from cadm_amd import synth
        from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel as CaDM
        from cadm_amd.dynamics.mlp_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel as Vanilla
        from cadm_amd.envs import make_env_spec
        self.cfg, self.m, self.world, self.rank, self.dist = cfg, m, world, rank, dist
        self.n_per_gpu, self.n = n_per_gpu, n_per_gpu * world
        self.p, self.E, self.H = cfg["p"], cfg["E"], cfg["H"]
        self.prob = prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=self.E, m=m, H=self.H, seed=0)
        kw = dict(hidden_sizes=prob["hidden_sizes"], hidden_nonlinearity="swish", n_forwards=self.H, n_candidates=self.n,
                  ensemble_size=self.E, n_particles=self.p, use_cem=True, deterministic=cfg["deterministic"], normalize_input=True,
                  device="cuda:%d" % local_rank, process_group=dist.group.WORLD if world > 1 else None)
        if lib is not None:
            kw["engine_lib"] = lib
        if cfg["context"]:
            self.model = CaDM("dyn_model", make_env_spec(cfg["env"]), cp_hidden_sizes=prob["cp_hidden_sizes"], context_out_dim=prob["C"],
                              history_length=prob["Hh"], state_diff=1, **kw)
        else:
            self.model = Vanilla("dyn_model", make_env_spec(cfg["env"]), **kw)
        self.eng = eng = self.model.engine
        if prob["cp"] is not None:
            eng.set_net("context_model", prob["cp"])
        eng.set_net("ff_model", prob["ff"])
        st = prob["stats"]
        self.model.set_normalization(OrderedDict((k, (st[k + "_mean"], st[k + "_std"])) for k in ("obs", "delta", "act", "cp_obs", "cp_act", "back_delta")))
        self.model._push_stats()
        eng.set_stats(st)       # the synthetic statistics verbatim (state_diff would zero the history statistics)
        self.obs, self.cp_obs, self.cp_act = eng._t(prob["obs"]), eng._t(prob["cp_obs"]), eng._t(prob["cp_act"])
        self.init_mean, self.init_var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
        self.collective, self.rccl_nranks, self.degraded = "none", 1, False
        self.iters = eng.num_cem_iters

    # ---- numpy in -> numpy out through the class (the reference's call, dynamics.py:344-367 / vanilla :191-207)
    def api_call(self, prev_sol):
        prob = self.prob
        if self.cfg["context"]:
            return self.model.get_action(prob["obs"], prob["cp_obs"], prob["cp_act"], prev_sol, prob["init_var"])
        return self.model.get_action(prob["obs"], prev_sol, prob["init_var"])

    def check_rccl(self):
        if self.world > 1:
            # RCCL communicator inside libcadm_hip.so: one ncclAllGather per CEM iteration, all on-stream -- north_star's design.
            # If it could not be created on some rank (the model's ranks then agree on torch.distributed.all_gather_into_tensor,
            # _sharding()), the run still measures, but the line SAYS SO: `collective` names the fallback, `rccl_nranks` is 0 and
            # `degraded` is set -- never a silent substitution.
            self.rccl_nranks, rccl_rank = self.eng.dist_info()
            if self.eng._ext is None and self.eng.dist_world == self.world and self.rccl_nranks == self.world and rccl_rank == self.rank:
                self.collective = "rccl all-gather in libcadm_hip.so"
            else:
                self.collective = ("FALLBACK: torch.distributed all_gather plugged into the library's sharded planner (in-library RCCL communicator unavailable: "
                                   "dist_world=%d rccl nranks=%d rank=%d, expected %d / %d)" % (self.eng.dist_world, self.rccl_nranks, rccl_rank, self.world, self.rank))
                self.rccl_nranks = 0
                self.degraded = True

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize(self.eng.device)

    def _max_over_ranks(self, elapsed):
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.eng.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    RAMP_PLANS = 300

    def ramp(self):
        """Untimed, before the W warmup steps: RAMP_PLANS back-to-back planner calls on HBM-resident inputs (~0.3 s of
        continuous work; the same count on every rank).  A synchronous 1 ms call pattern on a GPU that has idled through model
        construction can sit in a low clock state for the whole default run (one run in ~10 measured 2.2 ms per get_action with
        the kernels' own times unchanged); this is not one of the W or K steps."""
        eng = self.eng
        for c in range(self.RAMP_PLANS):
            self.plan_device(c)
        torch.cuda.synchronize(eng.device)

    def plan_device(self, c):
        """One planner call on HBM-resident inputs.  Sharded: the library's loop on every rank; the model's negotiation (`_sharding`) picks
        its collective -- the in-library RCCL communicator on every rank, or, all ranks together, torch.distributed's all-gather plugged in."""
        cp_obs, cp_act = (self.cp_obs, self.cp_act) if self.cfg["context"] else (None, None)
        self.model._sharding()
        return self.eng.cem_plan(self.obs, cp_obs, cp_act, self.init_mean, self.init_var, self.n, seed=0, call=c)

    def run_api(self, steps, warmup):
        """K get_action calls through the class, warm start shifted between calls (sampler.py:118-120)."""
        prev = self.prob["init_mean"].copy()
        self.ramp()
        for _ in range(warmup):
            plan = self.api_call(prev)
        self.check_rccl()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            plan = self.api_call(prev)
            prev[:, :-1] = plan[:, 1:]
            prev[:, -1] = 0.0
        self.barrier()
        elapsed = self._max_over_ranks(time.perf_counter() - t0)
        assert np.isfinite(plan).all() and np.abs(plan).max() <= 1.0
        return elapsed

    def run_device(self, steps, warmup, profile=True):
        """K planner calls on HBM-resident inputs (cadm_cem_plan only).  profile=True: the rollout launches and the all-gathers
        are bracketed by hipEvents on the launch stream inside the library (kernel / collective times; the events cost a few us
        per get_action, so the wall time of THIS pass is not the device-resident figure); profile=False: no events, wall time only.
        -> dict(elapsed [s, max over ranks], kern_ms, kern_launches, ag_*)."""
        eng = self.eng
        step = self.plan_device
        step(0)                  # (negotiates the sharded path on first use)
        self.check_rccl()
        if profile:
            # the warmup steps run bracketed too: the library creates its hipEvent pairs on first use, and a host that is busy creating
            # events falls behind the GPU -- the start event is then stamped before the kernel's dispatch packet has even arrived and the
            # bracket reads 10-15 us long (driver's command, 20 steps: 172 us per rollout against 162 with 50 steps; rocprofv3: 157-163)
            eng.profile_enable(True)
        for w in range(max(warmup, 10 if profile else 0)):
            step(w)
        if profile:
            torch.cuda.synchronize(eng.device)
            eng.profile_enable(True)      # (resets the counters; the event pool stays)
        self.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            plan = step(warmup + k)
        self.barrier()
        elapsed = self._max_over_ranks(time.perf_counter() - t0)
        kern_ms = ag_ms = 0.0
        kern_launches = ag_calls = 0
        if profile:
            kern_ms, kern_launches = eng.profile_read()
            ag_ms, ag_calls = eng.profile_read_collective()
            eng.profile_enable(False)
        assert torch.isfinite(plan).all()
        return dict(elapsed=elapsed, kern_ms=kern_ms, kern_launches=kern_launches, ag_ms=ag_ms, ag_calls=ag_calls)

    def summary(self, r, steps):
        row_steps = self.m * self.n * self.p * self.H * self.iters
        prob = self.prob
        fl = flops_per_row_step(prob["K0"], 200, prob["D"], 4)
        xfl = executed_mfma_flops_per_row_step(prob["K0"], 200, prob["D"], 4)
        rows_per_launch = self.m * self.n_per_gpu * self.p * self.H
        kavg = r["kern_ms"] / 1e3 / max(r["kern_launches"], 1) or float("nan")
        return dict(value=row_steps * steps / r["elapsed"], ms_per_get_action=r["elapsed"] / steps * 1e3,
                    kernel_avg_launch_ms=kavg * 1e3, launches=r["kern_launches"], row_steps_per_launch=rows_per_launch,
                    achieved_tflops=rows_per_launch * fl / kavg / 1e12, executed_f16_mfma_tflops=rows_per_launch * xfl / kavg / 1e12,
                    flops_per_row_step=fl, row_steps_per_get_action=row_steps,
                    allgather_us=(r["ag_ms"] * 1e3 / r["ag_calls"]) if r["ag_calls"] else None, allgather_calls=r["ag_calls"])

    def close(self):
        torch.cuda.synchronize(self.eng.device)
        self.eng.close()


DryRunPlanner.summary = Planner.summary


LEGS = {   # name: (config, m, candidates per GPU)
    "cfg3": ("cfg3", 1, 2000), "cfg4": ("cfg4", 1, 1000), "cfg5": ("cfg5", 1, 1000), "m10": ("cfg2", 10, 200),
}


def pmc_block(config, build_id):
    """Counters of the dominant kernel.  PMC counters cannot be read from inside this process; they come from separate
    `rocprofv3 --pmc` passes of THIS command (tools/profile_round.sh), committed under profiles/ as means per rollout dispatch
    together with the build id (cadm_build_id(): hash of the kernel sources) of the library they were measured on.  A file from
    another build is NOT quoted: the fields stay null and `pmc_note` says why.
    gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide coalesced reads at half their bytes -> doubled; KB -> bytes."""
    tpath = os.path.join(ROOT, "profiles", PMC_FILE % config)
    if not os.path.exists(tpath):
        return None, None, None, "no committed PMC pass for this workload"
    raw = json.load(open(tpath))
    src = "profiles/" + PMC_FILE % config
    if raw.get("build_id") != build_id:
        return None, None, src, ("%s was measured on build %s, the loaded library is build %s: counters not quoted (re-run "
                                 "tools/profile_round.sh)" % (src, raw.get("build_id"), build_id))
    traffic = None
    if "FETCH_SIZE" in raw and "WRITE_SIZE" in raw:
        traffic = (2.0 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"]) * 1024.0
    # SIMD-cycles the matrix pipe was busy / SIMD-cycles of the launch (tools/pmc_extract.py: SQ_VALU_MFMA_BUSY_CYCLES /
    # (4 SIMDs x CUs x GRBM_GUI_ACTIVE))
    return traffic, raw.get("mfma_busy_frac"), src, (
        "mfma_busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles of the launch) and traffic (bytes/launch = 2*FETCH_SIZE + WRITE_SIZE) are read "
        "from the committed rocprofv3 --pmc passes of this command on THIS build (build id %s), not measured in this run" % build_id)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", help="headline workload: cfg2 (BASELINE metric) | cfg3 | cfg4 | cfg5 | m10")
    ap.add_argument("--cand-per-gpu", type=int, default=None)
    ap.add_argument("--legs", default=None, help="comma list of extra legs (default: cfg3,cfg4,cfg5,m10 at N=1; cfg5 at N>1; 'none')")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the cpu_baseline leg (and the parity block measured inside it)")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no legs, no training-step leg, no cpu baseline / parity")
    ap.add_argument("--lib", default=None, help="DEVELOPER: bind the run to another build of the library (tools/ab.sh); default = the product")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rank / JSON plumbing only, on CPU over gloo, no planner (tests/test_bench_launcher.py)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    from cadm_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if not args.dry_run:
        if torch.cuda.device_count() < world:
            raise SystemExit("--gpus %d but only %d visible GPU(s)" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    lib = _lib.load_dev(args.lib) if args.lib else None
    build_id = "dry-run" if args.dry_run else (lib or _lib.load()).cadm_build_id().decode()

    def make(name, cand=None):
        cfgname, m, n_per_gpu = LEGS[name] if name in LEGS else (name, 1, synth.CONFIGS[name]["n"])
        if args.dry_run:
            return DryRunPlanner(dict(synth.CONFIGS[cfgname]), m, cand or n_per_gpu, world, rank, dist), cfgname, m
        return Planner(dict(synth.CONFIGS[cfgname]), m, cand or n_per_gpu, world, rank, local_rank, dist, lib), cfgname, m

    head, cfgname, m = make(args.config, args.cand_per_gpu)
    api_elapsed = head.run_api(args.steps, args.warmup)
    plain = head.run_device(args.steps, args.warmup, profile=False)      # device-resident wall time: no event bracketing
    # kernel / collective times from hipEvents: its own pass of at least 60 get_actions behind 20 bracketed warmup ones -- the brackets read
    # 166-169 us per rollout over the first 100 launches of a bracketed pass and 160 over 500 (rocprofv3 of the same command: 155-157);
    # `roofline.launches` says how many launches the average is over.  The K timed steps of the headline are the two passes above.
    ksteps = max(args.steps, 60)
    r = head.run_device(ksteps, max(args.warmup, 20), profile=True)
    s = head.summary(r, ksteps)
    cfg = head.cfg
    api_ms = api_elapsed / args.steps * 1e3
    dev_ms = plain["elapsed"] / args.steps * 1e3
    api_value = s["row_steps_per_get_action"] * args.steps / api_elapsed

    traffic = mfma_busy = pmc_src = None
    pmc_note = "PMC passes are recorded for the default single-GPU workloads only"
    if world == 1 and args.cand_per_gpu is None and not args.dry_run:
        traffic, mfma_busy, pmc_src, pmc_note = pmc_block(args.config, build_id)
    peak_equiv = F16_MFMA_PEAK_TFLOPS / 3.0
    out = {
        "metric": "CEM rollout row-steps/s (cand x part x horizon x 5 CEM iters per get_action; ens=%d members)" % cfg["E"],
        "value": api_value, "unit": "row-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": api_ms, "get_action_latency_ms": api_ms, "plans_per_s": 1e3 / api_ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "per_gpu_value": api_value / world,
        "value_note": "wall time of the drop-in class's get_action, numpy in -> numpy out, warm-start shift between calls (SURVEY.md 8d); "
                      "`device_resident` = the same planner with inputs already in HBM",
        "device_resident": {"value": s["row_steps_per_get_action"] / (dev_ms * 1e-3), "unit": "row-steps/s", "device_ms_per_get_action": dev_ms,
                            "note": "cadm_cem_plan back to back on HBM-resident inputs, no hipEvent bracketing; the pass that yields the "
                                    "kernel times (events on) took %.4f ms per get_action" % s["ms_per_get_action"]},
        "api_overhead_us": (api_ms - dev_ms) * 1e3,
        "dtype": "f32 (each fp32 product as 3 f16 MFMA products of 2-way split operands, fp32 accumulate; error <= 2^-22 per product; "
                 "measured parity: see `parity`)",
        "data": "dry-run (placeholders: no planner ran)" if args.dry_run else "synthetic",
        "build_id": build_id, "degraded": bool(head.degraded),
        "config": {"workload": "%s: %s PE-TS+CaDM get_action, ens=%d part=%d cand=%d (%d/GPU) H=%d m=%d, random-init weights"
                               % (args.config, cfg["env"], cfg["E"], cfg["p"], head.n, head.n_per_gpu, cfg["H"], m),
                   "global_candidates": head.n, "parallelism": "candidate-shard x%d" % world, "collective": head.collective,
                   "rccl_nranks": head.rccl_nranks, "allgather_us": s["allgather_us"], "allgathers_timed": s["allgather_calls"]},
        "roofline": {"bound": "mfma", "achieved": s["achieved_tflops"], "peak": peak_equiv, "unit": "TFLOP/s",
                     "frac": s["achieved_tflops"] / peak_equiv,
                     "peak_note": "fp32-equivalent ceiling of the pipe the kernel executes on: dense f16 MFMA peak 2500 TFLOP/s / 3 split "
                                  "products per fp32 product (v_mfma_f32_16x16x32_f16); `achieved` counts the algorithmic fp32 FLOPs of SURVEY 8d",
                     "pipe": "v_mfma_f32_16x16x32_f16", "pipe_peak": F16_MFMA_PEAK_TFLOPS,
                     "pipe_executed": s["executed_f16_mfma_tflops"], "pipe_frac": s["executed_f16_mfma_tflops"] / F16_MFMA_PEAK_TFLOPS,
                     "mfma_busy": mfma_busy, "traffic": traffic, "pmc_source": pmc_src, "pmc_note": pmc_note,
                     "traffic_note": "fabric bytes per launch summed over the 8 XCD L2s; algorithmic: the packed weight stream once per chip, ~3.4 MB incl. "
                                     "actions and returns.  A member's workgroups sit on two or three XCDs (XCD-affine work items since round 6: ~9 MB; with "
                                     "every member on every XCD it was 28 MB); < 0.1 TB/s, not what bounds the kernel",
                     "fp32_matrix_peak_for_context": FP32_MFMA_PEAK_TFLOPS,
                     "kernel": "rollout_xdl_kernel", "avg_launch_ms": s["kernel_avg_launch_ms"], "launches": s["launches"],
                     "launch_note": "one 'launch' = one rollout of all rows over the horizon, bracketed by hipEvents inside libcadm_hip.so on "
                                    "the launch stream (a single kernel at cfg2; at >= 2 row tiles per workgroup slot the launcher issues a "
                                    "tile-pair kernel plus a single-tile kernel for the remainder)",
                     "flops_per_row_step": s["flops_per_row_step"], "row_steps_per_launch": s["row_steps_per_launch"]},
    }
    prob_head, n_head, p_head = head.prob, head.n, head.p
    head.close()

    # extra legs, device-resident protocol (barrier + sync bracketed, max over ranks)
    legs = args.legs if args.legs is not None else ("cfg3,cfg4,cfg5,m10" if world == 1 else "cfg5")
    if args.no_extras:
        legs = "none"
    out["legs"] = {}
    for name in [x for x in legs.split(",") if x and x != "none" and x != args.config]:
        pl, lcfg, lm = make(name)
        lsteps = max(5, min(args.steps, 20))
        ls = pl.summary(pl.run_device(lsteps, 2), lsteps)
        out["legs"][name] = {"workload": "%s env=%s m=%d cand=%d (%d/GPU)" % (lcfg, pl.cfg["env"], lm, pl.n, pl.n_per_gpu),
                             "value": ls["value"], "per_gpu_value": ls["value"] / world, "unit": "row-steps/s",
                             "protocol": "device-resident inputs",
                             "ms_per_get_action": ls["ms_per_get_action"],
                             "kernel_avg_launch_ms": ls["kernel_avg_launch_ms"], "achieved_tflops": ls["achieved_tflops"],
                             "roofline_frac": ls["achieved_tflops"] / peak_equiv,
                             "pipe_frac": ls["executed_f16_mfma_tflops"] / F16_MFMA_PEAK_TFLOPS, "steps": lsteps,
                             "collective": pl.collective, "rccl_nranks": pl.rccl_nranks, "allgather_us": ls["allgather_us"]}
        pl.close()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.no_extras and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline(prob_head, n_head, p_head)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["parity"] = parity_block()
        if world == 1 and not args.no_extras and not args.dry_run:
            out["train_step"] = train_step_bench("cuda:%d" % local_rank, lib)
            out["context_batched"] = context_bench("cuda:%d" % local_rank, lib)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
