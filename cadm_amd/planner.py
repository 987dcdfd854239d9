"""Host-side orchestration of the CEM / random-shooting planners over HipEngine primitives.

Single GPU, no injected randomness  ->  ``HipEngine.cem_plan`` (one C call, `cadm_cem_plan`).
This module is the per-iteration form used
  * for multi-GPU planning: candidates shard contiguously over the ranks of a
    torch.distributed group (backend "nccl" == RCCL over xGMI; "gloo" in CPU tests), each rank
    rolls out its own shard and ONE all-gather of the per-candidate returns per CEM iteration
    ([m, n/G] floats per rank) gives every rank the full [m, n] vector; every rank then runs the
    identical top-k + refit.  Action sequences are never communicated and never sampled twice:
    a rank draws only ITS candidates from the counter-based RNG (keyed by global candidate id) and
    the refit draws the <= num_elites elite sequences again by id -- the work beside
    the rollout does not grow with the number of ranks;
  * for parity tests with injected ``z`` / ``eps`` (the reference's TF RNG streams are unseeded,
    SURVEY.md section 0).
Reference: /root/reference/cadm/dynamics/core/utils.py:398-488 (CEM), :490-561 (RS).
"""
import torch


class Shard:
    """Contiguous candidate shard of one rank."""

    def __init__(self, n, rank=0, world=1, group=None):
        if n % world != 0:
            raise ValueError("n_candidates (%d) must be divisible by the number of ranks (%d)" % (n, world))
        self.n, self.rank, self.world, self.group = n, rank, world, group
        self.n_local = n // world
        self.offset = rank * self.n_local

    @staticmethod
    def from_group(n, group):
        """Shard over the ranks of an EXPLICIT torch.distributed group.  Sharding is opt-in: a process that merely has
        torch.distributed initialised (parallel samplers planning for different observations, say) must not have its
        candidates mixed with other ranks'.  Every rank of `group` must call the planner with identical observations,
        weights and statistics (replicated state; checked by `check_replicated`)."""
        if group is None:
            return Shard(n)
        import torch.distributed as dist
        return Shard(n, dist.get_rank(group), dist.get_world_size(group), group)


def check_replicated(tensors, shard):
    """Guard for sharded planning: the replicated inputs must be the same on every rank of the group.  ONE all-reduce (MAX of
    [h, -h] gives max and -min of a position-weighted, NaN-aware checksum) plus a host sync -- so the planner runs it on the
    first few sharded calls of a model only (`MLPEnsembleCEMDynamicsModel(check_replicated_calls=...)`), not on every
    `get_action`: the steady-state call has exactly the path's own collectives (one all-gather per CEM iteration)."""
    if shard.world == 1:
        return
    import torch.distributed as dist
    sums = []
    for t in tensors:
        if t is None:
            continue
        x = torch.nan_to_num(t.double().reshape(-1), nan=12345.678, posinf=1.0e300, neginf=-1.0e300)
        w = torch.arange(1, x.numel() + 1, device=x.device, dtype=torch.float64)      # position-weighted: permutations differ
        sums.append((x * w).sum())
        sums.append(x.sum())
    h = torch.stack(sums)
    both = torch.cat([h, -h])
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=shard.group)
    hi, lo = both[:h.numel()], -both[h.numel():]
    if not torch.equal(lo, hi):
        raise RuntimeError("candidate-sharded planning needs identical obs / history / warm start on every rank of the "
                           "group (checksums differ: %s vs %s)" % (lo.tolist(), hi.tolist()))


def gather_cand_returns(cand_local, shard):
    """[m, n_local] per rank -> [G, m, n_local] on every rank (one collective)."""
    if shard.world == 1:
        return cand_local.unsqueeze(0)
    import torch.distributed as dist
    m, nl = cand_local.shape
    out = torch.empty((shard.world * m, nl), dtype=cand_local.dtype, device=cand_local.device)   # concatenated layout
    dist.all_gather_into_tensor(out, cand_local.contiguous(), group=shard.group)
    return out.view(shard.world, m, nl)


def cem_plan(engine, obs, cp_obs, cp_act, init_mean, init_var, n, seed=0, call=0, z=None, eps=None,
             shard=None, return_info=False):
    """z [iters,m,n,H,A] / eps [iters,H,m,n_local,p,D] optional injected draws."""
    shard = shard or Shard(n)
    obs = engine._t(obs)
    mean = engine._t(init_mean).clone()
    var = engine._t(init_var).clone()
    ctx_vec = engine.context_forward(cp_obs, cp_act) if engine.C > 0 else None
    info = []
    local = shard.world > 1 and z is None and not return_info       # (injected draws / diagnostics: the fully sampled form)
    for it in range(engine.num_cem_iters):
        if local:
            actions = engine.sample_actions(mean, var, n, seed=seed, call=call, it=it, cand_offset=shard.offset, n_local=shard.n_local)
        else:
            actions = engine.sample_actions(mean, var, n, z=None if z is None else z[it], seed=seed, call=call, it=it)
        rows = engine.rollout_returns(obs, ctx_vec, actions, eps=None if eps is None else eps[it], seed=seed,
                                      call=call, it=it, cand_offset=shard.offset, n_local=shard.n_local)
        cand = gather_cand_returns(engine.particle_mean(rows), shard)
        elites = engine.cem_refit(cand, actions, mean, var, G=shard.world, want_elites=return_info,
                                  regen=(seed, call, it) if local else None)
        if return_info:
            info.append(dict(actions=actions, rows=rows, cand=cand, elites=elites, mean=mean.clone(), var=var.clone()))
    plan = mean if engine.discrete else mean.clamp(-1.0, 1.0)   # dynamics.py:365-366
    return (plan, info, ctx_vec) if return_info else plan


def rs_plan(engine, obs, cp_obs, cp_act, n, seed=0, call=0, actions=None, raw=None, eps=None, shard=None):
    shard = shard or Shard(n)
    obs = engine._t(obs)
    m = obs.shape[0]
    ctx_vec = engine.context_forward(cp_obs, cp_act) if engine.C > 0 else None
    if actions is None:
        actions, raw = engine.sample_uniform(m, n, seed=seed, call=call)
    else:
        actions = engine._t(actions)
    rows = engine.rollout_returns(obs, ctx_vec, actions, eps=eps, norm_actions=not engine.discrete, seed=seed,
                                  call=call, it=0, cand_offset=shard.offset, n_local=shard.n_local)
    cand = gather_cand_returns(engine.particle_mean(rows), shard)
    first, best = engine.rs_select(cand, actions, G=shard.world)
    if engine.discrete:
        raw = engine._t(raw, dtype=torch.int32)
        return raw[torch.arange(m, device=raw.device), best.long(), 0], cand
    return first.clamp(-1.0, 1.0), cand
