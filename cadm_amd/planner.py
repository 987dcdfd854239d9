"""Host-side pieces of the CEM / random-shooting planners.

The planner itself is ONE implementation, in the library: ``HipEngine.cem_plan`` / ``cem_plan_host`` / ``rs_plan`` (`cadm_cem_plan`,
`cadm_cem_plan_staged`, `cadm_rs_plan`).  Candidate-sharded over several GPUs it is still that loop -- every rank draws and rolls out
its own contiguous shard, ONE all-gather of the per-candidate returns per CEM iteration ([m, n/G] floats + a checksum word per rank)
gives every rank the full vector, every rank runs the identical top-k + refit and regenerates the elite sequences by id.
Only the collective is a plug: the RCCL communicator the ctx owns (`HipEngine.dist_init`), or -- same loop, same payload, same
checks -- an all-gather this module supplies over any torch.distributed backend (`ExternalAllGather`, registered with
`cadm_dist_init_external`; "gloo" in the CPU tests and in the two-ranks-on-one-GPU test).

What is left here:
  * `Shard`: a rank's contiguous candidate range;  `check_replicated`: the host-side guard of the first sharded calls;
  * `ExternalAllGather`: the host-supplied collective;
  * `cem_plan` / `rs_plan`: the per-iteration, SINGLE-RANK form over the engine's primitives, for parity tests with injected ``z`` /
    ``eps`` and for diagnostics (`return_info`) -- the reference's TF RNG streams are unseeded.
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
    if both.is_cuda and dist.get_backend(shard.group) != "nccl":      # (gloo moves host tensors)
        both = both.cpu()
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=shard.group)
    hi, lo = both[:h.numel()], -both[h.numel():]
    if not torch.equal(lo, hi):
        raise RuntimeError("candidate-sharded planning needs identical obs / history / warm start on every rank of the "
                           "group (checksums differ: %s vs %s)" % (lo.tolist(), hi.tolist()))


class ExternalAllGather:
    """The all-gather the library calls where the RCCL path calls ncclAllGather (include/cadm_hip.h: cadm_allgather_fn), over a
    torch.distributed group of any backend.  `resolve(ptr, nbytes)` maps a raw pointer of the call to a float32 torch tensor viewing
    that memory -- both pointers lie inside the planner's workspace, which the caller owns as a tensor.  Backend "nccl": the collective
    is enqueued on the current stream like every kernel of the plan.  Any other backend (gloo): the payload -- m * n/G + 1 floats per
    rank -- takes a synchronous round trip through host memory.  An exception inside the callback cannot cross the C frames: it is kept
    in `.error` (the library call then fails and `HipEngine._check` re-raises it)."""

    def __init__(self, group, resolve):
        import torch.distributed as dist
        from . import _lib
        self.group, self.resolve = group, resolve
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.error, self.calls = None, 0
        self.cfunc = _lib.ALLGATHER_FN(self._call)      # (kept alive with the object: the library holds the raw function pointer)

    def _call(self, user, send, recv, count, stream):
        try:
            self.gather(self.resolve(send, 4 * count), self.resolve(recv, 4 * count * self.world))
            self.calls += 1
            return 0
        except BaseException as exc:      # noqa: B902 -- nothing may propagate into the C caller
            self.error = exc
            return 1

    def gather(self, send, recv):
        """[count] floats of every rank -> [world * count] floats, rank-major, on every rank"""
        import torch.distributed as dist
        if send.is_cuda and self.backend != "nccl":
            host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(host, send.cpu(), group=self.group)      # (.cpu() waits for the stream's work on `send`)
            recv.copy_(host)
        else:
            dist.all_gather_into_tensor(recv, send.contiguous(), group=self.group)


def cem_plan(engine, obs, cp_obs, cp_act, init_mean, init_var, n, seed=0, call=0, z=None, eps=None, return_info=False):
    """Single rank, one iteration at a time over the engine's primitives.  z [iters,m,n,H,A] / eps [iters,H,m,n,p,D]: optional injected draws."""
    obs = engine._t(obs)
    mean = engine._t(init_mean).clone()
    var = engine._t(init_var).clone()
    ctx_vec = engine.context_forward(cp_obs, cp_act) if engine.C > 0 else None
    info = []
    for it in range(engine.num_cem_iters):
        actions = engine.sample_actions(mean, var, n, z=None if z is None else z[it], seed=seed, call=call, it=it)
        rows = engine.rollout_returns(obs, ctx_vec, actions, eps=None if eps is None else eps[it], seed=seed, call=call, it=it)
        cand = engine.particle_mean(rows).unsqueeze(0)
        elites = engine.cem_refit(cand, actions, mean, var, want_elites=return_info)
        if return_info:
            info.append(dict(actions=actions, rows=rows, cand=cand, elites=elites, mean=mean.clone(), var=var.clone()))
    plan = mean if engine.discrete else mean.clamp(-1.0, 1.0)   # dynamics.py:365-366
    return (plan, info, ctx_vec) if return_info else plan


def rs_plan(engine, obs, cp_obs, cp_act, n, seed=0, call=0, actions=None, raw=None, eps=None):
    obs = engine._t(obs)
    m = obs.shape[0]
    ctx_vec = engine.context_forward(cp_obs, cp_act) if engine.C > 0 else None
    if actions is None:
        actions, raw = engine.sample_uniform(m, n, seed=seed, call=call)
    else:
        actions = engine._t(actions)
    rows = engine.rollout_returns(obs, ctx_vec, actions, eps=eps, norm_actions=not engine.discrete, seed=seed, call=call, it=0)
    cand = engine.particle_mean(rows).unsqueeze(0)
    first, best = engine.rs_select(cand, actions)
    if engine.discrete:
        raw = engine._t(raw, dtype=torch.int32)
        return raw[torch.arange(m, device=raw.device), best.long(), 0], cand
    return first.clamp(-1.0, 1.0), cand
