"""Host logic of the rollout launcher (cadm_amd/csrc/xdl_geo.h: xdl_plan_units): how a member's row tiles are cut between the kernel
flavours -- cooperative kernel with one / two row tiles per workgroup, wave-tile kernel with 4 / 8 tiles per workgroup -- by a small
dynamic programme over measured costs (profiles/r4_s3_flavour_table.txt).  Runs on the CPU: the developer library exports the plan
as a pure host function (no ctx, no device)."""
import ctypes as ct

import pytest

from cadm_amd import _lib

CAP = (1, 2, 4, 8)
COST = (1.0, 1.59, 3.2, 5.19)          # halfcheetah, HID 200 (xdl_geo.h: xdl_costs; profiles/r6_flavour_table.txt)
WT8P = (4.48, 4.52, 4.84)              # a wave-tile-8 round that covers only 5 / 6 / 7 units (its workgroups run fewer waves)


def plan(units, two=True, wave=True):
    out = (ct.c_int * 4)()
    rc = _lib.load_dev().cadm_dev_rollout_plan(units, int(two), int(wave), out)
    assert rc == 0
    return tuple(out)


def plan_cost(cnt, units, cost=None, wt8p=None):
    """what the launcher's launches cost: biggest flavour first, the last launch of a flavour takes the ragged rest; a wave-tile-8 round
    that ends up covering 5 - 7 units costs less than a full one"""
    cost, wt8p = cost or COST, wt8p or WT8P
    c0, c1, c2, c3 = cnt
    total = c0 * cost[0] + c1 * cost[1] + c2 * cost[2]
    if c3:
        rem8 = units - (c0 + 2 * c1 + 4 * c2)                  # what the wave-tile-8 launch has to cover
        if rem8 <= 8 * (c3 - 1):
            return None                                        # a round too many
        last = min(8, rem8 - 8 * (c3 - 1))
        total += (c3 - 1) * cost[3] + (wt8p[last - 5] if 5 <= last <= 7 else cost[3])
    return total


def brute(units, two=True, wave=True, cost=None, wt8p=None):
    """cheapest cover by exhaustive search over launch counts"""
    best = None
    ok = (True, two, wave, wave)
    for c3 in range(units // 8 + 2):
        for c2 in range(units // 4 + 2):
            for c1 in range(units // 2 + 2):
                cov = 8 * c3 + 4 * c2 + 2 * c1
                c0 = max(0, units - cov)
                cnt = (c0, c1, c2, c3)
                if any(n and not k for n, k in zip(cnt, ok)):
                    continue
                c = plan_cost(cnt, units, cost, wt8p)
                if c is not None and (best is None or c < best - 1e-6):
                    best = c
    return best


def test_named_workloads():
    assert plan(1) == (1, 0, 0, 0)                 # cfg2: 50 tiles per member on 51 CUs -> one cooperative one-tile launch
    assert plan(2) == (0, 1, 0, 0)
    assert plan(5) == (1, 2, 0, 0)                 # cfg5 per GPU (250 tiles per member): two two-tile rounds + a one-tile launch
    assert plan(6) == (0, 0, 0, 1)                 # ONE partly filled wave-tile round (741.8 us measured against 776.9 for three two-tile launches)
    assert plan(8) == (0, 0, 0, 1)                 # one full wave-tile round
    assert plan(10) == (0, 1, 0, 1)                # cfg3 / m = 10 (500 tiles per member): a wave-tile round + a two-tile launch for the rest
    assert plan(14) == (0, 0, 0, 2)                # a full round + a partial one of the same launch
    assert plan(0) == (0, 0, 0, 0)


@pytest.mark.parametrize("two,wave", [(True, True), (True, False), (False, True), (False, False)])
def test_plan_covers_and_is_cheapest(two, wave):
    ok = (True, two, wave, wave)
    for units in range(1, 41):
        cnt = plan(units, two, wave)
        assert all(n >= 0 for n in cnt)
        assert not any(n and not k for n, k in zip(cnt, ok)), "a flavour that does not exist for the geometry was planned"
        assert sum(n * c for n, c in zip(cnt, CAP)) >= units, "the plan does not cover the member's tiles"
        cost = plan_cost(cnt, units)
        assert cost is not None and cost <= brute(units, two, wave) + 1e-4, (units, cnt, cost)


def test_cost_table_is_per_instantiation():
    """VERDICT r5 #3a: the plan's costs are a constexpr of the launcher's instantiation (xdl_geo.h: xdl_costs(env, hid)), not four global
    constants: slim humanoid's measured table differs from halfcheetah's, and each table's plan is the cheapest cover under ITS costs."""
    lib = _lib.load_dev()
    tables = {}
    for env_kind, hid in ((0, 200), (2, 200), (1, 256)):
        cnt, costs = (ct.c_int * 4)(), (ct.c_float * 7)()
        assert lib.cadm_dev_rollout_plan_for(5, 1, 1, env_kind, hid, cnt, costs) == 0
        tab = tuple(round(float(c), 3) for c in costs)
        tables[(env_kind, hid)] = tab
        for units in range(1, 33):
            assert lib.cadm_dev_rollout_plan_for(units, 1, 1, env_kind, hid, cnt, None) == 0
            got = tuple(cnt)
            assert sum(n * c for n, c in zip(got, CAP)) >= units
            c = plan_cost(got, units, tab[:4], tab[4:])
            assert c is not None and c <= brute(units, cost=tab[:4], wt8p=tab[4:]) + 1e-4, (env_kind, hid, units, got)
    assert tables[(0, 200)] == COST + WT8P
    assert tables[(2, 200)] == (1.0, 1.62, 3.62, 5.5, 4.78, 4.8, 5.15)      # slim humanoid
    assert tables[(1, 256)] == tables[(0, 200)]               # unmeasured geometries use halfcheetah's
