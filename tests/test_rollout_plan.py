"""Host logic of the rollout launcher (cadm_amd/csrc/xdl_geo.h: xdl_plan_units): how a member's row tiles are cut between the kernel
flavours -- cooperative kernel with one / two row tiles per workgroup, wave-tile kernel with 4 / 8 tiles per workgroup -- by a small
dynamic programme over measured costs (profiles/r4_s3_flavour_table.txt).  Runs on the CPU: the developer library exports the plan
as a pure host function (no ctx, no device)."""
import ctypes as ct

import pytest

from cadm_amd import _lib

CAP = (1, 2, 4, 8)
COST = (1.0, 1.6, 3.6, 5.3)


def plan(units, two=True, wave=True):
    out = (ct.c_int * 4)()
    rc = _lib.load_dev().cadm_dev_rollout_plan(units, int(two), int(wave), out)
    assert rc == 0
    return tuple(out)


def brute(units, two=True, wave=True):
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
                cost = sum(n * c for n, c in zip(cnt, COST))
                if best is None or cost < best - 1e-6:
                    best = cost
    return best


def test_named_workloads():
    assert plan(1) == (1, 0, 0, 0)                 # cfg2: 50 tiles per member on 51 CUs -> one cooperative one-tile launch
    assert plan(2) == (0, 1, 0, 0)
    assert plan(5) == (1, 2, 0, 0)                 # cfg5 per GPU (250 tiles per member): two two-tile rounds + a one-tile launch
    assert plan(8) == (0, 0, 0, 1)                 # one full wave-tile round
    assert plan(10) == (0, 1, 0, 1)                # cfg3 / m = 10 (500 tiles per member): a wave-tile round + a two-tile launch for the rest
    assert plan(0) == (0, 0, 0, 0)


@pytest.mark.parametrize("two,wave", [(True, True), (True, False), (False, True), (False, False)])
def test_plan_covers_and_is_cheapest(two, wave):
    ok = (True, two, wave, wave)
    for units in range(1, 41):
        cnt = plan(units, two, wave)
        assert all(n >= 0 for n in cnt)
        assert not any(n and not k for n, k in zip(cnt, ok)), "a flavour that does not exist for the geometry was planned"
        assert sum(n * c for n, c in zip(cnt, CAP)) >= units, "the plan does not cover the member's tiles"
        cost = sum(n * c for n, c in zip(cnt, COST))
        assert cost <= brute(units, two, wave) + 1e-4, (units, cnt, cost)


def test_huge_batches_stay_bounded():
    for units in (65, 100, 1000, 20000):
        cnt = plan(units)
        assert sum(n * c for n, c in zip(cnt, CAP)) >= units
        assert cnt[3] >= (units - 64) // 8          # beyond the table: rounds of the biggest flavour
        cnt = plan(units, wave=False)
        assert cnt[2] == cnt[3] == 0 and sum(n * c for n, c in zip(cnt, CAP)) >= units


def test_cost_table_is_per_instantiation():
    """VERDICT r5 #3a: the plan's costs are a constexpr of the launcher's instantiation (xdl_geo.h: xdl_costs(env, hid)), not four global
    constants: slim humanoid's measured table differs from halfcheetah's, and each table's plan is the cheapest cover under ITS costs."""
    global COST
    lib = _lib.load_dev()
    tables = {}
    for env_kind, hid in ((0, 200), (2, 200), (1, 256)):
        cnt, costs = (ct.c_int * 4)(), (ct.c_float * 4)()
        assert lib.cadm_dev_rollout_plan_for(5, 1, 1, env_kind, hid, cnt, costs) == 0
        tables[(env_kind, hid)] = tuple(round(float(c), 3) for c in costs)
        saved = COST
        try:
            COST = tuple(float(c) for c in costs)
            for units in range(1, 33):
                assert lib.cadm_dev_rollout_plan_for(units, 1, 1, env_kind, hid, cnt, None) == 0
                got = tuple(cnt)
                assert sum(n * c for n, c in zip(got, CAP)) >= units
                assert sum(n * c for n, c in zip(got, COST)) <= brute(units) + 1e-4, (env_kind, hid, units, got)
        finally:
            COST = saved
    assert tables[(0, 200)] == (1.0, 1.6, 3.6, 5.3)
    assert tables[(2, 200)] == (1.0, 1.74, 3.8, 5.5)          # slim humanoid (profiles/r5: measured 1 / 1.74 / 3.8 / 5.5)
    assert tables[(1, 256)] == tables[(0, 200)]               # unmeasured geometries use halfcheetah's
