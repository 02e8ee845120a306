"""The bookkeeping of tools/fuzz_parity.py on synthetic iteration logs (no GPU): the horizon on which two oracle runs
agree, the conditioned count, and the classification of a one-sided factorisation breakdown -- which once tested the
trust-region radius AFTER the iteration's update and so missed a breakdown at 1.8e9 (the halved radius is below 1e9)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity as fz  # noqa: E402


def _log(cost, ok, radius=None, step=None, change=None):
    n = len(cost)
    return {"cost": np.asarray(cost, float), "step_is_successful": np.asarray(ok, int),
            "trust_region_radius": np.asarray(radius if radius is not None else [1e4] * n, float),
            "step_norm": np.asarray(step if step is not None else [1.0] * n, float),
            "cost_change": np.asarray(change if change is not None else [1.0] * n, float)}


def test_horizon_ends_where_the_oracle_disagrees_with_itself():
    a = _log([10, 5, 3, 2, 1.5], [0, 1, 1, 1, 1])
    assert fz.horizon(a, _log([10, 5, 3, 2, 1.5], [0, 1, 1, 1, 1]), 5) == 5
    assert fz.horizon(a, _log([10, 5, 3 * (1 + 1e-8), 2, 1.5], [0, 1, 1, 1, 1]), 5) == 2
    assert fz.horizon(a, _log([10, 5, 3, 2, 1.5], [0, 1, 1, 0, 1]), 5) == 3
    # the cost of a REJECTED candidate is a far-off point: not compared
    assert fz.horizon(_log([10, 5, 99, 2], [0, 1, 0, 1]), _log([10, 5, 77, 2], [0, 1, 0, 1]), 4) == 4
    assert fz.horizon(a, _log([10 * (1 + 1e-6), 5, 3, 2, 1.5], [0, 1, 1, 1, 1]), 5) == 1      # never below 1


def test_conditioned_agreement_scales_with_the_oracles_own_spread():
    ref = _log([10, 5, 3, 2], [0, 1, 1, 1])
    probe = _log([10, 5 * (1 + 1e-8), 3 * (1 + 1e-8), 2 * (1 + 1e-8)], [0, 1, 1, 1])
    n, worst = fz.conditioned_agreement(_log([10, 5 * (1 + 2e-8), 3 * (1 - 1e-7), 2], [0, 1, 1, 1]), ref, [probe], 4)
    assert n == 4 and 0.3 < worst < 0.4                         # 1e-7 against an allowance of 30 x 1e-8
    n, _ = fz.conditioned_agreement(_log([10, 5, 3 * (1 + 1e-6), 2], [0, 1, 1, 1]), ref, [probe], 4)
    assert n == 2                                               # beyond the allowance: the count stops, no verdict
    n, _ = fz.conditioned_agreement(_log([10, 5, 3, 2], [0, 1, 0, 1]), ref, [probe], 4)
    assert n == 2                                               # another decision
    n, _ = fz.conditioned_agreement(_log([10, 5, 3, 2], [0, 1, 1, 1]), ref, [_log([10, 5, 3 * 1.001, 2], [0, 1, 1, 1])], 4)
    assert n == 2                                               # the oracle itself is undetermined there (spread > 1e-5)
    n, worst = fz.conditioned_agreement(_log([10, 5 * (1 + 5e-9), 3, 2], [0, 1, 1, 1]), ref, [ref], 4)
    assert n == 4 and abs(worst - 0.5) < 1e-6                   # floor of the allowance: 1e-8


def test_one_sided_breakdown_is_classified_by_the_radius_the_step_was_computed_with():
    fz.SUMMARY["breakdown_first"] = {"gpu": 0, "oracle": 0}
    radius_gpu = [1e4, 6e8, 1.8e9, 8.9e8, 2.7e9]                # entry i: the radius AFTER iteration i
    gpu = _log([10, 5, 3, 3, 2.9], [0, 1, 1, 0, 1], radius=radius_gpu, step=[0, 1, 1, 0.0, 1], change=[0, 5, 2, 0.0, 0.1])
    orc_ = _log([10, 5, 3, 2.9, 2.8], [0, 1, 1, 1, 1], radius=[1e4, 6e8, 1.8e9, 5.3e9, 1.6e10])
    assert fz.solver_breakdown(gpu, orc_, 5) == 3               # iteration 3 ran at 1.8e9 and produced no step on one side
    assert fz.SUMMARY["breakdown_first"] == {"gpu": 1, "oracle": 0}
    # an ordinary rejected step (a step was taken, the cost went up) is not a breakdown
    ordinary = _log([10, 5, 3, 3, 2.9], [0, 1, 1, 0, 1], radius=radius_gpu, step=[0, 1, 1, 0.7, 1], change=[0, 5, 2, -0.3, 0.1])
    assert fz.solver_breakdown(ordinary, orc_, 5) == 5
    # nor is an invalid step at a moderate radius (that would be a finding, and stays inside the comparison)
    small = _log([10, 5, 3, 3, 2.9], [0, 1, 1, 0, 1], radius=[1e4, 6e4, 1.8e5, 9e4, 2.7e5], step=[0, 1, 1, 0.0, 1], change=[0, 5, 2, 0.0, 0.1])
    assert fz.solver_breakdown(small, orc_, 5) == 5
    fz.SUMMARY["breakdown_first"] = {"gpu": 0, "oracle": 0}


def test_a_count_ended_by_the_hip_run_is_not_agreement():
    """conditioned_agreement says WHY it stopped; a case whose count the HIP run ended before MIN_HORIZON iterations -- the
    oracle runs still agreeing with each other -- is recorded as a mismatch even when the strict verdict over a shorter
    horizon was 'ok' (r03 printed such cases as ok and counted them as compared)."""
    ref = _log([10, 5, 3, 2, 1.5], [0, 1, 1, 1, 1])
    probe = _log([10, 5, 3, 2, 1.5], [0, 1, 1, 1, 1])
    n, _ = fz.conditioned_agreement(_log([10, 5, 3 * (1 + 1e-5), 2, 1.5], [0, 1, 1, 1, 1]), ref, [probe], 5)
    assert n == 2 and fz.LAST_STOP[0] == "hip"
    fz.SUMMARY["classes"] = {}
    fz.record("x", 1, True, n)          # strict horizon 1 (judged ok there), conditioned count 2, ended by the HIP run
    assert fz.SUMMARY["classes"]["x"]["mismatch"] == 1 and fz.SUMMARY["classes"]["x"]["hip_left"] == 1
    n, _ = fz.conditioned_agreement(ref, ref, [_log([10, 5, 3 * 1.001, 2, 1.5], [0, 1, 1, 1, 1])], 5)
    assert n == 2 and fz.LAST_STOP[0] == "oracle"
    fz.record("y", 1, True, n)          # the oracle itself is undetermined there: uncompared, not a mismatch
    assert fz.SUMMARY["classes"]["y"]["mismatch"] == 0 and fz.SUMMARY["classes"]["y"]["uncompared"] == 1
    n, _ = fz.conditioned_agreement(ref, ref, [probe], 5)
    assert n == 5 and fz.LAST_STOP[0] == "end"
    fz.SUMMARY["classes"] = {}
