"""Diagnostic of the wide (144-row super-block) path: reduced system, right-hand side and LM step against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ceres_slam_amd import synth
from ceres_slam_amd.solver import StereoBA
from oracle import oracle as orc

def rel(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
for size in [(14, 300, 13), (30, 900, 24), (75, 1800, 20), (200, 3000, 17)]:
    prob = synth.make_problem(size[0], size[1], track_len=size[2], seed=3)
    ba = StereoBA.from_synth(prob)
    st = ba.stats()
    op = orc.OracleProblem.from_synth(prob)
    for radius in (1e4,):
        try:
            S, rhs, dp, dl, mcc = ba.lm_step(radius)
        except Exception as e:
            print(size, "lm_step failed:", e); continue
        S2, rhs2, _ = op.reduced_system(radius)
        dp2, dl2, mcc2 = op.lm_step(radius)
        n = S.shape[0]
        x = np.linalg.solve(S, rhs)
        print(size, "wide", st.wide_superblocks, "windows", st.num_windows, "S", rel(S, S2), "rhs", rel(rhs, rhs2), "dp", rel(dp, dp2),
              "dp vs numpy solve of own S", rel(dp[1:].ravel()[:n], x) if dp.shape[0] * 6 != n else rel(dp.ravel(), x), "dl", rel(dl, dl2), "mcc", mcc, mcc2)
        if rel(S, S2) > 1e-8:
            D = np.abs(S - S2)
            bad = np.argwhere(D > 1e-8 * np.abs(S2).max())
            print("   bad entries", len(bad), "first", bad[:10].tolist(), "rows range", bad[:, 0].min(), bad[:, 0].max(), "cols", bad[:, 1].min(), bad[:, 1].max())
            print("   sym err", np.abs(S - S.T).max())
