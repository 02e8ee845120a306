"""A/B check of the reduced-camera solve: matrix-core kernels (default) vs the register-tile kernels (SSBA_BCR_LEGACY=1).
Runs the LM step of a few problem sizes through the C ABI and compares the pose step with a dense numpy solve of the
reduced system the library reports.  python tools/bcr_ab.py [child]"""
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    from ceres_slam_amd import synth
    from ceres_slam_amd.solver import StereoBA
    out = {}
    for name, (P, L) in {"p50": (50, 2000), "p200": (200, 8000), "p420": (420, 16000), "p1000": (1000, 20000)}.items():
        prob = synth.make_problem(P, L, track_len=12, seed=3)
        ba = StereoBA.from_synth(prob)
        for radius in (1e4, 30.0):
            S, rhs, dp, dl, mcc = ba.lm_step(radius)
            x = np.linalg.solve(S, rhs)
            err = np.abs(dp[1:].ravel() - x).max() / np.abs(x).max()
            out[f"{name}_r{radius:g}"] = err
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for legacy in ("0", "1"):
            env = dict(os.environ, SSBA_BCR_LEGACY=legacy)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print("legacy" if legacy == "1" else "mfma  ", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:])
