"""Wall time of one drop-in call sequence at a given size: build the handle (host copies), ssba_finalize (layout
+ uploads), ssba_solve to convergence (includes the log / summary read-back), parameter read-back, destroy --
the whole of what replaces problem construction + ceres::Solve in a driver.  The second round reuses the
process-wide buffer pool.

usage: python tools/time_end_to_end.py [C2|C1|P,L] [rounds]      (SSBA_API_TIMING=1 prints the per-entry-point times)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ceres_slam_amd import capi, synth  # noqa: E402
from ceres_slam_amd.solver import StereoBA  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if "," in cfg:
        P, L = (int(v) for v in cfg.split(","))
        prob = synth.make_problem(P, L, track_len=12, seed=42)
    else:
        prob = synth.make_config(cfg)
    rows = []
    for r in range(rounds):
        t0 = time.perf_counter()
        ba = StereoBA.from_synth(prob)
        t1 = time.perf_counter()
        ba.finalize()
        t2 = time.perf_counter()
        s, log = ba.solve(capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1))
        t3 = time.perf_counter()
        poses, points = ba.poses, ba.points
        t4 = time.perf_counter()
        ba.close()
        t5 = time.perf_counter()
        rows.append(dict(round=r, build_ms=(t1 - t0) * 1e3, finalize_ms=(t2 - t1) * 1e3, solve_ms=(t3 - t2) * 1e3,
                         readback_ms=(t4 - t3) * 1e3, destroy_ms=(t5 - t4) * 1e3, total_ms=(t5 - t0) * 1e3,
                         iterations=int(s.num_iterations), final_cost=float(s.final_cost)))
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
