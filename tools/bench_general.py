"""Timing of the general-structure path (ssba_dense.hip): solver iterations per second on problems the windowed
layout cannot hold (long tracks), and on C2 itself forced onto the dense reduced system, next to the windowed path
on the same problem.  One JSON line per case.

    python tools/bench_general.py [--steps 30]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


TRACE = None


def _trace(msg):
    """--trace FILE: progress marks + the process's memory map (so that the raw addresses of a native stack dump -- e.g. the
    one rocprofv3's signal handler prints -- can be resolved to library + offset afterwards)."""
    if not TRACE:
        return
    with open(TRACE, "a") as fh:
        fh.write(f"[{time.time():.3f}] {msg}\n")
    with open(TRACE + ".maps", "w") as fh:
        fh.write(open("/proc/self/maps").read())


def run_case(name, prob, steps, force_dense):
    from ceres_slam_amd import capi
    from ceres_slam_amd.solver import StereoBA
    if force_dense:
        os.environ["SSBA_FORCE_DENSE"] = "1"
    try:
        ba = StereoBA.from_synth(prob)
    finally:
        os.environ.pop("SSBA_FORCE_DENSE", None)
    st = ba.stats()
    opts = capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1)
    _trace("finalized " + name)
    s, _ = ba.solve(opts)
    _trace(f"solved {name}: {int(s.num_iterations)} iterations")
    period = max(int(s.num_iterations) - 1, 1)
    ba.poses[:] = prob.poses_init
    ba.points[:] = prob.points_init
    ba.solve_begin(opts, ignore_convergence=True)
    _trace("solve_begin done")
    ba.step(3)
    ba.synchronize()
    _trace("three steps done")
    ba.restart()
    def loop():
        ba.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            if i and i % period == 0:
                ba.restart()
            if TRACE:
                with open(TRACE, "a") as fh:
                    fh.write(f"step {i}\n")
            ba.step(1)
        ba.synchronize()
        return time.perf_counter() - t0

    dt = loop()                       # graph replay, no event bracketing
    ba.restart()
    ba.set_kernel_timing(True)        # per-kernel-class HIP events (slower: two events per launch)
    loop()
    rows = {k: round(ms / steps, 4) for k, (n, ms) in ba.kernel_times().items() if n}
    ba.solve_end()
    print(json.dumps({"case": name, "general_structure": int(st.general_structure), "free_poses": int(st.num_free_poses),
                      "observations": int(st.num_observations), "reduced_blocks": int(st.num_reduced_blocks),
                      "solve_iterations": int(s.num_iterations), "final_cost": float(s.final_cost),
                      "ms_per_iteration": round(1e3 * dt / steps, 4), "iterations_per_s": round(steps / dt, 2),
                      "kernel_ms_per_iteration": rows,
                      "stats": {k: int(getattr(st, k)) for k in ("num_poses", "num_free_poses", "num_points", "num_active_points", "num_observations",
                                                                  "num_windows", "num_superblocks", "num_reduced_blocks", "pose_bandwidth",
                                                                  "general_structure", "wide_superblocks")}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--case", default="", help="run only the cases whose name contains this")
    ap.add_argument("--trace", default="", help="file for progress marks; FILE.maps gets /proc/self/maps at each mark")
    args = ap.parse_args()
    global run_case, TRACE
    TRACE = args.trace
    _run = run_case

    def run_case(name, make, steps, force):       # problems are built lazily: a filtered-out case costs nothing
        if args.case in name:
            _run(name, make(), steps, force)

    from ceres_slam_amd import synth
    run_case("P200_L20000_track24", lambda: synth.make_problem(200, 20000, track_len=24), args.steps, False)
    run_case("P600_L60000_track24", lambda: synth.make_problem(600, 60000, track_len=24), args.steps, False)
    # C2: 1000 states on a circle of 1000 x 0.5 m, so the last states see the landmarks of the first ones
    run_case("C2_loop_closure", lambda: synth.add_loop_closure(synth.make_config("C2"), num_landmarks=300), args.steps, False)
    # the same closure on landmarks that keep <= 12 observations: the closing states ride as a border of the
    # block-tridiagonal system (general_structure == 2, windowed kernels + ssba_border.hip)
    run_case("C2_loop_closure_border", lambda: synth.add_loop_closure(synth.make_config("C2"), num_landmarks=300, max_track=12), args.steps, False)
    run_case("C4_loop_closure_border", lambda: synth.add_loop_closure(synth.make_problem(10000, 1000000, track_len=12, seed=21), num_landmarks=300, max_track=12),
             args.steps, False)
    run_case("C2_windowed", lambda: synth.make_config("C2"), args.steps, False)
    run_case("C2_forced_dense", lambda: synth.make_config("C2"), args.steps, True)


if __name__ == "__main__":
    main()
