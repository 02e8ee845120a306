#!/usr/bin/env python3
"""Per-rank compute time of one trust-region iteration in an N-rank job, measured on ONE GPU.

One process plays rank r of N: it holds the rank's landmark shard of the N x C2 problem and runs the real
kernel sequence, with an exchange callback that does nothing (the sums over ranks are missing, so the
iterates are meaningless, but every kernel does its normal work on its normal sizes).  What is left out is
exactly the collective time; what is measured is the GPU + launch time per iteration that each rank pays:
  * "partitioned": chain elimination with pinned shared ends + separator system (ssba_set_partition),
  * "all-reduce":  the whole N x 84 super-block system solved by every rank.
usage: python tools/rank_compute_time.py [--partitioned-only] [N ...]   (--partitioned-only: for a kernel trace of that mode alone)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_slam_amd import capi, sharding, synth  # noqa: E402
from ceres_slam_amd.solver import StereoBA  # noqa: E402


def measure(world, rank, partitioned, steps=60):
    P1, L1 = synth.CONFIGS["C2"]
    prob = synth.make_problem(P1 * world, L1 * world)
    cut = sharding.aligned_partition(prob.obs_pose, prob.obs_point, prob.num_poses, prob.num_points, world) if world > 1 else None
    if world > 1:
        shard = sharding.shard_by_landmarks(prob, world, rank, ranges=cut[0] if (partitioned and cut) else None)
    else:
        shard = sharding.whole(prob)
    ba = StereoBA(prob.camera, shard.poses, shard.points, shard.obs_pose, shard.obs_point, shard.obs_uvd, prob.stiffness(),
                  world_size=world, rank=rank, partition=cut[1] if (partitioned and cut is not None) else None)
    if world > 1:
        ba.set_exchange(lambda ptr, count, op: None)
    opts = capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1)
    ba.solve_begin(opts, ignore_convergence=True)
    ba.step(10)
    ba.synchronize()
    ba.restart()
    ba.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if i and i % 20 == 0:
            ba.restart()
        ba.step(1)
    ba.synchronize()
    dt = time.perf_counter() - t0
    n_x = ba.exchange_size() if world > 1 else 0
    ba.solve_end()
    ba.close()
    return 1e3 * dt / steps, n_x


if __name__ == "__main__":
    part_only = "--partitioned-only" in sys.argv
    worlds = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 2, 4, 8]
    rows = []
    for w in worlds:
        r = w // 2            # a middle rank: two shared separators
        if w == 1:
            ms, nx = measure(1, 0, False)
            rows.append(dict(world=1, mode="single GPU", ms_per_iteration=ms, exchange_doubles=0))
        else:
            for part in ((True,) if part_only else (True, False)):
                ms, nx = measure(w, r, part)
                rows.append(dict(world=w, rank=r, mode="partitioned" if part else "all-reduce", ms_per_iteration=ms, exchange_doubles=int(nx)))
        print(json.dumps(rows[-1] if (w == 1 or part_only) else rows[-2:]), flush=True)
    print(json.dumps({"rank_compute_time": rows}))
