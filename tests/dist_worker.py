"""Worker for the multi-process tests (one process per rank; RANK / WORLD_SIZE / MASTER_* in env).

mode "cpu": gloo on CPU tensors -- the sharding logic + the all-reduce of the shard-local
            normal-equation sums, with the CPU oracle as the compute (no GPU needed);
mode "gpu": every rank drives the HIP library on its landmark shard (all ranks on cuda:0 when
            only one GPU exists) and the exchange points go through torch.distributed.
Each rank writes a JSON result file: <out>.<rank>.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    from ceres_slam_amd import sharding, synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    P, Lm, T = [int(v) for v in os.environ.get("SSBA_TEST_SIZE", "16,400,6").split(",")]
    huber_a = float(os.environ.get("SSBA_TEST_HUBER", "0"))      # with it: 30 % outlier observations (BASELINE.json configs[4])
    lighting = None
    shared_free = 0
    pose_factors = None
    use_bounds = False
    if mode.endswith("_phong") or mode.endswith("_phongfree") or mode.endswith("_phongfreeb"):     # BASELINE.json configs[2] sharded: lighting terms of a landmark live on its rank
        use_bounds = mode.endswith("b")        # tests/dataset_ba_phong.cpp:142-180: bounds on the Phong and texture blocks -> projected line search
        shared_free = 7 if (mode.endswith("free") or use_bounds) else 0
        mode = mode[:mode.rindex("_")]
        prob, ph = synth.make_phong_problem(P, Lm, track_len=T, seed=21)
        # shared light / Phong / texture blocks constant, or free (their border sums ride next to the reduced system)
        lighting = ph.as_oracle_dict("perturbed" if shared_free else "truth")
    elif mode.endswith("_sun"):        # unary pose residual blocks (pose prior + sun sensor, tests/dataset_vo_sun.cpp:80-124), no constant pose
        mode = mode[:mode.rindex("_")]
        from test_oracle_pose_factors import _sun_problem
        prob, pose_factors = _sun_problem(P=P, L=Lm, seed=5, huber=huber_a)
        huber_a = 0.0
    else:
        prob = synth.make_problem(P, Lm, track_len=T, seed=21, outlier_fraction=0.3 if huber_a > 0 else 0.0)
    partition = None
    if mode == "gpu_part":      # super-block-aligned landmark ranges + partitioned reduced solve
        cut = sharding.aligned_partition(prob.obs_pose, prob.obs_point, prob.num_poses, prob.num_points, world)
        if cut is None:         # no super-block-aligned cut exists: plain landmark sharding, whole reduced system summed (as bench.py does)
            shard = sharding.shard_by_landmarks(prob, world, rank)
        else:
            ranges, partition = cut
            shard = sharding.shard_by_landmarks(prob, world, rank, ranges=ranges)
    else:
        shard = sharding.shard_by_landmarks(prob, world, rank)
    res = {"rank": rank, "num_local_obs": int(shard.obs_pose.shape[0]), "num_local_points": int(shard.points.shape[0])}
    if mode == "cpu":
        from oracle import oracle as orc     # checker as compute: this is a test of the host logic
        op = orc.OracleProblem(prob.camera, shard.poses, shard.points, shard.obs_pose, shard.obs_point,
                               shard.obs_uvd, prob.stiffness())
        cost, g_p, g_l, H_pp, H_ll = op.linearize()
        buf = torch.from_numpy(np.concatenate([[cost], g_p.ravel(), H_pp.ravel()]))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        gmax = torch.tensor([np.abs(g_l).max()], dtype=torch.float64)
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
        res.update(cost=float(buf[0]), g_p=buf[1:1 + g_p.size].tolist(), H_pp_trace=float(np.trace(buf[1 + g_p.size:].numpy().reshape(-1, 6, 6).sum(0))),
                   gmax_l=float(gmax[0]))
    else:
        from ceres_slam_amd import capi
        from ceres_slam_amd.solver import StereoBA
        torch.cuda.set_device(0)
        lt = None
        if lighting is not None:        # this rank's landmarks and observations
            sel = np.isin(prob.obs_point, shard.point_ids)
            lt = dict(lighting, normals=lighting["normals"][shard.point_ids], material_of_point=lighting["material_of_point"][shard.point_ids],
                      intensity=lighting["intensity"][sel], normal_obs=lighting["normal_obs"][sel])
        ba = StereoBA(prob.camera, shard.poses, shard.points, shard.obs_pose, shard.obs_point, shard.obs_uvd,
                      prob.stiffness(), device=0, world_size=world, rank=rank, partition=partition, huber_a=huber_a, lighting=lt,
                      shared_free=shared_free, use_bounds=use_bounds, pose_factors=pose_factors,
                      pose_const=np.zeros(prob.num_poses, dtype=np.uint8) if pose_factors else None)
        sharding.attach_torch_exchange(ba, dist)
        okw = dict(max_num_iterations=int(os.environ.get("SSBA_TEST_MAXIT", "1000")), use_nonmonotonic_steps=1)
        if "SSBA_TEST_DOGLEG" in os.environ:        # 0 TRADITIONAL_DOGLEG, 1 SUBSPACE_DOGLEG (tests/dataset_ba_phong.cpp:85-86)
            okw.update(trust_region_strategy_type=1, dogleg_type=int(os.environ["SSBA_TEST_DOGLEG"]))
        s, log = ba.solve(capi.default_options(**okw))
        res.update(termination=int(s.termination_type), num_iterations=int(s.num_iterations),
                   final_cost=float(s.final_cost), initial_cost=float(s.initial_cost), cost=log["cost"].tolist(),
                   poses=ba.poses.tolist(), points=ba.points.tolist(), point_ids=shard.point_ids.tolist(),
                   partition=None if partition is None else partition.tolist(), accept=log["step_is_successful"].tolist(),
                   gmax=log["gradient_max_norm"].tolist(), step_norm=log["step_norm"].tolist())
        if use_bounds:
            res["line_search_steps"] = int(s.num_line_search_steps)
        if lt is not None:
            res["normals"] = ba.normals.tolist()
            res["light"] = np.asarray(ba.light).tolist()
            res["phong"] = np.asarray(ba.phong).tolist()
            res["texture"] = np.asarray(ba.texture).tolist()
    with open(f"{out}.{rank}.json", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
