"""Landmark sharding for multi-GPU runs.

Each stereo residual touches one pose and one landmark (tests/dataset_vo.cpp:51-53 in the
reference), so giving every rank a contiguous range of landmarks together with all their
observations makes the landmark blocks, their elimination and the back-substitution purely
local; only the reduced pose system has to be summed over ranks -- one all-reduce of the
block-tridiagonal exchange vector per iteration plus two scalar-sized ones.  Poses are
replicated.  Balance is by observation count.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Shard:
    poses: np.ndarray          # (P,12) all poses (replicated)
    points: np.ndarray         # (L_local,3)
    obs_pose: np.ndarray
    obs_point: np.ndarray      # local landmark index
    obs_uvd: np.ndarray
    point_ids: np.ndarray      # local -> global landmark index
    poses_init: np.ndarray
    points_init: np.ndarray


def landmark_ranges(obs_point: np.ndarray, num_points: int, world: int):
    """Contiguous landmark ranges with ~equal observation counts: list of (begin, end)."""
    cnt = np.bincount(obs_point, minlength=num_points).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(cnt)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r / world, side="left")))
    cuts.append(num_points)
    cuts = np.maximum.accumulate(np.asarray(cuts))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def whole(prob) -> Shard:
    return Shard(prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd,
                 np.arange(prob.num_points), prob.poses_init.copy(), prob.points_init.copy())


def shard_by_landmarks(prob, world: int, rank: int) -> Shard:
    b, e = landmark_ranges(prob.obs_point, prob.num_points, world)[rank]
    sel = (prob.obs_point >= b) & (prob.obs_point < e)
    pts = prob.points_init[b:e].copy()
    return Shard(prob.poses_init.copy(), pts, prob.obs_pose[sel], (prob.obs_point[sel] - b).astype(np.uint32),
                 prob.obs_uvd[sel], np.arange(b, e), prob.poses_init.copy(), pts.copy())


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def attach_torch_exchange(ba, dist, group=None):
    """Route the library's exchange points through torch.distributed (backend "nccl" = RCCL
    over xGMI on ROCm; "gloo" works for tests).  The library must run on torch's current
    stream (StereoBA.set_stream) so that the collective is ordered after the kernels that
    fill the buffer."""
    import torch
    cache = {}

    def exchange(ptr: int, count: int, op: int):
        key = (ptr, count)
        t = cache.get(key)
        if t is None:
            t = torch.as_tensor(_DevArray(ptr, count), device="cuda")
            cache[key] = t
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)

    ba.set_exchange(exchange)
    return exchange
