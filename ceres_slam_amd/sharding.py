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


def aligned_partition(obs_pose: np.ndarray, obs_point: np.ndarray, num_poses: int, num_points: int, world: int, pose_const=None,
                      sbp: int = 12):
    """Landmark ranges cut where the co-visibility band crosses a super-block boundary, for the partitioned
    reduced solve (ssba_set_partition).  Landmarks must be ordered along the trajectory (as the reference's
    datasets and synth.py are: by first observing state).  Returns (ranges, separator_superblocks) or None if no
    such cut exists near the balanced one (the caller then uses the plain all-reduce path).

    Free-pose index f = rank of the pose among the non-constant poses; super-block = f // sbp.  Rank r may only
    touch super-blocks [sep[r], sep[r+1]] and neighbours share exactly the super-block sep[r+1]."""
    if pose_const is None:
        pose_const = np.zeros(num_poses, dtype=bool)
        pose_const[0] = True
    free_idx = np.full(num_poses, -1, dtype=np.int64)
    free_idx[~np.asarray(pose_const, dtype=bool)] = np.arange(int((~np.asarray(pose_const, dtype=bool)).sum()))
    nsb = max(1, (int(free_idx.max()) + 1 + sbp - 1) // sbp)
    f = free_idx[obs_pose]
    ok = f >= 0
    lo = np.full(num_points, np.iinfo(np.int64).max, dtype=np.int64)
    hi = np.full(num_points, -1, dtype=np.int64)
    np.minimum.at(lo, obs_point[ok], f[ok] // sbp)
    np.maximum.at(hi, obs_point[ok], f[ok] // sbp)
    seen = hi >= 0
    # running envelopes: a cut before landmark c is valid for separator s iff every landmark < c stays in
    # super-blocks <= s and every landmark >= c stays in super-blocks >= s
    hi_prefix = np.maximum.accumulate(np.where(seen, hi, -1))
    lo_suffix = np.minimum.accumulate(np.where(seen, lo, np.iinfo(np.int64).max)[::-1])[::-1]
    balanced = landmark_ranges(obs_point, num_points, world)
    cuts, seps = [0], [0]
    for r in range(1, world):
        target = balanced[r][0]
        best = None
        for c in sorted(range(max(1, target - 4000), min(num_points, target + 4000)), key=lambda c: abs(c - target)):
            s = hi_prefix[c - 1]
            if s >= 0 and lo_suffix[c] >= s and s > seps[-1] and c > cuts[-1]:
                best = (c, int(s))
                break
        if best is None:
            return None
        cuts.append(best[0])
        seps.append(best[1])
    cuts.append(num_points)
    seps.append(nsb - 1)
    if seps[-1] <= seps[-2]:
        return None
    return [(cuts[r], cuts[r + 1]) for r in range(world)], np.asarray(seps, dtype=np.uint32)


def whole(prob) -> Shard:
    return Shard(prob.poses_init.copy(), prob.points_init.copy(), prob.obs_pose, prob.obs_point, prob.obs_uvd,
                 np.arange(prob.num_points), prob.poses_init.copy(), prob.points_init.copy())


def shard_by_landmarks(prob, world: int, rank: int, ranges=None) -> Shard:
    b, e = (ranges or landmark_ranges(prob.obs_point, prob.num_points, world))[rank]
    sel = (prob.obs_point >= b) & (prob.obs_point < e)
    pts = prob.points_init[b:e].copy()
    return Shard(prob.poses_init.copy(), pts, prob.obs_pose[sel], (prob.obs_point[sel] - b).astype(np.uint32),
                 prob.obs_uvd[sel], np.arange(b, e), prob.poses_init.copy(), pts.copy())


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def attach_torch_exchange(ba, dist, group=None):
    """Route the library's exchange points through torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
    "gloo" works for tests).  Kernels and collectives must be ordered on ONE stream: a dedicated torch stream
    is created, handed to the library (ssba_set_stream) and made current around every collective.  (The
    legacy default stream cannot be used: its handle is NULL, which ssba_set_stream reads as "the library's
    own non-blocking stream" -- the collectives would then race with the kernels that fill the buffers.)
    Returns the stream; synchronise it (or call StereoBA.synchronize) before reading results."""
    import torch
    stream = torch.cuda.Stream()
    assert stream.cuda_stream != 0
    ba.set_stream(stream.cuda_stream)
    cache = {}

    def exchange(ptr: int, count: int, op: int):
        key = (ptr, count)
        t = cache.get(key)
        if t is None:
            t = torch.as_tensor(_DevArray(ptr, count), device="cuda")
            cache[key] = t
        with torch.cuda.stream(stream):
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)

    ba.set_exchange(exchange)
    ba._exchange_stream = stream
    return stream


def attach_rccl_exchange(ba, dist=None, group=None):
    """Native exchange (ssba_set_rccl): the library calls ncclAllReduce itself on its own stream -- no Python between
    the kernel segments of an iteration.  Only the 128-byte RCCL unique id travels through `dist` (any
    torch.distributed backend; None for a single rank), once."""
    world = 1 if dist is None else dist.get_world_size(group)
    rank = 0 if dist is None else dist.get_rank(group)
    ids, err = [None], None
    if rank == 0:
        try:
            ids[0] = ba.rccl_unique_id()
        except Exception as e:      # noqa: BLE001 -- the other ranks sit in the broadcast below: they must get an answer
            err = e
    if world > 1:
        dist.broadcast_object_list(ids, src=0, group=group)
    if ids[0] is None:              # every rank leaves through the same sequence of collectives
        raise err if err is not None else RuntimeError("rank 0 could not create the RCCL unique id")
    ba.set_rccl(ids[0])
