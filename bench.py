#!/usr/bin/env python3
"""Bench harness: Gauss-Newton (Levenberg-Marquardt) iterations/s of the stereo-BA hot path.

Contract (one JSON line on rank 0):
  * a *step* is one trust-region iteration (linearise -> Schur -> reduced solve ->
    back-substitute -> candidate evaluation -> accept/reject) on the synthetic stereo-BA
    problem BASELINE.json quotes the metric on: C2 = 1 000 poses / 100 000 landmarks /
    ~1.19 M stereo observations, reprojection-only cost, fp64;
  * all inputs are resident in HBM before the timed region; the timed region runs the real
    solve repeatedly: whenever the solve has done as many iterations as the converging
    solve needs (measured in warm-up) the parameters are reset on the device and the
    solve restarts, so every timed iteration does the full work of a live iteration;
  * N > 1: `python bench.py --gpus N` starts its own N ranks as child processes (torch.distributed.run, one per GPU); under
    a launcher (WORLD_SIZE set) it is one of the ranks.  Landmarks are sharded, poses replicated; every rank eliminates
    its chain of the reduced system and only the separator blocks are summed over RCCL (all-reduce of the whole reduced
    system with SSBA_NO_PARTITION=1).  Workload: N = 2, 4 -> N x C2 (one C2-sized trajectory segment per rank); N = 8 ->
    BASELINE.json configs[3] exactly (C4 = 10 000 poses / 1 000 000 landmarks), with the 8 x C2 weak-scaling figure as the
    secondary key `weak_scaling_NxC2`.  `value` = joint-problem iterations/s x (landmarks / 100 000), i.e. C2-sized units
    of work per second; `config.joint_iters_per_sec` is the plain rate of the joint problem.  Every N > 1 line also carries
    `speedup_vs_1gpu_same_problem` (rank 0 measures the SAME joint problem on one GPU while the others wait: a larger
    problem scores more units per second on one GPU already, so value(N) / value(1) is not a scaling factor) and
    `strong_scaling_C2` (BASELINE.json's literal metric problem, 1 000 / 100 000, cut into N shards); `scaling` says
    "strong" for C4 and "weak" for N x C2.
  * `--config LT24`: the long-track workload (600 poses / 60 000 landmarks / 24 observations per landmark: the wide reduced
    system of ceres_slam_amd/csrc/ssba_wide.hip); `value` is then that problem's plain iterations/s.
  * `roofline`: the dominant kernel's algorithmic bytes or flops per launch / its average
    duration measured with HIP events on the library's stream during the timed region;
  * `cpu_baseline`: the CPU oracle (a port with Ceres-equivalent semantics, NOT Ceres --
    Ceres/Eigen are not installable here) timed on the host cores on the same problem.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP64_PEAK_TFLOPS = 78.6        # public MI355X spec, vector = matrix fp64


def bcr_levels(nsb):
    """Blocks per BCR level: n, ceil(n/2), ... , 1 (ceres_slam_amd/csrc/ssba_bcr.hip)."""
    out = [nsb]
    while out[-1] > 1:
        out.append((out[-1] + 1) // 2)
    return out


def algorithmic_work(stats, phong=False):
    """ALGORITHMIC bytes / flops of ONE launch of each kernel class:
    per-unit figure x units per launch; classes launched several times per iteration (the
    BCR levels) are averaged over those launches."""
    N, L, P = stats["num_observations"], stats["num_active_points"], stats["num_free_poses"]
    B, T = stats["num_reduced_blocks"], stats["num_observations"] / max(stats["num_active_points"], 1)
    lv = bcr_levels(stats["num_superblocks"])
    odd = [n // 2 for n in lv[:-1]] + [1]            # blocks factored per factor / backsub launch
    nxt = lv[1:]                                     # blocks produced per reduce launch
    bsub = list(odd)
    fused = bool(stats.get("pcr_fused", 0)) and os.environ.get("SSBA_NO_PCR_FUSED") != "1"
    fused_launches = 0
    if stats.get("pcr_blocks", 0):
        # plain levels down to the first one with <= 128 blocks, then parallel cyclic reduction of those n blocks:
        # ceil(log2 n) steps over ALL n blocks + one decoupled factor / solve, no back-substitution sweep there
        n = stats["pcr_blocks"]
        k = lv.index(n)
        steps = max(n - 1, 0).bit_length()
        odd = [m // 2 for m in lv[:k]] + [n] * (steps + 1)
        nxt = lv[1:k + 1] + [n] * steps
        bsub = [n] + [m // 2 for m in lv[:k]]
        # blocks moved per processed block: plain factor 3 in + 3 out, parallel step 3 in + 2 out (D stays), last 1 + 1;
        # plain reduce 3 in + 2 out, parallel reduce D in/out + 3 operand blocks (YU, YL of the folded block before, YL of
        # the one after) + the coupling and its transpose
        fact_blocks = 6 * sum(m // 2 for m in lv[:k]) + 5 * n * steps + 2 * n
        red_blocks = 5 * sum(lv[1:k + 1]) + 7 * n * steps
        if fused:
            # one launch per step (PcrFused): D + two Gram blocks + two couplings in, assembled D + four Gram blocks out =
            # 10 block moves per block and step; the decoupled last step 3 in + 1 out; no reduce launches in the plan.
            # Per block and step: Cholesky + 145 right-hand-side columns + the Gram products YL^T YL, YU^T YU (symmetric:
            # bd^3 each) and YU^T YL (2 bd^3).  (The workgroups of a block repeat the factorisation: not algorithmic work.)
            fact_blocks = 6 * sum(m // 2 for m in lv[:k]) + 10 * n * steps + 4 * n
            red_blocks = 5 * sum(lv[1:k + 1])
            nxt = lv[1:k + 1]
            fused_launches = steps
    else:
        fact_blocks, red_blocks = 6 * sum(odd), 5 * sum(nxt)
    bd = 72
    blk = bd * bd * 8
    if phong:
        # config 3: 56 B per observation (u,v,d,intensity,observed normal), landmark = position + normal +
        # material id (52 B), H_ll 21 + g_l 6 doubles out, C^-1 21 doubles; 7 residual rows, K = 6
        lm = {
            "k_linearize_landmarks": dict(bytes=56 * N + (52 + 216) * L, flops=650 * N),
            "k_linearize_poses": dict(bytes=(56 + 4 + 52) * N + 216 * P, flops=900 * N),
            "k_schur_windows": dict(bytes=56 * N + (52 + 216) * L + 23040 * stats["num_windows"],
                                    flops=L * (T * (T + 1) / 2 * 432 + T * 1400)),
            "k_backsub_eval": dict(bytes=2 * 56 * N + (52 + 216 + 168 + 96) * L, flops=1800 * N),
        }
    else:
        lm = None
    out = {
        # 24 B (u,v,d) per observation + landmark in (24 B) + H_ll,g_l out (72 B)
        "k_linearize_landmarks": dict(bytes=24 * N + 96 * L, flops=150 * N),
        # 24 B obs + 4 B ref + 24 B gathered point per observation, 216 B out per pose
        "k_linearize_poses": dict(bytes=52 * N + 216 * P, flops=330 * N),
        # per landmark: T(T+1)/2 pairs x 108 FMA + T half-linearisations (~330 flop)
        # + one 23 KB slab (78 blocks + 12 rhs) written per window
        "k_schur_windows": dict(bytes=24 * N + 120 * L + 23040 * stats["num_windows"], flops=L * (T * (T + 1) / 2 * 216 + T * 330)),
        "k_backsub_eval": dict(bytes=2 * 24 * N + 120 * L, flops=500 * N),
        # slabs in (23 KB per window) + S blocks out
        "k_assemble_reduced": dict(bytes=23040 * stats["num_windows"] + B * 288 * 2, flops=0),
        # per block: Cholesky bd^3/3 + two triangular solves with 2*bd+1 right-hand sides; 3 blocks in, 3 out
        "k_bcr_factor": dict(bytes=blk * fact_blocks / len(odd),
                             flops=((bd ** 3 / 3 + bd * bd * (2 * bd + 1)) * sum(odd) + 4 * bd ** 3 * stats.get("pcr_blocks", 0) * fused_launches) / len(odd)),
        # per new block: three bd^3 products (2 flop per FMA), 3 blocks in, 2 out
        "k_bcr_reduce": dict(bytes=blk * red_blocks / max(len(nxt), 1), flops=3 * 2 * bd ** 3 * sum(nxt) / max(len(nxt), 1)),
        # per block: two mat-vecs + one triangular solve; 3 blocks in
        "k_bcr_backsub": dict(bytes=3 * blk * sum(bsub) / len(bsub), flops=(4 * bd * bd + bd * bd) * sum(bsub) / len(bsub)),
    }
    if lm:
        out.update(lm)
    return out


def algorithmic_work_wide(stats):
    """The same per-launch figures for the wide reduced system (tracks of 13-24 observations: super-blocks of 24 poses =
    144 rows, ceres_slam_amd/csrc/ssba_wide.hip).  n blocks, s = ceil(log2 n) parallel-cyclic-reduction steps: every step is one
    k_wd_factor launch (per block: Cholesky of the 144 x 144 diagonal block + forward substitution of [L | U^T | r], 289
    columns; D, L, U in, G, YL, YU out) and one k_wd_reduce launch (per block three 144^3 products; the two neighbours'
    YL / YU and its own D, L in, D', L' out); the decoupled last step factors and solves."""
    N, L, P = stats["num_observations"], stats["num_active_points"], stats["num_free_poses"]
    T = N / max(L, 1)
    n = max(stats.get("wide_superblocks", 0), 1)
    bd = 144
    blk = bd * bd * 8
    out = algorithmic_work(dict(stats, num_superblocks=1, pcr_blocks=0))
    # per landmark T(T+1)/2 pairs x 108 FMA + T factor rows; one slab of 54 tiles per item (~21 landmarks' worth each)
    out["k_schur_windows"] = dict(bytes=24 * N + 120 * L + 54 * 256 * 8 * stats.get("num_windows", 0), flops=L * (T * (T + 1) / 2 * 216 + T * 330))
    out["k_assemble_reduced"] = dict(bytes=54 * 256 * 8 * stats.get("num_windows", 0) + stats["num_reduced_blocks"] * 288 * 2, flops=0)
    out["k_bcr_factor"] = dict(bytes=6 * blk * n, flops=(bd ** 3 / 3 + bd * bd * (2 * bd + 1)) * n)
    out["k_bcr_reduce"] = dict(bytes=7 * blk * n, flops=3 * 2 * bd ** 3 * n)
    # general layout: 24 B per observation + 4 B pose index; pose pass reads 32-byte pose-major records
    out["k_linearize_poses"] = dict(bytes=(32 + 24) * N + 216 * P, flops=330 * N)
    out["k_linearize_landmarks"] = dict(bytes=28 * N + 96 * L, flops=150 * N)
    out["k_backsub_eval"] = dict(bytes=2 * 28 * N + 120 * L, flops=500 * N)
    return out


def pmc_traffic(config):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of this workload (bench.py
    cannot collect PMC counters itself); tools/pmc_summarize.py documents the correction.  Returns (traffic by kernel
    class, the file it came from): the newest round's file under profiles/ -- a REPLAYED figure, not measured in this run."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{config.lower()}.json")))
    if not paths:
        return {}, None
    path = paths[-1]
    return {k: v["traffic_bytes_per_launch"] for k, v in json.load(open(path))["kernels"].items()}, os.path.relpath(path, ROOT)


def measured_peaks():
    """Peaks measured on this GPU model with tools/fp64_calib.hip (newest profiles/r*_fp64_calibration.txt), to read the
    roofline fractions against what the chip delivers rather than against the data-sheet figures."""
    import glob
    import re
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fp64_calibration.txt")))
    out = {"hbm_spec_GBs": HBM_PEAK_GBS, "fp64_spec_TFLOPs": FP64_PEAK_TFLOPS}
    if not paths:
        return out
    txt = open(paths[-1]).read()
    out["source"] = os.path.relpath(paths[-1], ROOT)
    # (no measured HBM figure: the copy kernels of tools/fp64_calib.hip reach 4.7 TB/s on this chip, below the 6.3 TB/s the
    # hardware guide measures for a 16-byte-per-lane copy -- a weak kernel is not a peak; fractions are priced against the
    # 8 TB/s spec only)
    m = re.search(r"fp64 MFMA 16x16x4 throughput, 1 wave\(s\)/SIMD, 4 indep acc: ([0-9.]+) TFLOP/s", txt)
    if m:
        out["fp64_mfma_measured_TFLOPs"] = float(m.group(1))
    m = re.search(r"fp64 FMA throughput, 4 wave\(s\)/SIMD, 8 indep acc: ([0-9.]+) TFLOP/s", txt)
    if m:
        out["fp64_valu_measured_TFLOPs"] = float(m.group(1))
    return out


def whole_iteration_work(stats):
    """compulsory HBM bytes and flops of ONE trust-region iteration of the matrix-free design,
    from the problem sizes alone (N observations, P free poses, L landmarks, B stored 6x6 blocks of S, track length T,
    half-bandwidth w of S in poses)."""
    N, L, P = stats["num_observations"], stats["num_active_points"], stats["num_free_poses"]
    B, T, w = stats["num_reduced_blocks"], stats["num_observations"] / max(stats["num_active_points"], 1), stats["pose_bandwidth"] + 1
    n = 6 * P
    byt = 128 * N + P * 96 * 2 + L * 24 * 2 + L * (48 + 24) * 2 + P * (168 + 48) * 2 + 3 * B * 288 + n * 8 * 4
    flo = 430 * N + L * (T * (T + 1) / 2 * 324 + T * 54) + n * (6 * w) ** 2 + 60 * N
    return dict(N=N, P=P, L=L, B=B, T=round(T, 3), half_bandwidth_poses=w, bytes=byt, flops=flo,
                formula="bytes = 128 N + 192 P + 48 L + 144 L + 432 P + 864 B + 32 n; flops = 430 N + L (T(T+1)/2 324 + 54 T) + n (6 w)^2 + 60 N; n = 6 P")


def roofline_of(work, avg_ms):
    avg_s = max(avg_ms, 1e-9) * 1e-3
    ai = work["flops"] / max(work["bytes"], 1)
    if ai > FP64_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        r = dict(bound="mfma", achieved=work["flops"] / avg_s / 1e12, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s")
    else:
        r = dict(bound="hbm", achieved=work["bytes"] / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
    r["frac"] = r["achieved"] / r["peak"]
    r["traffic"] = None
    r["avg_kernel_ms"] = avg_ms
    return r


def launch_ranks(args):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILD processes (torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1) before this process has made any GPU call, relay rank 0's JSON line and the
    exit code.  Nothing else happens in the parent."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what this pool's driver supports (RCCL, tensor sharing)
    # one node by contract: RCCL's bootstrap and socket transport stay on the loopback interface instead of whatever external
    # interface the container happens to have (the NCCL_DEBUG record of this pool shows it picking the container's veth)
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    raise SystemExit(rc if rc or line else 1)


def host_description():
    """nproc and CPU model of the box (the CPU baseline is only meaningful with both)."""
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"nproc": os.cpu_count() or 1, "usable_cpus": usable, "cpu_model": model}


def measure(args, name, P, L, ctx, kernel_timing, phong=False, robust=False):
    """One workload: build the (sharded) problem, solve it once to convergence, then time `args.steps` iterations of the
    production path between barriers.  Returns everything the bench line reports for it (rank 0 decides what to print)."""
    import torch
    from ceres_slam_amd import capi, sharding, synth
    from ceres_slam_amd.solver import StereoBA
    world, rank, local_rank, dist, backend = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["dist"], ctx["backend"]
    huber_a = 1.345 if robust else 0.0
    lighting = None
    if phong:
        if world > 1 and args.shared_free and (args.bounds or args.dogleg >= 0):
            raise SystemExit("config C3 with FREE shared blocks shards with LM and without bounds only")
        prob, ph = synth.make_phong_problem(P, L)
        lighting = ph.as_oracle_dict("perturbed" if args.shared_free else "truth")
    else:
        prob = synth.make_problem(P, L, outlier_fraction=0.3 if robust else 0.0, track_len=synth.CONFIG_TRACK.get(name, 12))
    partition = None
    if world > 1:
        # landmark ranges cut at super-block boundaries -> partitioned reduced solve (only the separator system is
        # exchanged); SSBA_NO_PARTITION=1 or an unalignable problem falls back to summing the whole reduced system
        # (free shared lighting blocks: their border sums are exchanged in the all-reduce mode only)
        # (long tracks -- the wide reduced system -- shard with the all-reduce too: the partitioned solve is built for 72-row blocks)
        cut = None if os.environ.get("SSBA_NO_PARTITION") == "1" or (phong and args.shared_free) or name in synth.CONFIG_TRACK else sharding.aligned_partition(
            prob.obs_pose, prob.obs_point, prob.num_poses, prob.num_points, world)
        if cut is not None:
            shard = sharding.shard_by_landmarks(prob, world, rank, ranges=cut[0])
            partition = cut[1]
        else:
            shard = sharding.shard_by_landmarks(prob, world, rank)
    else:
        shard = sharding.whole(prob)
    if phong and world > 1:      # this rank's landmarks and their lighting observations
        sel = np.isin(prob.obs_point, shard.point_ids)
        lighting = dict(lighting, normals=lighting["normals"][shard.point_ids], material_of_point=lighting["material_of_point"][shard.point_ids],
                        intensity=lighting["intensity"][sel], normal_obs=lighting["normal_obs"][sel])
    ba = StereoBA(prob.camera, shard.poses, shard.points, shard.obs_pose, shard.obs_point, shard.obs_uvd,
                  prob.stiffness(), device=local_rank, world_size=world, rank=rank, lighting=lighting,
                  shared_free=args.shared_free if phong else 0, use_bounds=bool(phong and args.bounds), partition=partition,
                  huber_a=huber_a)
    exchange, rccl_ranks, rccl_note = "none", 0, None
    if world > 1:
        # native exchange: libssba.so enqueues ncclAllReduce (RCCL over xGMI) itself on its stream; torch.distributed only
        # carries the 128-byte unique id.  SSBA_BENCH_EXCHANGE=torch (or a gloo rehearsal) routes the collectives through
        # torch.distributed instead; so does a failed or timed-out RCCL set-up on any rank (ssba_set_rccl is time-limited
        # and says where it sat: the text goes to stderr and into the bench line)
        want_native = backend == "nccl" and os.environ.get("SSBA_BENCH_EXCHANGE", "rccl") != "torch"
        ok = 0
        if want_native:
            try:
                sharding.attach_rccl_exchange(ba, dist)
                ok = 1
            except Exception as e:      # noqa: BLE001 -- any failure falls back, on every rank
                rccl_note = f"rank {rank}: {e}"
                print(f"[bench] rank {rank}: native RCCL exchange unavailable ({e}); using torch.distributed", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            exchange = "rccl (native, in libssba.so)"
            rccl_ranks = ba.rccl_ranks()
        else:
            ba.set_exchange(None)
            sharding.attach_torch_exchange(ba, dist)      # library kernels + collectives on one dedicated torch stream
            exchange = f"torch.distributed ({backend})"
    st = ba.stats()
    stats = {k: int(getattr(st, k)) for k, _ in capi.Stats._fields_}

    opts = capi.default_options(max_num_iterations=1000, use_nonmonotonic_steps=1)   # tests/dataset_vo.cpp:65-70
    if args.dogleg >= 0:
        opts.trust_region_strategy_type, opts.dogleg_type = 1, args.dogleg
    # ---- warm-up: one real solve to convergence (also gives the restart period) ----------
    s_conv, log_conv = ba.solve(opts)
    period = max(int(s_conv.num_iterations) - 1, 1)
    final_cost = float(s_conv.final_cost)
    solve_wall_s, solve_device_s = float(s_conv.total_time_s), float(s_conv.device_time_s)
    ba.poses[:] = shard.poses_init
    ba.points[:] = shard.points_init
    if phong:
        ba.normals[:] = lighting["normals"]
        ba.phong[:], ba.texture[:], ba.light[:] = lighting["phong"], lighting["texture"], lighting["light"]

    def run(n_steps, timing):
        ba.set_kernel_timing(timing)
        done = 0
        while done < n_steps:
            if done and done % period == 0:
                ba.restart()
            k = min(n_steps - done, period - done % period)      # up to the next restart in one call (graph replays of ten iterations)
            ba.step(k)
            done += k

    ba.solve_begin(opts, ignore_convergence=True)
    # set-up, not measurement: the two hipGraphs the production path replays (ten iterations per launch, and the single
    # iteration for the remainder) are captured here, whatever --warmup is (a warm-up shorter than ten iterations would
    # leave the capture + instantiation of the batched graph, ~1.5 ms, inside the K timed steps): ten iterations capture
    # the batch, the next seven run the six eager iterations a fresh handle starts with and capture the single one
    if world == 1:
        ba.step(17)
        ba.synchronize()
        ba.restart()
    run(args.warmup, False)
    ba.synchronize()
    ba.restart()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, False)          # production path: hipGraph replays
    ba.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # instrumented pass: the same K steps launched eagerly with every kernel bracketed by HIP
    # events on the library's stream (per-kernel durations for the roofline; agrees with
    # rocprofv3 --kernel-trace of this command)
    dt_instr = None
    if kernel_timing:
        ba.restart()
        ba.synchronize()
        t1 = time.perf_counter()
        run(args.steps, True)
        ba.synchronize()
        dt_instr = time.perf_counter() - t1
    ktimes = ba.kernel_times()
    ba.solve_end()
    xsize = int(ba.exchange_size()) if world > 1 else 0
    ba.close()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dict(name=name, prob=prob, lighting=lighting, huber_a=huber_a, stats=stats, dt=dt, dt_instr=dt_instr, ktimes=ktimes,
                period=period, final_cost=final_cost, solve_iterations=int(s_conv.num_iterations), solve_wall_s=solve_wall_s,
                line_search=dict(evaluations=int(s_conv.num_line_search_steps), searches_on_device=int(s_conv.num_line_searches_on_device),
                                 searches_by_host=int(s_conv.num_line_searches_by_host)),
                solve_device_s=solve_device_s, exchange=exchange, rccl_ranks=rccl_ranks, rccl_note=rccl_note,
                partitioned=partition is not None, exchange_doubles=xsize, poses=P, landmarks=L)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, help="C2 (default at N = 1), C3, C4, C5, LT24 (600 poses / 60 000 landmarks / 24 observations per landmark: the wide reduced system); default at N > 1: N x C2 for N = 2, 4 and "
                                                   "C4 exactly (10 000 poses / 1 M landmarks, BASELINE configs[3]) at N = 8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=1000, help="iteration cap of the CPU-oracle sample (it converges in ~80)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="no HIP-event bracketing (use under rocprofv3)")
    ap.add_argument("--no-weak-secondary", action="store_true", help="N = 8: skip the secondary 8 x C2 weak-scaling measurement")
    ap.add_argument("--strong", action="store_true", help="N > 1 with --config: also measure C2 itself (1 000 / 100 000) sharded over the N ranks "
                                                          "(secondary key strong_scaling_C2; measured by default when --config is not given)")
    ap.add_argument("--no-strong-secondary", action="store_true", help="N > 1: skip the strong-scaled C2 measurement")
    ap.add_argument("--no-single-reference", action="store_true", help="N > 1: skip rank 0's one-GPU measurement of the same joint problem")
    ap.add_argument("--shared-free", type=int, default=0, help="C3 only: free shared blocks (bit 0 light, 1 Phong, 2 texture)")
    ap.add_argument("--bounds", action="store_true", help="C3 only: the driver's bounds on the Phong / texture blocks")
    ap.add_argument("--dogleg", type=int, default=-1, help="-1 LM (dataset_vo), 0 TRADITIONAL_DOGLEG, 1 SUBSPACE_DOGLEG")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)         # children; never returns

    import torch
    import torch.distributed as dist
    from ceres_slam_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    # one rank per GPU; SSBA_BENCH_BACKEND=gloo + fewer devices than ranks is a rehearsal mode for
    # boxes with a single GPU (all ranks share cuda:0, collectives staged through the host)
    backend = os.environ.get("SSBA_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)       # (counting devices does not initialise the GPU)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the stereo-BA path is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    ctx = dict(world=world, rank=rank, local_rank=local_rank, dist=dist, backend=backend)

    cfg = args.config or ("C2" if world < 8 else "C4")
    phong = cfg == "C3"                   # BASELINE.json configs[2]: C2 + Phong lighting residual blocks
    robust = cfg == "C5"                  # BASELINE.json configs[4]: Huber loss, 30 % outlier observations (C2 shape per GPU)
    if cfg not in ("C2", "C3", "C4", "C5", "LT24"):
        raise SystemExit(f"unknown --config {cfg}")
    P1, L1 = synth.CONFIGS["C2" if (phong or robust) else cfg]
    C2P, C2L = synth.CONFIGS["C2"]
    # the joint problem: C4 is a fixed size (strong-scaled over the ranks); the C2-shaped configurations grow with the ranks
    P, L = (P1, L1) if cfg == "C4" else (P1 * world, L1 * world)
    m = measure(args, cfg, P, L, ctx, kernel_timing=not args.no_kernel_timing, phong=phong, robust=robust)
    weak = strong = single = single_c2 = None
    if world == 8 and args.config is None and not args.no_weak_secondary:
        weak = measure(args, "C2", C2P * world, C2L * world, ctx, kernel_timing=False)
    if world > 1 and (args.strong or args.config is None) and not args.no_strong_secondary and cfg != "C3":
        # BASELINE.json's literal metric problem (1 000 poses / 100 000 landmarks) sharded over the N ranks: strong scaling of C2
        strong = measure(args, "C2", C2P, C2L, ctx, kernel_timing=False, robust=robust)
    if world > 1 and not args.no_single_reference:
        # the SAME joint problem on ONE GPU, measured by rank 0 while the other ranks wait at the barrier: the honest
        # denominator of a scaling factor (a larger problem scores more C2-units per second on one GPU already)
        ctx1 = dict(ctx, world=1, rank=0)
        if rank == 0:
            single = measure(args, cfg, P, L, ctx1, kernel_timing=False, phong=phong, robust=robust)
            if strong is not None:
                single_c2 = single if (P, L) == (C2P, C2L) else measure(args, "C2", C2P, C2L, ctx1, kernel_timing=False, robust=robust)
        dist.barrier()

    if rank == 0:
        print(json.dumps(bench_line(args, m, weak, world, cfg, strong=strong, single=single, single_c2=single_c2)), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_line(args, m, weak, world, cfg, strong=None, single=None, single_c2=None):
    from ceres_slam_amd import synth
    C2L = synth.CONFIGS["C2"][1]
    stats, ktimes, prob, dt = m["stats"], m["ktimes"], m["prob"], m["dt"]
    phong, robust = cfg == "C3", cfg == "C5"
    ms = 1e3 * dt / args.steps
    joint_ips = args.steps / dt
    # `value` counts C2-sized units of work: one iteration of a joint problem with k x 100 000 landmarks is k units
    # (N x C2 at N ranks: N units; C4: 10 units)
    units = m["landmarks"] / C2L
    if cfg in synth.CONFIG_TRACK:         # another problem shape (24 observations per landmark): plain iterations/s of that problem
        units = 1.0
    out = {
        "metric": "gauss_newton_iters_per_sec",
        "value": joint_ips * units,
        # one C2-sized unit per iteration: plain iterations/s; a larger joint problem: C2-units/s (see value_definition)
        "unit": "iters/s" if units == 1 else "C2-units/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms,
        "ms_per_step_instrumented": (1e3 * m["dt_instr"] / args.steps) if m["dt_instr"] else None,
        "higher_is_better": True,
        # C4 is one fixed problem cut into N shards (strong); the C2-shaped workloads grow with the ranks (weak)
        "scaling": "strong" if (cfg == "C4" and world > 1) else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "value_definition": "joint-problem iterations/s x (landmarks of the joint problem / 100 000): C2-sized units of work per "
                            "second, so that N = 1 (C2), N x C2 and C4 (10 units per iteration) are one scale; joint_iters_per_sec "
                            "is the plain rate of the joint problem",
        "config": {"workload": f"{cfg}: {m['poses']} poses / {m['landmarks']} landmarks / "
                               f"{prob.num_obs} stereo observations, "
                               + ("stereo + Phong intensity + normal residual blocks, 6-D landmark blocks "
                                  f"(position + unit normal), shared_free={args.shared_free}, bounds={bool(args.bounds)}, "
                                  f"strategy={'LM' if args.dogleg < 0 else 'DOGLEG/%d' % args.dogleg}"
                                  if phong else ("reprojection-only LM (Ceres dataset_vo options)"
                                                 + (", HuberLoss(1.345) on every block, 30 % outlier observations" if robust else "")))
                               + f", {world} shard(s)",
                   "poses": m["poses"], "landmarks": m["landmarks"], "observations": int(prob.num_obs),
                   "exchange": m["exchange"], "rccl_ranks": m["rccl_ranks"],
                   "reduced_solve": ("partitioned: chain elimination per rank + separator exchange" if m["partitioned"]
                                     else ("replicated after an all-reduce of the reduced system" if world > 1 else "single GPU")),
                   "exchange_doubles_per_iteration": m["exchange_doubles"],
                   "restart_period_iters": m["period"], "joint_iters_per_sec": joint_ips, "c2_units_per_iteration": units,
                   "converged_final_cost": m["final_cost"],
                   # one blocking ssba_solve from host buffers: upload + loop + write-back (PCIe-inclusive)
                   "solve_iterations": m["solve_iterations"], "solve_wall_s_incl_pcie": m["solve_wall_s"],
                   "solve_device_s": m["solve_device_s"]},
        "stats": stats,
    }
    if m["line_search"]["evaluations"]:       # bounds: the projected line search of the converged solve above
        out["config"]["line_search"] = m["line_search"]
    if m["rccl_note"]:
        out["config"]["rccl_set_up_failure"] = m["rccl_note"]
    if weak is not None:
        w_ips = args.steps / weak["dt"]
        out["weak_scaling_NxC2"] = {"workload": f"{world} x C2: {weak['poses']} poses / {weak['landmarks']} landmarks / {weak['prob'].num_obs} "
                                                "observations (one C2-sized segment per rank)",
                                    "joint_iters_per_sec": w_ips, "ms_per_step": 1e3 * weak["dt"] / args.steps,
                                    "value": w_ips * weak["landmarks"] / C2L, "exchange": weak["exchange"], "rccl_ranks": weak["rccl_ranks"],
                                    "exchange_doubles_per_iteration": weak["exchange_doubles"]}
    if single is not None:
        s_ips = args.steps / single["dt"]
        out["speedup_vs_1gpu_same_problem"] = joint_ips / s_ips
        out["single_gpu_same_problem"] = {"joint_iters_per_sec": s_ips, "ms_per_step": 1e3 * single["dt"] / args.steps,
                                          "value": s_ips * units, "measured_by": "rank 0 on its own GPU, the other ranks waiting at a barrier"}
    if strong is not None:
        g_ips = args.steps / strong["dt"]
        out["strong_scaling_C2"] = {"workload": f"C2 itself: {strong['poses']} poses / {strong['landmarks']} landmarks / {strong['prob'].num_obs} "
                                                f"observations cut into {world} landmark shards (BASELINE.json's metric problem)",
                                    "iters_per_sec": g_ips, "ms_per_step": 1e3 * strong["dt"] / args.steps, "scaling": "strong",
                                    "exchange": strong["exchange"], "partitioned": strong["partitioned"],
                                    "exchange_doubles_per_iteration": strong["exchange_doubles"]}
        if single_c2 is not None:
            c_ips = args.steps / single_c2["dt"]
            out["strong_scaling_C2"]["single_gpu_iters_per_sec"] = c_ips
            out["strong_scaling_C2"]["speedup_vs_1gpu_same_problem"] = g_ips / c_ips
    if any(v[0] for v in ktimes.values()):
        work = algorithmic_work_wide(stats) if stats.get("wide_superblocks") else algorithmic_work(stats, phong)
        per_kernel = {k: (v[1] / v[0] if v[0] else 0.0) for k, v in ktimes.items()}   # avg ms
        if per_kernel.get("k_linearize_landmarks") and not per_kernel.get("k_linearize_poses") and "k_linearize_poses" in work:
            # window layout: both linearisation passes are ONE launch (k_linearize_both, timed in the landmark pass's class):
            # its time against the sum of the two passes' algorithmic work
            work = dict(work)
            work["k_linearize_landmarks"] = {q: work["k_linearize_landmarks"][q] + work["k_linearize_poses"][q] for q in ("bytes", "flops")}
        iter_kernel_ms = {k: v[1] / args.steps for k, v in ktimes.items()}
        dom = max((k for k in iter_kernel_ms if k in work), key=lambda k: iter_kernel_ms[k])
        roof = roofline_of(work[dom], per_kernel[dom])
        roof["kernel"] = dom
        roof_all = {k: roofline_of(work[k], per_kernel[k]) for k in work if per_kernel.get(k)}
        traffic, traffic_src = pmc_traffic(cfg) if world == 1 else ({}, None)
        for k, r in roof_all.items():
            r["traffic"] = traffic.get(k)
            r["algorithmic_bytes"] = work[k]["bytes"]
        roof["traffic"] = traffic.get(dom)
        roof["traffic_source"] = f"{traffic_src} (rocprofv3 --pmc passes of the same command; replayed, not measured in this run)" if traffic_src else None
        roof["algorithmic_bytes"] = work[dom]["bytes"]
        out["roofline"] = roof
        out["roofline_by_kernel"] = roof_all
        out["kernel_ms_per_iter"] = {k: round(v, 5) for k, v in iter_kernel_ms.items()}
    else:
        out["note"] = "kernel timing disabled"
    wi = whole_iteration_work(stats)
    wi["ms"] = ms
    wi["achieved_GBs"] = wi["bytes"] / (ms * 1e-3) / 1e9
    wi["achieved_TFLOPs"] = wi["flops"] / (ms * 1e-3) / 1e12
    wi["frac_hbm"] = wi["achieved_GBs"] / HBM_PEAK_GBS
    wi["frac_fp64"] = wi["achieved_TFLOPs"] / FP64_PEAK_TFLOPS
    wi["floor_us_at_hbm_peak"] = wi["bytes"] / (HBM_PEAK_GBS * 1e9) * 1e6
    if world == 1:
        out["roofline_whole_iteration"] = wi
    out["peaks"] = measured_peaks()
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(prob, args.cpu_iters, m["final_cost"], m["lighting"], args, m["huber_a"])
        # both sides in plain iterations/s of the SAME problem (`value` counts C2-sized units: ten per iteration at C4)
        out["gpu_over_cpu"] = joint_ips / out["cpu_baseline"]["value"]
    return out


def cpu_baseline(prob, iters, gpu_final_cost, lighting=None, args=None, huber_a=0.0):
    """CPU oracle = port with Ceres-equivalent semantics (kind "port"); bounded sample."""
    from oracle import oracle as orc
    host = host_description()
    cores = min(host["usable_cpus"], 16)
    if lighting is None:
        op = orc.OracleProblem.from_synth(prob, huber_a=huber_a)
    else:
        op = orc.OracleProblem(prob.camera, prob.poses_init, prob.points_init, prob.obs_pose, prob.obs_point, prob.obs_uvd,
                               prob.stiffness(), lighting=lighting, shared_free=args.shared_free, use_bounds=args.bounds)
    orc.lib()
    kw = dict(num_threads=cores, max_num_iterations=iters)
    if args is not None and args.dogleg >= 0:
        kw.update(trust_region_strategy_type=1, dogleg_type=args.dogleg)
    t0 = time.perf_counter()
    s, log = op.solve(orc.driver_options(**kw))
    dt = time.perf_counter() - t0
    n = max(int(s.num_iterations) - 1, 1)
    # also what the reference's evaluation mode costs (Jets instead of closed forms) and the reference's own
    # thread setting (num_threads = 8, dataset_ba_phong.cpp:81-82) -- bounded samples of the same solve
    variants = {}
    if lighting is None:
        def sample(threads, jets, cap=20):
            o = orc.OracleProblem.from_synth(prob, huber_a=huber_a)
            kv = dict(kw, num_threads=threads, max_num_iterations=cap)
            orc.set_jacobian_mode(1 if jets else 0)
            try:
                t = time.perf_counter()
                sv, _ = o.solve(orc.driver_options(**kv))
                el = time.perf_counter() - t
            finally:
                orc.set_jacobian_mode(0)
            m = max(int(sv.num_iterations) - 1, 1)
            return {"value": m / el, "unit": "iters/s", "cores": threads, "ms_per_iter": 1e3 * el / m, "iterations": m}
        variants["jet_autodiff_jacobians"] = sample(cores, True)
        variants["closed_form_8_threads"] = sample(min(8, cores), False)
        # SURVEY.md 8(d) "all host cores": `cores` is the CPU share of a one-GPU box (16); the host shows more (usable_cpus) -- a
        # bounded sample with 64 threads says what they add to this solve (memory-bound sweeps, a serial band Cholesky)
        if host["usable_cpus"] > cores:
            variants["closed_form_%d_threads" % min(host["usable_cpus"], 64)] = sample(min(host["usable_cpus"], 64), False, cap=8)
    return {"value": n / dt, "unit": "iters/s", "cores": cores, "kind": "port", "host": host, "variants": variants,
            "sample": f"{n} trust-region iterations (termination {int(s.termination_type)}) of the same "
                      f"{prob.num_obs}-observation solve: CPU restatement with Ceres-equivalent semantics, NOT Ceres "
                      f"(OpenMP, {cores} threads on a {host['nproc']}-CPU {host['cpu_model']}, analytic Jacobians, Schur + band Cholesky); {dt:.1f} s",
            "ms_per_iter": 1e3 * dt / n, "final_cost": float(s.final_cost), "iterations": int(s.num_iterations),
            "gpu_final_cost": gpu_final_cost,
            "final_cost_rel_diff_gpu_vs_cpu": abs(gpu_final_cost - float(s.final_cost)) / float(s.final_cost)}


if __name__ == "__main__":
    main()
