#!/usr/bin/env python3
"""bench.py — scored trajectories/s of the DWA rollout + social-force scoring
path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target]

One "step" = one BLOCKING scoring call through the C ABI, as SURVEY.md §8d defines the
metric (the reference's grid loop is one blocking loop, src/sfw_planner.cpp:345-417):
sfw_grid_stage (shared-prefix planning on the host, the arena of footprint, agents, sample
vectors and class tables packed in pinned memory — the launch's first kernel fetches it) +
sfw_grid_launch (K1a/K1b/K1c, the K2 dispatches, K3) + sfw_grid_fetch (wait for the stream;
the 8*T-byte cost vector and the selection are on the host when it returns, written there by
the selection kernels and read in place through sfw_grid_costs_view) — the three calls
sfw_score_grid is made of, their arguments marshalled once (HipScorer.prepared).  The world state (costmap, footprint, agents) is resident in the
library before the timed region starts.  Default workload: the north-star target
configuration (256 x 256 samples, 50 pedestrians, 40 steps).

`--gpus N` MEANS N ranks, one per GPU: under the driver's `torch.distributed.run` it checks WORLD_SIZE == N; started on its own it
re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (ensure_ranks); a mismatch, or fewer
devices than ranks over RCCL, exits 2 and prints no line.

N > 1 (one process per GPU under torch.distributed.run): the ranks share an N-times taller
grid (weak scaling; rows are the outer, sharded axis, ref :345) — contiguous blocks of linvel
rows of equal planned work, about one workload-sized block each — and one all-reduce(min) of the
[N,4] f64 key table per step picks the global best.  Every N also reports BASELINE.json config 5 (4096 x 4096, 100
pedestrians) sharded over the N ranks under extra.cfg5_strong (strong scaling).

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP64_VECTOR_PEAK_TFLOPS = 78.6   # AMD MI355X datasheet
FP32_VECTOR_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md

GRID_OVERRIDE = None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="target",
                    help="BASELINE.json config: target (default), cfg1..cfg5, cfg2_o64, ref5x9")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32", "f64_strict"],
                    help="f64 (default, the parity mode), f32 (forces in float), f64_strict (f64 with the longer polynomials)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads under `extra`")
    ap.add_argument("--extras", default="cfg5_strong,target_strong,inproc_multi,upload,resident,cfg2,cfg2_o64,cfg2_o240,target_o720,cfg3,cfg4,cfg4_spec,f32",
                    help="comma-separated secondary measurements to run (all by default)")
    ap.add_argument("--extra-steps", type=int, default=20, help="timed steps of each secondary workload")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (gloo: CPU tensors; lets several ranks share one GPU in tests)")
    ap.add_argument("--inproc-workload", default="cfg5",
                    help="N > 1: the workload rank 0 also scores through sfw_multi_score_grid over all N devices (extra.inproc_multi_*)")
    ap.add_argument("--inproc-child", default=None, help=argparse.SUPPRESS)  # workload:devices:exchange:steps:warmup
    ap.add_argument("--inproc-timeout", type=float, default=180.0,
                    help="N > 1: wall-clock limit of each one-process multi-device measurement (run in a child process)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)  # rank launch only (CPU test)
    ap.add_argument("--grid", default=None, help="override the sample grid, e.g. 512x512 (experiments only)")
    ap.add_argument("--verify", action="store_true", help=argparse.SUPPRESS)  # the default since round 4 (kept for old command lines)
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the oracle check of the headline launch (outside the timed region; on by default)")
    ap.add_argument("--verify-budget", type=float, default=75.0,
                    help="seconds of all-core oracle time the verify leg may take: the WHOLE grid when it fits (the target "
                         "grid: ~30 s on the 256-thread GPU box), else a sub-grid spread over the whole grid")
    ap.add_argument("--extra-verify-budget", type=float, default=6.0,
                    help="seconds of all-core oracle time the verify leg of EACH entry under `extra` may take (cfg2's whole grid "
                         "fits; the others: a sub-grid + the 32 lowest-cost samples); 0 switches those legs off")
    ap.add_argument("--resident", action="store_true",
                    help="time launch + selection fetch only (no stage, no cost-vector D2H): kernel tuning aid")
    return ap.parse_args()


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def ensure_ranks(args):
    """`--gpus N` MEANS N ranks, one per GPU (rows are the sharded axis, ref src/sfw_planner.cpp:345).
    * WORLD_SIZE unset and N > 1: this process becomes the launcher — it re-executes the same command line under
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on a free port of 127.0.0.1 and exits with the
      launcher's status; rank 0 of the children prints the one JSON line (stdout is inherited).
    * WORLD_SIZE set (the driver's torchrun form) and != N: exit 2 with a message — never a mislabelled line.
    * backend nccl with fewer than N visible devices: exit 2 (a `--gpus 8` command must not print `n_gpus: 1`).
    Returns only in a process that is one of exactly N ranks."""
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: the launcher's rank count and --gpus "
                             "disagree; refusing to print a line labelled with either\n")
            sys.exit(2)
        return
    if args.gpus == 1:
        return
    if args.backend == "nccl" and not args.launch_check:
        import torch

        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} over RCCL needs {args.gpus} visible devices, this node shows "
                             f"{n_dev}; nothing measured\n")
            print(json.dumps({"error": f"--gpus {args.gpus} but {n_dev} device(s) visible", "n_gpus_requested": args.gpus,
                              "visible_devices": n_dev}))
            sys.exit(2)
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SFW_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write(f"bench.py: --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks: {' '.join(cmd)}\n")
    sys.stderr.flush()
    sys.exit(subprocess.call(cmd, env=env))


def launch_check(args, rank, world, local_rank):
    """--launch-check: the rank launch alone, no scoring (runs without a GPU; tests/test_bench_launch.py): every rank joins
    the process group, rank 0 prints what the group is."""
    info = {"launch_check": True, "n_gpus": world, "gpus_arg": args.gpus,
            "launched_by": "bench.py (self-launch)" if os.environ.get("SFW_BENCH_SELF_LAUNCHED") else
                           ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "python")}
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "pid": os.getpid()})
        info["ranks"] = seen
        dist.barrier()
        dist.destroy_process_group()
    else:
        info["ranks"] = [{"rank": 0, "local_rank": 0, "pid": os.getpid()}]
    if rank == 0:
        print(json.dumps(info), flush=True)


def cpu_baseline(scene, params_kw, budget_s=12.0):
    """Time the CPU oracle (the line-faithful port of the reference's serial
    loop) on a bounded subsample of the same workload: whole linvel rows, evenly
    spaced, 1 thread (the reference loop is single-threaded) and all cores."""
    from oracle.sfw_oracle import OracleScorer, lib as olib
    from social_force_window_planner_amd._abi import default_params

    o = OracleScorer(default_params(**params_kw))
    o.load_scene(scene)
    lin, ang = scene.linvels, scene.angvels
    cols = ang
    # calibrate on a slice of one row
    ncal = min(len(ang), 16)
    t0 = time.perf_counter()
    o.score_grid(scene.robot_state, lin[-1:], ang[:ncal], scene.goal_args, n_threads=1)
    per_traj = (time.perf_counter() - t0) / ncal
    if per_traj * len(ang) > budget_s / 2:  # a whole row would blow the budget: evenly spaced columns instead
        ncol = int(max(8, min(len(ang), budget_s / 2 / per_traj)))
        cols = ang[np.unique(np.linspace(0, len(ang) - 1, ncol).round().astype(int))]
    rows = int(max(1, min(len(lin), budget_s / 2 / max(per_traj * len(cols), 1e-9))))
    sel = np.unique(np.linspace(0, len(lin) - 1, rows).round().astype(int))
    sub = lin[sel]
    n = len(sub) * len(cols)
    t0 = time.perf_counter()
    o.score_grid(scene.robot_state, sub, cols, scene.goal_args, n_threads=1)
    t1 = time.perf_counter() - t0
    cores = olib().sfwo_max_threads()
    rows_mt = int(max(1, min(len(lin), rows * max(1, cores // 2))))
    sel_mt = np.unique(np.linspace(0, len(lin) - 1, rows_mt).round().astype(int))
    sub_mt = lin[sel_mt]
    t0 = time.perf_counter()
    o.score_grid(scene.robot_state, sub_mt, cols, scene.goal_args, n_threads=cores)
    tm = time.perf_counter() - t0
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": n / t1,
        "unit": "trajectories/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{len(sub)} of {len(lin)} linvel rows x {len(cols)} of {len(ang)} angvels = {n} trajectories, "
                  f"oracle -O2 double, 1 thread",
        "all_cores": {"value": len(sub_mt) * len(cols) / tm, "cores": cores,
                      "sample": f"{len(sub_mt)} rows x {len(cols)} = {len(sub_mt) * len(cols)} trajectories, OpenMP"},
        "cpu_model": cpu_model,
        "nproc": os.cpu_count(),
    }


def verify_against_oracle(job, costs, best, budget_s, global_key=None, topk=0, tol=1e-9):
    """The launch bench.py times, checked against the CPU oracle OUTSIDE the timed region: the whole grid of this rank when
    the oracle finishes it within budget_s on all host threads, otherwise a sub-grid spread evenly over the whole grid.
    Compared: sentinel sets (-1 rejected, -2 the never-scored (0,0) sample), every valid cost (max relative error), and the
    selection (ref src/sfw_planner.cpp:394-414): on a whole grid index, vx, vtheta and n_valid against the oracle's own
    sequential scan; on a sub-grid (topk > 0) the oracle also scores the topk samples the launch ranks lowest, and the
    reference's selection order over everything the oracle scored must name the launch's sample (cmd_vel_scope says which
    of the two a line's cmd_vel_match is)."""
    from oracle.sfw_oracle import OracleScorer
    from social_force_window_planner_amd._abi import default_params

    o = OracleScorer(default_params(**job.params_kw))
    o.load_scene(job.scene)
    rs, ga, lin, ang = job.scene.robot_state, job.scene.goal_args, job.lin, job.ang
    cores = os.cpu_count() or 1
    grid = np.asarray(costs).reshape(len(lin), len(ang))
    # calibrate the all-core oracle on a few samples spread over the grid
    cr = np.unique(np.linspace(0, len(lin) - 1, 4).round().astype(int))
    cc = np.unique(np.linspace(0, len(ang) - 1, min(len(ang), max(16, cores // 2))).round().astype(int))
    t0 = time.perf_counter()
    o.score_grid(rs, lin[cr], ang[cc], ga, n_threads=cores)
    per = (time.perf_counter() - t0) / (len(cr) * len(cc))
    n_fit = int(budget_s / max(per, 1e-9))
    whole = n_fit >= len(lin) * len(ang)
    if whole:
        rows, cols = np.arange(len(lin)), np.arange(len(ang))
    else:
        side = max(2, int(math.sqrt(n_fit)))
        rows = np.unique(np.linspace(0, len(lin) - 1, min(len(lin), side)).round().astype(int))
        cols = np.unique(np.linspace(0, len(ang) - 1, min(len(ang), max(2, n_fit // len(rows)))).round().astype(int))
    t0 = time.perf_counter()
    oc, ob = o.score_grid(rs, lin[rows], ang[cols], ga, n_threads=cores)
    t_or = time.perf_counter() - t0
    gc = grid[np.ix_(rows, cols)].ravel()
    v = oc >= 0
    same_set = bool(np.array_equal(oc < 0, gc < 0) and np.array_equal(oc[oc < 0], gc[gc < 0]))
    rel = np.abs(gc[v] - oc[v]) / np.abs(oc[v]) if v.any() else np.zeros(1)
    out = {
        "samples": int(len(oc)),
        "coverage": ("the whole grid the timed launch scored" if whole else
                     f"{len(rows)} x {len(cols)} sub-grid spread over the whole grid, out of the full-size launch's cost vector"),
        "max_rel_err": float(rel.max()),
        "median_rel_err": float(np.median(rel)),
        "invalid_set_equal": same_set,
        "n_valid_oracle": int(v.sum()),
        "cmd_vel_match": None,
        "oracle": {"threads": cores, "seconds": t_or, "trajectories_per_s": len(oc) / t_or},
        "tolerance": "f64 parity mode: tests assert 1e-9 relative; north star asks 1e-4 and an identical cmd_vel",
    }
    if whole:
        # GridJob stages this rank's rows with index_base = row0 * nw; the oracle's index is local to the rows it was given
        out["cmd_vel_match"] = bool(best["index"] - job.index_base == ob["index"] and best["vx"] == ob["vx"]
                                    and best["vtheta"] == ob["vtheta"] and best["n_valid"] == ob["n_valid"])
        out["oracle_cmd_vel"] = {"vx": ob["vx"], "vtheta": ob["vtheta"], "cost": ob["cost"], "index": ob["index"] + job.index_base,
                                 "n_valid": ob["n_valid"]}
        out["cmd_vel_scope"] = "the oracle's sequential selection over the whole grid"
    if not whole or global_key:
        # the selected sample itself, wherever it lies: its cost under the oracle, and that no checked sample beats it
        gi = int(-global_key[3]) if global_key else best["index"]
        if gi >= 0:
            r, c = divmod(gi, len(ang))
            c1, _ = o.score_grid(rs, job.lin_all[r:r + 1], ang[c:c + 1], ga, n_threads=1)
            gcost = global_key[0] if global_key else best["cost"]
            out["selected_sample"] = {"index": gi, "oracle_cost": float(c1[0]), "rel_err": float(abs(c1[0] - gcost) / abs(c1[0])),
                                      "no_sampled_cost_below_it": bool(not v.any() or oc[v].min() >= c1[0] * (1 - 1e-9))}
    if not whole and topk > 0 and not global_key:
        # The selection on a grid the oracle cannot finish: the oracle scores the topk samples the launch ranks lowest as well
        # (row by row, all threads), and the reference's order (cost up, linvel down, |angvel| up, later iterate first; the
        # 10000.0 cap) over everything the oracle scored — sub-grid + those — must name the launch's sample.
        flat = np.asarray(costs)
        sel = (flat >= 0) & (flat <= 10000.0)
        cand = np.flatnonzero(sel)
        cand = cand[np.argsort(flat[cand], kind="stable")[:topk]]
        if best["index"] - job.index_base >= 0 and best["index"] - job.index_base not in cand:
            cand = np.append(cand, best["index"] - job.index_base)
        keys = {}
        for ri, r in enumerate(rows):  # the sub-grid's oracle costs
            for ci, c in enumerate(cols):
                keys[int(r) * len(ang) + int(c)] = float(oc[ri * len(cols) + ci])
        top_rel = 0.0
        for r in np.unique(cand // len(ang)):
            cs = np.sort(cand[cand // len(ang) == r] % len(ang))
            c_or, _ = o.score_grid(rs, lin[r:r + 1], ang[cs], ga, n_threads=min(cores, len(cs)))
            for c, val in zip(cs, c_or):
                keys[int(r) * len(ang) + int(c)] = float(val)
                g = flat[int(r) * len(ang) + int(c)]
                if val >= 0:
                    top_rel = max(top_rel, abs(g - val) / abs(val))
                else:
                    same_set = False
        pick, pk = -1, None
        for i, cst in keys.items():
            if cst < 0 or cst > 10000.0:
                continue
            r, c = divmod(i, len(ang))
            k = (cst, -float(lin[r]), abs(float(ang[c])), -i)
            if pk is None or k < pk:
                pick, pk = i, k
        out["invalid_set_equal"] = same_set
        out["max_rel_err"] = max(out["max_rel_err"], top_rel)
        out["cmd_vel_match"] = bool(pick == best["index"] - job.index_base)
        out["cmd_vel_scope"] = (f"the reference's selection order over the {len(keys)} samples the oracle scored: the sub-grid + the "
                                f"{len(cand)} samples the launch ranks lowest")
        out["oracle_cmd_vel"] = {"index": pick + job.index_base if pick >= 0 else -1, "cost": pk[0] if pk else -1.0}
    out["within_tolerance"] = bool(out["max_rel_err"] <= tol and out["invalid_set_equal"])
    out["tolerance_checked"] = tol
    return out


class GridJob:
    """One rank's share of a workload, world state resident on its GPU."""

    def __init__(self, workload_name, precision, rank, world, device, scaling="weak"):
        from social_force_window_planner_amd import multi_gpu
        from social_force_window_planner_amd import synthetic as syn
        from social_force_window_planner_amd._abi import SFW_PRECISION_F32, SFW_PRECISION_F64, SFW_PRECISION_F64_STRICT, default_params
        from social_force_window_planner_amd.planner import HipScorer

        w = syn.WORKLOADS[workload_name]
        if GRID_OVERRIDE:
            nv, nw = (int(v) for v in GRID_OVERRIDE.lower().split("x"))
            w = dataclasses.replace(w, nv=nv, nw=nw)
        if world > 1 and scaling == "weak":  # N-times more linvel rows, this rank takes its block
            w = dataclasses.replace(w, nv=w.nv * world)
        self.workload = w
        # the scene (costmap, people) does not depend on the grid size: build it with a small grid and
        # sample the full grid separately (cfg5's 4096-row vectors are only sliced here)
        self.scene = syn.make_scene(w if w.sampler == "reference" else dataclasses.replace(w, nv=2, nw=2))
        lin_all, ang = syn.generalised_sampler(w.nv, w.nw) if w.sampler != "reference" else syn.reference_sampler()
        self.lin_all = lin_all
        self.params_kw = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
        prec = {"f32": SFW_PRECISION_F32, "f64_strict": SFW_PRECISION_F64_STRICT}.get(precision, SFW_PRECISION_F64)
        self.scorer = HipScorer(default_params(precision=prec, **self.params_kw), device=device)
        self.scorer.set_timing(True)  # per-kernel HIP events (on the handle's stream) for the roofline object
        self.scorer.load_scene(self.scene)
        if world > 1:
            # the grid — cfg5 itself (strong scaling) or the N-times taller target grid (weak scaling) — cut over the ranks into
            # contiguous blocks of equal PLANNED work (sfw_plan_row_blocks, the cut sfw_multi_score_grid makes too): the
            # shared-prefix tree saves another share of the steps in every velocity range, equal row counts are unequal work
            from social_force_window_planner_amd.planner import plan_row_blocks

            cuts = plan_row_blocks(lin_all, ang, self.scene.robot_state, self.scene.goal_args, w.sim_time, w.n_steps,
                                   w.n_people + 1, world)
            self.row0, row1 = int(cuts[rank]), int(cuts[rank + 1])
        else:
            self.row0, row1 = multi_gpu.shard_rows(w.nv, rank, world)
        self.lin = lin_all[self.row0:row1]
        self.ang = ang
        self.index_base = self.row0 * len(self.ang)
        self.n_local = len(self.lin) * len(self.ang)
        zero = int(np.any(self.lin == 0.0) and np.any(self.ang == 0.0))
        self.n_scored = self.n_local - zero  # the (0,0) sample is never scored (ref :349-352)
        self.scorer.stage(self.scene.robot_state, self.lin, self.ang, self.scene.goal_args, self.index_base)
        self.plan = self.scorer.plan_info()
        self._prep = None

    def step(self, resident=False):
        # arguments marshalled once, the caller's cost buffer the same every cycle (as a C caller's): three foreign calls per step
        if self._prep is None:
            sc = self.scene
            self._prep = self.scorer.prepared(sc.robot_state, self.lin, self.ang, sc.goal_args, self.index_base, zero_copy=True)
        return self._prep.relaunch() if resident else self._prep.step()


def run_config(workload_name, precision, steps, warmup, ctx, scaling="weak", resident=False):
    import torch

    rank, world, device = ctx["rank"], ctx["world"], ctx["device"]
    job = GridJob(workload_name, precision, rank, world, device, scaling)
    dist = ctx.get("dist")
    from social_force_window_planner_amd import multi_gpu

    state = {"win": (0, None), "xchg": []}
    exchange = multi_gpu.KeyExchange(dist, rank, world, device=ctx["coll_device"]) if dist is not None else None

    def one_step():
        _, best, key = job.step(resident)
        if dist is not None:  # single all-reduce(min): every rank fills its own row, +inf elsewhere
            t0 = time.perf_counter()
            wr, wk, _ = exchange(key)
            state["xchg"].append(time.perf_counter() - t0)
            state["win"] = (wr, wk)
        return best, key

    for _ in range(warmup):
        one_step()
    state["xchg"].clear()
    k2_ms, k1_ms, k3_ms, all_ms, wall = [], [], [], [], []
    # The interpreter's cyclic garbage collector stays out of the timed region: with torch imported a full (generation-2)
    # collection walks millions of objects — 80 ms in the middle of twenty 1.3 ms steps (profiles/r06_bench.json's first
    # recording had it in extra.cfg2_o64: ms_per_step 5.5 against a median of 1.34; worst_step_ms shows such a step).
    import gc

    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tp = t0
    for _ in range(steps):
        best, key = one_step()
        # HIP events recorded on the handle's own stream around each kernel group: read on every fourth step (reading them is
        # host time inside the timed region, in front of the next stage)
        if len(wall) % 4 == 0:
            k2_ms.append(job.scorer.last_launch_ms(2))
            all_ms.append(job.scorer.last_launch_ms(0))
            k1_ms.append(job.scorer.last_launch_ms(1))
            k3_ms.append(job.scorer.last_launch_ms(3))
        tn = time.perf_counter()
        wall.append(tn - tp)
        tp = tn
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    per_rank = None
    if dist is not None:
        cd = ctx["coll_device"]
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cd)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([float(job.n_scored)], dtype=torch.float64, device=cd)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        n_scored_total = int(tot.item())
        mine = torch.zeros((world, 7), dtype=torch.float64, device=cd)
        mine[rank] = torch.tensor([float(np.mean(k2_ms)), float(np.median(wall)) * 1e3,
                                   float(np.median(state["xchg"])) * 1e6, executed_share_of(job),
                                   job.scorer.sustained_clock_ghz(), float(len(job.lin)), float(job.plan["levels"])],
                                  dtype=torch.float64)
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        tab = mine.cpu().numpy()
        # executed_share differs per rank: every block of rows spans another velocity range, so its shared-prefix tree
        # saves another share of the steps — K2 time per rank follows it (read the imbalance here, not as "efficiency")
        per_rank = {"social_kernel_ms": tab[:, 0].tolist(), "median_step_ms": tab[:, 1].tolist(),
                    "exchange_us": tab[:, 2].tolist(), "executed_share": tab[:, 3].tolist(),
                    "sustained_clock_ghz": tab[:, 4].tolist(), "rows": [int(v) for v in tab[:, 5]],
                    "levels": [int(v) for v in tab[:, 6]]}
        win_rank, win_key = state["win"]
    else:
        n_scored_total = job.n_scored
        win_rank, win_key = 0, key
    return {
        "job": job,
        "elapsed": elapsed,
        "steps": steps,
        "n_scored_total": n_scored_total,
        "median_step_ms": float(np.median(wall)) * 1e3,
        "worst_step_ms": float(np.max(wall)) * 1e3,
        "worst_step_index": int(np.argmax(wall)),
        "k1_ms": float(np.mean(k1_ms)),
        "k2_ms": float(np.mean(k2_ms)),
        "k3_ms": float(np.mean(k3_ms)),
        "launch_ms": float(np.mean(all_ms)),
        "best": best,
        "global_key": win_key,
        "winner_rank": win_rank,
        "per_rank": per_rank,
        "clock_ghz": job.scorer.sustained_clock_ghz(),
    }


def inproc_multi(workload_name, precision, devices, exchange, steps, warmup):
    """sfw_multi_score_grid itself — ONE process driving len(devices) ranks, the shape the reference's plugin has
    (one process, sfw_plugin.xml:1-9; rows are the outer axis, src/sfw_planner.cpp:345): blocking call incl. the cost
    vector, host wall-clock per phase from sfw_multi_last_us."""
    from social_force_window_planner_amd import synthetic as syn
    from social_force_window_planner_amd._abi import SFW_PRECISION_F32, SFW_PRECISION_F64, SFW_PRECISION_F64_STRICT, default_params
    from social_force_window_planner_amd.planner import MultiScorer, plan_info_of_rank

    w = syn.WORKLOADS[workload_name]
    scene = syn.make_scene(dataclasses.replace(w, nv=2, nw=2))
    lin, ang = syn.generalised_sampler(w.nv, w.nw)
    prec = {"f32": SFW_PRECISION_F32, "f64_strict": SFW_PRECISION_F64_STRICT}.get(precision, SFW_PRECISION_F64)
    m = MultiScorer(default_params(precision=prec, sim_time=w.sim_time, sim_granularity=w.sim_granularity),
                    devices=tuple(devices), exchange=exchange)
    try:
        m.load_scene(scene)
        for _ in range(warmup):
            m.score_grid(scene.robot_state, lin, ang, scene.goal_args)
        wall, us = [], []
        for _ in range(steps):
            t0 = time.perf_counter()
            _, best = m.score_grid(scene.robot_state, lin, ang, scene.goal_args)
            wall.append(time.perf_counter() - t0)
            us.append(m.last_us())
        zero = int(np.any(lin == 0.0) and np.any(ang == 0.0))
        med = float(np.median(wall))
        return {
            "ranks": len(devices), "devices": list(devices), "exchange": "rccl" if exchange == 0 else "host_reduce",
            "steps": steps, "ms_per_call": med * 1e3, "value": (w.nv * w.nw - zero) / med, "unit": "trajectories/s",
            "enqueue_us": float(np.median([u["enqueue_us"] for u in us])),
            "exchange_us": float(np.median([u["exchange_us"] for u in us])),
            "fetch_us": float(np.median([u["fetch_us"] for u in us])),
            "levels_rank0": plan_info_of_rank(m, 0)["levels"],
            "levels": [plan_info_of_rank(m, r)["levels"] for r in range(len(devices))],
            "rows": [m.rank_rows(r)[1] for r in range(len(devices))],
            "collective": m.describe(),  # communicators ncclCommInitAll returned, their size and devices, RCCL version
            "cmd_vel_index": best["index"],
        }
    finally:
        m.close()


def inproc_multi_guarded(workload_name, precision, devices, exchange, steps, warmup, limit_s):
    """inproc_multi in a CHILD process under a wall-clock limit: the RCCL leg (ncclCommInitAll over every device of the
    node while each device also carries a torchrun rank's context) has only ever run where a node was available — a hang
    there must cost this entry, not the bench line.  The child is this file with --inproc-child; it needs no torch."""
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                        "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "HIP_VISIBLE_DEVICES_RANK")}
    spec = f"{workload_name}:{','.join(str(d) for d in devices)}:{exchange}:{steps}:{warmup}"
    cmd = [sys.executable, os.path.abspath(__file__), "--inproc-child", spec, "--precision", precision]
    if GRID_OVERRIDE:
        cmd += ["--grid", GRID_OVERRIDE]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {limit_s:.0f} s (child stopped)"}
    for line in reversed(out.stdout.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": f"child exit {out.returncode}: {out.stderr.strip()[-400:]}"}


def collective_info(dist, backend, ctx):
    """What the process group behind the per-step all-reduce(min) really is (a SCALE record must say what it measured)."""
    import torch

    info = {"backend": dist.get_backend(), "requested_backend": backend, "world_size": dist.get_world_size(),
            "rank_devices": None, "rccl_version": None, "visible_devices": torch.cuda.device_count(),
            "op": "all_reduce(MIN) of the [N,4] f64 key table, once per step", "coll_device": ctx["coll_device"]}
    try:
        v = torch.cuda.nccl.version()  # RCCL's version on ROCm
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    try:
        devs = [None] * dist.get_world_size()
        dist.all_gather_object(devs, {"rank": ctx["rank"], "device": ctx["device"],
                                      "name": torch.cuda.get_device_name(ctx["device"]),
                                      "hip_visible": os.environ.get("HIP_VISIBLE_DEVICES")})
        info["rank_devices"] = devs
    except Exception as e:  # noqa: BLE001
        info["rank_devices"] = f"unavailable ({type(e).__name__}: {e})"
    return info


def host_wait(dist, rank, key):
    """Rank 0 releases the others through the process group's store (CPU-side wait); a barrier if there is no store."""
    try:
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set(key, "1")
        else:
            store.wait([key])
    except Exception:
        dist.barrier()


def measured_traffic(workload_name):
    """HBM bytes per launch from the committed PMC passes (profiles/r0N_traffic.json, newest first), or None."""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f).get(workload_name)
            if d:
                d = dict(d)
                d["source"] = "profiles/" + name
                return d
        except (OSError, ValueError):
            pass
    return None


def executed_share_of(job):
    """Share of the algorithmic sample-steps this rank really integrates (shared-prefix tree + suffix)."""
    plan, S = job.plan, job.workload.n_steps
    total = plan["samples"] * S
    if plan["levels"] <= 0 or not total:
        return 1.0
    return (plan["class_steps"] + plan["samples"] * (S - plan["split_step"])) / total


def k2_kernel_name(plan, precision, has_points):
    """The K2 kernel family that dominated the launch, from sfw_grid_plan_info().organisation (SFW_ORG_*: 1 / 2 = register
    form with one / two slots per lane, 3 = flat form) — the name the rocprofv3 kernel rows of profiles/ start with."""
    ty = "float" if precision == "f32" else "double"
    org = plan["organisation"]
    if org == 3:
        return f"sfw_social_kernel_flat<{ty}, ..., OBS={'true' if has_points else 'false'}>"
    if org in (1, 2):
        return f"sfw_social_kernel<{ty}, NS={org}, ...>" + (" + flat-form side launch" if plan.get("flat_samples") else "")
    return "sfw_no_social_kernel (robot alone)"


def roofline_for(job, k2_ms, precision, brief=False):
    from social_force_window_planner_amd import synthetic as syn

    w = job.workload
    flops_traj = syn.algorithmic_flops_per_traj(w.n_people, w.n_steps, w.n_obstacles)
    flops_launch = flops_traj * job.n_scored
    peak = FP32_VECTOR_PEAK_TFLOPS if precision == "f32" else FP64_VECTOR_PEAK_TFLOPS
    ach = flops_launch / (k2_ms * 1e-3) / 1e12 if k2_ms > 0 else 0.0
    # work actually integrated: class-steps of the shared-prefix levels + the samples' remaining steps
    plan = job.plan
    total_steps = plan["samples"] * w.n_steps
    executed_steps = plan["class_steps"] + plan["samples"] * (w.n_steps - plan["split_step"]) if plan["levels"] > 0 \
        else total_steps
    executed_share = executed_steps / total_steps if total_steps else 1.0
    if brief:
        return {"frac": ach / peak, "executed_frac": ach / peak * executed_share, "achieved_tflops": ach}
    bytes_launch = syn.algorithmic_bytes_per_call(dataclasses.replace(w, nv=len(job.lin)), len(job.scene.footprint))
    hbm = bytes_launch / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
    tr = measured_traffic(job.workload.name) if (precision == "f64" and not GRID_OVERRIDE) else None
    return {
        "bound": "valu",
        "kernel": k2_kernel_name(plan, precision, w.n_obstacles > 0),
        "achieved": ach,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": ach / peak,
        "executed_frac": ach / peak * executed_share,
        "executed_share_of_algorithmic_steps": executed_share,
        "traffic": tr["k2_bytes"] if tr else None,
        "traffic_note": ("HBM bytes per step through all K2 dispatches, rocprofv3 PMC passes committed in "
                         + tr["source"] + " (algorithmic bytes are in roofline.hbm)") if tr else None,
        "flops_per_trajectory": flops_traj,
        "kernel_ms": k2_ms,
        "launches_per_step": plan["chunks"] * (1 + plan["levels"]),
        "shared_prefix": ({"split_step": plan["split_step"], "levels": plan["levels"],
                           "class_steps": plan["class_steps"],
                           "sample_steps_replaced": plan["samples"] * plan["split_step"]}
                          if plan["levels"] > 0 else None),
        "note": "frac: algorithmic flops per SURVEY.md §8d (48 per ordered pair incl. exp/atan2/sqrt as 1 op each) / "
                "kernel_ms / non-MFMA vector peak of the dtype (the path is FP64 vector-ALU bound, neither HBM nor "
                "MFMA).  executed_frac: the same with only the steps actually integrated (the shared-prefix rollout "
                "simulates the first split_step steps once per class of samples with bit-identical robot "
                "trajectories).  kernel_ms = HIP events on the handle's stream from the end of the pose rollout "
                "(K1a) to the end of the last K2 dispatch",
        "hbm": {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                "traffic": tr["all_kernels_bytes"] if tr else None, "bytes_per_launch": bytes_launch,
                "note": "non-binding by construction: a few bytes per trajectory (SURVEY.md §8d)"},
    }


def workload_text(w):
    return (f"{w.name}: {w.nv}x{w.nw} (v,w) grid, {w.n_people} pedestrians, {w.map_size}x{w.map_size} costmap, "
            f"sim_time={w.sim_time} s, sim_granularity={w.sim_granularity} ({w.n_steps} steps), 16-gon footprint"
            + (f", {w.n_obstacles} laser points" if w.n_obstacles else "")
            + (", pedestrians from 2.1 m (NOT SURVEY §8d's 0.8 m, where every sample of this crowd ends in a contact: "
               "tests/test_parity_gpu.py::test_cfg4_spec_crowd_full_size)" if w.n_people >= 150 and w.people_r_in is None else "")
            + (f", pedestrians from {w.people_r_in} m (SURVEY §8d's crowd as specified)" if w.n_people >= 150 and w.people_r_in is not None else ""))


def brief_verify(job, timed_best, precision, budget_s):
    """One more call of the job the entry has just timed, its cost vector against the oracle under the entry's budget
    (verify_against_oracle: the whole grid when that fits, else a sub-grid + the 32 lowest-cost samples for the selection)."""
    costs, vbest, _ = job.step()
    v = verify_against_oracle(job, costs, vbest, budget_s, topk=32, tol=1e-4 if precision == "f32" else 1e-9)
    return {"samples": v["samples"], "coverage": v["coverage"], "max_rel_err": v["max_rel_err"],
            "invalid_set_equal": v["invalid_set_equal"], "cmd_vel_match": v["cmd_vel_match"], "cmd_vel_scope": v.get("cmd_vel_scope"),
            "within_tolerance": v["within_tolerance"], "tolerance_checked": v["tolerance_checked"],
            "same_cmd_vel_as_timed_steps": vbest["index"] == timed_best["index"], "oracle_seconds": v["oracle"]["seconds"]}


def extra_entry(name, precision, steps, warmup, ctx, verify_budget=0.0):
    r = run_config(name, precision, steps, warmup, ctx)
    j = r["job"]
    rf = roofline_for(j, r["k2_ms"], precision, brief=True)
    ver = brief_verify(j, r["best"], precision, verify_budget) if verify_budget > 0 else None
    return {
        "verify": ver,
        "workload": workload_text(j.workload),
        "value": r["n_scored_total"] * steps / r["elapsed"],
        "unit": "trajectories/s",
        "steps": steps,
        "ms_per_step": r["elapsed"] / steps * 1e3,
        "median_ms_per_step": r["median_step_ms"],
        "worst_step_ms": r["worst_step_ms"],  # (a single slow step — a box hiccup — shows here and in value, not in the median)
        "worst_step_index": r["worst_step_index"],
        "kernel_only_value": j.n_scored / (r["launch_ms"] * 1e-3),
        "kernel_ms": {"rollout": r["k1_ms"], "social": r["k2_ms"], "argmin": r["k3_ms"], "launch_total": r["launch_ms"]},
        "roofline_frac": rf["frac"],
        "roofline_executed_frac": rf["executed_frac"],
        "chunks": j.plan["chunks"],
        "k2_organisation": j.plan["organisation"],  # SFW_ORG_*: 1 / 2 register-resident slots, 3 flat
        "flat_samples": j.plan["flat_samples"],  # samples a register-form launch hands to concurrent flat-form waves
        "sustained_clock_ghz": r["clock_ghz"],
        "cmd_vel_index": r["best"]["index"],
        "n_valid": r["best"]["n_valid"],
    }


def main():
    args = parse_args()
    global GRID_OVERRIDE
    GRID_OVERRIDE = args.grid
    if args.inproc_child:
        name, devs, xchg, st, wu = args.inproc_child.split(":")
        try:
            res = inproc_multi(name, args.precision, [int(d) for d in devs.split(",")], int(xchg), int(st), int(wu))
        except Exception as e:
            res = {"error": repr(e)}
        print(json.dumps(res), flush=True)
        return
    ensure_ranks(args)  # --gpus N means N ranks: re-launches itself under torch.distributed.run, or refuses a mismatch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        launch_check(args, rank, world, local_rank)
        return
    import torch

    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible; bench.py measures the HIP path only"}))
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_dev = torch.cuda.device_count()
        if args.backend == "nccl":
            if local_rank >= n_dev:
                raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            local_rank = local_rank % n_dev
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group(backend="gloo")
        dist = dist_mod
    ctx = {"rank": rank, "world": world, "device": local_rank, "dist": dist,
           "coll_device": f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"}

    res = run_config(args.workload, args.precision, args.steps, args.warmup, ctx, resident=args.resident)
    job = res["job"]
    value = res["n_scored_total"] * args.steps / res["elapsed"]
    w = job.workload
    out = {
        "metric": "scored trajectories/sec (v x w samples x steps) @ N pedestrians; cmd_vel match",
        "value": value,
        "unit": "trajectories/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["elapsed"] / args.steps * 1e3,
        "median_ms_per_step": res["median_step_ms"],
        "value_at_median": job.n_scored * world / (res["median_step_ms"] * 1e-3) if world == 1 else None,
        "higher_is_better": True,
        "scaling": "weak" if world > 1 else None,
        "vs_baseline": None,
        "dtype": "f64" if args.precision == "f64_strict" else args.precision,
        "data": "synthetic (seeded costmap/people per SURVEY.md §8d)",
        "config": {
            "workload": workload_text(w),
            "samples_per_gpu": job.n_local,
            "timed_call": ("sfw_grid_launch + selection fetch only (--resident)" if args.resident else
                           "blocking sfw_grid_stage + sfw_grid_launch + sfw_grid_fetch incl. the 8*T-byte cost vector, which "
                           "the launch's selection kernels leave in the handle's pinned host buffer (read in place through "
                           "sfw_grid_costs_view; grids over 64 MB of costs: copied out); world state resident"),
            "parallelism": (f"{w.nv} linvel rows sharded over {world} GPU(s): contiguous blocks of equal planned work (about "
                            f"{w.nv // world} rows per rank, per_rank.rows), all-reduce(min) of the [N,4] f64 selection-key table "
                            "per step") if world > 1 else "single GPU",
        },
        "kernel_ms": {"rollout": res["k1_ms"], "social": res["k2_ms"], "argmin": res["k3_ms"],
                      "launch_total": res["launch_ms"]},
        "kernel_only_value": job.n_scored * world / (res["launch_ms"] * 1e-3),
        "cmd_vel": {"vx": res["best"]["vx"], "vtheta": res["best"]["vtheta"], "cost": res["best"]["cost"],
                    "index": res["best"]["index"], "n_valid": res["best"]["n_valid"]},
        "roofline": roofline_for(job, res["k2_ms"], args.precision),
        # shader clock the last social-force launch really ran at (a wave's own s_memtime / s_memrealtime): boxes and thermal
        # states differ by ~10 %, kernel times of different runs are only comparable next to it
        "sustained_clock_ghz": res["clock_ghz"],
    }
    if world > 1:
        from social_force_window_planner_amd import multi_gpu

        gvx, gvth, gidx = multi_gpu.cmd_from_key(res["global_key"], len(job.ang), job.lin_all, job.ang)
        out["global_cmd_vel"] = {"vx": gvx, "vtheta": gvth, "index": gidx,
                                 "cost": res["global_key"][0] if res["global_key"] else -1.0,
                                 "winner_rank": res["winner_rank"]}
        out["per_rank"] = res["per_rank"]
        out["collective"] = collective_info(dist, args.backend, ctx)
        if not args.no_verify:
            if rank == 0:  # rank 0's own row block against the oracle + the globally selected sample, outside the timed region
                costs, vbest, _ = job.step()
                out["verify"] = verify_against_oracle(job, costs, vbest, min(args.verify_budget, 45.0), global_key=res["global_key"])
                out["verify"]["scope"] = f"rank 0's rows [{job.row0}, {job.row0 + len(job.lin)}) of {w.nv}"
                del costs
            host_wait(dist, rank, "sfw_verify_done")
    extra = {}
    wanted = set() if (args.no_extra or args.resident) else set(args.extras.split(","))
    if "cfg5_strong" in wanted:
        # BASELINE.json config 5 sharded over the N ranks (strong scaling); N = 1: the whole
        # 16.8 M-sample grid on one GPU, in table chunks.  One probe step sizes the run: >= 5 timed steps when a step
        # takes < 3 s, 3 otherwise.
        del job, res
        probe = run_config("cfg5", args.precision, 1, 1, ctx, scaling="strong")
        n5 = 5 if probe["elapsed"] < 3.0 else 3
        del probe
        r5 = run_config("cfg5", args.precision, n5, 0, ctx, scaling="strong")
        j5 = r5["job"]
        extra["cfg5_strong"] = {
            "workload": workload_text(j5.workload) + f", rows sharded over {world} GPU(s)",
            "value": r5["n_scored_total"] * n5 / r5["elapsed"], "unit": "trajectories/s", "steps": n5, "warmup": 2,
            "ms_per_step": r5["elapsed"] / n5 * 1e3, "n_gpus": world, "scaling": "strong",
            "samples_per_gpu": j5.n_local, "chunks_per_gpu": j5.plan["chunks"],
            "per_rank": r5["per_rank"] or {"social_kernel_ms": [r5["k2_ms"]], "median_step_ms": [r5["median_step_ms"]],
                                           "exchange_us": [0.0], "executed_share": [executed_share_of(j5)]},
            "roofline_frac_rank0": roofline_for(j5, r5["k2_ms"], args.precision, brief=True),
        }
        del j5, r5
    if "target_strong" in wanted and world > 1:
        # The headline grid itself cut over the N ranks (strong scaling of a target-sized grid): every rank's block shares
        # less of its prefix (per_rank.levels / executed_share), so this curve is sub-linear by construction — shown as a number
        rt = run_config(args.workload, args.precision, max(5, args.extra_steps), 2, ctx, scaling="strong")
        jt = rt["job"]
        extra["target_strong"] = {
            "workload": workload_text(jt.workload) + f", rows sharded over {world} GPU(s)",
            "value": rt["n_scored_total"] * rt["steps"] / rt["elapsed"], "unit": "trajectories/s", "steps": rt["steps"],
            "ms_per_step": rt["elapsed"] / rt["steps"] * 1e3, "n_gpus": world, "scaling": "strong",
            "samples_per_gpu": jt.n_local, "per_rank": rt["per_rank"],
        }
        del jt, rt
    if "inproc_multi" in wanted and world > 1:
        # The plugin-shaped path at N > 1: rank 0 ALONE drives all N devices from one process through
        # sfw_multi_score_grid over SFW_MULTI_RCCL (ncclCommInitAll + one grouped ncclAllReduce(min) per call); the other
        # ranks wait at the barrier below.  Same cfg5 grid as extra.cfg5_strong, so the two paths' numbers sit side by side.
        # (--backend gloo = ranks sharing GPUs in tests: host-reduce ranks on this rank's device instead.)
        if rank == 0:
            from social_force_window_planner_amd._abi import SFW_MULTI_HOST_REDUCE, SFW_MULTI_RCCL

            devs, xchg = (list(range(world)), SFW_MULTI_RCCL) if args.backend == "nccl" else ([local_rank] * world, SFW_MULTI_HOST_REDUCE)
            for name, st, wu in ((args.inproc_workload, 3, 1), (args.workload, 20, 3)):
                # child process + wall-clock limit: neither a broken RCCL install nor a hang takes the bench line down
                extra["inproc_multi_" + name] = inproc_multi_guarded(name, args.precision, devs, xchg, st, wu,
                                                                    args.inproc_timeout)
        # the others wait on the HOST (a key of the rendezvous store): a collective would park a spinning RCCL kernel on
        # every device rank 0 is about to use
        host_wait(dist, rank, "sfw_inproc_multi_done")
    if rank == 0 and world == 1 and not args.resident:
        job = GridJob(args.workload, args.precision, 0, 1, local_rank)
        if not args.no_verify:
            # the oracle on the headline grid itself, outside the timed region (the metric says "cmd_vel match")
            costs, vbest, _ = job.step()
            out["verify"] = verify_against_oracle(job, costs, vbest, args.verify_budget)
            out["verify"]["same_cmd_vel_as_timed_steps"] = vbest["index"] == out["cmd_vel"]["index"]
            del costs
        if "inproc_multi" in wanted:
            # sfw_multi_score_grid on the one visible device: R host-reduce ranks sharing it (row blocks, worker thread per
            # rank, column plan shared).  What scales here is the HOST side: the enqueue phase must not grow with R.
            from social_force_window_planner_amd._abi import SFW_MULTI_HOST_REDUCE

            im = {}
            for R in (1, 2, 4, 8):
                im[f"R{R}"] = inproc_multi(args.workload, args.precision, [local_rank] * R, SFW_MULTI_HOST_REDUCE,
                                           max(5, args.steps // 2), 2)
            im["enqueue_us_R8_over_R1"] = im["R8"]["enqueue_us"] / im["R1"]["enqueue_us"]
            im["note"] = ("one process, R handles on ONE device (the box has one GPU): kernels of the ranks share the GPU, so "
                          "ms_per_call is not a scaling curve; enqueue_us is the host cost of staging + launching all ranks")
            extra["inproc_multi"] = im
        if "upload" in wanted:
            # costmap + footprint + agents handed over again before every step, the costmap CHANGED every time (one far cell
            # toggles: an unchanged snapshot would be recognised and not sent) — since round 6 the snapshot travels with the next
            # stage, so the price is: steps with the hand-over minus steps without, same count; never less than the host time
            # of the three sfw_set_* calls themselves (the difference of two 8 ms means is noisy)
            reps = 12
            cells = job.scene.cells
            c0 = int(cells[1, 1])

            def hand_over(k):
                cells[1, 1] = (c0 + 1 + (k & 1)) % 250
                t = time.perf_counter()
                job.scorer.load_scene(job.scene)
                return time.perf_counter() - t

            for k in range(2):
                hand_over(k)
                job.step()
            t0 = time.perf_counter()
            host = 0.0
            for k in range(reps):
                host += hand_over(k)
                job.step()
            t1 = time.perf_counter()
            for _ in range(reps):
                job.step()
            t2 = time.perf_counter()
            cells[1, 1] = c0
            job.scorer.load_scene(job.scene)
            extra["world_upload_ms"] = max(((t1 - t0) - (t2 - t1)) / reps, host / reps) * 1e3
            # SURVEY §8d's metric counts the H2D of the inputs: the headline with the per-cycle world upload added to every step
            out["value_incl_world_upload"] = job.n_scored / ((out["ms_per_step"] + extra["world_upload_ms"]) * 1e-3)
            out["config"]["world_state"] = ("resident: costmap + footprint + agents are uploaded before the timed region "
                                            "(extra.world_upload_ms per cycle; value_incl_world_upload adds it to every step)")
        if "resident" in wanted:
            r_res = run_config(args.workload, args.precision, max(5, args.steps // 2), 2, ctx, resident=True)
            extra["resident_launch"] = {
                "value": r_res["n_scored_total"] * r_res["steps"] / r_res["elapsed"], "unit": "trajectories/s",
                "ms_per_step": r_res["elapsed"] / r_res["steps"] * 1e3,
                "note": "launch + 48-byte selection fetch only, grid staged once (round-1 headline form)"}
        evb = 0.0 if args.no_verify else args.extra_verify_budget
        for name in ("cfg2", "cfg2_o64", "cfg2_o240", "target_o720", "cfg3", "cfg4"):
            if name in wanted and name != args.workload:
                extra[name] = extra_entry(name, args.precision, args.extra_steps, 2, ctx, evb)
                if name == "cfg4" and "cfg4_spec" in wanted:
                    # SURVEY §8d's cfg4 crowd as specified: no sample reaches the horizon without a pedestrian contact (n_valid
                    # 0), so this entry times rollouts that end early — reported so that the specified workload has a number
                    e = extra_entry("cfg4_spec", args.precision, max(3, args.extra_steps // 4), 1, ctx, evb)
                    for k in ("roofline_frac", "roofline_executed_frac"):
                        e.pop(k, None)
                    e["note"] = ("every sample ends in a pedestrian contact before the horizon (n_valid 0, no cmd_vel): rollouts "
                                 "cut short, not the 40-step rollout the cfg4 entry above times")
                    extra["cfg4_spec"] = e
        if "f32" in wanted:
            if args.precision == "f64":
                # opt-in fast mode (forces in float, state/thresholds in double): same workload
                st = max(5, args.steps // 3)
                r3 = run_config(args.workload, "f32", st, 1, ctx)
                extra["f32_forces_mode"] = {
                    "value": r3["n_scored_total"] * st / r3["elapsed"], "unit": "trajectories/s",
                    "kernel_ms": {"social": r3["k2_ms"]},
                    "same_cmd_vel_as_f64": r3["best"]["index"] == out["cmd_vel"]["index"],
                    "verify": brief_verify(r3["job"], r3["best"], "f32", evb) if evb > 0 else None,
                    "note": "not the headline: costs within 1e-4 (measured <= 3e-5) of the f64 oracle on the BASELINE "
                            "workloads (0.025 s steps); chaotic 0.25 s-step crowds differ more (profiles/r05_parity_sweep.txt)"}
                # the other opt-in: f64 with the pair term's polynomials one degree longer each (SFW_PRECISION_F64_STRICT)
                rs_ = run_config(args.workload, "f64_strict", st, 1, ctx)
                extra["f64_strict_mode"] = {
                    "value": rs_["n_scored_total"] * st / rs_["elapsed"], "unit": "trajectories/s",
                    "kernel_ms": {"social": rs_["k2_ms"]},
                    "same_cmd_vel_as_f64": rs_["best"]["index"] == out["cmd_vel"]["index"],
                    "verify": brief_verify(rs_["job"], rs_["best"], "f64_strict", evb) if evb > 0 else None,
                    "note": "asin 8 / exp 9 instead of the default's 7 / 9 (7 / 8 until round 4): the angle's polynomial one degree longer"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(_scene_with_grid(job), job.params_kw)
    if extra:
        out["extra"] = extra
    if rank == 0:
        print(json.dumps(scale_record_first(out, args, ctx)))
    if dist is not None:
        dist.destroy_process_group()


def scale_record_first(out, args, ctx):
    """The line in the order a reader of a SCALE record needs it: metric / value / n_gpus, then `launch` (how many ranks
    really ran, who launched them, every rank's device) and `cfg5_strong` (BASELINE.json config 5, the configuration the
    1/2/4/8 curve is quoted on; the full entry stays under extra.cfg5_strong), then everything else."""
    import torch

    co = out.get("collective")
    launch = {
        "world_size": co["world_size"] if co else 1,
        "gpus_arg": args.gpus,
        "launched_by": ("bench.py --gpus N (self-launch under torch.distributed.run)" if os.environ.get("SFW_BENCH_SELF_LAUNCHED")
                        else "torch.distributed.run" if "WORLD_SIZE" in os.environ else "python (single process)"),
        "backend": co["backend"] if co else None,
        "visible_devices": torch.cuda.device_count(),
        "rank_devices": co["rank_devices"] if co else [{"rank": 0, "device": ctx["device"],
                                                        "name": torch.cuda.get_device_name(ctx["device"]),
                                                        "hip_visible": os.environ.get("HIP_VISIBLE_DEVICES")}],
    }
    c5 = (out.get("extra") or {}).get("cfg5_strong")
    brief5 = None
    if c5:
        brief5 = {"value": c5["value"], "unit": c5["unit"], "n_gpus": c5["n_gpus"], "scaling": "strong",
                  "ms_per_step": c5["ms_per_step"], "steps": c5["steps"], "samples_per_gpu": c5["samples_per_gpu"],
                  "social_kernel_ms_per_rank": c5["per_rank"]["social_kernel_ms"],
                  "workload": c5["workload"], "full_entry": "extra.cfg5_strong"}
    first = {}
    for k in ("metric", "value", "unit", "n_gpus"):
        first[k] = out[k]
    first["launch"] = launch
    first["cfg5_strong"] = brief5
    for k, v in out.items():
        if k not in first:
            first[k] = v
    return first


def _scene_with_grid(job):
    """The job's scene with the job's sample vectors (GridJob builds the scene on a 2 x 2 grid)."""
    sc = dataclasses.replace(job.scene)
    sc.linvels, sc.angvels = job.lin, job.ang
    return sc


if __name__ == "__main__":
    main()
