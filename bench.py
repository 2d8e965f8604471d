#!/usr/bin/env python3
"""bench.py — scored trajectories/s of the DWA rollout + social-force scoring
path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2]

One "step" = one pass of the hot path over one (v,w) grid: rollout+costmap
kernel, social-force kernel, argmin kernels and the 48-byte D2H of the selected
command, with every input (costmap, agents, sample vectors) already resident in
HBM.  N > 1 (one process per GPU under torch.distributed.run): the linvel rows
of an N-times larger grid are sharded over the ranks (weak scaling, SURVEY.md
§8e) and one all-reduce(min) per step picks the global best.

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", help="BASELINE.json config: cfg1..cfg5, target, ref5x9")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (target-config) measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (gloo: CPU tensors; lets several ranks share one GPU in tests)")
    ap.add_argument("--grid", default=None, help="override the sample grid, e.g. 512x512 (experiments only)")
    ap.add_argument("--verify", action="store_true", help="check the GPU result against the oracle on a sample subset")
    return ap.parse_args()


def cpu_baseline(scene, params_kw, budget_s=12.0):
    """Time the CPU oracle (the line-faithful port of the reference's serial
    loop) on a bounded subsample of the same workload: whole linvel rows, evenly
    spaced, 1 thread (the reference loop is single-threaded) and all cores."""
    from oracle.sfw_oracle import OracleScorer, lib as olib
    from social_force_window_planner_amd._abi import default_params

    o = OracleScorer(default_params(**params_kw))
    o.load_scene(scene)
    lin, ang = scene.linvels, scene.angvels
    # calibrate on one row
    t0 = time.perf_counter()
    o.score_grid(scene.robot_state, lin[-1:], ang, scene.goal_args, n_threads=1)
    per_traj = (time.perf_counter() - t0) / len(ang)
    rows = int(max(1, min(len(lin), budget_s / 2 / max(per_traj * len(ang), 1e-9))))
    sel = np.unique(np.linspace(0, len(lin) - 1, rows).round().astype(int))
    sub = lin[sel]
    n = len(sub) * len(ang)
    t0 = time.perf_counter()
    o.score_grid(scene.robot_state, sub, ang, scene.goal_args, n_threads=1)
    t1 = time.perf_counter() - t0
    cores = olib().sfwo_max_threads()
    rows_mt = int(max(1, min(len(lin), rows * max(1, cores // 2))))
    sel_mt = np.unique(np.linspace(0, len(lin) - 1, rows_mt).round().astype(int))
    sub_mt = lin[sel_mt]
    t0 = time.perf_counter()
    o.score_grid(scene.robot_state, sub_mt, ang, scene.goal_args, n_threads=cores)
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
        "sample": f"{len(sub)} of {len(lin)} linvel rows x {len(ang)} angvels = {n} trajectories, oracle -O2 double, 1 thread",
        "all_cores": {"value": len(sub_mt) * len(ang) / tm, "cores": cores,
                      "sample": f"{len(sub_mt)} rows x {len(ang)} = {len(sub_mt) * len(ang)} trajectories, OpenMP"},
        "cpu_model": cpu_model,
        "nproc": os.cpu_count(),
    }


class GridJob:
    """One rank's share of a workload, resident on its GPU."""

    def __init__(self, workload_name, precision, rank, world, device):
        from social_force_window_planner_amd import synthetic as syn
        from social_force_window_planner_amd._abi import SFW_PRECISION_F32, SFW_PRECISION_F64, default_params
        from social_force_window_planner_amd.planner import HipScorer

        w = syn.WORKLOADS[workload_name]
        if GRID_OVERRIDE:
            nv, nw = (int(v) for v in GRID_OVERRIDE.lower().split("x"))
            w = dataclasses.replace(w, nv=nv, nw=nw)
        if world > 1:  # weak scaling: N-times more linvel rows, this rank takes its block
            w = dataclasses.replace(w, nv=w.nv * world)
        self.workload = w
        self.scene = syn.make_scene(w)
        self.params_kw = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
        prec = SFW_PRECISION_F32 if precision == "f32" else SFW_PRECISION_F64
        self.scorer = HipScorer(default_params(precision=prec, **self.params_kw), device=device)
        self.scorer.set_timing(True)  # per-kernel HIP events for the roofline object
        self.scorer.load_scene(self.scene)
        from social_force_window_planner_amd import multi_gpu

        self.row0, row1 = multi_gpu.shard_rows(w.nv, rank, world)
        self.lin = self.scene.linvels[self.row0:row1]
        self.ang = self.scene.angvels
        self.index_base = self.row0 * len(self.ang)
        self.n_local = len(self.lin) * len(self.ang)
        zero = int(np.any(self.lin == 0.0) and np.any(self.ang == 0.0))
        self.n_scored = self.n_local - zero  # the (0,0) sample is never scored (ref :349-352)
        self.scorer.stage(self.scene.robot_state, self.lin, self.ang, self.scene.goal_args, self.index_base)
        self.plan = self.scorer.plan_info()

    def step(self):
        self.scorer.launch()
        _, best, key = self.scorer.fetch(want_costs=False)
        return best, key


def run_single_config(workload_name, precision, steps, warmup, dist_ctx):
    import torch

    rank, world, device = dist_ctx["rank"], dist_ctx["world"], dist_ctx["device"]
    job = GridJob(workload_name, precision, rank, world, device)
    dist = dist_ctx.get("dist")
    from social_force_window_planner_amd import multi_gpu

    state = {"win": (0, None)}

    def one_step():
        best, key = job.step()
        if dist is not None:  # single all-reduce(min): every rank fills its own row, +inf elsewhere
            wr, wk, _ = multi_gpu.exchange_best(key, dist, rank, world, device=dist_ctx["coll_device"])
            state["win"] = (wr, wk)
        return best, key

    for _ in range(warmup):
        one_step()
    k2_ms, k1_ms, k3_ms, all_ms = [], [], [], []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        best, key = one_step()
        # HIP events recorded on the handle's own stream around each kernel; the dominant
        # kernel (K2) is read back on every timed step, the small ones on every 4th
        k2_ms.append(job.scorer.last_launch_ms(2))
        if len(k2_ms) % 4 == 1:
            all_ms.append(job.scorer.last_launch_ms(0))
            k1_ms.append(job.scorer.last_launch_ms(1))
            k3_ms.append(job.scorer.last_launch_ms(3))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dist_ctx["coll_device"])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([float(job.n_scored)], dtype=torch.float64, device=dist_ctx["coll_device"])
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        n_scored_total = int(tot.item())
        win_rank, win_key = state["win"]
    else:
        n_scored_total = job.n_scored
        win_rank, win_key = 0, key
    return {
        "job": job,
        "elapsed": elapsed,
        "n_scored_total": n_scored_total,
        "k1_ms": float(np.mean(k1_ms)),
        "k2_ms": float(np.mean(k2_ms)),
        "k3_ms": float(np.mean(k3_ms)),
        "launch_ms": float(np.mean(all_ms)),
        "best": best,
        "global_key": win_key,
        "winner_rank": win_rank,
    }


def measured_traffic(workload_name):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            return json.load(f).get(workload_name)
    except (OSError, ValueError):
        return None


def roofline_for(job, k2_ms, precision):
    from social_force_window_planner_amd import synthetic as syn

    w = job.workload
    flops_traj = syn.algorithmic_flops_per_traj(w.n_people, w.n_steps, w.n_obstacles)
    flops_launch = flops_traj * job.n_scored
    peak = FP32_VECTOR_PEAK_TFLOPS if precision == "f32" else FP64_VECTOR_PEAK_TFLOPS
    ach = flops_launch / (k2_ms * 1e-3) / 1e12 if k2_ms > 0 else 0.0
    bytes_launch = syn.algorithmic_bytes_per_call(dataclasses.replace(w, nv=len(job.lin)), len(job.scene.footprint))
    hbm = bytes_launch / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
    tr = measured_traffic(job.workload.name) if (precision == "f64" and not GRID_OVERRIDE) else None
    return {
        "bound": "valu",
        "kernel": "sfw_social_kernel<%s>" % ("float" if precision == "f32" else "double"),
        "achieved": ach,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": ach / peak,
        "traffic": tr["k2_bytes"] if tr else None,
        "traffic_note": ("HBM bytes per step through all K2 dispatches, rocprofv3 PMC passes committed in "
                         "profiles/r01_traffic.json (the K1->K2 robot-step table and the class records of the "
                         "shared-prefix levels; algorithmic bytes are in roofline.hbm)") if tr else None,
        "flops_per_trajectory": flops_traj,
        "kernel_ms": k2_ms,
        "launches_per_step": job.plan["chunks"] * (1 + job.plan["levels"]),
        "shared_prefix": ({"split_step": job.plan["split_step"], "levels": job.plan["levels"],
                           "class_steps": job.plan["class_steps"],
                           "sample_steps_replaced": job.plan["samples"] * job.plan["split_step"],
                           "note": "steps [0, split_step) are simulated along a tree of classes of samples whose "
                                   "robot trajectories coincide under the acceleration limits (bit-identical costs); "
                                   "kernel_ms spans the prefix launches and the suffix launch, `achieved` still "
                                   "prices the full algorithmic work"} if job.plan["levels"] > 0 else None),
        "note": "algorithmic flops per SURVEY.md §8d (48 per ordered pair incl. exp/atan2/sqrt as 1 op each); "
                "non-MFMA vector peak for the dtype (the path is FP64 vector-ALU bound, neither HBM nor MFMA); "
                "kernel_ms = HIP events from the end of the pose rollout (K1a) to the end of the last K2 dispatch, "
                "i.e. all K2 dispatches of a step incl. the footprint/costmap kernels that run beside the prefix levels",
        "hbm": {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                "traffic": tr["all_kernels_bytes"] if tr else None, "bytes_per_launch": bytes_launch,
                "note": "non-binding by construction: a few bytes per trajectory (SURVEY.md §8d)"},
    }


def main():
    args = parse_args()
    global GRID_OVERRIDE
    GRID_OVERRIDE = args.grid
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    res = run_single_config(args.workload, args.precision, args.steps, args.warmup, ctx)
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
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic (seeded costmap/people per SURVEY.md §8d)",
        "config": {
            "workload": f"{args.workload}: {w.nv}x{w.nw} (v,w) grid, {w.n_people} pedestrians, "
                        f"{w.map_size}x{w.map_size} costmap, sim_time={w.sim_time} s, "
                        f"sim_granularity={w.sim_granularity} ({w.n_steps} steps), 16-gon footprint",
            "samples_per_gpu": job.n_local,
            "parallelism": f"linvel rows sharded over {world} GPU(s), all-reduce(min) of the 4-double selection key"
                           if world > 1 else "single GPU",
        },
        "kernel_ms": {"rollout": res["k1_ms"], "social": res["k2_ms"], "argmin": res["k3_ms"],
                      "launch_total": res["launch_ms"]},
        "cmd_vel": {"vx": res["best"]["vx"], "vtheta": res["best"]["vtheta"], "cost": res["best"]["cost"],
                    "index": res["best"]["index"], "n_valid": res["best"]["n_valid"]},
        "roofline": roofline_for(job, res["k2_ms"], args.precision),
    }
    if world > 1:
        from social_force_window_planner_amd import multi_gpu

        gvx, gvth, gidx = multi_gpu.cmd_from_key(res["global_key"], len(job.ang), job.scene.linvels, job.ang)
        out["global_cmd_vel"] = {"vx": gvx, "vtheta": gvth, "index": gidx,
                                 "cost": res["global_key"][0] if res["global_key"] else -1.0,
                                 "winner_rank": res["winner_rank"]}
    if rank == 0 and world == 1:
        if args.verify:
            from oracle.sfw_oracle import OracleScorer
            from social_force_window_planner_amd._abi import default_params

            o = OracleScorer(default_params(**job.params_kw))
            o.load_scene(job.scene)
            costs, _, _ = job.scorer.fetch(want_costs=True)
            rows = np.unique(np.linspace(0, len(job.lin) - 1, 8).round().astype(int))
            oc, _ = o.score_grid(job.scene.robot_state, job.lin[rows], job.ang, job.scene.goal_args,
                                 n_threads=os.cpu_count())
            gc = costs.reshape(len(job.lin), len(job.ang))[rows].ravel()
            v = oc >= 0
            out["verify"] = {"max_rel_err": float((np.abs(gc[v] - oc[v]) / np.abs(oc[v])).max()),
                             "same_invalid_set": bool(np.array_equal(oc < 0, gc < 0)), "rows": len(rows)}
        if not args.no_extra:
            # PCIe-inclusive form: the blocking sfw_score_grid call (H2D of the sample vectors,
            # all kernels, D2H of the whole cost vector + selection), world state already uploaded
            t0 = time.perf_counter()
            reps = max(3, args.steps // 2)
            for _ in range(reps):
                job.scorer.score_grid(job.scene.robot_state, job.lin, job.ang, job.scene.goal_args)
            dtb = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                job.scorer.load_scene(job.scene)
            dtw = (time.perf_counter() - t0) / reps
            out.setdefault("extra", {})["host_buffers"] = {
                "score_grid_blocking_traj_per_s": job.n_scored / dtb, "score_grid_blocking_ms": dtb * 1e3,
                "world_upload_ms": dtw * 1e3,
                "note": "never `value`: includes the 8*T-byte cost-vector D2H; world_upload = costmap+footprint+agents H2D"}
            job.scorer.stage(job.scene.robot_state, job.lin, job.ang, job.scene.goal_args, job.index_base)
        if not args.no_extra and args.workload != "target":
            # the north-star target configuration (50 pedestrians, 40 steps), fewer steps
            r2 = run_single_config("target", args.precision, max(2, args.steps // 4), 1, ctx)
            j2 = r2["job"]
            out.setdefault("extra", {})["target"] = {
                "workload": f"target: {j2.workload.nv}x{j2.workload.nw} grid, {j2.workload.n_people} pedestrians, "
                            f"{j2.workload.n_steps} steps",
                "value": r2["n_scored_total"] * max(2, args.steps // 4) / r2["elapsed"],
                "unit": "trajectories/s",
                "kernel_ms": {"rollout": r2["k1_ms"], "social": r2["k2_ms"], "argmin": r2["k3_ms"]},
                "roofline_frac": roofline_for(j2, r2["k2_ms"], args.precision)["frac"],
            }
        if not args.no_extra and args.precision == "f64":
            # opt-in fast mode (forces in float, state/thresholds in double): same workload
            r3 = run_single_config(args.workload, "f32", max(2, args.steps // 4), 1, ctx)
            out.setdefault("extra", {})["f32_forces_mode"] = {
                "value": r3["n_scored_total"] * max(2, args.steps // 4) / r3["elapsed"], "unit": "trajectories/s",
                "kernel_ms": {"social": r3["k2_ms"]}, "same_cmd_vel_as_f64": r3["best"]["index"] == res["best"]["index"],
                "note": "not the headline: costs within 1e-4 (measured <= 3e-5) of the f64 oracle"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(job.scene, job.params_kw)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
