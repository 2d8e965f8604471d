"""world_size-2 (and 3) gloo test of the N>1 path: row sharding + the single
all-reduce(min) exchange reproduce the single-process selection exactly.
The per-shard costs come from the CPU oracle (this is a test of the sharding and
exchange logic, which is identical on gloo and RCCL)."""
import dataclasses
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from social_force_window_planner_amd import multi_gpu
from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, wl, out_dir, tie):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.sfw_oracle import OracleScorer

        scene = syn.make_scene(wl)
        o = OracleScorer(default_params(sim_time=wl.sim_time, sim_granularity=wl.sim_granularity))
        o.load_scene(scene)
        lo, hi = multi_gpu.shard_rows(wl.nv, rank, world)
        lin = scene.linvels[lo:hi]
        nw = len(scene.angvels)
        if len(lin):
            costs, best = o.score_grid(scene.robot_state, lin, scene.angvels, scene.goal_args)
            if tie:  # force an exact cross-rank tie so the tie-break order is exercised
                from oracle.sfw_oracle import select_best

                costs = np.where(costs >= 0, 1.0, costs)
                best = select_best(lin, scene.angvels, costs)
            key = multi_gpu.key_from_best(best, lin, scene.angvels, nw, lo * nw)
        else:
            costs = np.zeros(0)
            key = (multi_gpu.INF,) * 4
        win_rank, win_key, table = multi_gpu.exchange_best(key, dist, rank, world)
        # the preallocated form bench.py uses gives the same table, call after call
        ex = multi_gpu.KeyExchange(dist, rank, world)
        for _ in range(2):
            r2, k2, t2 = ex(key)
            assert r2 == win_rank and k2 == win_key and np.array_equal(t2, table)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), costs=costs, lo=lo, hi=hi, win_rank=-1 if win_rank is None else win_rank,
                 win_key=np.array(win_key if win_key else (np.inf,) * 4), table=table)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tie", [(2, False), (2, True), (3, False)])
def test_sharded_selection_matches_single_process(tmp_path, oracle_mod, world, tie):
    wl = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=6, nw=5, n_people=4, seed=42)
    mp.spawn(_worker, args=(world, _free_port(), wl, str(tmp_path), tie), nprocs=world, join=True)
    scene = syn.make_scene(wl)
    o = oracle_mod.OracleScorer(default_params(sim_time=wl.sim_time, sim_granularity=wl.sim_granularity))
    o.load_scene(scene)
    full, best = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    if tie:
        full = np.where(full >= 0, 1.0, full)
        best = oracle_mod.select_best(scene.linvels, scene.angvels, full)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    # shards tile the grid with no overlap and the concatenated costs equal the full grid
    assert [int(p["lo"]) for p in parts][0] == 0 and int(parts[-1]["hi"]) == wl.nv
    assert np.array_equal(np.concatenate([p["costs"] for p in parts]), full)
    # every rank sees the same winner, and it is the single-process selection
    for p in parts:
        assert np.array_equal(p["table"], parts[0]["table"])
        vx, vth, idx = multi_gpu.cmd_from_key(tuple(p["win_key"]), wl.nw, scene.linvels, scene.angvels)
        assert idx == best["index"] and vx == best["vx"] and vth == best["vtheta"]
        assert float(p["win_key"][0]) == best["cost"]


def test_shard_rows_cover_and_balance():
    for nv in (1, 5, 128, 4096):
        for world in (1, 2, 3, 8):
            spans = [multi_gpu.shard_rows(nv, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nv
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_invalid_everywhere():
    r, k = multi_gpu.lexicographic_min([(np.inf,) * 4, (np.inf,) * 4])
    assert r is None and k is None
    assert multi_gpu.cmd_from_key(None, 9, [0.0], [0.0]) == (0.0, 0.0, -1)


def _wait_worker(rank, world, port, out_dir):
    import time

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench

        t0 = time.perf_counter()
        if rank == 0:
            time.sleep(0.6)  # rank 0 alone drives the in-process multi-device run
        bench.host_wait(dist, rank, "sfw_test_key")
        np.save(os.path.join(out_dir, f"wait_{rank}.npy"), np.array([time.perf_counter() - t0]))
    finally:
        dist.destroy_process_group()


def test_other_ranks_wait_on_the_host_while_rank0_works(tmp_path):
    """bench.py --gpus N: while rank 0 scores through sfw_multi_score_grid on all N devices the other ranks wait on a
    key of the rendezvous store (no collective, hence no RCCL kernel parked on the devices rank 0 is using)."""
    world = 3
    mp.spawn(_wait_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    waits = [float(np.load(tmp_path / f"wait_{r}.npy")[0]) for r in range(world)]
    assert all(w >= 0.55 for w in waits[1:]), waits
