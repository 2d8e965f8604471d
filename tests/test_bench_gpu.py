"""bench.py contract: one JSON line with the required keys (N=1), and the N>1
code path (row sharding + all-reduce(min) exchange) exercised with two ranks
sharing the one visible GPU over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--extras",
                        "upload,resident,cfg2,cfg2_o64,cfg2_o240,inproc_multi", "--extra-steps", "3"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["launch"]["world_size"] == 1 and d["launch"]["rank_devices"][0]["device"] == 0
    assert d["roofline"]["kernel"].startswith("sfw_social_kernel_flat<double")  # the kernel that dominated, from the plan
    assert d["dtype"] == "f64" and d["higher_is_better"] is True
    # the default workload is the configuration the metric is quoted on: the north-star target
    assert d["config"]["workload"].startswith("target: 256x256 (v,w) grid, 50 pedestrians")
    assert "sfw_grid_fetch" in d["config"]["timed_call"]
    assert d["value"] > 1e6  # the north-star bar, through the blocking call incl. the cost-vector D2H
    assert d["kernel_only_value"] > 1e6  # from HIP events of a few sampled steps: no ordering asserted against the wall clock
    rf = d["roofline"]
    assert rf["bound"] == "valu" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0 < rf["executed_frac"] <= rf["frac"]
    ex = d["extra"]
    assert ex["cfg2"]["value"] > 1e6 and 0 < ex["cfg2"]["roofline_executed_frac"] <= ex["cfg2"]["roofline_frac"]
    assert "64 laser points" in ex["cfg2_o64"]["workload"] and ex["cfg2_o64"]["value"] < ex["cfg2"]["value"]
    assert ex["resident_launch"]["value"] > 1e6  # informational: launch + selection fetch only
    # sfw_multi_score_grid itself, R = 1..8 host-reduce ranks on the one device: same command, the host side does not grow with R
    im = ex["inproc_multi"]
    assert all(im[f"R{r}"]["cmd_vel_index"] == d["cmd_vel"]["index"] and im[f"R{r}"]["value"] > 1e6 for r in (1, 2, 4, 8))
    assert im["R8"]["enqueue_us"] < 4 * im["R1"]["enqueue_us"] and im["R8"]["ranks"] == 8
    assert d["scaling"] is None and d["sustained_clock_ghz"] > 1.0
    assert rf["hbm"]["unit"] == "GB/s" and rf["hbm"]["frac"] < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["all_cores"]["cores"] >= 1
    # the oracle on the headline grid itself, by default, outside the timed region
    vf = d["verify"]
    assert vf["samples"] == 65536 and vf["coverage"].startswith("the whole grid")
    assert vf["invalid_set_equal"] is True and vf["cmd_vel_match"] is True and vf["max_rel_err"] <= 1e-9
    assert vf["oracle_cmd_vel"]["index"] == d["cmd_vel"]["index"] and vf["same_cmd_vel_as_timed_steps"] is True
    # ... and every timed entry of `extra` carries its own: cfg2's whole grid, the laser-point variants inside --extra-verify-budget
    for name in ("cfg2", "cfg2_o64", "cfg2_o240"):
        ev = ex[name]["verify"]
        assert ev["invalid_set_equal"] is True and ev["cmd_vel_match"] is True and ev["max_rel_err"] <= 1e-9, (name, ev)
        assert ev["within_tolerance"] is True and ev["same_cmd_vel_as_timed_steps"] is True, (name, ev)
    assert ex["cfg2"]["verify"]["coverage"].startswith("the whole grid") and ex["cfg2"]["verify"]["samples"] == 16384
    # the per-cycle world upload is outside the timed step; the line says so and prices it
    assert "world_state" in d["config"] and 0 < d["value_incl_world_upload"] < d["value"]
    assert "240 laser points" in ex["cfg2_o240"]["workload"] and ex["cfg2_o240"]["value"] < ex["cfg2_o64"]["value"]
    assert 0 < ex["cfg2_o240"]["roofline_frac"] < 1
    assert im["R8"]["collective"]["exchange"] == "host_reduce" and im["R8"]["collective"]["devices"] == [0] * 8
    assert len(im["R8"]["levels"]) == 8 and sum(im["R8"]["rows"]) == 256
    assert d["value"] / cb["value"] > 100  # sanity: the GPU path is not the CPU path


def test_two_ranks_on_one_gpu_over_gloo():
    # the driver's single-process command shape: NO torchrun around it — `--gpus 2` itself launches the two ranks
    # (bench.ensure_ranks; tests/test_bench_launch.py covers the launch alone on CPU)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--no-cpu-baseline", "--extras", "inproc_multi,target_strong", "--inproc-workload", "cfg2",
           "--extra-steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d2 = _last_json(r.stdout)
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak" and sum(d2["per_rank"]["rows"]) == 512
    # what a reader of a SCALE record needs first: the rank count that really ran, who launched it, every rank's device
    assert list(d2)[:6] == ["metric", "value", "unit", "n_gpus", "launch", "cfg5_strong"]
    la = d2["launch"]
    assert la["world_size"] == 2 and la["gpus_arg"] == 2 and la["launched_by"].startswith("bench.py")
    assert sorted(x["rank"] for x in la["rank_devices"]) == [0, 1]
    assert abs(d2["config"]["samples_per_gpu"] - 65536) < 0.15 * 65536  # blocks of equal planned work, not of equal row counts
    assert len(d2["per_rank"]["social_kernel_ms"]) == 2 and min(d2["per_rank"]["exchange_us"]) > 0
    assert len(d2["per_rank"]["executed_share"]) == 2 and all(0 < v <= 1 for v in d2["per_rank"]["executed_share"])
    assert len(d2["per_rank"]["levels"]) == 2
    # the record says what the collective was: backend, group size, every rank's device
    co = d2["collective"]
    assert co["backend"] == "gloo" and co["world_size"] == 2 and len(co["rank_devices"]) == 2 and co["rccl_version"]
    # rank 0's row block against the oracle (budgeted sub-grid or the whole block) + the globally selected sample
    vf = d2["verify"]
    assert vf["invalid_set_equal"] is True and vf["max_rel_err"] <= 1e-9 and vf["selected_sample"]["rel_err"] <= 1e-9
    assert vf["selected_sample"]["index"] == d2["global_cmd_vel"]["index"]
    # strong scaling of the target grid itself next to the weak-scaling headline
    ts = d2["extra"]["target_strong"]
    assert ts["scaling"] == "strong" and sum(ts["per_rank"]["rows"]) == 256 and len(ts["per_rank"]["levels"]) == 2
    # rank 0 alone through sfw_multi_score_grid (one process, two ranks): printed next to the torchrun numbers
    im = d2["extra"]["inproc_multi_cfg2"]
    assert im["ranks"] == 2 and im["value"] > 1e6 and im["enqueue_us"] > 0, im
    assert d2["extra"]["inproc_multi_target"]["ranks"] == 2
    # the same 512x256 grid scored by one process selects the same command
    r1 = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--grid", "512x256",
                         "--no-cpu-baseline", "--no-extra", "--no-verify"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = _last_json(r1.stdout)
    assert d2["global_cmd_vel"]["index"] == d1["cmd_vel"]["index"]
    assert d2["global_cmd_vel"]["vx"] == d1["cmd_vel"]["vx"] and d2["global_cmd_vel"]["vtheta"] == d1["cmd_vel"]["vtheta"]
    assert d2["global_cmd_vel"]["cost"] == d1["cmd_vel"]["cost"]


@pytest.mark.gpu
def test_a_stuck_one_process_leg_costs_its_entry_not_the_line():
    """The one-process multi-device legs at N > 1 run in a child under a wall-clock limit (bench.inproc_multi_guarded): with
    a limit no child can meet, the entries carry an error and the headline line is still printed."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo",
           "--no-cpu-baseline", "--no-verify", "--extras", "inproc_multi", "--inproc-workload", "cfg2", "--inproc-timeout", "0.01"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "no result within" in d["extra"]["inproc_multi_cfg2"]["error"]
    assert "no result within" in d["extra"]["inproc_multi_target"]["error"]



def test_exchange_over_rccl_single_rank():
    """The all-reduce(min) exchange on the real backend (nccl == RCCL on ROCm) with
    device tensors; a 1-rank group is all a 1-GPU box can host."""
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
from social_force_window_planner_amd import multi_gpu
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%d")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
key = (3.25, -0.7, 0.125, -4242.0)
r, k, table = multi_gpu.exchange_best(key, dist, 0, 1, device="cuda:0")
assert r == 0 and k == key and table.shape == (1, 4), (r, k, table)
r, k, _ = multi_gpu.exchange_best((float("inf"),) * 4, dist, 0, 1, device="cuda:0")
assert r is None and k is None
ex = multi_gpu.KeyExchange(dist, 0, 1, device="cuda:0")
for _ in range(3):
    r, k, table = ex(key)
    assert r == 0 and k == key and table.shape == (1, 4), (r, k, table)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
dist.destroy_process_group()
print("rccl-ok")
''' % _port()
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, r.stderr[-3000:]
