"""`bench.py --gpus N` means N ranks (the sharded axis is the linvel rows, ref src/sfw_planner.cpp:345): without a launcher
around it the command launches its own N ranks under torch.distributed.run; a WORLD_SIZE that disagrees with --gpus, or too
few devices for RCCL, is an error — never a line labelled `n_gpus: 1`.  Runs without a GPU (`--launch-check` stops after the
process group is up; the scoring itself is covered by tests/test_bench_gpu.py on the GPU box)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--launch-check"], cwd=ROOT,
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout  # rank 0 alone prints
    d = lines[0]
    assert d["n_gpus"] == 2 and d["gpus_arg"] == 2 and d["launched_by"].startswith("bench.py")
    assert sorted(x["rank"] for x in d["ranks"]) == [0, 1] and len({x["pid"] for x in d["ranks"]}) == 2
    assert "launching 2 ranks" in r.stderr


def test_under_torchrun_the_same_command_is_not_relaunched():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--backend", "gloo",
                        "--launch-check"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    (d,) = _json_lines(r.stdout)
    assert d["n_gpus"] == 2 and d["launched_by"] == "torch.distributed.run" and "launching" not in r.stderr


def test_world_size_that_disagrees_with_gpus_fails_loudly():
    for world, gpus in (("1", "8"), ("4", "2"), ("2", "1")):
        r = subprocess.run([sys.executable, "bench.py", "--gpus", gpus, "--launch-check"], cwd=ROOT,
                           env=_env(WORLD_SIZE=world, RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and not _json_lines(r.stdout), (world, gpus, r.stdout)
        assert f"--gpus {gpus} but WORLD_SIZE={world}" in r.stderr


def test_more_ranks_than_devices_over_rccl_is_refused():
    """This container has no GPU: `--gpus 8` over RCCL must exit non-zero and print no line with n_gpus in it."""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        import pytest

        pytest.skip("eight devices visible")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stdout
    lines = _json_lines(r.stdout)
    assert all("n_gpus" not in d and "value" not in d for d in lines) and "needs 8 visible devices" in r.stderr
