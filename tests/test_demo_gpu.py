"""End-to-end C++ example (examples/drive_demo.cpp): SFMSensorInterface ->
transformGlobalPlan -> updatePlan -> findBestAction for hundreds of control cycles."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "social_force_window_planner_amd", "host")


@pytest.mark.parametrize("args", [["1500", "6"], ["2000", "12"], ["1500", "12", "9", "17"]])
def test_robot_crosses_the_room(args):
    r = subprocess.run(["make", "-C", HOST, "all", "demo"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(HOST, "drive_demo")] + args, capture_output=True, text=True, timeout=600)
    m = re.search(r"RESULT cycles=(\d+) reached=(\d) dist_goal=([\d.]+) min_clearance=([\d.]+) failed_cycles=(\d+)", r.stdout)
    assert m, r.stdout[-2000:] + r.stderr[-2000:]
    cycles, reached, dist, clear, failed = int(m[1]), int(m[2]), float(m[3]), float(m[4]), int(m[5])
    assert r.returncode == 0 and reached == 1 and dist < 0.1 + 1e-9   # inside xy_goal_tolerance
    assert clear > 0.35                                               # never inside the robot radius of a person
    assert failed <= cycles // 100


def test_plain_c_caller():
    """examples/minimal_c.c: the C ABI from a C99 translation unit, one control cycle."""
    csrc = os.path.join(ROOT, "social_force_window_planner_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "cdemo"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(ROOT, "build", "minimal_c")], capture_output=True, text=True, timeout=120)
    m = re.search(r"RESULT index=(-?\d+) vx=([\d.]+) vtheta=(-?[\d.]+) cost=([\d.]+) valid=(\d+)", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:] + r.stderr[-2000:]
    assert int(m[1]) >= 0 and int(m[5]) >= 1 and float(m[4]) >= 0.0
    assert "-2.000" in r.stdout  # the never-scored (0,0) sample
