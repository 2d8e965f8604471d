"""Seeded synthetic inputs for the hot path and the velocity
samplers.

The reference builds its sample sets once in the SFWPlanner constructor
(reference src/sfw_planner.cpp:64-85): 5 linear velocities i*max_vel_x/4 and 9
angular velocities [0, +s, -s, +2s, -2s, +3s, -3s, +4s, -4s], s = max_vel_th/4.
`reference_sampler` reproduces that bit-for-bit; `generalised_sampler` extends
the same construction (same iteration order: |w| ascending, + before -) to any
(nv, nw), which the BASELINE.json grids (128, 256, 1024, 4096: even) need.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

from ._abi import SfwAgent


def reference_sampler(max_vel_x=0.7, max_vel_th=0.5):
    """linvels_ / angvels_ exactly as reference src/sfw_planner.cpp:64-85."""
    n_lin, n_ang = 4, 4
    lin_step = max_vel_x / n_lin
    lin = [i * lin_step for i in range(n_lin + 1)]
    ang_step = max_vel_th / n_ang
    ang = [0.0]
    for i in range(1, n_ang + 1):
        ang.append(i * ang_step)
        ang.append(i * (-ang_step))
    return np.array(lin, dtype=np.float64), np.array(ang, dtype=np.float64)


def generalised_sampler(nv, nw, max_vel_x=0.7, max_vel_th=0.5):
    """Same construction for arbitrary counts.

    linvels: i * (max_vel_x / (nv-1)), i = 0..nv-1   (nv == 1 -> [max_vel_x]).
    angvels, odd nw = 1+2n: [0, +s, -s, ..., +n s, -n s], s = max_vel_th / n
             (identical to the reference for nw = 9);
             even nw = 2n: [+s/2, -s/2, +3s/2, -3s/2, ...], s = max_vel_th / n
             (symmetric about 0, never contains 0, |w| <= max_vel_th - s/2).
    """
    if nv < 1 or nw < 1:
        raise ValueError("nv and nw must be >= 1")
    if nv == 1:
        lin = np.array([max_vel_x], dtype=np.float64)
    else:
        step = max_vel_x / (nv - 1)
        lin = np.array([i * step for i in range(nv)], dtype=np.float64)
    ang = []
    if nw % 2 == 1:
        n = (nw - 1) // 2
        ang.append(0.0)
        if n:
            s = max_vel_th / n
            for i in range(1, n + 1):
                ang.append(i * s)
                ang.append(i * (-s))
    else:
        n = nw // 2
        s = max_vel_th / n
        for i in range(1, n + 1):
            ang.append((i - 0.5) * s)
            ang.append((i - 0.5) * (-s))
    return lin, np.array(ang, dtype=np.float64)


@dataclasses.dataclass(frozen=True)
class Workload:
    """One BASELINE.json configuration (or a parity-test variant of it)."""

    name: str
    nv: int
    nw: int
    n_people: int
    map_size: int
    sim_time: float
    sim_granularity: float = 0.025
    seed: int = 0
    footprint: str = "polygon16"  # "polygon16" | "point"
    n_obstacles: int = 0
    sampler: str = "generalised"  # "generalised" | "reference"
    n_discs: int = 10
    people_r_in: float | None = None  # inner radius of the pedestrian annulus (None: see make_people)

    @property
    def n_samples(self):
        return self.nv * self.nw

    @property
    def n_steps(self):
        n = int(self.sim_time / self.sim_granularity + 0.5)
        return n if n else 1


# BASELINE.json "configs" (index = seed) + the north-star target.
WORKLOADS = {
    "cfg1": Workload("cfg1", 21, 21, 0, 100, 0.5, seed=1, n_discs=0),
    "cfg2": Workload("cfg2", 128, 128, 20, 200, 1.0, seed=2),
    "cfg2_o64": Workload("cfg2_o64", 128, 128, 20, 200, 1.0, seed=2, n_obstacles=64),  # SURVEY §8d secondary case
    # what the reference's laserCb really hands over (src/sensor_interface.cpp:117-127: every beam below max_obstacle_dist,
    # every agent gets the whole scan, :513-524): a 240-point and a 720-point scan
    "cfg2_o240": Workload("cfg2_o240", 128, 128, 20, 200, 1.0, seed=2, n_obstacles=240),
    "target_o720": Workload("target_o720", 256, 256, 50, 500, 1.0, seed=6, n_obstacles=720),
    "cfg3": Workload("cfg3", 256, 256, 50, 500, 2.0, seed=3),
    "cfg4": Workload("cfg4", 1024, 1024, 200, 500, 1.0, seed=4),
    # ... with SURVEY §8d's crowd as specified (pedestrians from 0.8 m): every sample ends in a pedestrian contact before the
    # horizon, so a launch times rollouts cut short and selects nothing (tests/test_parity_gpu.py::test_cfg4_spec_crowd_full_size)
    "cfg4_spec": Workload("cfg4_spec", 1024, 1024, 200, 500, 1.0, seed=4, people_r_in=0.8),
    "cfg5": Workload("cfg5", 4096, 4096, 100, 500, 1.0, seed=5),
    "target": Workload("target", 256, 256, 50, 500, 1.0, seed=6),
    "ref5x9": Workload("ref5x9", 5, 9, 5, 200, 1.0, seed=7, sampler="reference"),
}


@dataclasses.dataclass
class Scene:
    workload: Workload
    cells: np.ndarray  # uint8 [size_y, size_x]
    origin_x: float
    origin_y: float
    resolution: float
    footprint: np.ndarray  # float64 [K, 2]
    agents: np.ndarray  # ctypes array of SfwAgent, index 0 = robot
    obstacles: np.ndarray  # float64 [O, 2]
    robot_state: tuple  # x, y, theta, vx, vy, vtheta
    goal_args: tuple  # acc_x, acc_y, acc_theta, wpx, wpy
    linvels: np.ndarray
    angvels: np.ndarray


def make_costmap(size, resolution, rng, n_discs, robot_xy=(0.0, 0.0)):
    """Free space with inflated discs (core 254, linear falloff to 0 over 0.5 m),
    centres >= 1 m from the robot, 1-cell NO_INFORMATION (255) border."""
    origin = -(size * resolution) / 2.0
    cells = np.zeros((size, size), dtype=np.uint8)
    half = size * resolution / 2.0
    ys, xs = np.meshgrid(
        origin + (np.arange(size) + 0.5) * resolution,
        origin + (np.arange(size) + 0.5) * resolution,
        indexing="ij",
    )
    core, falloff = 0.15, 0.5
    placed = 0
    guard = 0
    while placed < n_discs and guard < 10000:
        guard += 1
        cx, cy = rng.uniform(-half + 0.5, half - 0.5, size=2)
        if math.hypot(cx - robot_xy[0], cy - robot_xy[1]) < 1.0:
            continue
        d = np.hypot(xs - cx, ys - cy)
        val = np.where(d <= core, 254.0, np.clip(253.0 * (1.0 - (d - core) / falloff), 0.0, 253.0))
        cells = np.maximum(cells, val.astype(np.uint8))
        placed += 1
    if n_discs > 0:
        cells[0, :] = 255
        cells[-1, :] = 255
        cells[:, 0] = 255
        cells[:, -1] = 255
    return cells, origin, origin


def make_footprint(kind, radius=0.35):
    if kind == "point":
        return np.zeros((0, 2), dtype=np.float64)
    if kind == "polygon16":
        k = 16
        a = np.arange(k) * (2.0 * math.pi / k)
        return np.stack([radius * np.cos(a), radius * np.sin(a)], axis=1).astype(np.float64)
    if kind == "box":
        return np.array([[0.4, 0.3], [-0.4, 0.3], [-0.4, -0.3], [0.4, -0.3]], dtype=np.float64)
    raise ValueError(kind)


def make_people(n, rng, robot_xy=(0.0, 0.0), naive_goal_time=2.0, person_radius=0.35,
                people_velocity=1.0, r_in=None):
    """N pedestrians in an annulus 0.8..r_out m around the robot, >= 0.7 m apart
    (r_out = 5 m, widened for dense crowds so rejection sampling terminates),
    speed U(0.2,1.3), heading U(-pi,pi), goal = pos + naive_goal_time*vel
    (reference src/sensor_interface.cpp:494-503)."""
    # annulus from 0.8 m.  Dense crowds (N >= 150, BASELINE cfg4) start from
    # 2.1 m instead: closer than robot 0.7 m/s + pedestrian 1.0 m/s can close in the 1 s horizon
    # plus the 0.35 m contact radius, every one of the ~1e6 samples is rejected by an early contact
    # and the "LDS people-tiling stress" configuration would measure early exits, not the rollout.
    if r_in is None:
        r_in = 0.8 if n < 150 else 2.1
    r_out = max(5.0, math.sqrt(n * person_radius**2 / 0.35 + r_in**2))
    pts = []
    guard = 0
    while len(pts) < n:
        guard += 1
        if guard > 2000000:
            raise RuntimeError("people placement did not converge")
        r = math.sqrt(rng.uniform(r_in**2, r_out**2))
        a = rng.uniform(-math.pi, math.pi)
        p = (robot_xy[0] + r * math.cos(a), robot_xy[1] + r * math.sin(a))
        if all((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 >= 0.49 for q in pts):
            pts.append(p)
    people = []
    for i, (px, py) in enumerate(pts):
        sp = rng.uniform(0.2, 1.3)
        hd = rng.uniform(-math.pi, math.pi)
        vx, vy = sp * math.cos(hd), sp * math.sin(hd)
        ag = SfwAgent()
        ag.x, ag.y, ag.vx, ag.vy = px, py, vx, vy
        ag.goal_x, ag.goal_y = px + naive_goal_time * vx, py + naive_goal_time * vy
        ag.goal_radius = person_radius
        ag.desired_velocity = people_velocity
        ag.radius = person_radius
        ag.has_goal = 1
        ag.id = i + 1
        ag.group_id = -1
        people.append(ag)
    return people


def make_robot_agent(x, y, vx, vy, max_vel_x=0.7, robot_radius=0.35, robot_id=0):
    """agents_[0] as SFMSensorInterface builds it (reference
    src/sensor_interface.cpp:31-37, 552-580): no goal, local-frame twist."""
    ag = SfwAgent()
    ag.x, ag.y, ag.vx, ag.vy = x, y, vx, vy
    ag.has_goal = 0
    ag.desired_velocity = max_vel_x
    ag.radius = robot_radius
    ag.id = robot_id
    ag.group_id = -1
    return ag


def make_scene(workload: Workload | str, max_vel_x=0.7, max_vel_th=0.5) -> Scene:
    if isinstance(workload, str):
        workload = WORKLOADS[workload]
    w = workload
    rng = np.random.default_rng(w.seed)
    res = 0.05
    cells, ox, oy = make_costmap(w.map_size, res, rng, w.n_discs)
    fp = make_footprint(w.footprint)
    # robot at the map centre, heading 0, moving at 0.3 m/s; pose/velocity are
    # float-representable on purpose (the reference truncates them to float,
    # src/sfw_planner.cpp:145-152)
    rs = (0.0, 0.0, 0.0, float(np.float32(0.3)), 0.0, 0.0)
    robot = make_robot_agent(rs[0], rs[1], rs[3], rs[4], max_vel_x=max_vel_x)
    people = make_people(w.n_people, rng, r_in=w.people_r_in)
    arr = (SfwAgent * (1 + len(people)))(robot, *people)
    if w.n_obstacles > 0:
        a = np.arange(w.n_obstacles) * (2.0 * math.pi / w.n_obstacles)
        obs = np.stack([3.0 * np.cos(a), 3.0 * np.sin(a)], axis=1).astype(np.float64)
    else:
        obs = np.zeros((0, 2), dtype=np.float64)
    if w.sampler == "reference":
        lin, ang = reference_sampler(max_vel_x, max_vel_th)
        assert (len(lin), len(ang)) == (w.nv, w.nw)
    else:
        lin, ang = generalised_sampler(w.nv, w.nw, max_vel_x, max_vel_th)
    goal_args = (1.0, 0.0, 1.0, 2.0, 0.5)
    return Scene(w, cells, ox, oy, res, fp, arr, obs, rs, goal_args, lin, ang)


def algorithmic_flops_per_traj(n_people, n_steps, n_obstacles=0):
    """SURVEY.md §8d flop convention: 48 per ordered pair interaction, 30 per
    agent-step, 12 per agent-obstacle-step."""
    a = n_people + 1
    return 48.0 * n_steps * n_people * (n_people + 2) + 30.0 * n_steps * a + 12.0 * n_steps * a * n_obstacles


def algorithmic_bytes_per_call(w: Workload, k_footprint=16):
    """SURVEY.md §8d byte convention: in = costmap + 80 B/agent + 16 B/obstacle
    + 16 B/footprint vertex + ~256; out = 8 B/sample (double costs) + 16."""
    a = w.n_people + 1
    return w.map_size * w.map_size + 80 * a + 16 * w.n_obstacles + 16 * k_footprint + 256 + 8 * w.n_samples + 16
