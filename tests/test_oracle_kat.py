"""Known-answer tests that pin the CPU oracle (oracle/sfw_oracle.cpp).

Sources of truth, in decreasing strength:
  * the reference's own line_iterator.hpp / trajectory.cpp compiled from
    /root/reference into oracle/_ref (real reference code);
  * SURVEY.md Appendix B vectors (generated from the same header);
  * closed-form cases derived from the reference text (file:line in each test);
  * an independent numpy restatement of the published social-force term.
The lightsfm / Costmap2D boundaries stay "parity unpinned" (DESIGN.md).
"""
import ctypes as C
import math

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import SfwAgent, default_params

# SURVEY.md Appendix B (generated from the reference's line_iterator.hpp)
APPENDIX_B = {
    (0, 0, 5, 2): [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2), (5, 2)],
    (5, 2, 0, 0): [(5, 2), (4, 2), (3, 1), (2, 1), (1, 0), (0, 0)],
    (0, 0, 2, 5): [(0, 0), (0, 1), (1, 2), (1, 3), (2, 4), (2, 5)],
    (3, 3, 3, 3): [(3, 3)],
    (0, 0, 4, 4): [(0, 0), (1, 1), (2, 2), (3, 3), (4, 4)],
    (4, 0, 0, 3): [(4, 0), (3, 1), (2, 2), (1, 2), (0, 3)],
    (0, 0, 7, 1): [(0, 0), (1, 0), (2, 0), (3, 0), (4, 1), (5, 1), (6, 1), (7, 1)],
    (10, 10, 4, 13): [(10, 10), (9, 11), (8, 11), (7, 12), (6, 12), (5, 13), (4, 13)],
    (0, 0, 1, 0): [(0, 0), (1, 0)],
    (2, 7, 2, 1): [(2, 7), (2, 6), (2, 5), (2, 4), (2, 3), (2, 2), (2, 1)],
}


@pytest.mark.parametrize("ends", list(APPENDIX_B))
def test_line_iterator_appendix_b(oracle_mod, ends):
    got = [tuple(c) for c in oracle_mod.line_cells(*ends)]
    assert got == APPENDIX_B[ends]
    assert len(got) == max(abs(ends[2] - ends[0]), abs(ends[3] - ends[1])) + 1


def test_line_iterator_matches_real_reference_header(oracle_mod):
    """oracle Bresenham == reference LineIterator (compiled from /root/reference)
    on 2000 random segments + all Appendix B vectors."""
    if oracle_mod.ref_lib() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    rng = np.random.default_rng(7)
    cases = list(APPENDIX_B) + [tuple(int(v) for v in rng.integers(-40, 200, size=4)) for _ in range(2000)]
    for ends in cases:
        assert np.array_equal(oracle_mod.line_cells(*ends), oracle_mod.line_cells(*ends, use_ref=True)), ends
    for ends, cells in APPENDIX_B.items():
        assert [tuple(c) for c in oracle_mod.line_cells(*ends, use_ref=True)] == cells


def test_trajectory_container_real_reference(oracle_mod):
    """reference src/trajectory.cpp:16 (cost_ = -1 on construction) and the
    addPoint/getPoint/getEndpoint round trip."""
    R = oracle_mod.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built")
    pts = np.arange(15, dtype=np.float64).reshape(5, 3)
    out = np.zeros_like(pts)
    end = np.zeros(3)
    cost0 = C.c_double()
    n = R.ref_trajectory_roundtrip(pts.ctypes.data, 5, out.ctypes.data, end.ctypes.data, C.byref(cost0))
    assert n == 5 and cost0.value == -1.0
    assert np.array_equal(out, pts) and np.array_equal(end, pts[-1])


# ---------------------------------------------------------------------------
def _new_velocity(vg, vi, a, dt):  # reference sfw_planner.hpp:457-463
    return min(vg, vi + a * dt) if (vg - vi) >= 0 else max(vg, vi - a * dt)


def _closed_form_cost(p, rs, vs, ws, ga):
    """Independent Python restatement of scoreTrajectory for the pedestrian-free,
    empty-map, K<3 case (reference src/sfw_planner.cpp:519-527, :581-588, :643-667)."""
    x, y, th, vx, vy, vth = rs
    S = int(p.sim_time / p.sim_granularity + 0.5) or 1
    dt = p.sim_time / S
    for _ in range(S):
        vx = _new_velocity(vs, vx, ga[0], dt)
        vy = _new_velocity(0.0, vy, ga[1], dt)
        vth = _new_velocity(ws, vth, ga[2], dt)
        x, y = (x + (vx * math.cos(th) + vy * math.cos(math.pi / 2 + th)) * dt,
                y + (vx * math.sin(th) + vy * math.sin(math.pi / 2 + th)) * dt)
        th = th + vth * dt
    dx, dy = ga[3] - x, ga[4] - y
    d = dx * dx + dy * dy
    ang = np.float32(math.atan2(dy, dx) - th)
    mn, mx = np.float32(-math.pi), np.float32(math.pi)
    if ang >= mn:
        ang = mn + np.fmod(np.float32(ang - mn), np.float32(mx - mn))
    else:
        ang = mx - np.fmod(np.float32(mn - ang), np.float32(mx - mn))
    ang = abs(float(np.float32(ang))) / math.pi
    vel = abs(p.max_vel_x - vx) / p.max_vel_x
    return p.vel_weight * vel + p.distance_weight * d + p.angle_weight * ang


def test_closed_form_no_pedestrians(oracle_mod):
    """cfg1 (21x21, 0 pedestrians, empty 100x100 map) with the centre-cell
    footprint: social_work = costmap_cost = 0, cost is the analytic 3-term sum."""
    import dataclasses

    w = dataclasses.replace(syn.WORKLOADS["cfg1"], footprint="point")
    scene = syn.make_scene(w)
    p = default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
    o = oracle_mod.OracleScorer(p)
    o.load_scene(scene)
    costs, best = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    assert costs[0] == -2.0  # (0,0) sample is skipped, reference :349-352
    for i in range(1, len(costs), 7):
        iv, iw = divmod(i, len(scene.angvels))
        exp = _closed_form_cost(p, scene.robot_state, scene.linvels[iv], scene.angvels[iw], scene.goal_args)
        assert costs[i] == pytest.approx(exp, rel=1e-13, abs=1e-15)
    # same numbers with the 16-gon on the empty map (footprint cost 0 everywhere)
    scene2 = syn.make_scene(syn.WORKLOADS["cfg1"])
    o2 = oracle_mod.OracleScorer(p)
    o2.load_scene(scene2)
    costs2, best2 = o2.score_grid(scene2.robot_state, scene2.linvels, scene2.angvels, scene2.goal_args)
    assert np.array_equal(costs, costs2) and best == best2
    assert best["index"] == int(np.argmin(np.where(costs >= 0, costs, np.inf)))


def test_num_steps_rule(oracle_mod):
    """reference :519-527: int(sim_time/granularity + 0.5), at least 1."""
    L = oracle_mod.lib()
    for st, g, exp in [(1.0, 0.025, 40), (1.5, 0.25, 6), (2.0, 0.025, 80), (0.5, 0.025, 20), (0.01, 0.025, 1),
                       (0.0, 0.025, 1), (0.1, 0.03, 3)]:
        assert L.sfwo_num_steps(C.byref(default_params(sim_time=st, sim_granularity=g))) == exp


def test_normalize_angle_float(oracle_mod):
    """reference sfw_planner.hpp:399-407: float arithmetic, fmod, asymmetric branches."""
    L = oracle_mod.lib()
    pi = np.float32(math.pi)
    for v in [0.0, 1.0, -1.0, 3.0, -3.0, 4.0, -4.0, 7.0, -7.0, 100.0, -100.0, math.pi, -math.pi]:
        v32 = np.float32(v)
        if v32 >= -pi:
            exp = -pi + np.fmod(np.float32(v32 + pi), np.float32(pi + pi))
        else:
            exp = pi - np.fmod(np.float32(-pi - v32), np.float32(pi + pi))
        got = L.sfwo_normalize_angle(float(v32), float(-pi), float(pi))
        assert np.float32(got) == np.float32(exp)
        assert abs(math.remainder(got - v, 2 * math.pi)) < 2e-5 or abs(abs(got) - math.pi) < 2e-5


# ---------------------------------------------------------------------------
def _pair_force_numpy(p, me, other):
    """Independent restatement of the published social-force interaction
    (Moussaid et al.; lightsfm computeSocialForce), written
    with explicit trig instead of the oracle's vector helpers."""
    diff = np.array([other.x - me.x, other.y - me.y])
    dn = math.hypot(*diff)
    dd = diff / dn
    w = np.array([me.vx - other.vx, me.vy - other.vy])
    inter = p.sfm_lambda * w + dd
    il = math.hypot(*inter)
    ih = inter / il
    theta = math.atan2(dd[1], dd[0]) - math.atan2(ih[1], ih[0])
    theta = (theta + math.pi) % (2 * math.pi) - math.pi
    if theta <= -math.pi:
        theta += 2 * math.pi
    B = p.sfm_gamma * il
    fv = -math.exp(-dn / B - (p.sfm_n_prime * B * theta) ** 2)
    fa = -np.sign(theta) * math.exp(-dn / B - (p.sfm_n * B * theta) ** 2)
    normal = np.array([-ih[1], ih[0]])
    return p.sfm_force_factor_social * (fv * ih + fa * normal)


def _agent(x, y, vx, vy):
    a = SfwAgent()
    a.x, a.y, a.vx, a.vy = x, y, vx, vy
    return a


def test_pair_force_hand_case(oracle_mod):
    """Two agents on the x axis, 2 m apart, walking towards each other with a
    small lateral offset in velocity: every intermediate value written out."""
    p = default_params()
    me, other = _agent(0, 0, 1.0, 0.0), _agent(2, 0, -0.5, 0.1)
    # diff=(2,0) dhat=(1,0); w=(1.5,-0.1); I = 2w + dhat = (4,-0.2)
    il = math.hypot(4.0, -0.2)
    theta = 0.0 - math.atan2(-0.2, 4.0)
    B = 0.35 * il
    fv = -math.exp(-2.0 / B - (3.0 * B * theta) ** 2)
    fa = -1.0 * math.exp(-2.0 / B - (2.0 * B * theta) ** 2)
    ih = np.array([4.0, -0.2]) / il
    exp = 2.1 * (fv * ih + fa * np.array([0.2, 4.0]) / il)
    got = oracle_mod.pair_force(p, me, other)
    assert got == pytest.approx(exp, rel=1e-13)


def test_pair_force_matches_independent_restatement_and_is_antisymmetric(oracle_mod):
    p = default_params()
    rng = np.random.default_rng(11)
    for _ in range(500):
        a = _agent(*rng.uniform(-5, 5, 2), *rng.uniform(-1.3, 1.3, 2))
        b = _agent(*rng.uniform(-5, 5, 2), *rng.uniform(-1.3, 1.3, 2))
        fab = oracle_mod.pair_force(p, a, b)
        assert fab == pytest.approx(_pair_force_numpy(p, a, b), rel=1e-10, abs=1e-300)
        # Newton's third law holds when both agents carry the same sfm::Parameters
        # (the HIP kernel relies on it to evaluate each unordered pair once)
        fba = oracle_mod.pair_force(p, b, a)
        assert fba == pytest.approx(-fab, rel=1e-10, abs=1e-300)


# ---------------------------------------------------------------------------
def _scorer_with_map(oracle_mod, cells, res=1.0, ox=0.0, oy=0.0, footprint=None, params=None):
    o = oracle_mod.OracleScorer(params or default_params())
    o.set_costmap(cells, ox, oy, res)
    o.set_footprint(footprint if footprint is not None else np.zeros((0, 2)))
    return o


def test_footprint_cost_code_table(oracle_mod):
    """reference src/costmap_model.cpp:36-48 (K<3: 255->-2, 254|253->-1) and
    :112-121 (edges: 255->-2, 254->-1, 253 is NOT rejected), off-map -> -3."""
    cells = np.zeros((20, 20), dtype=np.uint8)
    cells[5, 5], cells[5, 6], cells[5, 7], cells[5, 8] = 255, 254, 253, 77
    o = _scorer_with_map(oracle_mod, cells)
    assert o.footprint_cost(5.5, 5.5, 0.0) == -2.0
    assert o.footprint_cost(6.5, 5.5, 0.0) == -1.0
    assert o.footprint_cost(7.5, 5.5, 0.0) == -1.0
    assert o.footprint_cost(8.5, 5.5, 0.0) == 77.0
    assert o.footprint_cost(-0.1, 5.0, 0.0) == -3.0
    assert o.footprint_cost(20.0, 5.0, 0.0) == -3.0
    assert o.footprint_cost(0.0, 0.0, 0.0) == 0.0  # exactly on the origin is on the map
    # polygon: a 2x2 square whose edge crosses the 253 cell is legal, returns the max
    sq = np.array([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]])
    o = _scorer_with_map(oracle_mod, cells, footprint=sq)
    assert o.footprint_cost(7.5, 6.5, 0.0) == -1.0       # the left edge ends on the 254 cell (6,5)
    cells2 = cells.copy()
    cells2[5, 6] = 0
    o = _scorer_with_map(oracle_mod, cells2, footprint=sq)
    assert o.footprint_cost(7.5, 6.5, 0.0) == 253.0      # 253 on an edge: max, not rejected
    cells3 = cells2.copy()
    cells3[5, 7] = 254
    o = _scorer_with_map(oracle_mod, cells3, footprint=sq)
    assert o.footprint_cost(7.5, 6.5, 0.0) == -1.0
    cells3[5, 7] = 255
    o = _scorer_with_map(oracle_mod, cells3, footprint=sq)
    assert o.footprint_cost(7.5, 6.5, 0.0) == -2.0
    # a vertex off the map -> -3 even when the centre is on it
    assert o.footprint_cost(0.5, 6.5, 0.0) == -3.0
    # obstacles strictly inside the polygon are not seen (only edges are rasterised)
    cells4 = np.zeros((20, 20), dtype=np.uint8)
    cells4[10, 10] = 254
    big = 3.0 * sq
    o = _scorer_with_map(oracle_mod, cells4, footprint=big)
    assert o.footprint_cost(10.5, 10.5, 0.3) == 0.0


def test_footprint_first_edge_code_wins(oracle_mod):
    """src/costmap_model.cpp:56-72: edges are walked in order and the first
    negative line cost is returned as is."""
    cells = np.zeros((20, 20), dtype=np.uint8)
    sq = np.array([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]])
    cells[11, 10] = 255   # on the first edge (top, y = 11.5)
    cells[9, 10] = 254    # on the third edge (bottom, y = 9.5)
    o = _scorer_with_map(oracle_mod, cells, footprint=sq)
    assert o.footprint_cost(10.5, 10.5, 0.0) == -2.0
    cells[11, 10], cells[9, 10] = 254, 255
    o = _scorer_with_map(oracle_mod, cells, footprint=sq)
    assert o.footprint_cost(10.5, 10.5, 0.0) == -1.0


# ---------------------------------------------------------------------------
def test_selection_rule(oracle_mod):
    """reference src/sfw_planner.cpp:338-417, :426-468."""
    lin = np.array([0.0, 0.35, 0.7])
    ang = np.array([0.0, 0.25, -0.25])
    sel = oracle_mod.select_best

    def costs(**kw):
        c = np.full(9, -1.0)
        c[0] = -2.0
        for k, v in kw.items():
            c[int(k[1:])] = v
        return c

    # plain minimum
    b = sel(lin, ang, costs(i4=3.0, i5=2.0, i7=2.5))
    assert (b["index"], b["vx"], b["vtheta"], b["cost"], b["n_valid"]) == (5, 0.35, -0.25, 2.0, 3)
    # equal cost: higher linear velocity wins (:397)
    assert sel(lin, ang, costs(i4=2.0, i7=2.0))["index"] == 7
    assert sel(lin, ang, costs(i7=2.0, i4=2.0))["index"] == 7
    # equal cost and linvel: smaller |angvel| wins (:403-404)
    assert sel(lin, ang, costs(i6=2.0, i7=2.0))["index"] == 6
    # full tie: the LATER iterate replaces (-w beats +w)
    assert sel(lin, ang, costs(i7=2.0, i8=2.0))["index"] == 8
    # costs above 10000 are never selectable (best_cost starts at 10000.0, :344)
    b = sel(lin, ang, costs(i4=10000.5, i5=20000.0))
    assert b["index"] == -1 and b["n_valid"] == 2 and (b["vx"], b["vtheta"], b["cost"]) == (0.0, 0.0, -1.0)
    # exactly 10000.0: selectable with linvel > 0, not with linvel == 0 (xv_=0, thetav_=0 initial best)
    assert sel(lin, ang, costs(i4=10000.0))["index"] == 4
    assert sel(lin, ang, costs(i1=10000.0))["index"] == -1
    # all invalid -> stop the robot (:456-468)
    b = sel(lin, ang, costs())
    assert b["index"] == -1 and b["n_valid"] == 0
    # zero cost is valid
    assert sel(lin, ang, costs(i3=0.0))["index"] == 3
    # NaN is never >= 0
    assert sel(lin, ang, costs(i3=float("nan"), i4=5.0))["index"] == 4


# ---------------------------------------------------------------------------
def _free_scene(n_people=0, size=400):
    import dataclasses

    w = dataclasses.replace(syn.WORKLOADS["cfg1"], map_size=size, n_people=n_people, footprint="point", sim_time=1.0)
    return syn.make_scene(w)


def _person(x, y, vx, vy, goal=None, goal_radius=0.35, dv=1.0, pid=1):
    a = SfwAgent()
    a.x, a.y, a.vx, a.vy = x, y, vx, vy
    a.has_goal = 1 if goal is not None else 0
    if goal is not None:
        a.goal_x, a.goal_y = goal
    a.goal_radius, a.desired_velocity, a.radius = goal_radius, dv, 0.35
    a.id, a.group_id = pid, -1
    return a


def test_collision_rejects_trajectory(oracle_mod):
    """reference :613-627: a person within robot_radius of the robot's post-step
    position -> -1.0, using the FLOAT product robot_radius*robot_radius."""
    scene = _free_scene()
    robot = syn.make_robot_agent(0, 0, 0.3, 0)
    p = default_params()
    o = oracle_mod.OracleScorer(p)
    o.load_scene(scene)
    # person standing 0.5 m ahead on the robot's path, robot drives at 0.7 m/s
    person = _person(0.6, 0.0, 0.0, 0.0)
    o.set_agents((SfwAgent * 2)(robot, person))
    c, _ = o.score_one(scene.robot_state, 0.7, 0.0, 0.0, scene.goal_args)
    assert c == -1.0
    # same person 5 m to the side: valid
    o.set_agents((SfwAgent * 2)(robot, _person(0.6, 5.0, 0.0, 0.0)))
    c, pts = o.score_one(scene.robot_state, 0.7, 0.0, 0.0, scene.goal_args)
    assert c > 0 and len(pts) == 40


def test_lone_person_closed_form(oracle_mod):
    """One pedestrian 1 km from the robot (pair forces underflow to exactly 0):
    walking at desiredVelocity straight at its goal the desired force is 0, so it
    moves uniformly until the goal is reached, then brakes with v *= (1 - dt/tau).
    Social work is exactly 0 => cost equals the pedestrian-free cost."""
    scene = _free_scene()
    p = default_params()
    o = oracle_mod.OracleScorer(p)
    o.load_scene(scene)
    robot = syn.make_robot_agent(0, 0, 0.3, 0)
    o.set_agents((SfwAgent * 1)(robot))
    base, _ = o.score_one(scene.robot_state, 0.5, 0.0, 0.1, scene.goal_args)
    far = _person(1000.0, 0.0, 1.0, 0.0, goal=(1000.6, 0.0))
    o.set_agents((SfwAgent * 2)(robot, far))
    c, _ = o.score_one(scene.robot_state, 0.5, 0.0, 0.1, scene.goal_args)
    assert c == base
    exp = _closed_form_cost(p, scene.robot_state, 0.5, 0.1, scene.goal_args)
    assert c == pytest.approx(exp, rel=1e-13)


def test_robot_id_equal_to_person_id_skips_wp(oracle_mod):
    """reference :697 via lightsfm computeSocialForce(Agent&, ...): the person
    whose id equals the robot's is skipped in Wp (robot id is never set in the
    reference)."""
    scene = _free_scene()
    p = default_params()
    o = oracle_mod.OracleScorer(p)
    o.load_scene(scene)
    robot = syn.make_robot_agent(0, 0, 0.3, 0, robot_id=0)
    person = _person(1.5, 0.8, -0.3, 0.0, goal=(0.0, 0.8), pid=1)
    o.set_agents((SfwAgent * 2)(robot, person))
    c_diff, _ = o.score_one(scene.robot_state, 0.5, 0.0, 0.0, scene.goal_args)
    person.id = 0
    o.set_agents((SfwAgent * 2)(robot, person))
    c_same, _ = o.score_one(scene.robot_state, 0.5, 0.0, 0.0, scene.goal_args)
    assert 0 < c_same < c_diff


def test_obstacle_force_adds_robot_work(oracle_mod):
    """reference :681-682: Wr includes the robot's obstacle-force norm.  Robot at
    rest rotating in place, one laser point 1 m away, no people: every step adds
    |F_obs| = forceFactorObstacle * exp(-(1.0 - radius)/sigma) to the social work."""
    scene = _free_scene()
    p = default_params()
    o = oracle_mod.OracleScorer(p)
    o.load_scene(scene)
    rs = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    robot = syn.make_robot_agent(0, 0, 0.0, 0)
    o.set_agents((SfwAgent * 1)(robot))
    base, _ = o.score_one(rs, 0.0, 0.0, 0.2, scene.goal_args)
    o.set_agents((SfwAgent * 1)(robot), np.array([[1.0, 0.0]]))
    c, _ = o.score_one(rs, 0.0, 0.0, 0.2, scene.goal_args)
    f_obs = 10.0 * math.exp(-(1.0 - 0.35) / 0.2)
    assert c - base == pytest.approx(p.social_weight * 40 * f_obs, rel=1e-12)
    # two symmetric points: the MEAN of the two forces cancels exactly
    o.set_agents((SfwAgent * 1)(robot), np.array([[1.0, 0.0], [-1.0, 0.0]]))
    c2, _ = o.score_one(rs, 0.0, 0.0, 0.2, scene.goal_args)
    assert c2 == pytest.approx(base, rel=1e-14)


def test_person_braking_after_goal_pop(oracle_mod):
    """lightsfm updatePosition/computeDesiredForce: once a goal is reached it is
    popped and the desired force becomes -v/tau.  A far-away person (no pair
    forces) starting inside its goal radius therefore brakes geometrically; the
    robot never feels it, so the cost equals the pedestrian-free cost."""
    scene = _free_scene()
    p = default_params()
    o = oracle_mod.OracleScorer(p)
    o.load_scene(scene)
    robot = syn.make_robot_agent(0, 0, 0.3, 0)
    o.set_agents((SfwAgent * 1)(robot))
    base, _ = o.score_one(scene.robot_state, 0.4, 0.0, -0.1, scene.goal_args)
    o.set_agents((SfwAgent * 2)(robot, _person(900.0, 900.0, 0.5, 0.0, goal=(900.1, 900.0))))
    c, _ = o.score_one(scene.robot_state, 0.4, 0.0, -0.1, scene.goal_args)
    assert c == base


# ---------------------------------------------------------------------------
# group forces (lightsfm computeGroupForce)
# ---------------------------------------------------------------------------
def _group_force_numpy(p, agents, idx):
    """Independent restatement: gaze + coherence + repulsion for agent idx."""
    a = agents[idx]
    members = [i for i, b in enumerate(agents) if b.group_id == a.group_id and b.group_id >= 0]
    if a.group_id < 0 or len(members) < 2:
        return np.zeros(2)
    n = len(members)
    pos = np.array([[b.x, b.y] for b in agents])
    center = pos[members].sum(axis=0) / n
    pa = pos[idx]
    goal = np.array([a.goal_x, a.goal_y])
    e = goal - pa
    dd = e / np.linalg.norm(e) if (a.has_goal and np.linalg.norm(e) > a.goal_radius) else np.zeros(2)
    f = np.zeros(2)
    com_others = (n * center - pa) / (n - 1)
    rel = com_others - pa
    ep = dd @ rel
    if np.linalg.norm(dd) > 0 and math.acos(ep / (np.linalg.norm(dd) * np.linalg.norm(rel))) > math.pi / 2:
        f += p.sfm_force_factor_group_gaze * (ep / (dd @ dd)) * dd
    rel = center - pa
    f += rel * p.sfm_force_factor_group_coherence * (math.tanh(np.linalg.norm(rel) - (n - 1) / 2) + 1) / 2
    rep = np.zeros(2)
    for m in members:
        if m != idx and np.linalg.norm(pa - pos[m]) < a.radius + agents[m].radius:
            rep += pa - pos[m]
    return f + p.sfm_force_factor_group_repulsion * rep


def _group_force_oracle(oracle_mod, p, agents, idx):
    arr = (SfwAgent * len(agents))(*agents)
    out = np.zeros(2)
    fn = oracle_mod.lib().sfwo_group_force
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    fn(C.addressof(p), C.addressof(arr), len(agents), idx, out.ctypes.data)
    return out


def test_group_force_hand_case(oracle_mod):
    """Three people in group 7 on a line, agent 0 walking AWAY from the others
    (centre of mass behind it -> gaze force pulls it back), agents 1 and 2 overlapping."""
    p = default_params()
    a0 = _person(0.0, 0.0, 0.0, 0.0, goal=(-5.0, 0.0), pid=1)
    a1 = _person(2.0, 0.0, 0.0, 0.0, goal=(10.0, 0.0), pid=2)
    a2 = _person(2.5, 0.0, 0.0, 0.0, goal=(10.0, 0.0), pid=3)
    for a in (a0, a1, a2):
        a.group_id = 7
    # agent 0: dd = (-1,0); COM of the others = (2.25, 0); ep = -2.25 < 0 -> gaze = 3 * (-2.25) * dd = (6.75, 0)
    # coherence: centre = (1.5,0), dist 1.5, maxd 1 -> (1.5,0) * 2 * (tanh(0.5)+1)/2
    exp0 = np.array([6.75 + 1.5 * (math.tanh(0.5) + 1.0), 0.0])
    assert _group_force_oracle(oracle_mod, p, [a0, a1, a2], 0) == pytest.approx(exp0, rel=1e-14)
    # agent 1: walks towards +x, the others' COM (1.25,0) is behind it (rel = -0.75): gaze = 3*(-0.75)*(1,0)
    # coherence: rel = (-0.5,0), dist 0.5 -> (-0.5)*(tanh(-0.5)+1); repulsion: overlaps agent 2 only: (-0.5, 0)
    exp1 = np.array([-2.25 - 0.5 * (math.tanh(-0.5) + 1.0) - 0.5, 0.0])
    assert _group_force_oracle(oracle_mod, p, [a0, a1, a2], 1) == pytest.approx(exp1, rel=1e-14)
    # a lone member or groupId < 0 feels nothing
    a2.group_id = 8
    a1.group_id = -1
    for i in range(3):
        assert not _group_force_oracle(oracle_mod, p, [a0, a1, a2], i).any()


def test_group_force_matches_independent_restatement(oracle_mod):
    p = default_params()
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(2, 8))
        agents = []
        for i in range(n):
            x, y = rng.uniform(-2, 2, 2)
            a = _person(x, y, *rng.uniform(-1, 1, 2), goal=tuple(rng.uniform(-6, 6, 2)), pid=i + 1)
            a.group_id = int(rng.integers(-1, 3))
            a.has_goal = int(rng.random() > 0.2)
            agents.append(a)
        for i in range(n):
            got = _group_force_oracle(oracle_mod, p, agents, i)
            assert got == pytest.approx(_group_force_numpy(p, agents, i), rel=1e-11, abs=1e-13)
