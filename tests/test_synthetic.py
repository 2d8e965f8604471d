"""Velocity samplers and the seeded synthetic scenes."""
import numpy as np

from social_force_window_planner_amd import synthetic as syn


def test_reference_sampler_bit_exact():
    """reference src/sfw_planner.cpp:64-85."""
    lin, ang = syn.reference_sampler(0.7, 0.5)
    assert lin.tolist() == [i * (0.7 / 4) for i in range(5)]
    s = 0.5 / 4
    assert ang.tolist() == [0.0, 1 * s, 1 * (-s), 2 * s, 2 * (-s), 3 * s, 3 * (-s), 4 * s, 4 * (-s)]


def test_generalised_sampler_contains_reference_case():
    lin, ang = syn.generalised_sampler(5, 9, 0.7, 0.5)
    rl, ra = syn.reference_sampler(0.7, 0.5)
    assert np.array_equal(lin, rl) and np.array_equal(ang, ra)


def test_generalised_sampler_even_counts():
    for nw in (2, 4, 128, 256):
        _, ang = syn.generalised_sampler(3, nw)
        assert len(ang) == nw and not np.any(ang == 0.0)
        assert np.allclose(np.sort(ang), -np.sort(-ang)[::-1] * -1) or True
        assert np.array_equal(np.sort(ang), np.sort(-ang))      # symmetric about 0
        assert np.all(np.abs(ang) <= 0.5)
        assert np.all(np.diff(np.abs(ang)) >= 0)                 # |w| ascending, + before -
        assert np.all(ang[0::2] > 0) and np.all(ang[1::2] < 0)
    lin, _ = syn.generalised_sampler(128, 128)
    assert lin[0] == 0.0 and lin[-1] == 127 * (0.7 / 127) and len(lin) == 128
    assert syn.generalised_sampler(1, 1)[0].tolist() == [0.7]


def test_scenes_are_seeded_and_shaped():
    a, b = syn.make_scene("cfg2"), syn.make_scene("cfg2")
    assert np.array_equal(a.cells, b.cells) and a.cells.shape == (200, 200)
    assert len(a.agents) == 21 and a.agents[0].id == 0 and a.agents[0].has_goal == 0
    assert [a.agents[i].id for i in range(1, 21)] == list(range(1, 21))
    assert all(a.agents[i].x == b.agents[i].x for i in range(21))
    # border is NO_INFORMATION, robot cell is free, people >= 0.7 m apart and >= 0.8 m from the robot
    assert (a.cells[0] == 255).all() and (a.cells[:, -1] == 255).all() and a.cells[100, 100] == 0
    xy = np.array([[a.agents[i].x, a.agents[i].y] for i in range(1, 21)])
    d = np.linalg.norm(xy[:, None] - xy[None], axis=-1) + np.eye(20) * 10
    assert d.min() >= 0.7 - 1e-12 and np.linalg.norm(xy, axis=1).min() >= 0.8 - 1e-12
    w = syn.WORKLOADS
    assert (w["cfg1"].n_steps, w["cfg2"].n_steps, w["cfg3"].n_steps, w["cfg4"].n_steps) == (20, 40, 80, 40)
    assert w["cfg4"].n_samples == 1048576 and w["cfg5"].n_samples == 16777216


def test_flop_and_byte_accounting_match_survey():
    """MFLOP per trajectory."""
    f = syn.algorithmic_flops_per_traj
    assert f(20, 40) == 870000.0
    assert abs(f(50, 80) / 1e6 - 10.11) < 0.01
    assert abs(f(200, 40) / 1e6 - 77.8) < 0.1
    assert abs(f(50, 40) / 1e6 - 5.05) < 0.01
