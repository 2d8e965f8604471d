"""Loads a tests/golden/*.npz fixture (pure data) into the ABI types."""
import glob
import os

import numpy as np

from social_force_window_planner_amd._abi import SfwAgent, default_params

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INT_FIELDS = {"has_goal", "id", "group_id", "reserved"}


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.cells = z["cells"]
        self.origin_x, self.origin_y, self.resolution = (float(v) for v in z["origin"])
        self.footprint = z["footprint"]
        fields = [str(f) for f in z["agent_fields"]]
        rows = z["agents"]
        self.agents = (SfwAgent * len(rows))()
        for i, row in enumerate(rows):
            for f, v in zip(fields, row):
                setattr(self.agents[i], f, int(v) if f in INT_FIELDS else float(v))
        self.obstacles = z["obstacles"]
        self.robot_state = tuple(float(v) for v in z["robot_state"])
        self.goal_args = tuple(float(v) for v in z["goal_args"])
        self.linvels, self.angvels = z["linvels"], z["angvels"]
        self.param_kw = {str(f): float(v) for f, v in zip(z["param_fields"], z["params"])}
        self.costs = z["costs"]
        b = z["best"]
        self.best = {"index": int(b[0]), "cost": float(b[1]), "vx": float(b[2]), "vtheta": float(b[3]),
                     "n_valid": int(b[4])}

    def params(self, **kw):
        return default_params(**{**self.param_kw, **kw})

    def load_into(self, scorer):
        scorer.set_costmap(self.cells, self.origin_x, self.origin_y, self.resolution)
        scorer.set_footprint(self.footprint)
        scorer.set_agents(self.agents, self.obstacles)
