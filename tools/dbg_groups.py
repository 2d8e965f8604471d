import dataclasses, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.sfw_oracle import OracleScorer
from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer

def run(scene, tag):
    o = OracleScorer(default_params()); o.load_scene(scene)
    g = HipScorer(default_params()); g.load_scene(scene)
    oc, ob = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    gc, gb = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    v = (oc >= 0) & (gc >= 0)
    print(tag, "maxrel", (np.abs(gc[v]-oc[v])/np.abs(oc[v])).max() if v.any() else None, "inv_eq", np.array_equal(oc<0, gc<0))

w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=3, nw=3, n_people=6, seed=301)
# (a) far apart groups
sc = syn.make_scene(w)
for i in (1,2): sc.agents[i].group_id = 0
run(sc, "groups far apart (coherence/gaze only)")
# (b) close agents, no groups
sc = syn.make_scene(w)
sc.agents[2].x, sc.agents[2].y = sc.agents[1].x + 0.3, sc.agents[1].y + 0.2
run(sc, "close agents, no groups")
# (c) close agents in a group
for i in (1,2): sc.agents[i].group_id = 0
run(sc, "close agents in one group")
# (d) group but no goals
sc = syn.make_scene(w)
for i in (1,2): sc.agents[i].group_id = 0; sc.agents[i].has_goal = 0
run(sc, "group, members without goals")
# (e) one step only
sc = syn.make_scene(dataclasses.replace(w, sim_time=0.025))
for i in (1,2): sc.agents[i].group_id = 0
o = OracleScorer(default_params(sim_time=0.025)); o.load_scene(sc)
g = HipScorer(default_params(sim_time=0.025)); g.load_scene(sc)
oc,_ = o.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args); gc,_ = g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
print("one step", np.abs(gc-oc).max())
sc = syn.make_scene(dataclasses.replace(w, sim_time=0.05))
for i in (1,2): sc.agents[i].group_id = 0
o = OracleScorer(default_params(sim_time=0.05)); o.load_scene(sc)
g = HipScorer(default_params(sim_time=0.05)); g.load_scene(sc)
oc,_ = o.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args); gc,_ = g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
print("two steps", np.abs(gc-oc).max(), oc[1], gc[1])
