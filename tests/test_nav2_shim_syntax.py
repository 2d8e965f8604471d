"""The nav2 plugin shell type-checks against the interface it claims.

No ROS 2 / nav2 in the image: the shim is compiled with -fsyntax-only against tests/nav2_stubs/ (declarations of the
names it uses, with the Foxy signatures of nav2_core::Controller that the reference overrides —
reference include/social_force_window_planner/sfw_planner_node.hpp:73-116).  A compile guard, not an oracle."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "social_force_window_planner_amd", "host", "nav2_shim")
FLAGS = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "nav2_stubs"),
         "-I" + os.path.join(ROOT, "include")]


def _check(path):
    return subprocess.run(FLAGS + [path], capture_output=True, text=True)


def test_shim_type_checks_against_the_foxy_controller_interface():
    r = _check(os.path.join(SHIM, "sfw_planner_node.cpp"))
    assert r.returncode == 0, r.stderr[-4000:]


@pytest.mark.parametrize("old,new,expect", [
    # signature drift against the base class is caught by `override`
    ("void setPlan(const nav_msgs::msg::Path &path) override;", "void setPlan(nav_msgs::msg::Path &path) override;", "override"),
    # a pure virtual left open is caught by the plugin-export assertion
    ("  void cleanup() override;\n", "", "cleanup"),
    # a host-library name that does not exist is caught
    ("sfw_planner_->updatePlan(local);", "sfw_planner_->updateThePlan(local);", "updateThePlan"),
    # a ROS parameter helper used with a wrong type is caught
    ("sensor_iface_->laserCb(s);", "sensor_iface_->laserCb(m);", "laserCb"),
])
def test_the_guard_really_guards(tmp_path, old, new, expect):
    """Negative controls: seeded mistakes in a copy of the shim must fail the same command."""
    d = tmp_path / "host" / "nav2_shim"
    shutil.copytree(os.path.dirname(SHIM), tmp_path / "host", ignore=shutil.ignore_patterns("*.so", "drive_demo", "__pycache__"))
    hit = False
    for name in ("sfw_planner_node.hpp", "sfw_planner_node.cpp"):
        p = d / name
        s = p.read_text()
        if old in s:
            p.write_text(s.replace(old, new))
            hit = True
    assert hit, f"seed pattern not found: {old!r}"
    r = _check(str(d / "sfw_planner_node.cpp"))
    assert r.returncode != 0 and expect in r.stderr, r.stderr[-2000:]


def test_plugin_descriptor_names_match_the_reference():
    """Class / base / library names as the reference ships them (reference sfw_plugin.xml:1-9, CMakeLists.txt:76)."""
    cm = open(os.path.join(SHIM, "CMakeLists.txt")).read()
    assert "project(social_force_window_planner)" in cm
    assert 'social_force_window_planner::SFWPlannerNode' in cm and 'nav2_core::Controller' in cm
    src = open(os.path.join(SHIM, "sfw_planner_node.cpp")).read()
    assert "PLUGINLIB_EXPORT_CLASS(social_force_window_planner::SFWPlannerNode, nav2_core::Controller)" in src
