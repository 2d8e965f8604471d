"""The C-ABI library loads, exports every symbol include/sfw_hip.h declares, and
refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from social_force_window_planner_amd import planner
from social_force_window_planner_amd._abi import (EXPORTED_SYMBOLS, SFW_ERR_INVALID_ARG, SFW_ERR_NO_DEVICE, SFW_OK,
                                                   SfwParams, default_params)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu():
    import torch

    return torch.cuda.is_available()


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "sfw_hip.h")).read()
    declared = set(re.findall(r"\b(sfw_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"sfw_planner_s"}
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    exported = planner.exported_symbols()
    assert all(exported.values()), [n for n, ok in exported.items() if not ok]
    assert planner.lib().sfw_abi_version() == 2


def test_params_default_matches_python_mirror():
    p = SfwParams()
    planner.lib().sfw_params_default(C.byref(p))
    q = default_params()
    for name, _ in SfwParams._fields_:
        assert getattr(p, name) == getattr(q, name), name
    # reference sfw_planner.hpp:56-66 + lightsfm defaults
    assert (p.max_vel_x, p.sim_time, p.sim_granularity) == (0.7, 1.0, 0.025)
    assert (p.social_weight, p.costmap_weight, p.angle_weight, p.distance_weight, p.vel_weight) == (1.2, 2.0, 0.7, 1.0, 1.0)
    assert (p.sfm_force_factor_social, p.sfm_lambda, p.sfm_gamma, p.sfm_n, p.sfm_n_prime) == (2.1, 2.0, 0.35, 2.0, 3.0)


def test_argument_validation_without_device():
    L = planner.lib()
    h = C.c_void_p()
    assert L.sfw_create(None, 0, C.byref(h)) == SFW_ERR_INVALID_ARG
    bad = default_params(sim_granularity=0.0)
    assert L.sfw_create(C.byref(bad), 0, C.byref(h)) == SFW_ERR_INVALID_ARG
    bad = default_params(precision=7)
    assert L.sfw_create(C.byref(bad), 0, C.byref(h)) == SFW_ERR_INVALID_ARG
    assert L.sfw_destroy(None) == SFW_OK
    assert L.sfw_last_error(None) == b"null handle"


@pytest.mark.skipif(_gpu(), reason="a GPU is visible")
def test_no_cpu_fallback():
    """On a box without a GPU the product path must fail loudly."""
    L = planner.lib()
    h = C.c_void_p()
    p = default_params()
    assert L.sfw_create(C.byref(p), 0, C.byref(h)) == SFW_ERR_NO_DEVICE
    assert not h.value
    with pytest.raises(planner.SfwError) as e:
        planner.HipScorer()
    assert e.value.status == SFW_ERR_NO_DEVICE


def test_product_does_not_link_the_oracle():
    """Neither product library depends on oracle/, and no product source mentions it."""
    import subprocess

    for so in (planner.LIB_PATH, os.path.join(ROOT, "social_force_window_planner_amd", "host", "libsfw_host.so")):
        out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
        assert "oracle" not in out
    pkg = os.path.join(ROOT, "social_force_window_planner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libsfw_oracle" not in src, f


def test_header_is_plain_c99(tmp_path):
    """include/sfw_hip.h is the drop-in boundary: it has to compile as C (a cgo / JNI / ctypes binding sees
    nothing else), with every declared function callable from a C translation unit."""
    import os
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "sfw_hip.h")).read()
    names = sorted(set(re.findall(r"\b(sfw_[a-z_0-9]+)\s*\(", header)))
    assert set(names) == set(EXPORTED_SYMBOLS)
    src = tmp_path / "abi_check.c"
    src.write_text('#include "sfw_hip.h"\n'
                   + "".join(f"void *use_{n}(void) {{ return (void *){n}; }}\n" for n in names)
                   + "int main(void) { sfw_params p; sfw_params_default(&p); return (int)sizeof(sfw_agent) + (int)sizeof(sfw_plan_info); }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Wno-pedantic", "-I", os.path.join(root, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
