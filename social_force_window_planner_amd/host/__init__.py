"""C++ host mirror of the reference's SFWPlanner (ROS-free) over the C ABI."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsfw_host.so")


def build(force=False):
    mk = os.path.join(_HERE, "Makefile")
    if not os.path.exists(mk):
        return None
    r = subprocess.run(["make", "-C", _HERE, "all"] + (["-B"] if force else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libsfw_host.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH
