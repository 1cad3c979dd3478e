"""Builds the C++ host-side mirror's test driver (fit-slam_amd/host) with g++ against the C ABI."""
from __future__ import annotations

import os
import subprocess

from . import _build

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(_HERE, "host")
DRIVER_SRC = os.path.join(HOST, "host_mirror_driver.cpp")
DRIVER = os.path.join(HOST, "host_mirror_driver")
HEADERS = [os.path.join(HOST, "frontier_scoring.hpp"), os.path.join(_HERE, "..", "include", "fitslam_frontier.h")]


def build(force: bool = False) -> str | None:
    if not os.path.exists(DRIVER_SRC):
        return None
    lib = _build.build()
    deps = [DRIVER_SRC, lib] + [h for h in HEADERS if os.path.exists(h)]
    if not force and os.path.exists(DRIVER) and all(os.path.getmtime(d) <= os.path.getmtime(DRIVER) for d in deps):
        return DRIVER
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(_HERE, "..", "include"), "-I", HOST,
           DRIVER_SRC, "-o", DRIVER, "-L", os.path.dirname(lib), "-lfitslam_frontier",
           "-Wl,-rpath,$ORIGIN/../csrc"]
    subprocess.check_call(cmd)
    return DRIVER
