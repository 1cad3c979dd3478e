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


def build(force: bool = False, sanitize: bool = False) -> str | None:
    """sanitize: the mirror and its driver instrumented with gcc's AddressSanitizer + UndefinedBehaviorSanitizer (a binary of its own,
    `host_mirror_driver_asan`; the product library stays as built) — tests/test_host_mirror.py runs it once on the GPU box."""
    if not os.path.exists(DRIVER_SRC):
        return None
    lib = _build.build()
    exe = DRIVER + ("_asan" if sanitize else "")
    deps = [DRIVER_SRC, lib] + [h for h in HEADERS if os.path.exists(h)]
    if not force and os.path.exists(exe) and all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps):
        return exe
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"] if sanitize else ["-O2"]
    cmd = ["g++", *san, "-std=c++17", "-Wall", "-I", os.path.join(_HERE, "..", "include"), "-I", HOST,
           DRIVER_SRC, "-o", exe, "-L", os.path.dirname(lib), "-l" + os.path.basename(lib)[3:-3],
           "-Wl,-rpath,$ORIGIN/../csrc"]
    subprocess.check_call(cmd)
    return exe
