"""The ROS 2 adapter under fit-slam_amd/host/ros2/ parses and type-checks.

What this proves, exactly: `g++ -std=c++17 -fsyntax-only -Wall -Wextra` accepts the three adapter sources when every ROS 2 /
nav2 / tf2 / BehaviorTree.CPP / pluginlib / slam_msgs / reference header they include is answered by the declarations-only
file tests/ros2_decls/ros2_decls.hpp (<= 250 lines: the ~40 types and signatures the sources name).  So: the files are
well-formed C++17, every fs_* call matches include/fitslam_frontier.h (the REAL header is on the include path), every call
into ROS matches the SHAPE written down in that file, the plugin class derives from the plugin base and overrides its
registerNodes.  It proves nothing about ROS itself, about linking, or about behaviour; row a23 stays "partial" and the
string checks of test_ros2_adapter_sources.py stay."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fit-slam_amd", "host", "ros2")
DECLS = os.path.join(ROOT, "tests", "ros2_decls", "ros2_decls.hpp")
SOURCES = ["CostAssignerGPU.cpp", "FisherInfoManagerGPU.cpp", "FisherInfoBTPluginGPU.cpp"]
OURS = ("fitslam_frontier.h", "fitslam_frontier_ros2/")


def _external_includes():
    """Every #include of the adapter's sources and headers that is neither the C++ standard library nor ours."""
    found = set()
    files = [os.path.join(PKG, "src", s) for s in SOURCES]
    inc = os.path.join(PKG, "include", "fitslam_frontier_ros2")
    files += [os.path.join(inc, f) for f in os.listdir(inc)]
    for path in files:
        for name in re.findall(r'^\s*#include\s*[<"]([^>"]+)[>"]', open(path).read(), flags=re.M):
            if name.startswith(OURS) or ("/" not in name and "." not in name):      # ours / <vector>, <mutex>, ...
                continue
            found.add(name)
    return sorted(found)


def test_declarations_file_stays_small_and_has_no_bodies():
    text = open(DECLS).read()
    assert len(text.splitlines()) <= 250
    # declarations only: the one brace-body allowed is the logging macro's do { } while (0)
    code = re.sub(r"//.*", "", text)
    assert not re.search(r"\)\s*(const)?\s*\{[^}]*\breturn\b", code), "a function body crept into the declarations"


def test_adapter_sources_parse_against_the_declarations(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    ext = _external_includes()
    assert "rclcpp/rclcpp.hpp" in ext and "nav2_costmap_2d/costmap_2d_ros.hpp" in ext and "pluginlib/class_list_macros.hpp" in ext
    shim = tmp_path / "decls"
    for name in ext:                              # every external include path -> the one declarations file
        p = shim / name
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(f'#include "{DECLS}"\n')
    for src in SOURCES:
        cmd = [gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror=return-type", "-Wno-unused-parameter",
               "-I", str(shim), "-I", os.path.join(PKG, "include"), "-I", os.path.join(ROOT, "include"), os.path.join(PKG, "src", src)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, f"{src} does not parse:\n{res.stderr[-4000:]}"
        assert "warning" not in res.stderr, f"{src}:\n{res.stderr[-4000:]}"


def test_a_wrong_abi_call_is_caught(tmp_path):
    """The check has teeth: the same sources with one fs_* call given a wrong argument list must fail to parse."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    shim = tmp_path / "decls"
    for name in _external_includes():
        p = shim / name
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(f'#include "{DECLS}"\n')
    src = open(os.path.join(PKG, "src", "FisherInfoManagerGPU.cpp")).read()
    assert "fs_multi_upload_landmarks(scorer_, xyz_world.data()," in src
    bad = tmp_path / "bad.cpp"
    bad.write_text(src.replace("fs_multi_upload_landmarks(scorer_, xyz_world.data(),", "fs_multi_upload_landmarks(xyz_world.data(), scorer_,"))
    res = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-I", str(shim), "-I", os.path.join(PKG, "include"),
                          "-I", os.path.join(ROOT, "include"), str(bad)], capture_output=True, text=True)
    assert res.returncode != 0
