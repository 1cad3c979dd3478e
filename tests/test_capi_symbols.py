"""The C-ABI library builds for gfx950, loads without a GPU, exports every symbol
include/fitslam_frontier.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", "fitslam_frontier.h"), os.path.join(ROOT, "include", "fitslam_frontier_dev.h")]


def _declared(headers=HEADERS):
    names = set()
    for h in headers:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_diagnostics_live_in_the_dev_header():
    """The drop-in header declares the scoring path only; timing, counters, knobs and the self test are in the dev header."""
    public, dev = set(_declared(HEADERS[:1])), set(_declared(HEADERS[1:]))
    assert dev == {"fs_enable_kernel_timing", "fs_kernel_time", "fs_set_option", "fs_get_counter", "fs_selftest_fp64"}
    assert not (public & dev)


def test_header_declares_expected_entry_points(fs):
    names = _declared()
    assert "fs_score_candidates" in names and "fs_score_arrival" in names and "fs_score_fim" in names
    assert set(names) == set(fs.capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(fs):
    lib = fs.load_library()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.fs_abi_version() == 1


def test_library_is_a_gfx950_code_object(fs):
    path = fs._build.LIB
    assert os.path.exists(path)
    blob = open(path, "rb").read()
    assert b"gfx950" in blob and b"fs_fim_kernel" in blob and b"fs_raymarch_kernel" in blob


def test_record_layout(fs):
    assert fs.RECORD_DTYPE.itemsize == 32
    assert [fs.RECORD_DTYPE.fields[n][1] for n in ("arrival", "argmax", "yaw", "info_ref", "trace", "logdet", "n_visible", "flags")] == \
        [0, 4, 8, 12, 16, 20, 24, 28]
    assert ctypes.sizeof(fs.capi.RayParamsC) == 8 * 4 + 4 * 2 + 8 * 16 + 4 * 4 + 8 * 2 + 8 * 4
    assert ctypes.sizeof(fs.capi.FimParamsC) == 16


def test_no_cpu_fallback_without_gpu(fs):
    """On a box without an MI355X the context cannot be created — the product never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fs.FsError) as e:
        fs.FrontierScorer(device=0)
    assert e.value.code == fs.capi.FS_E_NO_DEVICE
    lib = fs.load_library()
    assert lib.fs_score_candidates(None, 0, None, None, None, None, None) == fs.capi.FS_E_INVALID


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under fit-slam_amd/ may reference it."""
    pkg = os.path.join(ROOT, "fit-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "fso_" not in src and "libfso_oracle" not in src, os.path.join(dirpath, f)
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_synth_is_deterministic(fs):
    a = fs.synth.make_workload("C1")
    b = fs.synth.make_workload("C1")
    for k in ("cells", "goals", "landmarks", "frontier_size", "blacklisted"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k))
    assert a.cells.shape == (64, 64, 64) and a.goals.shape == (200, 3) and a.landmarks.shape == (2000, 3)
    assert a.rays_per_candidate == 32
    v = a.cells[np.clip(((a.goals[:, 2] - a.origin[2]) / a.resolution).astype(int), 0, 63),
                ((a.goals[:, 1] - a.origin[1]) / a.resolution).astype(int),
                ((a.goals[:, 0] - a.origin[0]) / a.resolution).astype(int)]
    assert np.all(v == 0)          # candidates are free frontier cells


def test_fim_kernels_keep_their_register_budget():
    """The FIM workers need <= 128 VGPRs and no scratch to keep 4 waves per SIMD (the build records the compiler's
    kernel-resource remarks and refuses to link a library that regressed)."""
    import importlib
    import json
    b = importlib.import_module("fit-slam_amd._build")
    b.build()
    usage = json.load(open(b.RESOURCES))["fs_fim.hip"]
    seen = 0
    for name, u in usage.items():
        for key, (max_vgprs, max_scratch) in b.RESOURCE_LIMITS.items():
            if key in name:
                seen += 1
                assert u["vgprs"] <= max_vgprs and u["scratch"] <= max_scratch, (name, u)
    assert seen >= 4


def test_concurrent_builds_take_turns(tmp_path):
    """N ranks of one launcher may all decide to build: the build runs under an exclusive lock and re-checks the content stamp
    once it holds it (fit-slam_amd/_build.py), so the second process compiles nothing.  Here: two processes call build() at
    once on an up-to-date tree and both return the library; needs_build() goes by content, not by timestamps."""
    import subprocess
    import sys
    code = ("import importlib, sys, os; sys.path.insert(0, %r); b = importlib.import_module('fit-slam_amd._build'); "
            "os.utime(os.path.join(b.CSRC, 'fs_capi.hip')); "          # a newer timestamp alone must not trigger a build
            "assert not b.needs_build(); print(b.build())") % ROOT
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        assert out.strip().endswith("libfitslam_frontier.so")


def test_build_stamps_go_by_content_never_by_mtime(tmp_path, monkeypatch):
    """VERDICT r04 weak #6 / ADVICE: needs_build() went by content, but which OBJECTS to recompile went by modification time —
    a copied tree whose .o files are newer than an edited source linked stale objects under a fresh stamp.  Now every object
    carries a stamp of its inputs (source + headers + flags) and of the file the compile produced.  On a one-source copy of the
    tree: (1) an object touched into the future is still recompiled when its source changes; (2) an object file swapped for
    another one under an unchanged stamp is recompiled; (3) a library swapped under an unchanged stamp is relinked; (4) with
    nothing changed, nothing is compiled whatever the timestamps say."""
    import importlib
    import shutil
    import time
    b = importlib.import_module("fit-slam_amd._build")
    csrc = tmp_path / "pkg" / "csrc"
    csrc.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    for h in ("fitslam_frontier.h", "fitslam_frontier_dev.h"):
        shutil.copy(os.path.join(ROOT, "include", h), tmp_path / "include" / h)
    shutil.copy(os.path.join(b.CSRC, "fs_internal.h"), csrc / "fs_internal.h")
    shutil.copy(os.path.join(b.CSRC, "fs_keyframes.hip"), csrc / "fs_keyframes.hip")
    monkeypatch.setattr(b, "CSRC", str(csrc))
    monkeypatch.setattr(b, "SOURCES", ["fs_keyframes.hip"])
    monkeypatch.setattr(b, "HEADERS", ["fs_internal.h", os.path.join("..", "..", "include", "fitslam_frontier.h"),
                                       os.path.join("..", "..", "include", "fitslam_frontier_dev.h")])
    monkeypatch.setattr(b, "LIB", str(csrc / "libt.so"))
    monkeypatch.setattr(b, "STAMP", str(csrc / "libt.so.stamp"))
    monkeypatch.setattr(b, "OBJ_STAMPS", str(csrc / "objects.stamp.json"))
    monkeypatch.setattr(b, "RESOURCES", str(csrc / "kernel_resources.json"))
    monkeypatch.setattr(b, "RESOURCE_LIMITS", {})
    compiled = []
    real_run = b.subprocess.run
    monkeypatch.setattr(b.subprocess, "run", lambda cmd, **kw: (compiled.append(cmd[-3]), real_run(cmd, **kw))[1])
    obj = csrc / "fs_keyframes.o"

    assert b.needs_build()
    b.build()
    assert len(compiled) == 1 and not b.needs_build()
    first = b._sha256_file(str(obj))
    # (4) every timestamp scrambled, nothing changed: no build, and a forced look compiles nothing
    past, future = time.time() - 86400, time.time() + 86400
    os.utime(csrc / "fs_keyframes.hip", (future, future)); os.utime(obj, (past, past)); os.utime(b.LIB, (past, past))
    assert not b.needs_build()
    b._build_locked(False, False)
    assert len(compiled) == 1
    # (1) the object far in the future, the source edited: recompiled, and the stamp names the new inputs
    os.utime(obj, (future + 86400, future + 86400))
    with open(csrc / "fs_keyframes.hip", "a") as f:
        f.write("\n// an edit\n__global__ void fs_stamp_test_kernel(int *p) { *p = 7; }\n")
    assert b.needs_build()
    b.build()
    assert len(compiled) == 2 and not b.needs_build()
    assert b._sha256_file(str(obj)) != first
    assert b._load_obj_stamps()["fs_keyframes.o"]["inputs"] == b.object_digest("fs_keyframes.hip")
    # (2) a different object file under the unchanged stamp
    with open(obj, "ab") as f:
        f.write(b"\0stale")
    assert not b._object_current("fs_keyframes.hip", str(obj), b._load_obj_stamps())
    b._build_locked(False, False)
    assert len(compiled) == 3
    # (3) a different library under the unchanged stamp
    with open(b.LIB, "ab") as f:
        f.write(b"\0stale")
    assert b.needs_build()
    b.build()
    assert len(compiled) == 3 and not b.needs_build()      # relinked from the current object, nothing recompiled


def test_headers_are_plain_c_and_a_c_program_links_against_the_library(tmp_path, fs):
    """The drop-in boundary is a C ABI: both headers compile as C99 (-pedantic, no C++ construct outside the extern "C" guards),
    and a C program that names every declared entry point links against the built library (no GPU needed to link: nothing runs)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    names = _declared()
    src = tmp_path / "abi.c"
    body = "\n".join(f"    p[{i}] = (fn){n};" for i, n in enumerate(names))
    src.write_text('#include <stddef.h>\n#include "fitslam_frontier.h"\n#include "fitslam_frontier_dev.h"\n'
                   f"typedef void (*fn)(void);\nint main(void)\n{{\n    fn p[{len(names)}];\n    fs_record r; fs_ray_params rp; fs_fim_params fp; fs_frontier_cluster cl;\n"
                   f"{body}\n    (void)r; (void)rp; (void)fp; (void)cl;\n    return p[0] && sizeof(fs_record) == 32 ? 0 : 1;\n}}\n")
    inc = os.path.join(ROOT, "include")
    res = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    lib = fs._build.LIB
    exe = tmp_path / "abi"
    res = subprocess.run([gcc, "-std=c99", "-I", inc, str(src), "-o", str(exe), lib, f"-Wl,-rpath,{os.path.dirname(lib)}",
                          "-Wl,--unresolved-symbols=ignore-in-shared-libs"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]


def test_window_arguments_of_the_binding(fs):
    """fs_update_grid_region through the binding: a packed window passes (0, 0) strides; a numpy VIEW into the caller's whole map
    passes the view's pointer and strides (nothing is copied); 2-D windows get a z extent of one.  Host logic only."""
    import numpy as np
    cells = np.arange(4 * 10 * 12, dtype=np.uint8).reshape(4, 10, 12)
    ptr, sx, sy, sz, rs, ss, keep = fs.capi._window_args(cells[1:3, 2:7, 3:8].copy(), False)
    assert (sx, sy, sz, rs, ss) == (5, 5, 2, 0, 0) and keep.flags["C_CONTIGUOUS"]
    view = cells[1:3, 2:7, 3:8]
    ptr, sx, sy, sz, rs, ss, keep = fs.capi._window_args(view, True)
    assert (sx, sy, sz, rs, ss) == (5, 5, 2, 12, 120) and ptr.value == view.ctypes.data and keep.base is not None
    ptr, sx, sy, sz, rs, ss, keep = fs.capi._window_args(cells[0, 4:5, 2:9], True)               # one row of a 2-D map
    assert (sx, sy, sz, rs, ss) == (7, 1, 1, 0, 0)
    ptr, sx, sy, sz, rs, ss, keep = fs.capi._window_args(np.zeros((0, 7), dtype=np.uint8), True)  # empty
    assert (sx, sy, sz) == (7, 0, 1)
    import pytest
    with pytest.raises(ValueError):
        fs.capi._window_args(cells[:, :, ::2], True)                                              # not contiguous along x
