"""Builds libfitslam_frontier.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the
GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["fs_raymarch.hip", "fs_fim.hip", "fs_rank.hip", "fs_sort.hip", "fs_gridops.hip", "fs_cloud.hip", "fs_frontier.hip", "fs_keyframes.hip", "fs_multi.hip", "fs_capi.hip"]
HEADERS = ["fs_internal.h", os.path.join("..", "..", "include", "fitslam_frontier.h"), os.path.join("..", "..", "include", "fitslam_frontier_dev.h")]

# -ffp-contract=off: the ray set-up (fp64) and the landmark transform (fp32) must round exactly like
# the specification; fused multiply-adds appear only where the code calls fma explicitly.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage"]
# The FIM workers run two 512-thread (or one 1024-thread) workgroups per CU = 4 waves per SIMD: more than 128 VGPRs
# halves the occupancy, and a spill lands in the innermost loop.  The register allocation of these kernels has flipped
# on innocent-looking edits, so the build checks it.
RESOURCE_LIMITS = {"fs_fim_kernel": (128, 0)}

# ---- development builds.  Every knob below switches on code behind `#ifdef FS_DEV` (cycle stamps, the schedule recorder,
# range checks) or changes a tuning constant.  A build with any of them NEVER overwrites the production library: it goes to
# libfitslam_frontier_dev<hash of the flags>.so with objects of its own, and a process only loads it while the same
# environment variables are set — so a half-finished experiment cannot leave a doctored library behind for tests or bench.
DEV_FLAGS = []
if os.environ.get("FS_BOUNDS"):           # range-checked global accesses in the FIM (counter 30) and ray (29) kernels
    DEV_FLAGS += ["-DFS_FIM_BOUNDS", "-DFS_RAY_BOUNDS"]
    RESOURCE_LIMITS = {}                  # (the checks cost registers: a few bytes of scratch are fine in this build)
if os.environ.get("FS_POISON"):           # growing buffers are retired (never freed) and 0xCD-filled: stale pointers and reads of fresh memory show
    DEV_FLAGS.append("-DFS_POISON")
LINK_FLAGS = []
if os.environ.get("FS_HOST_ASAN"):        # AddressSanitizer on the HOST side of the library only (-fno-gpu-sanitize: the code objects stay plain gfx950);
    DEV_FLAGS += ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "-g"]   # run with LD_PRELOAD=<clang's libclang_rt.asan-x86_64.so>
    LINK_FLAGS = ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan"]
    RESOURCE_LIMITS = {}
if os.environ.get("FS_HOST_COV"):         # gcov-format line coverage of the HOST side (clang's --coverage for the host compile only; read with gcov-11)
    DEV_FLAGS += ["-Xarch_host", "-fprofile-arcs", "-Xarch_host", "-ftest-coverage"]
    LINK_FLAGS = LINK_FLAGS + ["--coverage"]
    RESOURCE_LIMITS = {}
if os.environ.get("FS_T1_WAVES_PER_EU"):  # occupancy target of the FIM worker's register allocation
    DEV_FLAGS.append("-DFS_T1_WAVES_PER_EU=" + os.environ["FS_T1_WAVES_PER_EU"])
    RESOURCE_LIMITS = {}
if os.environ.get("FS_T1_THREADS"):       # workgroup size of the FIM worker (default 512)
    DEV_FLAGS.append("-DFS_T1_THREADS=" + os.environ["FS_T1_THREADS"])
if os.environ.get("FS_FIM_STAMPS"):       # per-phase cycle counters of the FIM worker in counters 16..26 (tools/fim_stamps.py)
    DEV_FLAGS.append("-DFS_FIM_STAMPS")
    if os.environ.get("FS_FIM_STAMPS") == "wave":   # ... and the barrier wait / scoring time by wave index
        DEV_FLAGS.append("-DFS_FIM_STAMPS_PER_WAVE")
if os.environ.get("FS_RAY_UNROLL"):       # speculative cell loads in flight per lane of the ray walks (default 4)
    DEV_FLAGS.append("-DFS_RAY_UNROLL=" + os.environ["FS_RAY_UNROLL"])
if os.environ.get("FS_RAY_WAVES"):        # fans (waves) per workgroup of the ray-march kernel (default 4)
    DEV_FLAGS.append("-DFS_RAY_WAVES=" + os.environ["FS_RAY_WAVES"])
if os.environ.get("FS_FIM_SCHEDULE"):     # per-candidate start / duration / workgroup of the persistent FIM grid (tools/fim_schedule.py)
    DEV_FLAGS.append("-DFS_FIM_SCHEDULE")
if os.environ.get("FS_EXTRA_FLAGS"):      # anything else an experiment wants to define (A/B variants inside the sources)
    DEV_FLAGS += os.environ["FS_EXTRA_FLAGS"].split()
    RESOURCE_LIMITS = {} if os.environ.get("FS_NO_LIMITS") else RESOURCE_LIMITS
SUFFIX = ""
if DEV_FLAGS:
    import hashlib
    SUFFIX = "_dev" + hashlib.sha1(" ".join(DEV_FLAGS).encode()).hexdigest()[:8]
    HIPCC_FLAGS = HIPCC_FLAGS + ["-DFS_DEV"] + DEV_FLAGS
LIB = os.path.join(CSRC, f"libfitslam_frontier{SUFFIX}.so")
RESOURCES = os.path.join(CSRC, f"kernel_resources{SUFFIX}.json")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


STAMP = LIB + ".stamp"
OBJ_STAMPS = os.path.join(CSRC, f"objects{SUFFIX}.stamp.json")


def _sha256_file(path: str) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _inputs_digest(names) -> str:
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for name in names:
        h.update(name.encode())
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


def source_digest() -> str:
    """What the library was built from: SHA-256 over the sources, the headers and the compiler flags."""
    return _inputs_digest(SOURCES + HEADERS)


def object_digest(src: str) -> str:
    """What ONE object was compiled from: its source, every header and the flags."""
    return _inputs_digest([src] + HEADERS)


def _read_stamp():
    """(source digest, library SHA-256 or None) as the last successful link wrote them."""
    lines = open(STAMP).read().split()
    return (lines[0] if lines else ""), (lines[1] if len(lines) > 1 else None)


def needs_build() -> bool:
    """By CONTENT, not by modification time: a copy of the tree (the GPU box's snapshot, a fresh checkout next to a built
    library) may order the timestamps any way it likes — N ranks starting there must not all decide to rebuild.  The stamp
    names the sources AND the library file it was written for: a library that is not the one the stamp describes (a stale
    file copied over a fresh stamp, a truncated copy) is out of date too."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        src, lib = _read_stamp()
        return src != source_digest() or lib != _sha256_file(LIB)
    except OSError:
        return True


def _load_obj_stamps() -> dict:
    import json
    try:
        d = json.load(open(OBJ_STAMPS))
        return d if isinstance(d, dict) else {}
    except Exception:
        return {}


def _object_current(src: str, obj: str, stamps: dict) -> bool:
    """An object is reused only when the stamp says it was compiled from exactly these inputs AND the file on disk is the
    one that compile produced.  Modification times decide nothing."""
    rec = stamps.get(os.path.basename(obj))
    if not rec or not os.path.exists(obj):
        return False
    return rec.get("inputs") == object_digest(src) and rec.get("object") == _sha256_file(obj)


def _parse_resource_remarks(text: str) -> dict:
    """kernel -> {vgprs, sgprs, scratch, lds, occupancy} from clang's -Rpass-analysis=kernel-resource-usage remarks."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def _record_resources(src: str, usage: dict) -> None:
    import json
    allr = {}
    if os.path.exists(RESOURCES):
        try:
            allr = json.load(open(RESOURCES))
        except Exception:
            allr = {}
    allr[src] = usage
    json.dump(allr, open(RESOURCES, "w"), indent=1, sort_keys=True)
    for name, u in usage.items():
        for key, (max_vgprs, max_scratch) in RESOURCE_LIMITS.items():
            if key in name and (u.get("vgprs", 0) > max_vgprs or u.get("scratch", 0) > max_scratch):
                raise RuntimeError(f"{src}: {name} uses {u.get('vgprs')} VGPRs and {u.get('scratch')} B of scratch per lane "
                                   f"(limits {max_vgprs} / {max_scratch}): the register allocation regressed")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what is out of date and link.  Safe to call from several processes at once (N ranks of one launcher, pytest-xdist
    workers): the work happens under an exclusive lock on the source directory, and whoever gets the lock second finds the
    library up to date and returns."""
    if not force and not needs_build():
        return LIB
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():       # another process built it while this one waited
                return LIB
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    import json
    cc = hipcc()
    objs = []
    stamps = _load_obj_stamps()
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", SUFFIX + ".o"))
        src_path = os.path.join(CSRC, src)
        if force or not _object_current(src, obj, stamps):
            cmd = [cc, *HIPCC_FLAGS, "-c", src_path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr, flush=True)
            res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            usage = _parse_resource_remarks(res.stderr)
            import re as _re
            # (the remarks come with source-context lines — "  160 | {", "      | ^" — that are noise without them)
            rest = "\n".join(l for l in res.stderr.splitlines()
                             if "kernel-resource-usage" not in l and l.strip() and not _re.match(r"^\s*\d*\s*\|", l))
            if rest:
                print(rest, file=sys.stderr, flush=True)       # (never stdout: bench.py's one JSON line lives there)
            if res.returncode != 0:
                raise subprocess.CalledProcessError(res.returncode, cmd)
            _record_resources(src, usage)
            # (written object by object: an interrupted build keeps what it finished)
            stamps[os.path.basename(obj)] = {"inputs": object_digest(src), "object": _sha256_file(obj)}
            tmp_s = OBJ_STAMPS + f".tmp{os.getpid()}"
            json.dump(stamps, open(tmp_s, "w"), indent=1, sort_keys=True)
            os.replace(tmp_s, OBJ_STAMPS)
        objs.append(obj)
    # linked under a temporary name and renamed into place: a process that loads the library while another one links never
    # sees a half-written file
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *LINK_FLAGS, "-o", tmp, *objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr, flush=True)
    subprocess.check_call(cmd, stdout=sys.stderr)
    os.replace(tmp, LIB)
    with open(STAMP + f".tmp{os.getpid()}", "w") as f:
        f.write(source_digest() + "\n" + _sha256_file(LIB) + "\n")
    os.replace(STAMP + f".tmp{os.getpid()}", STAMP)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
