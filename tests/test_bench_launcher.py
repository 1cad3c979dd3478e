"""bench.py as its own launcher (SURVEY.md §8(e) "Reporting"): `python bench.py --gpus N` without
a launcher around it must start N ranks itself, relay ONE result line that says n_gpus = N, and refuse a mismatch
between --gpus and WORLD_SIZE.  The GPU-free part of that is exercised here through --dry-run (gloo rendezvous + the
all-gather of dummy records; nothing is scored and no value is reported)."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_launcher_command_is_one_process_per_gpu_under_torchrun():
    b = _bench_module()
    argv = ["--gpus", "4", "--steps", "7", "--scaling", "strong"]
    cmd = b.launcher_command(b.parse(argv), argv, port=12345)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-len(argv) - 1] == BENCH and cmd[-len(argv):] == argv          # the ranks get the same arguments


def test_defaults_are_the_headline_configuration():
    b = _bench_module()
    a = b.parse([])
    assert (a.gpus, a.workload, a.scaling) == (1, "C3", "weak") and a.repeats >= 5
    # one stream, batches strictly back to back: with more contexts the FIM kernels of two batches share the CUs and the
    # per-launch durations behind `roofline` stop measuring the kernel
    assert a.pipeline == 1 and b.parse(["--pipeline", "2"]).pipeline == 2


def test_two_ranks_are_spawned_and_rank0_line_is_relayed():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["order_restored"] is True and j["value"] is None
    assert "dry-run rank 0/2" in r.stderr and "dry-run rank 1/2" in r.stderr              # both ranks ran


def test_gpus_must_match_world_size():
    env = _clean_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE 1" in r.stderr and not r.stdout.strip()


def test_source_hash_tracks_the_kernel_sources():
    b = _bench_module()
    h = b.source_hash()
    assert len(h) == 16 and h == b.source_hash()
    # a counter profile is only replayed into the bench line when it names exactly these sources
    assert b.counter_profile("no-such-workload", 0) is None


def test_counter_profiles_are_replayed_only_for_what_they_measured():
    """bench.py copies counter-derived figures (HBM traffic, what limits the kernel) into its line only from a profile that was
    collected on exactly these kernel sources, this workload AND this visibility volume — the cone-off request runs another FIM
    worker instantiation, with counters of its own (profiles/pmc_summary_ref_request.json)."""
    b = _bench_module()
    for name, angle in (("pmc_summary.json", 1.0), ("pmc_summary_ref_request.json", 4.0)):
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert float(j.get("fim_angle", 1.0)) == angle and j["workload"] == "C3"
        if j["source_hash"] == b.source_hash():                      # (a committed profile of older sources is simply not replayed)
            assert b.counter_profile("C3", 0, angle)["fim_angle"] == angle
    assert b.counter_profile("C3", 0, 2.5) is None and b.counter_profile("C5", 0, 1.0) is None and b.counter_profile("C3", 160, 1.0) is None
