"""The N > 1 data path of bench.py on one GPU: scoring on an explicit torch stream followed by the RCCL all-gather of the
32-byte records (a one-rank "nccl" group: same calls, same stream ordering, no second device needed)."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_scoring_then_rccl_all_gather_on_one_stream(fs):
    import torch
    import torch.distributed as dist
    shard = importlib.import_module("fit-slam_amd.shard")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        w = fs.synth.make_workload("C2", n_cand=3000)
        stream = torch.cuda.Stream(device=dev)
        prev = torch.cuda.current_stream(dev)
        torch.cuda.set_stream(stream)
        try:
            sc = fs.FrontierScorer(device=0, stream=stream.cuda_stream)
            sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
            sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate()
            sc.set_fim_params(14.0, 1.0); sc.max_arrival()
            want = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)          # host-buffer entry point
            n = w.goals.shape[0]
            cap = shard.shard_capacity(n, 1)
            d_goal = torch.from_numpy(w.goals).to(dev); d_fs = torch.from_numpy(w.frontier_size).to(dev)
            d_bl = torch.from_numpy(w.blacklisted).to(dev)
            # as in bench.py: two record buffers, the gather of batch k is asynchronous and overlaps the scoring of
            # batch k+1; no host synchronisation in between
            bufs = [torch.zeros((cap, 8), dtype=torch.int32, device=dev) for _ in range(2)]
            pending = [None, None]
            outs = []
            for k in range(5):
                b = k & 1
                if pending[b] is not None:
                    pending[b].wait()
                bufs[b].zero_()
                sc.score_candidates_dev(n, d_goal.data_ptr(), d_fs.data_ptr(), d_bl.data_ptr(), 0, bufs[b].data_ptr())
                full, pending[b] = shard.gather_records(bufs[b], n, async_op=True)
                outs.append(full)
            for p in pending:
                p.wait()
            torch.cuda.synchronize(dev)
            got = shard.records_to_numpy(outs[-1])
            # every batch scored the same list: integers identical, float sums equal up to their summation order
            for full in outs[:-1] + [shard.gather_records(bufs[0], n)]:                    # ... and the blocking form
                r = shard.records_to_numpy(full)
                for f in ("arrival", "argmax", "n_visible", "flags"):
                    np.testing.assert_array_equal(r[f], got[f])
                np.testing.assert_allclose(r["info_ref"], got["info_ref"], rtol=5e-6, atol=1e-6)
            for f in ("arrival", "argmax", "n_visible", "flags"):
                np.testing.assert_array_equal(got[f], want[f])
            np.testing.assert_allclose(got["info_ref"], want["info_ref"], rtol=5e-6, atol=1e-6)
            sc.close()
        finally:
            torch.cuda.set_stream(prev)
    finally:
        dist.destroy_process_group()


def test_bench_rehearses_the_multi_gpu_line_on_one_gpu():
    """`bench.py --rehearse-multi`: one rank, but a real (one-rank) RCCL group and every code path the N > 1 run takes — the
    asynchronous all-gather behind every step, fs_rank_candidates_dev on the gathered list, the multi_gpu keys (per-rank time,
    barrier skew, the all-gather by itself) and the block-sampled parity gate.  Not a scaling figure; the proof that the first
    real multi-GPU run will not die in code no one-GPU box had executed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-multi", "--workload", "C2", "--steps", "3", "--warmup", "2",
                        "--repeats", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1, out_lines[:5]                          # ONE line on stdout: RCCL's banner went to stderr
    j = json.loads(out_lines[0])
    assert j["n_gpus"] == 1 and j["rehearse_multi"] is True and "nccl" in j["config"]["sharding"]
    assert j["parity"]["ok"] and "rank blocks" in j["parity"]["sample"], j["parity"]
    mg = j["multi_gpu"]
    assert len(mg["per_rank_ms_per_step"]) == 1 and mg["all_gather_ms"] > 0 and mg["barrier_skew_ms"] == 0.0
    rs = j["ranked_step"]
    assert rs["order_is_a_permutation"] and rs["costs_ascending"] and rs["range_error"] == 0 and rs["ms_per_step"] >= j["ms_per_step"] * 0.9
