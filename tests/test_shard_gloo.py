"""The N > 1 path on CPU: two gloo ranks shard a candidate list, each produces its block of 32-byte
records (the oracle stands in for the GPU here — this test covers partitioning, the single
all-gather and order restoration, not the kernels) and every rank ends with the full, ordered list."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records_from_oracle(fs, O, w, lo, hi, table):
    G = O.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = O.RayParams(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    arr = O.arrival_information(G, P, w.goals[lo:hi], w.frontier_size[lo:hi], w.blacklisted[lo:hi])
    fim = O.pose_information(table, w.landmarks, O.poses_from_yaw(w.goals[lo:hi], arr["yaw"]), 14.0, 1.0)
    rec = np.zeros(hi - lo, dtype=fs.RECORD_DTYPE)
    ok = arr["status"] == 0
    rec["arrival"], rec["argmax"], rec["yaw"] = arr["arrival"], arr["argmax"], arr["yaw"]
    rec["info_ref"] = np.where(ok, fim["info_ref"], 0)
    rec["trace"] = np.where(ok, fim["trace"], 0)
    rec["logdet"] = np.where(ok, fim["logdet"], -np.inf)
    rec["n_visible"] = np.where(ok, fim["n_visible"], 0)
    rec["flags"] = arr["achievable"].astype(np.uint32) | (arr["status"].astype(np.uint32) << 8) | \
        (np.minimum(np.where(ok, fim["n_voxels"], 0), 65535).astype(np.uint32) << 16)
    return rec


def _worker(rank, world, port, n_total, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    fs = importlib.import_module("fit-slam_amd")
    shard = importlib.import_module("fit-slam_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = fs.synth.make_workload("C1", n_cand=n_total)
    lo, hi = shard.shard_bounds(n_total, world, rank)
    cap = shard.shard_capacity(n_total, world)
    rec = _records_from_oracle(fs, O, w, lo, hi, O.Table.generate((0.0, 21.0, -14.7, 14.7, -14.7, 14.7)))
    local = torch.zeros((cap, 8), dtype=torch.int32)
    local[: hi - lo] = torch.from_numpy(rec.view(np.int32).reshape(-1, 8).copy())
    full = shard.gather_records(local, n_total)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), shard.records_to_numpy(full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [101, 64])
def test_two_rank_gloo_gather(fs, oracle, tmp_path, n_total):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(a, b)
    w = fs.synth.make_workload("C1", n_cand=n_total)
    want = _records_from_oracle(fs, oracle, w, 0, n_total, oracle.Table.generate((0.0, 21.0, -14.7, 14.7, -14.7, 14.7)))
    np.testing.assert_array_equal(a.view(np.int32), want.view(np.int32))      # order restored, bytes identical


def test_shard_bounds_cover_and_order():
    shard = importlib.import_module("fit-slam_amd.shard")
    for n in (0, 1, 7, 8, 9, 160_000, 20_001):
        for world in (1, 2, 3, 8):
            cap = shard.shard_capacity(n, world)
            spans = [shard.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for r, (lo, hi) in enumerate(spans):
                assert 0 <= hi - lo <= cap and (lo == min(n, r * cap))
                if r:
                    assert lo == spans[r - 1][1]


def test_weak_scaling_blocks_are_functions_of_config_and_rank(fs):
    """bench.py's weak scaling: every rank builds the same map and cloud and ONLY its own block of the n x world list
    (synth.candidate_block).  Block 0 is the N = 1 list itself; block r is the same on whichever rank computes it (rank 0 rebuilds
    the sampled blocks for the parity gate) and differs from the others; all of them are frontier cells of the same grid."""
    w = fs.synth.make_workload("C1")
    g0, f0, b0 = fs.synth.candidate_block(w, "C1", 0)
    assert g0 is w.goals and f0 is w.frontier_size and b0 is w.blacklisted
    g1a, f1a, b1a = fs.synth.candidate_block(w, "C1", 1)
    g1b, f1b, b1b = fs.synth.candidate_block(fs.synth.make_workload("C1"), "C1", 1)
    np.testing.assert_array_equal(g1a, g1b); np.testing.assert_array_equal(f1a, f1b); np.testing.assert_array_equal(b1a, b1b)
    g2, _, _ = fs.synth.candidate_block(w, "C1", 2)
    assert g1a.shape == w.goals.shape == g2.shape and not np.array_equal(g1a, g2) and not np.array_equal(g1a, w.goals)
    # frontier cells: free cells (cost 0) of the grid, at cell centres
    cell = np.floor((g1a - np.asarray(w.origin)) / w.resolution).astype(int)
    assert np.all(w.cells[cell[:, 2], cell[:, 1], cell[:, 0]] == 0)
