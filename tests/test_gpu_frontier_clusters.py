"""fs_frontier_clusters (frontier detection + clustering on the GPU) against the oracle's restatement of
FrontierSearch::searchFrom / buildNewFrontier (DEP/src/FrontierSearch.cpp:21-216): cluster membership (labels), the set
of frontier cells found (every_frontier_list), cluster sizes / centroids / bounding boxes, and the reference's piece
sizes as they follow from the component sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_labels(r):
    seed = r["cell_seed"]
    ny, nx = seed.shape
    idx = np.arange(ny * nx).reshape(ny, nx)
    out = np.full((ny, nx), -1, dtype=np.int32)
    for s in np.unique(seed[seed >= 0]):
        m = seed == s
        out[m] = idx[m].min()
    return out


def _check(fs, oracle, scorer, cells, origin, res, pos, lethal=160, max_cluster=20, max_d=50.0, min_cluster=1):
    ny, nx = cells.shape
    r = oracle.frontier_search(cells, origin, res, pos, lethal_threshold=lethal, min_cluster=min_cluster, max_cluster=max_cluster, max_distance=max_d)
    labels, cl, n_cl, n_cells = scorer.frontier_clusters((ny, nx), pos, lethal_threshold=lethal, max_frontier_distance=max_d,
                                                         max_frontier_cluster_size=max_cluster)
    want = _oracle_labels(r)
    np.testing.assert_array_equal(labels, want)                                  # membership, cell for cell
    assert n_cells == r["n_every"] == int((want >= 0).sum())
    ids, counts = np.unique(want[want >= 0], return_counts=True)
    assert n_cl == len(ids) == cl.shape[0]
    np.testing.assert_array_equal(cl["label"], ids)                              # ascending labels
    np.testing.assert_array_equal(cl["size"], counts)
    yy, xx = np.mgrid[0:ny, 0:nx]
    for c in cl:
        m = want == c["label"]
        assert (c["min_x"], c["max_x"], c["min_y"], c["max_y"]) == (xx[m].min(), xx[m].max(), yy[m].min(), yy[m].max())
        assert abs(c["centroid_x"] - (origin[0] + (xx[m].mean() + 0.5) * res)) <= 1e-12 * max(1.0, abs(c["centroid_x"])) + 1e-12
        assert abs(c["centroid_y"] - (origin[1] + (yy[m].mean() + 0.5) * res)) <= 1e-12 * max(1.0, abs(c["centroid_y"])) + 1e-12
    # the reference's Frontier records: pieces of max + 1 cells in queue order, then a remainder if it exceeds min
    step = max_cluster + 1
    expect = sorted(sum(([step] * (int(n) // step) + ([int(n) % step] if int(n) % step > min_cluster else []) for n in cl["size"]), []))
    assert sorted(r["sizes"].tolist()) == expect
    return r, cl


def test_clusters_match_the_reference_search_on_synthetic_costmaps(fs, oracle, scorer):
    rng = np.random.default_rng(11)
    for seed, n in ((3, 160), (4, 200), (9, 512)):
        w = fs.synth.make_small_2d(seed, n=n, n_cand=40)
        cells = w.cells[0]
        scorer.upload_grid(w.cells, w.origin, w.resolution)
        free = np.argwhere(cells == 0)
        total = 0
        for k in rng.choice(len(free), size=5, replace=False):
            y, x = free[k]
            pos = (w.origin[0] + (x + 0.3) * w.resolution, w.origin[1] + (y + 0.6) * w.resolution)
            for max_d, max_cluster, lethal in ((50.0, 20, 160), (1.5, 20, 160), (50.0, 5, 250), (3.0, 40, 1)):
                r, cl = _check(fs, oracle, scorer, cells, w.origin, w.resolution, pos, lethal=lethal, max_cluster=max_cluster, max_d=max_d)
                total += cl.shape[0]
        assert total > 20


def test_robot_on_obstacle_off_map_and_hand_built_maps(fs, oracle, scorer):
    m = np.full((40, 60), 255, np.uint8)
    m[5:15, 5:25] = 0; m[5:15, 35:55] = 0; m[25:35, 5:25] = 0; m[15:25, 10:12] = 0
    m[14:26, 9] = 254; m[14:26, 12] = 254; m[4, 5:25] = 254
    scorer.upload_grid(m[None], (0.0, 0.0, 0.0), 0.05)
    r, cl = _check(fs, oracle, scorer, m, (0.0, 0.0), 0.05, (0.5, 0.5))
    assert cl.shape[0] >= 2
    _check(fs, oracle, scorer, m, (0.0, 0.0), 0.05, (0.5, 0.5), max_d=0.2, max_cluster=2)
    _check(fs, oracle, scorer, m, (0.0, 0.0), 0.05, (0.5, 0.5), min_cluster=8)
    # robot inside a lethal blob: the search starts from nearestFreeCell's cell (DEP/src/Helpers.cpp:285-329)
    b = np.full((30, 30), 255, np.uint8); b[5:25, 5:25] = 0; b[14:17, 14:17] = 254
    scorer.upload_grid(b[None], (0.0, 0.0, 0.0), 0.05)
    _check(fs, oracle, scorer, b, (0.0, 0.0), 0.05, (15.5 * 0.05, 15.5 * 0.05))
    # robot in unknown space far from anything free: nearestFreeCell walks a long queue first
    u = np.full((64, 64), 255, np.uint8); u[40:50, 30:45] = 0; u[45, 44] = 200
    scorer.upload_grid(u[None], (-1.0, -2.0, 0.0), 0.05)
    _check(fs, oracle, scorer, u, (-1.0, -2.0), 0.05, (-1.0 + 3.2 * 0.05, -2.0 + 2.7 * 0.05))
    # no free cell at all: the robot cell itself is the start (FrontierSearch.cpp:51-54)
    z = np.full((16, 16), 255, np.uint8)
    scorer.upload_grid(z[None], (0.0, 0.0, 0.0), 0.05)
    _check(fs, oracle, scorer, z, (0.0, 0.0), 0.05, (0.4, 0.4))
    # off the map: nothing (FrontierSearch.cpp:28-33)
    labels, cl, n_cl, n_cells = scorer.frontier_clusters((16, 16), (-0.5, 0.2))
    assert n_cl == 0 and n_cells == 0 and cl.shape[0] == 0 and (labels == -1).all()
    # more clusters than the caller's array holds: the count is still reported
    w = fs.synth.make_small_2d(9, n=512, n_cand=4)
    scorer.upload_grid(w.cells, w.origin, w.resolution)
    y, x = np.argwhere(w.cells[0] == 0)[100]
    pos = (w.origin[0] + (x + 0.5) * w.resolution, w.origin[1] + (y + 0.5) * w.resolution)
    _, cl_all, n_all, _ = scorer.frontier_clusters((512, 512), pos)
    _, cl_few, n_few, _ = scorer.frontier_clusters((512, 512), pos, max_clusters=3, want_labels=False)
    assert n_few == n_all > 3 and cl_few.shape[0] == 3
    with pytest.raises(fs.FsError):                                # the reference's search is 2-D
        c3 = fs.synth.make_workload("C1", n_cand=4)
        scorer.upload_grid(c3.cells, c3.origin, c3.resolution)
        scorer.frontier_clusters((64, 64), (0.0, 0.0))


import os

N_RANDOM = int(os.environ.get("FS_CLUSTER_SEEDS", "40"))      # by hand: thousands


@pytest.mark.parametrize("seed", range(N_RANDOM))
def test_random_costmaps(fs, oracle, scorer, seed):
    """Random byte maps (salt-and-pepper, so that components are many, small and oddly shaped; every cost value occurs), sizes from
    1 x 1 up, random robot cells (free, lethal, unknown — the reference searches for the nearest free one), thresholds, distance
    limits and cluster sizes: membership cell for cell, the cluster records and the reference's piece sizes."""
    rng = np.random.default_rng(40_000 + seed + 1000 * int(os.environ.get("FS_CLUSTER_BASE", "0")))
    nx, ny = (int(rng.integers(1, 10)), int(rng.integers(1, 10))) if rng.random() < 0.15 else (int(rng.integers(10, 140)), int(rng.integers(10, 140)))
    res = float(rng.choice([0.05, 0.1, 0.25]))
    origin = (float(rng.uniform(-5, 1)), float(rng.uniform(-5, 1)), 0.0)
    p_unknown = float(rng.choice([0.2, 0.5, 0.8]))
    vals = rng.choice(np.array([0, 0, 0, 1, 50, 159, 160, 200, 253, 254], np.uint8), size=(ny, nx))
    cells = np.where(rng.random((ny, nx)) < p_unknown, np.uint8(255), vals).astype(np.uint8)
    if rng.random() < 0.5:                                                  # blobs instead of salt and pepper
        k = int(rng.integers(2, 9))
        coarse = rng.random(((ny + k - 1) // k, (nx + k - 1) // k)) < p_unknown
        cells = np.where(np.kron(coarse, np.ones((k, k), bool))[:ny, :nx], np.uint8(255), vals).astype(np.uint8)
    scorer.upload_grid(cells[None], origin, res)
    for _ in range(3):
        x, y = int(rng.integers(0, nx)), int(rng.integers(0, ny))
        pos = (origin[0] + (x + float(rng.uniform(0.05, 0.95))) * res, origin[1] + (y + float(rng.uniform(0.05, 0.95))) * res)
        _check(fs, oracle, scorer, cells, origin, res, pos, lethal=int(rng.choice([1, 160, 250, 254])), max_cluster=int(rng.choice([1, 5, 20, 40])),
               max_d=float(rng.choice([0.3, 1.5, 50.0])))
