// host_mirror_driver.cpp — exercises the C++ host-side mirror (frontier_scoring.hpp) on the GPU.
// tests/test_host_mirror.py writes a workload file, runs this program and compares its output with the
// oracle.  Usage: host_mirror_driver <workload.bin> <result.bin>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "frontier_scoring.hpp"

using namespace frontier_exploration;

template <typename T>
static void rd(FILE *f, T *p, size_t n)
{
    if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

static int expect_throw(const char *what, const std::function<void()> &fn)
{
    try { fn(); } catch (const std::exception &e) { printf("CHECK %s: threw (%s)\n", what, e.what()); return 0; }
    printf("CHECK %s: DID NOT THROW\n", what);
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s workload.bin result.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("workload"); return 2; }
    int32_t nx, ny, n, m;
    double res, ox, oy, start[3], poly[4];
    rd(f, &nx, 1); rd(f, &ny, 1); rd(f, &res, 1); rd(f, &ox, 1); rd(f, &oy, 1);
    auto costmap = std::make_shared<Costmap2D>((unsigned)nx, (unsigned)ny, res, ox, oy, 0);
    rd(f, costmap->getCharMap(), (size_t)nx * ny);
    rd(f, &n, 1);
    std::vector<double> goals(2 * (size_t)n);
    std::vector<int32_t> sizes(n);
    std::vector<uint8_t> black(n);
    rd(f, goals.data(), goals.size()); rd(f, sizes.data(), n); rd(f, black.data(), n);
    rd(f, &m, 1);
    std::vector<float> lm(3 * (size_t)m);
    rd(f, lm.data(), lm.size());
    rd(f, start, 3); rd(f, poly, 4);
    fclose(f);

    int failures = 0;
    auto ctx = std::make_shared<ScoringContext>(0);
    CostAssigner assigner(ctx, costmap);
    std::vector<Point> boundary(4);
    boundary[0].x = poly[0]; boundary[0].y = poly[1]; boundary[1].x = poly[0]; boundary[1].y = poly[3];
    boundary[2].x = poly[2]; boundary[2].y = poly[3]; boundary[3].x = poly[2]; boundary[3].y = poly[1];
    assigner.updateBoundaryPolygon(boundary);

    auto req = std::make_shared<GetFrontierCostsRequest>();
    auto res_ = std::make_shared<GetFrontierCostsResponse>();
    req->start_pose.pose.position.x = start[0];
    req->start_pose.pose.position.y = start[1];
    req->start_pose.pose.orientation = orientationAroundZAxis(start[2]);
    for (int32_t i = 0; i < n; ++i) {
        auto fr = std::make_shared<Frontier>();
        fr->setUID((size_t)i + 1); fr->setSize(sizes[i]); fr->setGoalPoint(goals[2 * i], goals[2 * i + 1]);
        req->frontier_list.push_back(fr);
        if (black[i]) req->prohibited_frontiers.push_back(fr);
    }
    const bool ok = assigner.getFrontierCosts(req, res_);
    printf("getFrontierCosts -> %d, %zu frontiers\n", (int)ok, res_->frontier_list.size());
    if (!ok || !res_->success) ++failures;

    // the same request through the ONE-call path (fs_get_frontier_costs: planner first, then arrival + U1 in one device call) on
    // fresh Frontier objects: every field the reference's response carries must come out bit for bit as above
    {
        auto req2 = std::make_shared<GetFrontierCostsRequest>();
        auto res2 = std::make_shared<GetFrontierCostsResponse>();
        req2->start_pose = req->start_pose;
        for (int32_t i = 0; i < n; ++i) {
            auto fr = std::make_shared<Frontier>();
            fr->setUID((size_t)i + 1); fr->setSize(sizes[i]); fr->setGoalPoint(goals[2 * i], goals[2 * i + 1]);
            req2->frontier_list.push_back(fr);
            if (black[i]) req2->prohibited_frontiers.push_back(fr);
        }
        CostAssigner fused(ctx, costmap);
        fused.updateBoundaryPolygon(boundary);
        fused.setFused(true);
        const bool ok2 = fused.getFrontierCosts(req2, res2);
        int bad = (ok2 && res2->success) ? 0 : 1;
        for (int32_t i = 0; i < n && ok2; ++i) {
            const auto &p = res_->frontier_list[i];
            const auto &q = res2->frontier_list[i];
            bad += res2->frontier_costs[i] != res_->frontier_costs[i];
            bad += res2->frontier_arrival_information[i] != res_->frontier_arrival_information[i];
            bad += res2->frontier_distances[i] != res_->frontier_distances[i];
            bad += q->isAchievable() != p->isAchievable();
            bad += q->getGoalYaw() != p->getGoalYaw();
            bad += q->getCost("arrival_gain_utility") != p->getCost("arrival_gain_utility");
            bad += q->getCost("distance_utility") != p->getCost("distance_utility");
        }
        printf("getFrontierCosts through fs_get_frontier_costs: %d mismatches against the three-call path\n", bad);
        if (bad) ++failures;
    }

    // Fisher information at (goal, best yaw), built like isPoseSafe(Point, Point)
    roadmap_explorer::FisherInformationManager fim(ctx);
    failures += expect_throw("loadLookupTable(missing file)", [&] { fim.loadLookupTable("/nonexistent/fisher_information_lookup_table.dat"); });
    fim.generateLookupTable();
    fim.setLandmarks(lm);
    std::vector<Pose> poses(n);
    for (int32_t i = 0; i < n; ++i) {
        poses[i].position = req->frontier_list[i]->getGoalPoint();
        poses[i].orientation = req->frontier_list[i]->getGoalOrientation();
    }
    std::vector<float> info;
    fim.poseInformation(poses, info);
    float single = -1;
    const bool safe = fim.isPoseSafe(poses[0], false, single);
    printf("isPoseSafe(pose0) -> %d, information %.6f (batch %.6f)\n", (int)safe, single, info[0]);
    if (single != info[0] || safe != (single > 550.0f)) ++failures;

    // key-frame pose information (computeInformationForPose): the first 20 frontier poses act as key-frames, landmark j
    // is a word point of key-frame j % 20.  setMapData pairs graph pose i with the node of the same id
    // (getNodeDataAndOptTransform, deprecated/util.hpp:791-818), so the node list is deliberately NOT in pose order:
    // nodes are stored in reverse, one more graph pose (id 999) has no node data at all (it is a key-frame without
    // points), and a second node re-uses id 100 with junk points far away — the first node of that id in list order wins.
    std::vector<float> kf_info;
    {
        slam_msgs_lite::MapData map_data;
        const int n_kf = n < 20 ? n : 20;
        for (int k = 0; k < n_kf; ++k) {
            PoseStamped ps; ps.pose = poses[k];
            map_data.graph.poses.push_back(ps);
            map_data.graph.poses_id.push_back(100 + k);
        }
        for (int k = n_kf - 1; k >= 0; --k) {
            slam_msgs_lite::KeyFrame kf; kf.id = 100 + k;
            for (int32_t j = k; j < m; j += n_kf) { Point w; w.x = lm[3 * j]; w.y = lm[3 * j + 1]; w.z = lm[3 * j + 2]; kf.word_pts.push_back(w); }
            map_data.nodes.push_back(kf);
        }
        if (n > n_kf) {
            PoseStamped ps; ps.pose = poses[n_kf];
            map_data.graph.poses.push_back(ps);
            map_data.graph.poses_id.push_back(999);
        }
        {
            slam_msgs_lite::KeyFrame junk; junk.id = 100;
            for (int j = 0; j < 50; ++j) { Point w; w.x = poses[0].position.x + 0.01 * j; w.y = poses[0].position.y; w.z = 0.5; junk.word_pts.push_back(w); }
            map_data.nodes.push_back(junk);
        }
        frontier_exploration_information_affine::setMapData(*ctx, map_data);
        kf_info = frontier_exploration_information_affine::computeInformationForPoses(*ctx, poses, 2.0, 1.089, 0.5, 0.01f, 4.5);
        printf("computeInformationForPoses: pose0 %.4f\n", kf_info[0]);
    }

    // FrontierRoadMap::isConnectable for the pairs (i, i + 7): radius_to_decide_edges 6.1 m (DEP/params/exploration.yaml),
    // max_connection_length = 1.5 x that (FrontierRoadmap.cpp:21)
    std::vector<bool> connectable;
    {
        std::vector<std::pair<FrontierPtr, FrontierPtr>> pairs;
        for (int32_t i = 0; i < n; ++i) pairs.emplace_back(req->frontier_list[i], req->frontier_list[(i + 7) % n]);
        connectable = isConnectable(*ctx, *costmap, pairs, 6.1 * 1.5, 6.1);
        size_t yes = 0;
        for (bool b : connectable) yes += b;
        printf("isConnectable: %zu of %zu pairs\n", yes, connectable.size());
        if (connectable.size() != (size_t)n) ++failures;
    }

    // FrontierSearch::searchFrom from the start pose -> std::vector<FrontierPtr>, in both seed orders; the records of the
    // reference order are written out for the comparison with the oracle (goal points and sizes, record for record)
    size_t fs_clusters = 0, fs_cells = 0, fs_pieces = 0;
    std::vector<double> frontier_rows;                 // [k][3]: goal x, goal y, size — SeedOrder::Reference
    {
        FrontierSearch search(ctx, costmap);
        Point robot; robot.x = start[0]; robot.y = start[1];
        const auto nearest = search.searchFrom(robot);
        fs_clusters = search.lastClusters().size();
        fs_cells = search.getAllFrontiers().size();
        fs_pieces = nearest.size();
        size_t cells_in_clusters = 0;
        for (const auto &cl : search.lastClusters()) cells_in_clusters += (size_t)cl.size;
        if (cells_in_clusters != fs_cells) ++failures;
        search.reset();
        search.setSeedOrder(FrontierSearch::SeedOrder::Reference);
        const auto exact = search.searchFrom(robot);
        if (exact.size() != nearest.size() || search.getAllFrontiers().size() != fs_cells) ++failures;
        // the seed rule moves goal points inside their pieces, it never changes how many records of which size there are
        std::vector<int> sa, sb;
        for (const auto &fr : nearest) sa.push_back(fr->getSize());
        for (const auto &fr : exact) sb.push_back(fr->getSize());
        std::sort(sa.begin(), sa.end()); std::sort(sb.begin(), sb.end());
        if (sa != sb) ++failures;
        for (const auto &fr : exact) {
            frontier_rows.push_back(fr->getGoalPoint().x); frontier_rows.push_back(fr->getGoalPoint().y); frontier_rows.push_back((double)fr->getSize());
            if (fr->getUID() != generateUID(fr)) ++failures;
        }
        printf("FrontierSearch: %zu clusters, %zu cells, %zu Frontier records\n", fs_clusters, fs_cells, fs_pieces);
    }

    // ShardedScorer: two scoring contexts on the one GPU of the test box, one call from this one thread — the fused records
    // of the live (not blacklisted) frontiers must equal the single-context results above
    {
        ShardedScorer sharded({0, 0});
        fs_ray_params rp{};
        rp.max_camera_depth = 2.0; rp.delta_theta = 0.10; rp.camera_fov = 1.04; rp.robot_radius = 0.60;
        rp.n_rays = 0; rp.n_elev = 1; rp.elev[0] = 0.0;
        rp.obst_min = 240; rp.obst_max = 254; rp.trace_min = 255; rp.trace_max = 255;
        rp.factor_max = 1.2; rp.factor_min = 0.70;
        const float p32[4] = {(float)poly[0], (float)poly[1], (float)poly[2], (float)poly[3]};   // the polygon travels as Point32
        for (int i = 0; i < 4; ++i) rp.polygon[i] = (double)p32[i];
        sharded.setRayParams(rp);
        sharded.updateCostmap(*costmap);
        sharded.setLandmarks(lm);
        sharded.generateLookupTable();
        sharded.setVisibility(14.0, 1.0);
        sharded.setMaxArrivalInformation();
        std::vector<FrontierPtr> copies;
        for (int32_t i = 0; i < n; ++i) {
            auto fr = std::make_shared<Frontier>();
            fr->setUID((size_t)i + 1); fr->setSize(sizes[i]); fr->setGoalPoint(goals[2 * i], goals[2 * i + 1]);
            copies.push_back(fr);
        }
        const auto rec = sharded.scoreFrontiers(copies, black);
        int bad = 0;
        for (int32_t i = 0; i < n; ++i) {
            if (black[i]) { bad += (FS_RECORD_STATUS(rec[i].flags) != FS_STATUS_BLACKLISTED); continue; }
            bad += (double)rec[i].arrival != res_->frontier_arrival_information[i];
            bad += (double)rec[i].yaw != (double)(float)res_->frontier_list[i]->getGoalYaw();
            bad += std::fabs(rec[i].info_ref - info[i]) > 2e-6f * std::max(1.0f, std::fabs(info[i]));
        }
        printf("ShardedScorer({0,0}): %d devices, %d mismatches against the single context\n", sharded.numDevices(), bad);
        if (bad || sharded.numDevices() != 2) ++failures;

        // the whole cost assignment as ONE call over both members (fs_multi_get_frontier_costs: blocks gathered device to device,
        // ranked on the first): path columns as the planner of the run above left them.  Every response field bit for bit as from
        // the single context — through each way a block can travel (written in place / device copy + event / page-locked bounce).
        std::vector<double> plen(n, 0.0), phead(n, 0.0);
        std::vector<uint8_t> ach_in(n, 1);
        for (int32_t i = 0; i < n; ++i) {
            const auto &p = res_->frontier_list[i];
            if (black[i] || !p->isAchievable()) continue;
            plen[i] = p->getPathLength(); phead[i] = p->getPathHeading();
        }
        for (const int mode : {0, 3, 2}) {
            sharded.setOption("multi.gather", (double)mode);
            std::vector<FrontierPtr> fresh;
            for (int32_t i = 0; i < n; ++i) {
                auto fr = std::make_shared<Frontier>();
                fr->setUID((size_t)i + 1); fr->setSize(sizes[i]); fr->setGoalPoint(goals[2 * i], goals[2 * i + 1]);
                fresh.push_back(fr);
            }
            sharded.assignCosts(fresh, black, ach_in, plen, phead, 0.10, 1.04);
            int bad2 = sharded.gatherMode() == (mode ? mode : 1) ? 0 : 1;
            for (int32_t i = 0; i < n; ++i) {
                const auto &p = res_->frontier_list[i];
                const auto &q = fresh[i];
                bad2 += q->getWeightedCost() != res_->frontier_costs[i];
                if (black[i]) continue;
                bad2 += q->getArrivalInformation() != res_->frontier_arrival_information[i];
                bad2 += q->isAchievable() != p->isAchievable();
                bad2 += q->getGoalYaw() != p->getGoalYaw();
                bad2 += q->getCost("arrival_gain_utility") != p->getCost("arrival_gain_utility");
                bad2 += q->getCost("distance_utility") != p->getCost("distance_utility");
            }
            printf("ShardedScorer::assignCosts (multi.gather %d): %d mismatches against the single context\n", mode, bad2);
            if (bad2) ++failures;
        }
        std::vector<float> info2;
        sharded.poseInformation(poses, info2);
        int bad3 = 0;
        for (int32_t i = 0; i < n; ++i) bad3 += std::fabs(info2[i] - info[i]) > 2e-6f * std::max(1.0f, std::fabs(info[i]));
        printf("ShardedScorer::poseInformation: %d mismatches against the single context\n", bad3);
        if (bad3) ++failures;
    }

    // A costmap update cycle between two ticks: a layer rewrites a window of the master grid (Layer::updateCosts' bounds) and only that
    // window is sent (FrontierCostCalculator::updateCostmapWindow -> fs_update_grid_region; ShardedScorer likewise, on every device).
    // Arrival information, yaw and achievability must equal those of a calculator that snapshots the whole rewritten map.
    {
        std::vector<double> poly_v(poly, poly + 4);
        auto fresh_list = [&]() {
            std::vector<FrontierPtr> v;
            for (int32_t i = 0; i < n; ++i) {
                auto fr = std::make_shared<Frontier>();
                fr->setUID((size_t)i + 1); fr->setSize(sizes[i]); fr->setGoalPoint(goals[2 * i], goals[2 * i + 1]); fr->setAchievability(true);
                v.push_back(fr);
            }
            return v;
        };
        const std::vector<uint8_t> saved(costmap->getCharMap(), costmap->getCharMap() + (size_t)nx * ny);
        FrontierCostCalculator windowed(std::make_shared<ScoringContext>(0), costmap);       // snapshot of the map as it was
        windowed.setMaxArrivalInformation();
        ShardedScorer sharded({0, 0});
        fs_ray_params rp{};
        rp.max_camera_depth = 2.0; rp.delta_theta = 0.10; rp.camera_fov = 1.04; rp.robot_radius = 0.60;
        rp.n_elev = 1; rp.obst_min = 240; rp.obst_max = 254; rp.trace_min = 255; rp.trace_max = 255; rp.factor_max = 1.2; rp.factor_min = 0.70;
        for (int k = 0; k < 4; ++k) rp.polygon[k] = poly[k];
        sharded.setRayParams(rp);
        sharded.updateCostmap(*costmap);
        sharded.setMaxArrivalInformation();
        int bad = 0;
        const int wins[3][4] = {{nx / 3, ny / 4, nx / 3 + 37, ny / 4 + 21}, {0, ny - 13, 29, ny}, {nx - 9, 3, nx, 50 < ny ? 50 : ny}};
        for (const auto &b : wins) {
            for (int j = b[1]; j < b[3]; ++j)
                for (int i = b[0]; i < b[2]; ++i)
                    costmap->getCharMap()[(size_t)j * nx + i] = (uint8_t)(((i * 7 + j * 13) % 5 == 0) ? 254 : ((i + j) % 3 == 0 ? 0 : 255));
            windowed.updateCostmapWindow(b[0], b[1], b[2], b[3]);
            sharded.updateCostmapWindow(*costmap, b[0], b[1], b[2], b[3]);
            FrontierCostCalculator snap(std::make_shared<ScoringContext>(0), costmap);       // snapshot of the rewritten map
            snap.setMaxArrivalInformation();
            auto a = fresh_list(), c = fresh_list();
            windowed.setArrivalInformationForFrontiers(a, poly_v);
            snap.setArrivalInformationForFrontiers(c, poly_v);
            std::vector<double> g3(3 * (size_t)n, 0.0);
            for (int32_t i = 0; i < n; ++i) { g3[3 * i] = goals[2 * i]; g3[3 * i + 1] = goals[2 * i + 1]; }
            std::vector<int32_t> arr(n), amax(n), st(n); std::vector<double> yw(n); std::vector<uint8_t> ac(n);
            sharded.check(fs_multi_score_arrival(sharded.get(), n, g3.data(), sizes.data(), nullptr, nullptr, nullptr, arr.data(), amax.data(), yw.data(), ac.data(), st.data()),
                          "fs_multi_score_arrival");
            for (int32_t i = 0; i < n; ++i) {
                bad += a[i]->getArrivalInformation() != c[i]->getArrivalInformation();
                bad += a[i]->getGoalYaw() != c[i]->getGoalYaw();
                bad += a[i]->isAchievable() != c[i]->isAchievable();
                bad += (double)arr[i] != c[i]->getArrivalInformation();
                bad += (ac[i] != 0) != c[i]->isAchievable();
            }
        }
        printf("updateCostmapWindow (3 windows, one context and ShardedScorer({0,0})): %d mismatches against whole-map snapshots\n", bad);
        if (bad) ++failures;
        std::memcpy(costmap->getCharMap(), saved.data(), saved.size());
    }

    // error behaviour of the reference interface
    failures += expect_throw("Frontier getter on unset field", [] { Frontier fr; (void)fr.getArrivalInformation(); });
    failures += expect_throw("duplicate frontiers", [&] {
        auto r2 = std::make_shared<GetFrontierCostsRequest>(*req);
        r2->frontier_list.push_back(r2->frontier_list[0]);
        auto o2 = std::make_shared<GetFrontierCostsResponse>();
        assigner.getFrontierCosts(r2, o2);
    });
    {
        auto r3 = std::make_shared<GetFrontierCostsRequest>();
        auto o3 = std::make_shared<GetFrontierCostsResponse>();
        const bool empty_ok = assigner.getFrontierCosts(r3, o3);
        printf("CHECK empty frontier list: returned %d\n", (int)empty_ok);
        if (empty_ok || o3->success) ++failures;
    }

    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror("result"); return 2; }
    for (int32_t i = 0; i < n; ++i) {
        const auto &fr = res_->frontier_list[i];
        const double row[10] = {res_->frontier_arrival_information[i], fr->getGoalYaw(), (double)fr->isAchievable(),
                                res_->frontier_costs[i], fr->getCost("arrival_gain_utility"), fr->getCost("distance_utility"),
                                res_->frontier_distances[i], (double)info[i], (double)kf_info[i], (double)connectable[i]};
        fwrite(row, sizeof(double), 10, o);
    }
    const double tail[4] = {(double)fs_clusters, (double)fs_cells, (double)fs_pieces, (double)(frontier_rows.size() / 3)};
    fwrite(tail, sizeof(double), 4, o);
    fwrite(frontier_rows.data(), sizeof(double), frontier_rows.size(), o);
    fclose(o);
    printf("failures: %d\n", failures);
    return failures ? 1 : 0;
}
