// adapter_driver.cpp — runs the ROS 2 adapter (fit-slam_amd/host/ros2/src/*.cpp, compiled unchanged) against the test doubles of
// ros2_fakes.hpp on the GPU box; tests/test_gpu_ros2_adapter_run.py builds it, feeds it the workload file of the host-mirror
// test (same layout) and compares what it writes with the ctypes route and the oracle.
//
//   adapter_driver workload.bin lookup_table.dat result.bin
//
// result.bin (float64): for each of the two routes (three-step, fused) n x 8 columns
//   [arrival information, goal orientation, achievable, weighted cost, arrival utility, distance utility, path length (m),
//    response.frontier_costs]; then n_poses x 2 [isPoseSafe verdict, information]; then the tail (see the end of main).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include <pluginlib/class_list_macros.hpp>          // (answered by ros2_fakes.hpp, like every external include)

#include "fitslam_frontier_ros2/CostAssignerGPU.hpp"
#include "fitslam_frontier_ros2/FisherInfoBTPluginGPU.hpp"
#include "fitslam_frontier_ros2/FisherInfoManagerGPU.hpp"

template <typename T>
static void rd(FILE *f, T *p, size_t n)
{
    if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

static int failures = 0;
#define EXPECT(cond, what) do { if (!(cond)) { printf("CHECK FAILED: %s\n", what); ++failures; } else printf("check ok: %s\n", what); } while (0)

template <typename F>
static bool throws(F fn, const char *what)
{
    try { fn(); } catch (const std::exception &e) { printf("check ok: %s threw (%s)\n", what, e.what()); return true; }
    printf("CHECK FAILED: %s DID NOT THROW\n", what);
    ++failures;
    return false;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s workload.bin lookup_table.dat result.bin [regenerate]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t nx, ny, n, m;
    double res, ox, oy, start[3], poly[4];
    rd(f, &nx, 1); rd(f, &ny, 1); rd(f, &res, 1); rd(f, &ox, 1); rd(f, &oy, 1);
    std::vector<unsigned char> cells((size_t)nx * ny);
    rd(f, cells.data(), cells.size());
    rd(f, &n, 1);
    std::vector<double> goals(2 * (size_t)n);
    std::vector<int32_t> fsize(n);
    std::vector<uint8_t> black(n);
    rd(f, goals.data(), goals.size()); rd(f, fsize.data(), n); rd(f, black.data(), n);
    rd(f, &m, 1);
    std::vector<float> lm(3 * (size_t)m);
    rd(f, lm.data(), lm.size());
    rd(f, start, 3); rd(f, poly, 4);
    fclose(f);

    // ---- the world the adapter sees
    auto &prm = fakes::globals().parameters;                      // DEP/params/exploration.yaml
    prm["costCalculator/max_camera_depth"] = 2.0; prm["costCalculator/delta_theta"] = 0.10; prm["costCalculator/camera_fov"] = 1.04;
    prm["frontierCostsManager/alpha"] = 0.25; prm["frontierCostsManager/beta"] = 1.0; prm["frontierCostsManager/planner_allow_unknown"] = 1.0;
    prm["frontierCostsManager/vx_max"] = 0.5; prm["frontierCostsManager/wz_max"] = 0.5;
    auto costmap_ros = std::make_shared<nav2_costmap_2d::Costmap2DROS>();
    {
        fakes::CostmapRosState &r = fakes::state_of<fakes::CostmapRosState>(costmap_ros.get());
        r.robot_radius = 0.60;
        fakes::CostmapState &c = fakes::state_of<fakes::CostmapState>(&r.costmap);
        c.cells = cells; c.nx = (unsigned)nx; c.ny = (unsigned)ny; c.ox = ox; c.oy = oy; c.res = res;
    }
    std::shared_ptr<nav2_util::LifecycleNode> node = costmap_ros;
    node->declare_parameter("fitslam_frontier.gpu_devices", std::vector<int64_t>{0, 0});    // two contexts of the one GPU: the multi-device forms run
    node->declare_parameter("fisherInformation.lookup_table_file", std::string(argv[2]));

    auto make_list = [&] {
        std::vector<FrontierPtr> list;
        for (int32_t i = 0; i < n; ++i) list.push_back(fakes::make_frontier(goals[2 * i], goals[2 * i + 1], fsize[i]));
        return list;
    };
    geometry_msgs::msg::PoseStamped start_pose;
    start_pose.pose.position = {start[0], start[1], 0.0};
    start_pose.pose.orientation = nav2_util::geometry_utils::orientationAroundZAxis(start[2]);

    std::vector<double> out;
    std::vector<double> yaw_of(n, 0.0);
    // ================= CostAssignerGPU: both routes
    for (int fused = 0; fused < 2; ++fused) {
        fitslam_frontier_ros2::CostAssignerGPU assigner(costmap_ros);        // device list from the node parameter
        assigner.setPlannerMethod("EuclideanDistance");
        assigner.setFused(fused != 0);
        auto req = std::make_shared<frontier_exploration::GetFrontierCostsRequest>();
        auto resp = std::make_shared<frontier_exploration::GetFrontierCostsResponse>();
        req->start_pose = start_pose;
        req->frontier_list = make_list();
        // no polygon yet: the reference refuses (FrontierCostsManager.cpp:61-65) — success false, nothing thrown
        EXPECT(!assigner.getFrontierCosts(req, resp) && !resp->success, "no boundary polygon yet -> getFrontierCosts returns false");
        geometry_msgs::msg::PolygonStamped boundary;
        for (const auto &xy : {std::pair<double, double>{poly[0], poly[1]}, {poly[0], poly[3]}, {poly[2], poly[3]}, {poly[2], poly[1]}}) {
            geometry_msgs::msg::Point32 p; p.x = (float)xy.first; p.y = (float)xy.second; p.z = 0.0f;
            boundary.polygon.points.push_back(p);
        }
        EXPECT(assigner.updateBoundaryPolygon(boundary), "updateBoundaryPolygon");
        for (int32_t i = 0; i < n; ++i) if (black[i]) req->prohibited_frontiers.push_back(req->frontier_list[i]);
        const int resets0 = fakes::globals().planner_resets, plans0 = fakes::globals().planner_calls;
        const bool ok = assigner.getFrontierCosts(req, resp);
        EXPECT(ok && resp->success, fused ? "getFrontierCosts (fused route)" : "getFrontierCosts (three-step route)");
        EXPECT(resp->frontier_list.size() == (size_t)n && resp->frontier_list == req->frontier_list, "response carries the request's pointers in order");
        EXPECT(fakes::globals().planner_resets == resets0 + 1, "planner reset once per tick");
        EXPECT(fakes::globals().planner_calls > plans0, "the reference's planner was called");
        for (int32_t i = 0; i < n; ++i) {
            const fakes::FrontierState &s = fakes::frontier(req->frontier_list[i].get());
            const auto au = s.costs.find("arrival_gain_utility"), du = s.costs.find("distance_utility");
            out.insert(out.end(), {s.arrival, s.goal_orientation, s.achievable ? 1.0 : 0.0, s.weighted_cost,
                                   au == s.costs.end() ? -1000.0 : au->second, du == s.costs.end() ? -1000.0 : du->second, s.path_length_m,
                                   resp->frontier_costs[i]});
            if (!fused) yaw_of[i] = s.goal_orientation;
            if (resp->frontier_arrival_information[i] != s.arrival || resp->frontier_distances[i] != s.path_length_m) { printf("CHECK FAILED: response columns of frontier %d\n", i); ++failures; }
        }
        // an empty list: false, nothing thrown (:55-59); the same frontier twice: throws (:25-45,69-72)
        auto empty_req = std::make_shared<frontier_exploration::GetFrontierCostsRequest>();
        auto empty_resp = std::make_shared<frontier_exploration::GetFrontierCostsResponse>();
        empty_req->start_pose = start_pose;
        EXPECT(!assigner.getFrontierCosts(empty_req, empty_resp), "empty frontier list -> false");
        auto dup_req = std::make_shared<frontier_exploration::GetFrontierCostsRequest>();
        dup_req->start_pose = start_pose;
        dup_req->frontier_list = make_list();
        dup_req->frontier_list.push_back(dup_req->frontier_list[0]);
        throws([&] { assigner.getFrontierCosts(dup_req, empty_resp); }, "duplicate frontier");
    }
    {
        fakes::CostmapRosState &r = fakes::state_of<fakes::CostmapRosState>(costmap_ros.get());
        EXPECT(fakes::state_of<fakes::CostmapState>(&r.costmap).locked_reads >= 2, "the costmap was snapshotted once per tick");
    }

    // ================= FisherInformationManagerGPU + the BT plugin
    const int n_poses = n < 12 ? n : 12;
    {
        fakes::globals().parameters["fisherInformation.fisher_information_threshold"] = 550.0;
        fitslam_frontier_ros2::FisherInformationManagerGPU fim(node);        // table file and devices from the node parameters
        geometry_msgs::msg::Pose pose0;
        pose0.position = {goals[0], goals[1], 0.0};
        pose0.orientation = nav2_util::geometry_utils::orientationAroundZAxis(yaw_of[0]);
        float info = -1.0f;
        EXPECT(!fim.isPoseSafe(pose0, false, info), "isPoseSafe before any map_data -> false");
        if (argc > 4 && std::string(argv[4]) == "regenerate") {
            // generateLookupTable with gen_fi_lookup's bounds (DEP/src/fisher_information/GenerateLookupMain.cpp:9): rewrites the file the
            // node parameter names — the test compares its bytes with the oracle's records — and reads it back
            const float h = (float)(8.5 * 1.732);                 // (double literals narrowed at the call, as GenerateLookupMain.cpp:9 has them)
            fim.generateLookupTable(0.0f, 21.0f, -h, h, -h, h);
            fim.loadLookupTable();
            printf("check ok: generateLookupTable + loadLookupTable through the manager\n");
        }
        // the SLAM front end publishes the map: every landmark twice (two key-frames see it), a NaN point, over two nodes
        auto msg = std::make_shared<slam_msgs::msg::MapData>();
        msg->nodes.resize(2);
        for (int32_t j = 0; j < m; ++j) {
            geometry_msgs::msg::Point p{lm[3 * j], lm[3 * j + 1], lm[3 * j + 2]};
            msg->nodes[j % 2].word_pts.push_back(p);
            if (j % 3 == 0) msg->nodes[(j + 1) % 2].word_pts.push_back(p);
        }
        msg->nodes[0].word_pts.push_back(geometry_msgs::msg::Point{std::nan(""), 0.0, 0.0});
        fakes::publish<slam_msgs::msg::MapData>("map_data", msg);
        std::vector<geometry_msgs::msg::Pose> poses(n_poses);
        for (int i = 0; i < n_poses; ++i) {
            poses[i].position = {goals[2 * i], goals[2 * i + 1], 0.0};
            poses[i].orientation = nav2_util::geometry_utils::orientationAroundZAxis(yaw_of[i]);
        }
        std::vector<float> batch;
        EXPECT(fim.poseInformation(poses, batch) && batch.size() == (size_t)n_poses, "poseInformation (batch over the device list)");
        for (int i = 0; i < n_poses; ++i) {
            float one = -1.0f;
            const bool safe = fim.isPoseSafe(poses[i], true, one);
            out.push_back(safe ? 1.0 : 0.0); out.push_back((double)one);
            const float d = std::fabs(one - batch[i]), sc = std::fabs(batch[i]) > 1e-6f ? std::fabs(batch[i]) : 1e-6f;
            if (d / sc > 1e-5f) { printf("CHECK FAILED: single pose %d %.6f against batch %.6f\n", i, one, batch[i]); ++failures; }
            if (safe != (one > 550.0f)) { printf("CHECK FAILED: verdict of pose %d\n", i); ++failures; }
        }
        // the two-point overload: pose at `from`, looking at `to`
        geometry_msgs::msg::Point from{goals[0], goals[1], 0.0}, to{goals[0] + std::cos(yaw_of[0]), goals[1] + std::sin(yaw_of[0]), 0.0};
        float direct = 0.0f;
        const bool v1 = fim.isPoseSafe(from, to, true), v2 = fim.isPoseSafe(pose0, true, direct);
        EXPECT(v1 == v2, "isPoseSafe(point_from, point_to) == isPoseSafe(pose)");
    }
    double tick_info_status[2] = {0, 0};
    {
        // the plugin, loaded INSTEAD of the reference plugin, then NEXT TO it: one builder per ID either way
        for (int next_to = 0; next_to < 2; ++next_to) {
            BT::BehaviorTreeFactory factory;
            fakes::forget<fakes::FactoryState>(&factory);                      // (same stack slot as the previous round's factory)
            auto tf = std::make_shared<tf2_ros::Buffer>();
            if (next_to) { roadmap_explorer::FisherInfoBTPlugin reference; reference.registerNodes(factory, node, costmap_ros, tf); }
            roadmap_explorer::FisherInfoBTPluginGPU plugin;
            plugin.registerNodes(factory, node, costmap_ros, tf);
            EXPECT(factory.builders().size() == 2 && factory.builders().count("EvaluateFisherInformation") && factory.builders().count("MarkLethalFOV"),
                   next_to ? "plugin next to the reference plugin: both IDs, once" : "plugin instead of the reference plugin: both IDs, once");
            BT::NodeConfiguration config;
            config.blackboard = std::make_shared<BT::Blackboard>();
            std::unique_ptr<BT::TreeNode> tree_node = factory.builders().at("EvaluateFisherInformation")("EvaluateFisherInformation", config);
            auto *action = dynamic_cast<BT::SyncActionNode *>(tree_node.get());
            EXPECT(action != nullptr, "the builder makes a SyncActionNode");
            if (!action) continue;
            throws([&] { action->tick(); }, "tick without latest_robot_pose on the blackboard");
            config.blackboard->set<geometry_msgs::msg::PoseStamped>("latest_robot_pose", start_pose);
            auto msg = std::make_shared<slam_msgs::msg::MapData>();
            msg->nodes.resize(1);
            for (int32_t j = 0; j < m; ++j) msg->nodes[0].word_pts.push_back(geometry_msgs::msg::Point{lm[3 * j], lm[3 * j + 1], lm[3 * j + 2]});
            fakes::publish<slam_msgs::msg::MapData>("map_data", msg);          // (the node's manager subscribed when it was built)
            // camera pose from TF: pose 0 -> whatever the verdict of pose 0 was above; then a pose far from every landmark -> FAILURE + 700 ms
            auto &t = fakes::globals().tf_map_base.transform;
            t.translation = {goals[0], goals[1], 0.0};
            t.rotation = nav2_util::geometry_utils::orientationAroundZAxis(yaw_of[0]);
            fakes::globals().input_ports["exhaustive_landmark_search"] = "true";
            const BT::NodeStatus near_status = action->tick();
            t.translation = {1.0e4, 1.0e4, 0.0};
            const double slept0 = fakes::globals().slept_ms;
            const BT::NodeStatus far_status = action->tick();
            EXPECT(far_status == BT::NodeStatus::FAILURE && fakes::globals().slept_ms == slept0 + 700.0, "tick far from every landmark: FAILURE after the 700 ms back-off");
            roadmap_explorer::ExplorationErrorCode code;
            EXPECT(config.blackboard->get<roadmap_explorer::ExplorationErrorCode>("error_code_id", code), "FAILURE leaves error_code_id on the blackboard");
            tick_info_status[next_to] = near_status == BT::NodeStatus::SUCCESS ? 1.0 : 0.0;
        }
    }
    out.push_back(tick_info_status[0]); out.push_back(tick_info_status[1]);
    out.push_back((double)n_poses);
    out.push_back((double)failures);
    FILE *o = fopen(argv[3], "wb");
    if (!o) { perror(argv[3]); return 2; }
    fwrite(out.data(), sizeof(double), out.size(), o);
    fclose(o);
    printf("failures: %d\n", failures);
    return failures ? 1 : 0;
}
