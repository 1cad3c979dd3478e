// FisherInfoManagerGPU.cpp — see the header.
#include "fitslam_frontier_ros2/FisherInfoManagerGPU.hpp"

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <unordered_set>

#include <nav2_util/geometry_utils.hpp>

#include "roadmap_explorer/Parameters.hpp"
#include "roadmap_explorer/util/Logger.hpp"

namespace fitslam_frontier_ros2
{

FisherInformationManagerGPU::FisherInformationManagerGPU(std::shared_ptr<nav2_util::LifecycleNode> node, std::vector<int> device_ids,
                                                         const std::string &lookup_file)
    : lookup_file_(lookup_file), node_(node)
{
    if (device_ids.empty()) {                                                    // the list CostAssignerGPU reads, too
        std::vector<int64_t> ids{0};
        if (!node_->has_parameter("fitslam_frontier.gpu_devices")) node_->declare_parameter("fitslam_frontier.gpu_devices", ids);
        node_->get_parameter("fitslam_frontier.gpu_devices", ids);
        for (const int64_t d : ids) device_ids.push_back(static_cast<int>(d));
    }
    // Where the table lives.  The reference hard-codes the path (FisherInfoManager.cpp:120,234) and that path is the default here;
    // the node parameter `fisherInformation.lookup_table_file` lets an install that keeps the file elsewhere say so (an explicit
    // constructor argument wins over it).
    if (lookup_file_ == kReferenceLookupFile) {
        if (!node_->has_parameter("fisherInformation.lookup_table_file")) node_->declare_parameter("fisherInformation.lookup_table_file", lookup_file_);
        node_->get_parameter("fisherInformation.lookup_table_file", lookup_file_);
    }
    if (fs_multi_create(device_ids.data(), static_cast<int>(device_ids.size()), &scorer_) != FS_OK)
        throw std::runtime_error("fitslam_frontier: no MI355X (gfx950) under one of fitslam_frontier.gpu_devices; there is no CPU fallback");
    // Which landmarks a pose "sees" is the reference's REQUEST to the SLAM server (FisherInfoManager.cpp:63-64):
    // max_dist_pose_observation = 14.0, max_angle_pose_observation = 4.0 — "greater than pi to disregard angle of
    // observation".  Those two numbers are this manager's defaults, as node parameters of the same names, so that
    // isPoseSafe's verdicts against the threshold of 550 are taken over the set the reference asks for (a value >= pi
    // switches the cone off).  What the un-vendored server does beyond the request is unknown; in-tree hints
    // for a tighter camera model are LoadLookupMain.cpp:115 (1.0 rad about +x) — set max_angle_pose_observation to 1.0 for it.
    double max_dist = 14.0, max_angle = 4.0;
    if (!node_->has_parameter("fisherInformation.max_dist_pose_observation")) node_->declare_parameter("fisherInformation.max_dist_pose_observation", max_dist);
    if (!node_->has_parameter("fisherInformation.max_angle_pose_observation")) node_->declare_parameter("fisherInformation.max_angle_pose_observation", max_angle);
    node_->get_parameter("fisherInformation.max_dist_pose_observation", max_dist);
    node_->get_parameter("fisherInformation.max_angle_pose_observation", max_angle);
    fs_fim_params vis;
    vis.max_dist = max_dist;
    vis.max_angle = max_angle;
    check(fs_multi_set_fim_params(scorer_, &vis), "fs_multi_set_fim_params");
    loadLookupTable();                                                           // :10

    client_node_ = rclcpp::Node::make_shared("FIMManagerGPUClient");             // :9 (same pattern: a node of its own)
    map_data_subscription_ = client_node_->create_subscription<slam_msgs::msg::MapData>(
        "map_data", 10, std::bind(&FisherInformationManagerGPU::mapDataCallback, this, std::placeholders::_1));
    executor_ = std::make_shared<rclcpp::executors::SingleThreadedExecutor>();
    executor_->add_node(client_node_);
    spin_thread_ = std::thread([this]() { executor_->spin(); });
}

FisherInformationManagerGPU::~FisherInformationManagerGPU()
{
    // (the reference's destructor calls rclcpp::shutdown(), FisherInfoManager.cpp:22-28 — a latent bug that takes the
    // whole process down with the plugin; not reproduced)
    if (executor_) executor_->cancel();
    if (spin_thread_.joinable()) spin_thread_.join();
    map_data_subscription_.reset();
    client_node_.reset();
    fs_multi_destroy(scorer_);
}

void FisherInformationManagerGPU::check(int rc, const char *what) const
{
    if (rc != FS_OK) throw std::runtime_error(std::string(what) + ": " + fs_multi_last_error(scorer_));
}

void FisherInformationManagerGPU::generateLookupTable(float minX, float maxX, float minY, float maxY, float minZ, float maxZ)
{
    std::lock_guard<std::mutex> lock(ctx_mutex_);
    const float bounds[6] = {minX, maxX, minY, maxY, minZ, maxZ};
    check(fs_multi_lookup_generate(scorer_, bounds), "fs_multi_lookup_generate");   // FisherInfoManager.cpp:117-229 (generated once, handed to every device)
    fs_ctx *first = fs_multi_ctx(scorer_, 0);
    if (fs_lookup_save(first, lookup_file_.c_str()) != FS_OK)                    // byte-compatible 16-B records
        throw std::runtime_error(std::string("fs_lookup_save: ") + fs_last_error(first));
}

void FisherInformationManagerGPU::loadLookupTable()
{
    std::lock_guard<std::mutex> lock(ctx_mutex_);
    if (fs_multi_lookup_load(scorer_, lookup_file_.c_str()) != FS_OK)            // :231-262
        throw std::runtime_error("Cannot load lookup table. Does it exist in the path?");
}

void FisherInformationManagerGPU::setVisibility(double max_dist, double max_angle)
{
    std::lock_guard<std::mutex> lock(ctx_mutex_);
    fs_fim_params vis;
    vis.max_dist = max_dist;
    vis.max_angle = max_angle;
    check(fs_multi_set_fim_params(scorer_, &vis), "fs_multi_set_fim_params");
}

void FisherInformationManagerGPU::setLandmarks(const std::vector<float> &xyz_world)
{
    std::lock_guard<std::mutex> lock(ctx_mutex_);
    check(fs_multi_upload_landmarks(scorer_, xyz_world.data(), static_cast<int32_t>(xyz_world.size() / 3)), "fs_multi_upload_landmarks");
    have_landmarks_ = true;
}

void FisherInformationManagerGPU::mapDataCallback(const slam_msgs::msg::MapData::SharedPtr map_data)
{
    // The map points of every key-frame (nodes[].word_pts, world frame).  A map point observed from several key-frames
    // appears once per key-frame in the message; the crowding discount counts landmarks per voxel, so duplicates are
    // removed here (exact coordinates: they are copies of one map point).
    struct Key { float x, y, z; bool operator==(const Key &o) const { return x == o.x && y == o.y && z == o.z; } };
    struct KeyHash {
        size_t operator()(const Key &k) const
        {
            uint32_t b[3];
            std::memcpy(b, &k, sizeof b);
            size_t h = 0;
            for (uint32_t v : b) h ^= std::hash<uint32_t>{}(v) + 0x9e3779b9u + (h << 6) + (h >> 2);
            return h;
        }
    };
    std::unordered_set<Key, KeyHash> seen;
    std::vector<float> xyz;
    for (const auto &node : map_data->nodes) {
        for (const auto &p : node.word_pts) {
            const Key k{static_cast<float>(p.x), static_cast<float>(p.y), static_cast<float>(p.z)};
            if (!std::isfinite(k.x) || !std::isfinite(k.y) || !std::isfinite(k.z)) continue;
            if (!seen.insert(k).second) continue;
            xyz.push_back(k.x); xyz.push_back(k.y); xyz.push_back(k.z);
        }
    }
    setLandmarks(xyz);
    LOG_INFO("FisherInformationManagerGPU: staged " << xyz.size() / 3 << " landmarks from map_data");
}

bool FisherInformationManagerGPU::poseInformation(const std::vector<geometry_msgs::msg::Pose> &poses, std::vector<float> &information)
{
    information.assign(poses.size(), 0.0f);
    if (poses.empty()) return true;
    if (!have_landmarks_) {                                                      // the reference's "service not found" (:52-57)
        LOG_ERROR("No map_data received yet for fisher information");
        return false;
    }
    std::vector<double> pose7(poses.size() * 7);
    for (size_t i = 0; i < poses.size(); ++i) {
        const auto &p = poses[i];
        double *o = &pose7[7 * i];
        o[0] = p.position.x; o[1] = p.position.y; o[2] = p.position.z;
        o[3] = p.orientation.x; o[4] = p.orientation.y; o[5] = p.orientation.z; o[6] = p.orientation.w;
    }
    std::lock_guard<std::mutex> lock(ctx_mutex_);
    // info_ref alone (NULL for every other column): the INFO_ONLY worker — and, for the few poses of a tick, each pose spread over
    // several workgroups of its device (include/fitslam_frontier_dev.h, "fim.split")
    return fs_multi_score_fim(scorer_, static_cast<int32_t>(poses.size()), pose7.data(), information.data(),
                              nullptr, nullptr, nullptr, nullptr, nullptr) == FS_OK;
}

bool FisherInformationManagerGPU::isPoseSafe(geometry_msgs::msg::Pose &given_pose, bool exhaustiveSearch, float &information)
{
    // `exhaustiveSearch` travels in the reference's request (FisherInfoManager.cpp:65) and is acted on by the SLAM server:
    // false lets it restrict the search to its local map, true makes it go through every map point.  Here the whole
    // staged cloud is tested against the visibility volume on every call — the exhaustive search IS the only mode, it
    // costs microseconds — so `true` is honoured as asked and `false` cannot narrow anything: said once, not silently.
    if (!exhaustiveSearch && !told_about_exhaustive_.exchange(true))
        LOG_INFO("FisherInformationManagerGPU: exhaustive_search = false has no narrower mode on the GPU path; every staged landmark is tested on every query");
    const double fisher_information_threshold =
        parameterInstance.getValue<double>("fisherInformation.fisher_information_threshold");    // FisherInfoManager.cpp:41
    std::vector<float> info;
    if (!poseInformation({given_pose}, info)) return false;                      // :52-57,73-77: failure -> false
    LOG_WARN("Total information: " << info[0]);                                  // :99
    information = info[0];                                                       // :100
    return info[0] > fisher_information_threshold;                               // :112-114
}

bool FisherInformationManagerGPU::isPoseSafe(geometry_msgs::msg::Point point_from, geometry_msgs::msg::Point point_to, bool exhaustiveSearch)
{
    // FisherInfoManager.cpp:31-37 with getRelativePoseGivenTwoPoints (roadmap_explorer/util/GeometryUtils.hpp): position
    // = point_from, orientation = yaw of the segment about Z
    geometry_msgs::msg::Pose relative_pose;
    relative_pose.position = point_from;
    const double yaw = std::atan2(point_to.y - point_from.y, point_to.x - point_from.x);
    relative_pose.orientation = nav2_util::geometry_utils::orientationAroundZAxis(yaw);
    float information;
    return isPoseSafe(relative_pose, exhaustiveSearch, information);
}

}  // namespace fitslam_frontier_ros2
