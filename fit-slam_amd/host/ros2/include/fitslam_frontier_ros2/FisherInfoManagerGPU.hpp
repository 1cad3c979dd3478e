// FisherInfoManagerGPU.hpp — roadmap_explorer::FisherInformationManager's run-time interface
// (FIP/include/fisher_information_plugins/fisher_information/FisherInfoManager.hpp:72-145) on an MI355X.
//
// What changes against the reference: the per-pose service round trip `orb_slam3/get_landmarks_in_view`
// (FIP/src/fisher_information/FisherInfoManager.cpp:52-77) is replaced by the landmark cloud staged ONCE per SLAM map
// update from topic `map_data` (slam_msgs/msg/MapData — the message the roadmap already subscribes to,
// DEP/src/planners/FrontierRoadmap.cpp:28); the camera-frame transform and the visibility predicate the server applied
// are explicit — node parameters fisherInformation.max_dist_pose_observation / max_angle_pose_observation, defaulting to the
// reference's request (14.0 m, 4.0 rad = no cone; FisherInfoManager.cpp:63-64); the table look-up + crowding discount + sum (:83-100,287-324) run in
// fs_score_fim.  Same lookup-table file, same `information > threshold` decision (:112-114).
#ifndef FITSLAM_FRONTIER_ROS2_FISHER_INFO_MANAGER_GPU_HPP_
#define FITSLAM_FRONTIER_ROS2_FISHER_INFO_MANAGER_GPU_HPP_

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <geometry_msgs/msg/point.hpp>
#include <geometry_msgs/msg/pose.hpp>
#include <nav2_util/lifecycle_node.hpp>
#include <rclcpp/rclcpp.hpp>
#include <slam_msgs/msg/map_data.hpp>

#include "fitslam_frontier.h"

namespace fitslam_frontier_ros2
{

class FisherInformationManagerGPU
{
public:
    // FisherInfoManager.cpp:6-16: loads the lookup table at construction (throws when the file is missing, :240)
    // device_ids: HIP ordinals to score on; empty = the node's parameter `fitslam_frontier.gpu_devices` (default [0]) — the same
    // list CostAssignerGPU reads: one isPoseSafe pose runs on the first device, a batch (poseInformation) is cut into one
    // contiguous block per device (fs_multi_score_fim)
    static constexpr const char *kReferenceLookupFile = "/root/dev_ws/src/lookup_table_fi/fisher_information_lookup_table.dat";   // FisherInfoManager.cpp:120,234
    explicit FisherInformationManagerGPU(std::shared_ptr<nav2_util::LifecycleNode> node, std::vector<int> device_ids = {},
                                         const std::string &lookup_file = kReferenceLookupFile);
    ~FisherInformationManagerGPU();
    FisherInformationManagerGPU(const FisherInformationManagerGPU &) = delete;
    FisherInformationManagerGPU &operator=(const FisherInformationManagerGPU &) = delete;

    // FisherInfoManager.hpp:94-96
    void generateLookupTable(float minX, float maxX, float minY, float maxY, float minZ, float maxZ);
    void loadLookupTable();

    // FisherInfoManager.hpp:125-127
    bool isPoseSafe(geometry_msgs::msg::Pose &given_pose, bool exhaustiveSearch, float &information);
    bool isPoseSafe(geometry_msgs::msg::Point point_from, geometry_msgs::msg::Point point_to, bool exhaustiveSearch);

    // batch form for callers that score many poses per tick (one launch)
    bool poseInformation(const std::vector<geometry_msgs::msg::Pose> &poses, std::vector<float> &information);

    // stage a cloud directly (tests, or a SLAM front end that already holds the map points)
    void setLandmarks(const std::vector<float> &xyz_world);
    void setVisibility(double max_dist, double max_angle);

private:
    void mapDataCallback(const slam_msgs::msg::MapData::SharedPtr map_data);
    void check(int rc, const char *what) const;

    fs_multi *scorer_ = nullptr;                 // one fs_ctx per GPU of the device list, one calling thread at a time (ctx_mutex_)
    std::string lookup_file_;
    std::shared_ptr<nav2_util::LifecycleNode> node_;
    rclcpp::Node::SharedPtr client_node_;
    rclcpp::Subscription<slam_msgs::msg::MapData>::SharedPtr map_data_subscription_;
    std::shared_ptr<rclcpp::executors::SingleThreadedExecutor> executor_;
    std::thread spin_thread_;
    std::mutex ctx_mutex_;                       // the context is single-caller: map updates vs. BT ticks
    std::atomic<bool> have_landmarks_{false};
    std::atomic<bool> told_about_exhaustive_{false};
};

}  // namespace fitslam_frontier_ros2

#endif
