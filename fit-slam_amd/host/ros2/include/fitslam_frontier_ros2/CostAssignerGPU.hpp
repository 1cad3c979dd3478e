// CostAssignerGPU.hpp — drop-in for frontier_exploration::CostAssigner
// (DEP/include/frontier_exploration/CostAssigner.hpp:61-100) whose per-frontier scoring loop
// (FrontierCostsManager::assignCosts, DEP/src/FrontierCostsManager.cpp:47-213) runs on an MI355X through the C ABI of
// include/fitslam_frontier.h: ONE fs_multi_score_arrival call for the whole frontier list — cut into one block per GPU of
// the node's parameter `fitslam_frontier.gpu_devices` (default [0]) inside this one process —, the reference's own planner
// per frontier (roadmap A*, out of the accelerated path), ONE fs_rank_candidates call for the U1 utilities and costs.
//
// Same constructor argument, same public methods, same GetFrontierCostsRequest / GetFrontierCostsResponse (the
// reference's own structs, not copies): ProcessFrontierCosts (DEP/src/ExplorationBT.cpp:376-441) only needs the type of
// its `bel_ptr_` member changed.  Source only — not compilable in the scorer's build image (no ROS 2 there).
#ifndef FITSLAM_FRONTIER_ROS2_COST_ASSIGNER_GPU_HPP_
#define FITSLAM_FRONTIER_ROS2_COST_ASSIGNER_GPU_HPP_

#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <geometry_msgs/msg/polygon.hpp>
#include <geometry_msgs/msg/polygon_stamped.hpp>
#include <geometry_msgs/msg/pose.hpp>
#include <nav2_costmap_2d/costmap_2d.hpp>
#include <nav2_costmap_2d/costmap_2d_ros.hpp>
#include <nav2_costmap_2d/layered_costmap.hpp>

#include "frontier_exploration/CostAssigner.hpp"      // GetFrontierCostsRequest / Response, FrontierPtr, FrontierHash, ...
#include "frontier_exploration/CostCalculator.hpp"    // the reference's planner entry points (setPlanForFrontierRoadmap)
#include "frontier_exploration/Parameters.hpp"

#include "fitslam_frontier.h"

namespace fitslam_frontier_ros2
{

class CostAssignerGPU
{
public:
    // DEP/src/CostAssigner.cpp:9-20 + DEP/src/FrontierCostsManager.cpp:6-23 + DEP/src/CostCalculator.cpp:5-21
    // device_ids: HIP ordinals to score on; empty = the node's parameter `fitslam_frontier.gpu_devices` (default [0])
    explicit CostAssignerGPU(std::shared_ptr<nav2_costmap_2d::Costmap2DROS> explore_costmap_ros, std::vector<int> device_ids = {});
    ~CostAssignerGPU();
    CostAssignerGPU(const CostAssignerGPU &) = delete;
    CostAssignerGPU &operator=(const CostAssignerGPU &) = delete;

    // DEP/src/CostAssigner.cpp:121-167
    bool updateBoundaryPolygon(geometry_msgs::msg::PolygonStamped &explore_boundary);

    // DEP/src/CostAssigner.cpp:73-119
    bool getFrontierCosts(std::shared_ptr<frontier_exploration::GetFrontierCostsRequest> requestData,
                          std::shared_ptr<frontier_exploration::GetFrontierCostsResponse> resultData);

    // DEP/src/FrontierCostsManager.cpp:215-222
    void setFrontierBlacklist(std::vector<FrontierPtr> &blacklist);

    // which planner fills path length / heading: "RoadmapPlannerDistance" (the reference's choice,
    // DEP/src/CostAssigner.cpp:38), "A*PlannerDistance" or "EuclideanDistance"
    void setPlannerMethod(const std::string &method) { planner_method_ = method; }

    // true: the whole cost assignment as ONE device call, fs_multi_get_frontier_costs — the planner runs first (on every live
    // frontier, as if achievable), then arrival information on every GPU of gpu_devices, the blocks' records gathered device to
    // device over xGMI onto the first GPU, U1 costs there, one transfer back: the records never visit the host between scoring
    // and ranking.  Same costs, utilities, arrival information and achievability as the three-step route (the default, which
    // plans only what arrival information left achievable: FrontierCostsManager.cpp:88-91); what differs is spare planning work.
    void setFused(bool on) { fused_ = on; }

private:
    // FrontierCostsManager::assignCosts with the arrival-information loop batched on the GPU
    bool assignCosts(std::vector<FrontierPtr> &frontier_list, geometry_msgs::msg::Pose start_pose_w);
    bool assignCostsFused(std::vector<FrontierPtr> &frontier_list, geometry_msgs::msg::Pose start_pose_w);   // the same through ONE call
    bool prepareTick(std::vector<FrontierPtr> &frontier_list);   // what both routes do first (:51-72): snapshot, limits, list checks
    void plan(geometry_msgs::msg::Pose start_pose_w, FrontierPtr &frontier);   // the reference's planner named by planner_method_ (:98-109)
    void snapshotCostmap();                                  // fs_upload_grid under the costmap mutex
    void pushRayParams();                                    // fs_set_ray_params (+ cached arrival limits)
    void restoreArrivalLimits();                             // hands the cached limits back to every device
    void check(int rc, const char *what) const;              // FS_E_* -> std::runtime_error with fs_multi_last_error

    fs_multi *scorer_ = nullptr;                             // one fs_ctx per GPU, one calling thread (the BT tick thread)
    nav2_costmap_2d::LayeredCostmap *layered_costmap_ = nullptr;
    nav2_costmap_2d::Costmap2D *costmap_ = nullptr;
    std::shared_ptr<frontier_exploration::FrontierCostCalculator> planner_;   // planning only; never scores

    geometry_msgs::msg::Polygon polygon_;
    std::vector<double> polygon_xy_min_max_;
    std::string planner_method_ = "RoadmapPlannerDistance";
    bool fused_ = false;

    // parameters (same keys as the reference)
    double max_camera_depth_, delta_theta_, camera_fov_, robot_radius_;
    double alpha_, beta_, max_vx_, max_wx_;
    bool planner_allow_unknown_;

    // setMaxArrivalInformation's cache (DEP/src/CostCalculator.cpp:125,185-188)
    bool arrival_info_limits_set_ = false;
    double max_arrival_info_gt_ = 0.0, min_arrival_info_gt_ = 0.0;

    std::mutex blacklist_mutex_;
    std::unordered_map<FrontierPtr, bool, FrontierHash, FrontierGoalPointEquality> frontier_blacklist_;
};

}  // namespace fitslam_frontier_ros2

#endif
