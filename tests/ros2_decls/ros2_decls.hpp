// ros2_decls.hpp — DECLARATIONS ONLY: the shapes (types, member names, signatures) that the three adapter sources under
// fit-slam_amd/host/ros2/src/ assume of ROS 2 Humble, nav2, tf2, BehaviorTree.CPP, pluginlib, slam_msgs and of the reference's
// own headers.  None of those exists in this image; tests/test_ros2_adapter_parse.py points every such #include at this file
// and runs `g++ -std=c++17 -fsyntax-only`.  That proves the adapter PARSES AND TYPE-CHECKS AGAINST THESE SHAPES — nothing
// about ROS, linking or behaviour.  No function here has a body that does anything; nothing here is ever linked.
// Sources of the shapes: upstream Humble APIs as the reference calls them (call sites cited) and, for the reference's own
// types, its headers (DEP/ = dev_ws/src/DEPRECATED/frontier_exploration/frontier_exploration/, FIP/ = dev_ws/src/fit-slam2/
// fisher_information_plugins/).
#pragma once
#include <chrono>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

namespace geometry_msgs::msg {                                   // geometry_msgs/msg/*.hpp
struct Point { double x, y, z; bool operator==(const Point &) const; };
struct Point32 { float x, y, z; };
struct Vector3 { double x, y, z; };
struct Quaternion { double x, y, z, w; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { Pose pose; };
struct Polygon { std::vector<Point32> points; };
struct PolygonStamped { Polygon polygon; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { Transform transform; };
}
namespace slam_msgs::msg {                                       // slam_msgs/msg/map_data.hpp (DEP/src/planners/FrontierRoadmap.cpp:28)
struct KeyFrame { std::vector<geometry_msgs::msg::Point> word_pts; };
struct MapData { using SharedPtr = std::shared_ptr<MapData>; std::vector<KeyFrame> nodes; };
}
namespace slam_msgs::srv {                                       // slam_msgs/srv/get_map.hpp (DEP/include/.../CostCalculator.hpp:66-73)
struct GetMap_Response {};
struct GetMap { using Response = GetMap_Response; };
}
namespace rclcpp {                                               // rclcpp/rclcpp.hpp (FIP/src/fisher_information/FisherInfoManager.cpp:9-16)
template <typename M> struct Subscription { using SharedPtr = std::shared_ptr<Subscription>; };
struct Node {
    using SharedPtr = std::shared_ptr<Node>;
    static SharedPtr make_shared(const std::string &name);
    template <typename M, typename F> typename Subscription<M>::SharedPtr create_subscription(const std::string &topic, int qos_depth, F &&callback);
};
namespace executors { struct SingleThreadedExecutor { void add_node(Node::SharedPtr); void spin(); void cancel(); }; }
void sleep_for(std::chrono::nanoseconds);
}
namespace nav2_util {                                            // nav2_util/lifecycle_node.hpp, nav2_util/geometry_utils.hpp
struct LifecycleNode : std::enable_shared_from_this<LifecycleNode> {
    virtual ~LifecycleNode();
    bool has_parameter(const std::string &) const;
    template <typename T> void declare_parameter(const std::string &, const T &default_value);
    template <typename T> bool get_parameter(const std::string &, T &value) const;
};
namespace geometry_utils { geometry_msgs::msg::Quaternion orientationAroundZAxis(double angle); }   // DEP/.../util/GeometryUtils.hpp:123
}
namespace nav2_costmap_2d {                                      // nav2_costmap_2d/costmap_2d[_ros].hpp, layered_costmap.hpp
struct Costmap2D {
    typedef std::recursive_mutex mutex_t;
    mutex_t *getMutex();
    unsigned char *getCharMap() const;
    unsigned int getSizeInCellsX() const; unsigned int getSizeInCellsY() const;
    double getSizeInMetersX() const; double getSizeInMetersY() const;
    double getOriginX() const; double getOriginY() const; double getResolution() const;
};
struct LayeredCostmap { Costmap2D *getCostmap(); };
struct Costmap2DROS : nav2_util::LifecycleNode {                 // DEP/src/CostAssigner.cpp:11, DEP/src/CostCalculator.cpp:19
    LayeredCostmap *getLayeredCostmap(); Costmap2D *getCostmap(); double getRobotRadius();
};
}
namespace tf2 { using TimePoint = std::chrono::time_point<std::chrono::system_clock>; extern const TimePoint TimePointZero; }
namespace tf2_ros {                                              // tf2_ros/buffer.h (FIP/src/fisher_information/FisherInfoBTPlugin.cpp:34-39)
struct Buffer { geometry_msgs::msg::TransformStamped lookupTransform(const std::string &target, const std::string &source, const tf2::TimePoint &) const; };
}
namespace BT {                                                   // behaviortree_cpp_v3 (FIP/src/fisher_information/FisherInfoBTPlugin.cpp:12-70,199-214)
enum class NodeStatus { IDLE, RUNNING, SUCCESS, FAILURE };
struct PortInfo {};
using PortsList = std::unordered_map<std::string, PortInfo>;
template <typename T> std::pair<std::string, PortInfo> InputPort(const std::string &name);
struct Blackboard {
    using Ptr = std::shared_ptr<Blackboard>;
    template <typename T> bool get(const std::string &key, T &value) const;
    template <typename T> void set(const std::string &key, const T &value);
};
struct NodeConfiguration { Blackboard::Ptr blackboard; };
struct TreeNode {
    TreeNode(const std::string &name, const NodeConfiguration &config);
    virtual ~TreeNode();
    const NodeConfiguration &config() const;
    template <typename T> bool getInput(const std::string &key, T &destination) const;
};
struct SyncActionNode : TreeNode { using TreeNode::TreeNode; virtual NodeStatus tick() = 0; };
using NodeBuilder = std::function<std::unique_ptr<TreeNode>(const std::string &, const NodeConfiguration &)>;
struct BehaviorTreeFactory {
    const std::unordered_map<std::string, NodeBuilder> &builders() const;
    bool unregisterBuilder(const std::string &ID);
    template <typename T> void registerBuilder(const std::string &ID, const NodeBuilder &builder);
};
}
// pluginlib/class_list_macros.hpp: the export macro registers `cls` as an implementation of `base` — checked here as inheritance
#define PLUGINLIB_EXPORT_CLASS(cls, base) static_assert(std::is_base_of<base, cls>::value, #cls " must derive from " #base);

// ---- the reference's own headers
// DEP/include/frontier_exploration/Parameters.hpp:22-65 (and roadmap_explorer/Parameters.hpp: same interface)
class ParameterHandler {
public:
    template <typename T> T getValue(std::string parameterKey);
    template <typename T> void setValue(const std::string &parameterKey, const T &value);
    static ParameterHandler &getInstance();
};
inline ParameterHandler &parameterInstance = ParameterHandler::getInstance();
// DEP/include/frontier_exploration/util/logger.hpp:92-117 (stream-style macros); roadmap_explorer/util/Logger.hpp likewise
#define FS_DECL_LOG(X) do { std::ostringstream fs_decl_log_; fs_decl_log_ << X; } while (0)
#define LOG_INFO(X) FS_DECL_LOG(X)
#define LOG_WARN(X) FS_DECL_LOG(X)
#define LOG_ERROR(X) FS_DECL_LOG(X)
#define LOG_FATAL(X) FS_DECL_LOG(X)
// DEP/include/frontier_exploration/Frontier.hpp:60-175
class Frontier {
public:
    void setGoalOrientation(double theta); void setArrivalInformation(double info); void setPathLength(double pl);
    void setPathLengthInM(double pl); void setFisherInformation(double fi); void setCost(std::string costName, double value);
    void setWeightedCost(double cost); void setAchievability(bool value);
    bool operator==(const Frontier &other) const;
    int getSize() const; geometry_msgs::msg::Point &getGoalPoint() const; double getArrivalInformation() const;
    double getPathLength() const; double getPathLengthInM() const; double getPathHeading() const; double getWeightedCost() const;
    bool isAchievable() const;
};
using FrontierPtr = std::shared_ptr<Frontier>;
struct FrontierGoalPointEquality { bool operator()(const FrontierPtr &lhs, const FrontierPtr &rhs) const; };
struct FrontierHash { size_t operator()(const FrontierPtr &key) const; };
namespace frontier_exploration {
// DEP/include/frontier_exploration/CostAssigner.hpp:43-59
struct GetFrontierCostsRequest {
    geometry_msgs::msg::PoseStamped start_pose; std::vector<FrontierPtr> frontier_list;
    std::vector<std::vector<double>> every_frontier; std::vector<FrontierPtr> prohibited_frontiers;
};
struct GetFrontierCostsResponse {
    bool success; std::vector<FrontierPtr> frontier_list; std::vector<double> frontier_costs, frontier_distances,
    frontier_arrival_information, frontier_path_information;
};
// DEP/include/frontier_exploration/CostCalculator.hpp:45-129 (the planner entry points the adapter keeps calling)
class FrontierCostCalculator {
public:
    explicit FrontierCostCalculator(std::shared_ptr<nav2_costmap_2d::Costmap2DROS> explore_costmap_ros);
    void setPlanForFrontier(geometry_msgs::msg::Pose start_pose_w, FrontierPtr &goal_point_w, std::shared_ptr<slam_msgs::srv::GetMap_Response> map_data, bool compute_information, bool planner_allow_unknown_);
    void setPlanForFrontierEuclidean(geometry_msgs::msg::Pose start_pose_w, FrontierPtr &goal_point_w, std::shared_ptr<slam_msgs::srv::GetMap_Response> map_data, bool compute_information, bool planner_allow_unknown_);
    void setPlanForFrontierRoadmap(geometry_msgs::msg::Pose start_pose_w, FrontierPtr &goal_point_w, std::shared_ptr<slam_msgs::srv::GetMap_Response> map_data, bool compute_information, bool planner_allow_unknown_);
    void recomputeNormalizationFactors(FrontierPtr &frontier);
    void reset();
};
}
namespace roadmap_explorer {
// roadmap_explorer/bt_plugins/interface_pluginlib.hpp (un-vendored): the base as FIP/include/.../FisherInfoBTPlugin.hpp:13-21 overrides it
class BTPlugin {
public:
    virtual ~BTPlugin();
    virtual void registerNodes(BT::BehaviorTreeFactory &factory, std::shared_ptr<nav2_util::LifecycleNode> node,
                               std::shared_ptr<nav2_costmap_2d::Costmap2DROS> explore_costmap_ros, std::shared_ptr<tf2_ros::Buffer> tf_buffer) = 0;
};
enum class ExplorationErrorCode { NO_ERROR };                    // FIP/src/fisher_information/FisherInfoBTPlugin.cpp:52
// FIP/include/fisher_information_plugins/fisher_information/FisherInfoBTPlugin.hpp:13-21 (the reference plugin, for MarkLethalFOV)
class FisherInfoBTPlugin : public BTPlugin {
public:
    FisherInfoBTPlugin(); ~FisherInfoBTPlugin();
    void registerNodes(BT::BehaviorTreeFactory &factory, std::shared_ptr<nav2_util::LifecycleNode> node,
                       std::shared_ptr<nav2_costmap_2d::Costmap2DROS> explore_costmap_ros, std::shared_ptr<tf2_ros::Buffer> tf_buffer) override;
};
}
