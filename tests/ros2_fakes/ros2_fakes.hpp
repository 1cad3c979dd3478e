// ros2_fakes.hpp — TEST DOUBLES for the shapes of tests/ros2_decls/ros2_decls.hpp, so that the ROS 2 adapter under
// fit-slam_amd/host/ros2/ can be LINKED AND RUN on the GPU box (tests/test_gpu_ros2_adapter_run.py): the three adapter sources are
// compiled unchanged, every external #include answered by this file, and driven by tests/ros2_fakes/adapter_driver.cpp.
//
// What this is: the minimum behaviour each declared shape needs for the adapter's own code to execute — a costmap that is an
// array, a node that is a parameter map, a blackboard that is a map, a factory that is a map of builders, a tf buffer that returns
// one transform, a subscription that remembers its callback, the reference's Frontier record as a plain struct, and the
// reference's EUCLIDEAN planner (DEP/src/CostCalculator.cpp:445-484 restated; the roadmap / NavFn planners are out of scope,
// SURVEY.md §2) behind all three planner entry points.  What this is NOT: ROS, nav2, tf2 or BehaviorTree.CPP.  It says nothing
// about how the adapter behaves inside a ROS graph; row a23 stays "partial".  It does execute every line of the adapter that
// talks to the C ABI, on real inputs, and the test compares what comes out with the ctypes route and the oracle.
//
// The declarations file stays the ONE statement of the shapes (the parse test uses it alone): nothing is re-declared here.
// Its classes have no data members, so state lives in side tables keyed by the object's address.
#pragma once
#include "../ros2_decls/ros2_decls.hpp"

#include <any>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <limits>
#include <map>
#include <thread>

namespace fakes {

template <typename S>
inline S &state_of(const void *self)
{
    static std::mutex m;
    static std::unordered_map<const void *, std::unique_ptr<S>> table;
    std::lock_guard<std::mutex> lock(m);
    std::unique_ptr<S> &p = table[self];
    if (!p) p = std::make_unique<S>();
    return *p;
}

struct CostmapState {
    std::vector<unsigned char> cells; unsigned int nx = 0, ny = 0; double ox = 0, oy = 0, res = 0.05;
    nav2_costmap_2d::Costmap2D::mutex_t mutex;
    int locked_reads = 0;                       // getCharMap calls (the adapter must hold the mutex around its snapshot)
};
struct LayeredState { nav2_costmap_2d::Costmap2D *costmap = nullptr; };
struct CostmapRosState { nav2_costmap_2d::LayeredCostmap layered; nav2_costmap_2d::Costmap2D costmap; double robot_radius = 0.6; };
struct NodeParams { std::map<std::string, std::any> values; };
struct FrontierState {
    geometry_msgs::msg::Point goal{0, 0, 0}; int size = 0;
    double goal_orientation = 0, arrival = 0, path_length = 0, path_length_m = 0, path_heading = 0, fisher = 0, weighted_cost = 0;
    bool achievable = true;
    std::map<std::string, double> costs;
};
struct BlackboardState { std::map<std::string, std::any> values; };
struct TreeNodeState { BT::NodeConfiguration config; std::string name; };
struct FactoryState { std::unordered_map<std::string, BT::NodeBuilder> builders; };
struct ExecutorState { std::mutex m; std::condition_variable cv; bool cancelled = false; };

// an object of a shape without a destructor that lives on the stack (a factory in a loop) must be forgotten by hand before its
// address is used again
template <typename S>
inline void forget(const void *self) { state_of<S>(self) = S(); }

struct Globals {
    std::map<std::string, double> parameters;                         // the reference's ParameterHandler (double / bool)
    std::map<std::string, std::string> input_ports;                   // BT input ports of the node under test
    geometry_msgs::msg::TransformStamped tf_map_base;                 // what tf2_ros::Buffer::lookupTransform returns
    std::map<std::string, std::function<void(std::shared_ptr<void>)>> topics;
    double slept_ms = 0;
    int planner_calls = 0, planner_resets = 0, normalisation_calls = 0;
    std::vector<std::string> log;
};
inline Globals &globals() { static Globals g; return g; }

inline FrontierPtr make_frontier(double x, double y, int size)
{
    FrontierPtr f = std::make_shared<Frontier>();
    FrontierState &s = state_of<FrontierState>(f.get());
    s = FrontierState();
    s.goal.x = x; s.goal.y = y; s.size = size;
    return f;
}
inline FrontierState &frontier(const Frontier *f) { return state_of<FrontierState>(f); }

template <typename M>
inline void publish(const std::string &topic, std::shared_ptr<M> msg)
{
    auto it = globals().topics.find(topic);
    if (it != globals().topics.end()) it->second(std::static_pointer_cast<void>(msg));
}

}  // namespace fakes

// ---- geometry_msgs
inline bool geometry_msgs::msg::Point::operator==(const Point &o) const { return x == o.x && y == o.y && z == o.z; }

// ---- rclcpp
inline rclcpp::Node::SharedPtr rclcpp::Node::make_shared(const std::string &) { return std::make_shared<rclcpp::Node>(); }
template <typename M, typename F>
typename rclcpp::Subscription<M>::SharedPtr rclcpp::Node::create_subscription(const std::string &topic, int, F &&callback)
{
    std::function<void(const std::shared_ptr<M>)> cb = std::forward<F>(callback);
    fakes::globals().topics[topic] = [cb](std::shared_ptr<void> msg) { cb(std::static_pointer_cast<M>(msg)); };
    return std::make_shared<rclcpp::Subscription<M>>();
}
inline void rclcpp::executors::SingleThreadedExecutor::add_node(rclcpp::Node::SharedPtr)
{
    // (state is keyed by address and the shapes have no destructors to clear it: a new executor may sit where a cancelled one sat)
    fakes::ExecutorState &s = fakes::state_of<fakes::ExecutorState>(this);
    std::lock_guard<std::mutex> lock(s.m);
    s.cancelled = false;
}
inline void rclcpp::executors::SingleThreadedExecutor::spin()
{
    fakes::ExecutorState &s = fakes::state_of<fakes::ExecutorState>(this);
    std::unique_lock<std::mutex> lock(s.m);
    s.cv.wait(lock, [&] { return s.cancelled; });
}
inline void rclcpp::executors::SingleThreadedExecutor::cancel()
{
    fakes::ExecutorState &s = fakes::state_of<fakes::ExecutorState>(this);
    { std::lock_guard<std::mutex> lock(s.m); s.cancelled = true; }
    s.cv.notify_all();
}
inline void rclcpp::sleep_for(std::chrono::nanoseconds d) { fakes::globals().slept_ms += std::chrono::duration<double, std::milli>(d).count(); }

// ---- nav2_util
inline nav2_util::LifecycleNode::~LifecycleNode() {}
inline bool nav2_util::LifecycleNode::has_parameter(const std::string &k) const { return fakes::state_of<fakes::NodeParams>(this).values.count(k) > 0; }
template <typename T> void nav2_util::LifecycleNode::declare_parameter(const std::string &k, const T &v) { fakes::state_of<fakes::NodeParams>(this).values[k] = v; }
template <typename T> bool nav2_util::LifecycleNode::get_parameter(const std::string &k, T &v) const
{
    auto &vals = fakes::state_of<fakes::NodeParams>(this).values;
    auto it = vals.find(k);
    if (it == vals.end()) return false;
    v = std::any_cast<T>(it->second);
    return true;
}
inline geometry_msgs::msg::Quaternion nav2_util::geometry_utils::orientationAroundZAxis(double angle)
{
    // tf2::Quaternion::setRPY(0, 0, angle) (upstream Humble)
    geometry_msgs::msg::Quaternion q;
    q.x = 0.0; q.y = 0.0; q.z = std::sin(angle * 0.5); q.w = std::cos(angle * 0.5);
    return q;
}

// ---- nav2_costmap_2d
inline nav2_costmap_2d::Costmap2D::mutex_t *nav2_costmap_2d::Costmap2D::getMutex() { return &fakes::state_of<fakes::CostmapState>(this).mutex; }
inline unsigned char *nav2_costmap_2d::Costmap2D::getCharMap() const
{
    fakes::CostmapState &s = fakes::state_of<fakes::CostmapState>(this);
    // the snapshot must be taken under the costmap's mutex: a recursive mutex the calling thread owns can be locked again, one
    // that another thread could take cannot be told apart from here — so count the reads and let the driver check the lock itself
    ++s.locked_reads;
    return s.cells.data();
}
inline unsigned int nav2_costmap_2d::Costmap2D::getSizeInCellsX() const { return fakes::state_of<fakes::CostmapState>(this).nx; }
inline unsigned int nav2_costmap_2d::Costmap2D::getSizeInCellsY() const { return fakes::state_of<fakes::CostmapState>(this).ny; }
// upstream Humble costmap_2d.cpp: (size - 1 + 0.5) * resolution
inline double nav2_costmap_2d::Costmap2D::getSizeInMetersX() const { auto &s = fakes::state_of<fakes::CostmapState>(this); return (s.nx - 1 + 0.5) * s.res; }
inline double nav2_costmap_2d::Costmap2D::getSizeInMetersY() const { auto &s = fakes::state_of<fakes::CostmapState>(this); return (s.ny - 1 + 0.5) * s.res; }
inline double nav2_costmap_2d::Costmap2D::getOriginX() const { return fakes::state_of<fakes::CostmapState>(this).ox; }
inline double nav2_costmap_2d::Costmap2D::getOriginY() const { return fakes::state_of<fakes::CostmapState>(this).oy; }
inline double nav2_costmap_2d::Costmap2D::getResolution() const { return fakes::state_of<fakes::CostmapState>(this).res; }
inline nav2_costmap_2d::Costmap2D *nav2_costmap_2d::LayeredCostmap::getCostmap() { return fakes::state_of<fakes::LayeredState>(this).costmap; }
inline nav2_costmap_2d::LayeredCostmap *nav2_costmap_2d::Costmap2DROS::getLayeredCostmap()
{
    fakes::CostmapRosState &s = fakes::state_of<fakes::CostmapRosState>(this);
    fakes::state_of<fakes::LayeredState>(&s.layered).costmap = &s.costmap;
    return &s.layered;
}
inline nav2_costmap_2d::Costmap2D *nav2_costmap_2d::Costmap2DROS::getCostmap() { return &fakes::state_of<fakes::CostmapRosState>(this).costmap; }
inline double nav2_costmap_2d::Costmap2DROS::getRobotRadius() { return fakes::state_of<fakes::CostmapRosState>(this).robot_radius; }

// ---- tf2
inline const tf2::TimePoint tf2::TimePointZero{};
inline geometry_msgs::msg::TransformStamped tf2_ros::Buffer::lookupTransform(const std::string &, const std::string &, const tf2::TimePoint &) const
{
    return fakes::globals().tf_map_base;
}

// ---- BehaviorTree.CPP
template <typename T> std::pair<std::string, BT::PortInfo> BT::InputPort(const std::string &name) { return {name, BT::PortInfo{}}; }
template <typename T> bool BT::Blackboard::get(const std::string &key, T &value) const
{
    auto &vals = fakes::state_of<fakes::BlackboardState>(this).values;
    auto it = vals.find(key);
    if (it == vals.end()) return false;
    value = std::any_cast<T>(it->second);
    return true;
}
template <typename T> void BT::Blackboard::set(const std::string &key, const T &value) { fakes::state_of<fakes::BlackboardState>(this).values[key] = value; }
inline BT::TreeNode::TreeNode(const std::string &name, const BT::NodeConfiguration &config)
{
    fakes::TreeNodeState &s = fakes::state_of<fakes::TreeNodeState>(this);
    s.config = config; s.name = name;
}
inline BT::TreeNode::~TreeNode() {}
inline const BT::NodeConfiguration &BT::TreeNode::config() const { return fakes::state_of<fakes::TreeNodeState>(this).config; }
template <typename T> bool BT::TreeNode::getInput(const std::string &key, T &destination) const
{
    auto it = fakes::globals().input_ports.find(key);
    if (it == fakes::globals().input_ports.end()) return false;
    if constexpr (std::is_same<T, bool>::value) destination = (it->second == "true" || it->second == "1");
    else { std::istringstream in(it->second); in >> destination; }
    return true;
}
inline const std::unordered_map<std::string, BT::NodeBuilder> &BT::BehaviorTreeFactory::builders() const { return fakes::state_of<fakes::FactoryState>(this).builders; }
inline bool BT::BehaviorTreeFactory::unregisterBuilder(const std::string &id) { return fakes::state_of<fakes::FactoryState>(this).builders.erase(id) > 0; }
template <typename T> void BT::BehaviorTreeFactory::registerBuilder(const std::string &id, const BT::NodeBuilder &builder)
{
    auto &b = fakes::state_of<fakes::FactoryState>(this).builders;
    if (b.count(id)) throw std::runtime_error("ID [" + id + "] already registered");      // BehaviorTree.CPP v3 throws BehaviorTreeException here
    T::providedPorts();                                                                    // (the manifest: the node class must offer it)
    b[id] = builder;
}

// ---- the reference's own headers
inline ParameterHandler &ParameterHandler::getInstance() { static ParameterHandler h; return h; }
template <typename T> T ParameterHandler::getValue(std::string key)
{
    auto it = fakes::globals().parameters.find(key);
    if (it == fakes::globals().parameters.end()) throw std::runtime_error("Parameter " + key + " is not found in the map");   // Parameters.hpp:38-41
    return static_cast<T>(it->second);
}
template <typename T> void ParameterHandler::setValue(const std::string &key, const T &value) { fakes::globals().parameters[key] = static_cast<double>(value); }

inline void Frontier::setGoalOrientation(double v) { fakes::frontier(this).goal_orientation = v; }
inline void Frontier::setArrivalInformation(double v) { fakes::frontier(this).arrival = v; }
inline void Frontier::setPathLength(double v) { fakes::frontier(this).path_length = v; }
inline void Frontier::setPathLengthInM(double v) { fakes::frontier(this).path_length_m = v; }
inline void Frontier::setFisherInformation(double v) { fakes::frontier(this).fisher = v; }
inline void Frontier::setCost(std::string name, double v) { fakes::frontier(this).costs[name] = v; }
inline void Frontier::setWeightedCost(double v) { fakes::frontier(this).weighted_cost = v; }
inline void Frontier::setAchievability(bool v) { fakes::frontier(this).achievable = v; }
inline bool Frontier::operator==(const Frontier &o) const                                  // Frontier.hpp:121-131: goal point and size
{
    const fakes::FrontierState &a = fakes::frontier(this), &b = fakes::frontier(&o);
    return a.goal == b.goal && a.size == b.size;
}
inline int Frontier::getSize() const { return fakes::frontier(this).size; }
inline geometry_msgs::msg::Point &Frontier::getGoalPoint() const { return fakes::frontier(this).goal; }
inline double Frontier::getArrivalInformation() const { return fakes::frontier(this).arrival; }
inline double Frontier::getPathLength() const { return fakes::frontier(this).path_length; }
inline double Frontier::getPathLengthInM() const { return fakes::frontier(this).path_length_m; }
inline double Frontier::getPathHeading() const { return fakes::frontier(this).path_heading; }
inline double Frontier::getWeightedCost() const { return fakes::frontier(this).weighted_cost; }
inline bool Frontier::isAchievable() const { return fakes::frontier(this).achievable; }
inline bool FrontierGoalPointEquality::operator()(const FrontierPtr &l, const FrontierPtr &r) const { return l->getGoalPoint() == r->getGoalPoint(); }
inline size_t FrontierHash::operator()(const FrontierPtr &k) const
{
    // Frontier.hpp:177-189: a hash of the goal point's coordinates
    const geometry_msgs::msg::Point &g = k->getGoalPoint();
    return std::hash<double>{}(g.x) ^ (std::hash<double>{}(g.y) << 1);
}

namespace frontier_exploration {
inline FrontierCostCalculator::FrontierCostCalculator(std::shared_ptr<nav2_costmap_2d::Costmap2DROS>) {}
// DEP/src/CostCalculator.cpp:445-484, restated (quatToEuler(...)[2] = tf2's yaw of the quaternion)
inline void FrontierCostCalculator::setPlanForFrontierEuclidean(geometry_msgs::msg::Pose start, FrontierPtr &goal, std::shared_ptr<slam_msgs::srv::GetMap_Response>, bool, bool)
{
    ++fakes::globals().planner_calls;
    fakes::FrontierState &f = fakes::frontier(goal.get());
    const double dmax = std::numeric_limits<double>::max();
    auto give_up = [&] { f.achievable = false; f.path_length = dmax; f.path_length_m = dmax; f.path_heading = dmax; f.fisher = 0; };
    if (!f.achievable) { give_up(); return; }
    const double length = std::sqrt(std::pow(start.position.x - f.goal.x, 2) + std::pow(start.position.y - f.goal.y, 2));
    if (length < 0.5) { give_up(); return; }
    f.achievable = true;
    const auto &q = start.orientation;
    double robot_yaw = std::atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z));
    if (robot_yaw < 0) robot_yaw = robot_yaw + (M_PI * 2);
    double goal_yaw = std::atan2(f.goal.y - start.position.y, f.goal.x - start.position.x);
    if (goal_yaw < 0) goal_yaw = goal_yaw + (M_PI * 2);
    double heading = std::abs(robot_yaw - goal_yaw);
    if (heading > M_PI) heading = (2 * M_PI) - heading;
    f.path_length = length; f.path_length_m = length; f.path_heading = heading; f.fisher = 0.0;
}
inline void FrontierCostCalculator::setPlanForFrontier(geometry_msgs::msg::Pose s, FrontierPtr &g, std::shared_ptr<slam_msgs::srv::GetMap_Response> m, bool c, bool u) { setPlanForFrontierEuclidean(s, g, m, c, u); }
inline void FrontierCostCalculator::setPlanForFrontierRoadmap(geometry_msgs::msg::Pose s, FrontierPtr &g, std::shared_ptr<slam_msgs::srv::GetMap_Response> m, bool c, bool u) { setPlanForFrontierEuclidean(s, g, m, c, u); }
inline void FrontierCostCalculator::recomputeNormalizationFactors(FrontierPtr &) { ++fakes::globals().normalisation_calls; }
inline void FrontierCostCalculator::reset() { ++fakes::globals().planner_resets; }
}  // namespace frontier_exploration

namespace roadmap_explorer {
inline BTPlugin::~BTPlugin() {}
// stand-ins for the reference plugin's two nodes: only their IDs matter to the adapter's registration logic
struct FakeReferenceNode : BT::SyncActionNode {
    using BT::SyncActionNode::SyncActionNode;
    BT::NodeStatus tick() override { return BT::NodeStatus::IDLE; }
    static BT::PortsList providedPorts() { return {}; }
};
inline FisherInfoBTPlugin::FisherInfoBTPlugin() {}
inline FisherInfoBTPlugin::~FisherInfoBTPlugin() {}
inline void FisherInfoBTPlugin::registerNodes(BT::BehaviorTreeFactory &factory, std::shared_ptr<nav2_util::LifecycleNode>, std::shared_ptr<nav2_costmap_2d::Costmap2DROS>, std::shared_ptr<tf2_ros::Buffer>)
{
    BT::NodeBuilder b = [](const std::string &name, const BT::NodeConfiguration &config) { return std::make_unique<FakeReferenceNode>(name, config); };
    factory.registerBuilder<FakeReferenceNode>("EvaluateFisherInformation", b);      // FisherInfoBTPlugin.cpp:199-214
    factory.registerBuilder<FakeReferenceNode>("MarkLethalFOV", b);
}
}  // namespace roadmap_explorer
