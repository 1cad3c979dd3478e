// frontier_scoring.hpp — C++17 host-side mirror of the reference's scoring interface, implemented on
// the C ABI of include/fitslam_frontier.h (header-only; link with -lfitslam_frontier).
//
// FIT-SLAM's own host code is ROS 2 C++ (rclcpp / nav2 / pluginlib); that toolchain is absent from the
// build image, so this mirror keeps the reference's class and method names, argument meaning and error
// behaviour with plain C++ types in place of the ROS message types:
//   frontier_exploration::Costmap2D                 <- nav2_costmap_2d::Costmap2D accessors the path uses
//   frontier_exploration::Frontier / FrontierPtr    <- DEP/include/frontier_exploration/Frontier.hpp:39-134
//   frontier_exploration::FrontierCostCalculator    <- DEP/include/.../CostCalculator.hpp:46-136, DEP/src/CostCalculator.cpp:5-191,512-520
//   frontier_exploration::FrontierCostsManager      <- DEP/src/FrontierCostsManager.cpp:47-223
//   frontier_exploration::FrontierSearch             <- DEP/include/.../FrontierSearch.hpp:31-127, DEP/src/FrontierSearch.cpp:21-216 (clusters)
//   frontier_exploration::CostAssigner (+ Request/Response) <- DEP/include/.../CostAssigner.hpp:43-100, DEP/src/CostAssigner.cpp:29-167
//   frontier_exploration_information_affine::computeInformationForPose <- DEP/include/.../deprecated/util.hpp:840-916 (batched)
//   roadmap_explorer::FisherInformationManager      <- FIP/include/.../FisherInfoManager.hpp:72-145, FIP/src/.../FisherInfoManager.cpp:31-324
// (DEP/ = dev_ws/src/DEPRECATED/frontier_exploration/frontier_exploration/, FIP/ = dev_ws/src/fit-slam2/fisher_information_plugins/)
//
// What differs from the reference, by design: the per-frontier loop of assignCosts
// (FrontierCostsManager.cpp:74-119) is ONE batched GPU call; the path planner (roadmap A*, out of scope)
// is a pluggable callback defaulting to setPlanForFrontierEuclidean (CostCalculator.cpp:446-484); the
// landmark service round-trip (FisherInfoManager.cpp:52-77) is replaced by a cloud staged once with
// setLandmarks().  No scoring is ever computed on the CPU here.
#ifndef FITSLAM_FRONTIER_SCORING_HPP_
#define FITSLAM_FRONTIER_SCORING_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <queue>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "fitslam_frontier.h"

namespace frontier_exploration
{
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { Pose pose; };

// nav2_util::geometry_utils::orientationAroundZAxis == tf2::Quaternion::setRPY(0, 0, angle)
inline Quaternion orientationAroundZAxis(double angle)
{
    Quaternion q;
    q.x = 0.0; q.y = 0.0; q.z = std::sin(angle * 0.5); q.w = std::cos(angle * 0.5);
    return q;
}

// DEP/include/.../util/GeometryUtils.hpp:112-124
inline void getRelativePoseGivenTwoPoints(const Point &point_from, const Point &point_to, Pose &oriented_pose)
{
    const double dx = point_to.x - point_from.x, dy = point_to.y - point_from.y;
    oriented_pose.position = point_from;
    oriented_pose.orientation = orientationAroundZAxis(std::atan2(dy, dx));
}

// The subset of nav2_costmap_2d::Costmap2D the path touches, with an optional z extent (3-D extension).
class Costmap2D
{
public:
    Costmap2D(unsigned int size_x, unsigned int size_y, double resolution, double origin_x, double origin_y,
              unsigned char default_value = 0, unsigned int size_z = 1, double origin_z = 0.0)
        : size_x_(size_x), size_y_(size_y), size_z_(size_z), resolution_(resolution),
          origin_x_(origin_x), origin_y_(origin_y), origin_z_(origin_z),
          costmap_(static_cast<size_t>(size_x) * size_y * size_z, default_value) {}
    unsigned char getCost(unsigned int mx, unsigned int my) const { return costmap_[getIndex(mx, my)]; }
    unsigned char getCost(unsigned int index) const { return costmap_[index]; }
    void setCost(unsigned int mx, unsigned int my, unsigned char cost) { costmap_[getIndex(mx, my)] = cost; }
    unsigned char *getCharMap() { return costmap_.data(); }
    const unsigned char *getCharMap() const { return costmap_.data(); }
    unsigned int getIndex(unsigned int mx, unsigned int my) const { return my * size_x_ + mx; }
    void indexToCells(unsigned int index, unsigned int &mx, unsigned int &my) const { my = index / size_x_; mx = index - (my * size_x_); }
    bool worldToMap(double wx, double wy, unsigned int &mx, unsigned int &my) const
    {
        if (wx < origin_x_ || wy < origin_y_) return false;
        mx = static_cast<unsigned int>((wx - origin_x_) / resolution_);
        my = static_cast<unsigned int>((wy - origin_y_) / resolution_);
        return mx < size_x_ && my < size_y_;
    }
    void mapToWorld(unsigned int mx, unsigned int my, double &wx, double &wy) const
    {
        wx = origin_x_ + (mx + 0.5) * resolution_;
        wy = origin_y_ + (my + 0.5) * resolution_;
    }
    unsigned int getSizeInCellsX() const { return size_x_; }
    unsigned int getSizeInCellsY() const { return size_y_; }
    unsigned int getSizeInCellsZ() const { return size_z_; }
    double getSizeInMetersX() const { return (size_x_ - 1 + 0.5) * resolution_; }
    double getSizeInMetersY() const { return (size_y_ - 1 + 0.5) * resolution_; }
    double getOriginX() const { return origin_x_; }
    double getOriginY() const { return origin_y_; }
    double getOriginZ() const { return origin_z_; }
    double getResolution() const { return resolution_; }
    std::mutex &getMutex() { return mutex_; }

private:
    unsigned int size_x_, size_y_, size_z_;
    double resolution_, origin_x_, origin_y_, origin_z_;
    std::vector<unsigned char> costmap_;
    std::mutex mutex_;
};

// DEP/include/.../Frontier.hpp:39-134 — every getter throws std::runtime_error while its field is unset.
class Frontier
{
public:
    Frontier() : is_achievable(true), is_blacklisted(false) {}
    void setUID(size_t uid) { unique_id = uid; }
    void setSize(int sz) { size = sz; }
    void setGoalPoint(Point gp) { goal_point = gp; }
    void setGoalPoint(double x, double y) { Point p; p.x = x; p.y = y; goal_point = p; }
    void setGoalOrientation(double theta) { theta_s_star = theta; best_orientation = orientationAroundZAxis(theta); }
    void setArrivalInformation(double info) { information = info; }
    void setPathLength(double pl) { path_length = pl; }
    void setPathLengthInM(double pl) { path_length_m = pl; }
    void setPathHeading(double h) { path_heading = h; }
    void setFisherInformation(double fi) { fisher_information_in_path = fi; }
    void setCost(std::string name, double value) { costs[name] = value; }
    void setWeightedCost(double c) { weighted_cost = c; }
    void setAchievability(bool v) { is_achievable = v; }
    void setBlacklisted(bool v) { is_blacklisted = v; }
    size_t getUID() const { return need(unique_id, "unique_id"); }
    int getSize() const { return need(size, "size"); }
    const Point &getGoalPoint() const { return need(goal_point, "goal_point"); }
    const Quaternion &getGoalOrientation() const { return need(best_orientation, "best_orientation"); }
    double getGoalYaw() const { return need(theta_s_star, "theta_s_star"); }
    double getArrivalInformation() const { return need(information, "information"); }
    double getPathLength() const { return need(path_length, "path_length"); }
    double getPathLengthInM() const { return need(path_length_m, "path_length_m"); }
    double getPathHeading() const { return need(path_heading, "path_heading"); }
    double getFisherInformation() const { return need(fisher_information_in_path, "fisher_information_in_path"); }
    double getCost(const std::string &name) const
    {
        auto it = costs.find(name);
        if (it == costs.end()) throw std::runtime_error("Cost " + name + " is not set");
        return it->second;
    }
    double getWeightedCost() const { return need(weighted_cost, "weighted_cost"); }
    bool isAchievable() const { return is_achievable; }
    bool isBlacklisted() const { return is_blacklisted; }
    bool operator==(const Frontier &o) const { return getUID() == o.getUID(); }

private:
    template <typename T>
    static const T &need(const std::optional<T> &v, const char *what)
    {
        if (!v) throw std::runtime_error(std::string("Frontier field is null: ") + what);
        return *v;
    }
    std::optional<size_t> unique_id;
    std::optional<int> size;
    std::optional<Point> goal_point;
    std::optional<Quaternion> best_orientation;
    std::optional<double> theta_s_star, information, path_length, path_length_m, path_heading,
        fisher_information_in_path, weighted_cost;
    bool is_achievable, is_blacklisted;
    std::map<std::string, double> costs;
};
using FrontierPtr = std::shared_ptr<Frontier>;

// RAII owner of one fs_ctx (one GPU, one stream); shared by the mirror classes.
class ScoringContext
{
public:
    explicit ScoringContext(int device = 0)
    {
        const int rc = fs_ctx_create(device, nullptr, &ctx_);
        if (rc != FS_OK) throw std::runtime_error("fs_ctx_create failed: no MI355X (gfx950) device or HIP runtime; there is no CPU fallback");
    }
    ~ScoringContext() { fs_ctx_destroy(ctx_); }
    ScoringContext(const ScoringContext &) = delete;
    ScoringContext &operator=(const ScoringContext &) = delete;
    fs_ctx *get() const { return ctx_; }
    void check(int rc, const char *what) const
    {
        if (rc != FS_OK) throw std::runtime_error(std::string(what) + ": " + fs_last_error(ctx_));
    }

private:
    fs_ctx *ctx_ = nullptr;
};

class FrontierCostCalculator
{
public:
    // DEP/src/CostCalculator.cpp:5-21 — parameters costCalculator/{max_camera_depth,delta_theta,camera_fov} and
    // Costmap2DROS::getRobotRadius(); the costmap pointer is kept, the grid is snapshotted by updateCostmap().
    FrontierCostCalculator(std::shared_ptr<ScoringContext> ctx, std::shared_ptr<Costmap2D> costmap,
                           double max_camera_depth = 2.0, double delta_theta = 0.10, double camera_fov = 1.04,
                           double robot_radius = 0.60, double factor_of_max_is_min = 0.70)
        : ctx_(std::move(ctx)), exploration_costmap_(std::move(costmap)), MAX_CAMERA_DEPTH(max_camera_depth),
          DELTA_THETA(delta_theta), CAMERA_FOV(camera_fov), robot_radius_(robot_radius), factor_min_(factor_of_max_is_min)
    {
        polygon_[0] = polygon_[1] = -std::numeric_limits<double>::max();
        polygon_[2] = polygon_[3] = std::numeric_limits<double>::max();
        pushParams();
        updateCostmap();
    }

    // Snapshot the costmap into HBM (the reference reads the live costmap without taking its mutex,
    // ExplorationBT.cpp:171,195 lock only the search; here the copy is the explicit snapshot point).
    void updateCostmap()
    {
        std::lock_guard<std::mutex> lock(exploration_costmap_->getMutex());
        const double origin[3] = {exploration_costmap_->getOriginX(), exploration_costmap_->getOriginY(), exploration_costmap_->getOriginZ()};
        ctx_->check(fs_upload_grid(ctx_->get(), exploration_costmap_->getCharMap(), (int32_t)exploration_costmap_->getSizeInCellsX(),
                                   (int32_t)exploration_costmap_->getSizeInCellsY(), (int32_t)exploration_costmap_->getSizeInCellsZ(),
                                   origin, exploration_costmap_->getResolution()), "fs_upload_grid");
        if (arrival_info_limits_set_) ctx_->check(fs_set_arrival_limits(ctx_->get(), max_arrival_info_gt_, min_arrival_info_gt_), "fs_set_arrival_limits");
    }

    // The window [min_i, max_i) x [min_j, max_j) a costmap update cycle rewrote (the bounds every layer's updateCosts is called
    // with, DEP/src/nav2_plugins/lethal_marker.cpp:305-306): sent straight from the live character map, strides passed on.
    // The map must still have the staged shape, origin and resolution — otherwise take a new snapshot (updateCostmap).
    void updateCostmapWindow(int min_i, int min_j, int max_i, int max_j)
    {
        std::lock_guard<std::mutex> lock(exploration_costmap_->getMutex());
        const int64_t size_x = (int64_t)exploration_costmap_->getSizeInCellsX(), size_y = (int64_t)exploration_costmap_->getSizeInCellsY();
        ctx_->check(fs_update_grid_region(ctx_->get(), min_i, min_j, 0, max_i - min_i, max_j - min_j, (int32_t)exploration_costmap_->getSizeInCellsZ(),
                                          exploration_costmap_->getCharMap() + (int64_t)min_j * size_x + min_i, size_x, size_x * size_y),
                    "fs_update_grid_region");
    }

    // DEP/src/CostCalculator.cpp:23-121 for one frontier.
    void setArrivalInformationForFrontier(FrontierPtr &frontier, std::vector<double> &polygon_xy_min_max)
    {
        std::vector<FrontierPtr> one{frontier};
        setArrivalInformationForFrontiers(one, polygon_xy_min_max);
    }

    // The same for a whole list in one GPU call (replaces the serial loop of FrontierCostsManager.cpp:74-119).
    void setArrivalInformationForFrontiers(std::vector<FrontierPtr> &frontiers, std::vector<double> &polygon_xy_min_max)
    {
        if (polygon_xy_min_max.size() >= 4) setPolygon(polygon_xy_min_max);
        const int32_t n = (int32_t)frontiers.size();
        if (n == 0) return;
        std::vector<double> goal(3 * (size_t)n);
        std::vector<int32_t> fsize(n), arrival(n), argmax(n), status(n);
        std::vector<uint8_t> ach_in(n), ach(n);
        std::vector<double> yaw(n);
        for (int32_t i = 0; i < n; ++i) {
            const Point &g = frontiers[i]->getGoalPoint();
            goal[3 * i] = g.x; goal[3 * i + 1] = g.y; goal[3 * i + 2] = g.z;
            fsize[i] = frontiers[i]->getSize();
            ach_in[i] = frontiers[i]->isAchievable();
        }
        ctx_->check(fs_score_arrival(ctx_->get(), n, goal.data(), fsize.data(), nullptr, ach_in.data(), nullptr, arrival.data(),
                                     argmax.data(), yaw.data(), ach.data(), status.data()), "fs_score_arrival");
        for (int32_t i = 0; i < n; ++i) {
            frontiers[i]->setArrivalInformation((double)arrival[i]);      // :52 / :112
            frontiers[i]->setGoalOrientation(yaw[i]);                      // :53 / :119
            frontiers[i]->setAchievability(ach[i] != 0);                   // :78-82, :114-118
        }
    }

    // DEP/src/CostCalculator.cpp:123-191 — cached after the first success, returns 0.0 afterwards.
    double setMaxArrivalInformation()
    {
        if (arrival_info_limits_set_) return 0.0;
        double max_value = 0, max_gt = 0, min_gt = 0;
        ctx_->check(fs_max_arrival(ctx_->get(), &max_value, &max_gt, &min_gt), "fs_max_arrival");
        if (max_value == 0.0) return 0;                                     // (0,0) off the map: limits stay unset
        arrival_info_limits_set_ = true;
        max_arrival_info_gt_ = max_gt;
        min_arrival_info_gt_ = min_gt;
        return max_value;
    }

    // CostCalculator.cpp:446-484 — the stand-in planner (the roadmap A* is out of scope).
    void setPlanForFrontierEuclidean(Pose start_pose_w, FrontierPtr &goal_point_w)
    {
        const double dmax = std::numeric_limits<double>::max();
        auto unreachable = [&]() {
            goal_point_w->setAchievability(false);
            goal_point_w->setPathLength(dmax); goal_point_w->setPathLengthInM(dmax); goal_point_w->setPathHeading(dmax);
            goal_point_w->setFisherInformation(0);
        };
        if (goal_point_w->isAchievable() == false) { unreachable(); return; }
        const Point &g = goal_point_w->getGoalPoint();
        const double length = std::sqrt(std::pow(start_pose_w.position.x - g.x, 2) + std::pow(start_pose_w.position.y - g.y, 2));
        if (length < 0.5) { unreachable(); return; }
        goal_point_w->setAchievability(true);
        const Quaternion &q = start_pose_w.orientation;
        double robot_yaw = std::atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z));
        if (robot_yaw < 0) robot_yaw += M_PI * 2;
        double goal_yaw = std::atan2(g.y - start_pose_w.position.y, g.x - start_pose_w.position.x);
        if (goal_yaw < 0) goal_yaw += M_PI * 2;
        double path_heading = std::abs(robot_yaw - goal_yaw);
        if (path_heading > M_PI) path_heading = (2 * M_PI) - path_heading;
        goal_point_w->setPathLength(length); goal_point_w->setPathLengthInM(length); goal_point_w->setPathHeading(path_heading);
        goal_point_w->setFisherInformation(0.0);
    }

    // CostCalculator.cpp:512-520 / CostCalculator.hpp:90-116
    void recomputeNormalizationFactors(FrontierPtr &frontier)
    {
        if (!frontier->isAchievable()) return;
        min_traversable_distance = std::min(min_traversable_distance, frontier->getPathLength());
        max_traversable_distance = std::max(max_traversable_distance, frontier->getPathLength());
        min_arrival_info_per_frontier = std::min(min_arrival_info_per_frontier, frontier->getArrivalInformation());
        max_arrival_info_per_frontier = std::max(max_arrival_info_per_frontier, frontier->getArrivalInformation());
    }
    double getMinPlanDistance() { return min_traversable_distance; }
    double getMaxPlanDistance() { return max_traversable_distance; }
    double getMinArrivalInformation() { return min_arrival_info_per_frontier; }
    double getMaxArrivalInformation() { return max_arrival_info_gt_; }      // sic: the *_gt_ value (CostCalculator.hpp:105-108)
    double getMinArrivalInformationGT() { return min_arrival_info_gt_; }
    void reset()
    {
        min_traversable_distance = std::numeric_limits<double>::max(); max_traversable_distance = -1.0;
        min_arrival_info_per_frontier = std::numeric_limits<double>::max(); max_arrival_info_per_frontier = -1.0;
    }
    std::shared_ptr<ScoringContext> context() const { return ctx_; }
    double deltaTheta() const { return DELTA_THETA; }
    double cameraFov() const { return CAMERA_FOV; }
    // polygon_xy_min_max of the current call (CostAssigner.cpp:148-165): a change re-stages the ray parameters
    void setPolygon(const std::vector<double> &p)
    {
        bool changed = false;
        for (int i = 0; i < 4; ++i) { changed |= (polygon_[i] != p[i]); polygon_[i] = p[i]; }
        if (changed) pushParams();
    }

private:
    void pushParams()
    {
        fs_ray_params p{};
        p.max_camera_depth = MAX_CAMERA_DEPTH; p.delta_theta = DELTA_THETA; p.camera_fov = CAMERA_FOV; p.robot_radius = robot_radius_;
        p.n_rays = 0; p.n_elev = 1; p.elev[0] = 0.0;
        p.obst_min = 240; p.obst_max = 254; p.trace_min = 255; p.trace_max = 255;    // CostCalculator.cpp:40
        p.factor_max = 1.2; p.factor_min = factor_min_;                              // :186-188
        for (int i = 0; i < 4; ++i) p.polygon[i] = polygon_[i];
        ctx_->check(fs_set_ray_params(ctx_->get(), &p), "fs_set_ray_params");
        if (arrival_info_limits_set_) ctx_->check(fs_set_arrival_limits(ctx_->get(), max_arrival_info_gt_, min_arrival_info_gt_), "fs_set_arrival_limits");
    }

    std::shared_ptr<ScoringContext> ctx_;
    std::shared_ptr<Costmap2D> exploration_costmap_;
    double MAX_CAMERA_DEPTH, DELTA_THETA, CAMERA_FOV, robot_radius_, factor_min_;
    double polygon_[4];
    double min_traversable_distance = std::numeric_limits<double>::max(), max_traversable_distance = 0.0;
    double min_arrival_info_per_frontier = std::numeric_limits<double>::max(), max_arrival_info_per_frontier = 0.0;
    double max_arrival_info_gt_ = 0.0, min_arrival_info_gt_ = 0.0;
    bool arrival_info_limits_set_ = false;
};

class FrontierCostsManager
{
public:
    using Planner = std::function<void(const Pose &start_pose_w, FrontierPtr &frontier)>;

    // DEP/src/FrontierCostsManager.cpp:6-23 — parameters frontierCostsManager/{alpha,beta,vx_max,wz_max}.
    FrontierCostsManager(std::shared_ptr<ScoringContext> ctx, std::shared_ptr<Costmap2D> costmap,
                         double alpha = 0.25, double beta = 1.0, double vx_max = 0.5, double wz_max = 0.5)
        : alpha_(alpha), beta_(beta), max_vx_(vx_max), max_wx_(wz_max),
          costCalculator_(std::make_shared<FrontierCostCalculator>(std::move(ctx), std::move(costmap))) {}

    void setPlanner(Planner p) { planner_ = std::move(p); }
    std::shared_ptr<FrontierCostCalculator> getCostCalculator() { return costCalculator_; }

    // DEP/src/FrontierCostsManager.cpp:47-213.
    bool assignCosts(std::vector<FrontierPtr> &frontier_list, std::vector<double> polygon_xy_min_max, Pose start_pose_w)
    {
        costCalculator_->reset();                               // :51
        costCalculator_->setMaxArrivalInformation();            // :52
        if (frontier_list.size() == 0) return false;            // :55-59
        if (polygon_xy_min_max.size() <= 0) return false;       // :61-65
        for (size_t i = 0; i < frontier_list.size(); ++i)       // :25-45,69-72
            for (size_t j = i + 1; j < frontier_list.size(); ++j)
                if (frontier_list[i] == frontier_list[j]) throw std::runtime_error("Duplicate frontiers found.");

        const double dmax = std::numeric_limits<double>::max();
        std::vector<FrontierPtr> live;
        std::vector<uint8_t> black(frontier_list.size(), 0);
        {
            std::lock_guard<std::mutex> lock(blacklist_mutex_);
            for (size_t i = 0; i < frontier_list.size(); ++i) {
                auto &frontier = frontier_list[i];
                if (frontier_blacklist_.count(frontier.get()) > 0) {   // :77-86
                    frontier->setArrivalInformation(0.0); frontier->setGoalOrientation(0.0); frontier->setFisherInformation(0.0);
                    frontier->setPathLength(dmax); frontier->setPathLengthInM(dmax); frontier->setWeightedCost(dmax);
                    black[i] = 1;
                    continue;
                }
                live.push_back(frontier);
            }
        }
        costCalculator_->setArrivalInformationForFrontiers(live, polygon_xy_min_max);   // :96 batched
        for (auto &frontier : live) {                                                    // :98-109
            if (planner_) planner_(start_pose_w, frontier);
            else costCalculator_->setPlanForFrontierEuclidean(start_pose_w, frontier);
            costCalculator_->recomputeNormalizationFactors(frontier);                    // :118
        }
        // U1 + weighted cost on the GPU (:126-205)
        const int32_t n = (int32_t)frontier_list.size();
        std::vector<fs_record> rec(n);
        std::vector<double> plen(n), phead(n), cost(n), au(n), du(n);
        for (int32_t i = 0; i < n; ++i) {
            auto &f = frontier_list[i];
            rec[i] = fs_record{};
            rec[i].arrival = (int32_t)f->getArrivalInformation();
            rec[i].flags = (f->isAchievable() ? FS_FLAG_ACHIEVABLE : 0u);
            plen[i] = f->getPathLength();
            phead[i] = black[i] ? 0.0 : f->getPathHeading();
        }
        auto ctx = costCalculator_->context();
        const int rc = fs_rank_candidates(ctx->get(), n, rec.data(), black.data(), plen.data(), phead.data(), alpha_, beta_, max_vx_, max_wx_,
                                          cost.data(), au.data(), du.data(), nullptr);
        if (rc == FS_E_RANGE) throw std::runtime_error("Cost out of bounds");   // :148-149,173-174
        ctx->check(rc, "fs_rank_candidates");
        for (int32_t i = 0; i < n; ++i) {
            frontier_list[i]->setWeightedCost(cost[i]);
            frontier_list[i]->setCost("arrival_gain_utility", au[i]);
            frontier_list[i]->setCost("distance_utility", du[i]);
        }
        return true;
    }

    // The same contract as assignCosts through ONE device call (fs_get_frontier_costs): the planner runs FIRST — on every live
    // frontier, as if achievable — and what it decides (a frontier it cannot reach: achievable = false) enters the call as
    // achievable_in, which the arrival step can only clear further; the final flag is the AND of the same conditions in either
    // order, and the path columns of a frontier that ends up unachievable are never read.  Costs, utilities, arrival information,
    // orientation and achievability come out bit for bit as from assignCosts (tests/test_host_mirror.py); what differs is that the
    // planner also plans frontiers the reference would have skipped (FrontierCostsManager.cpp:88-91) — spare work, same result.
    bool assignCostsFused(std::vector<FrontierPtr> &frontier_list, std::vector<double> polygon_xy_min_max, Pose start_pose_w)
    {
        costCalculator_->reset();
        costCalculator_->setMaxArrivalInformation();
        if (frontier_list.size() == 0) return false;
        if (polygon_xy_min_max.size() <= 0) return false;
        for (size_t i = 0; i < frontier_list.size(); ++i)
            for (size_t j = i + 1; j < frontier_list.size(); ++j)
                if (frontier_list[i] == frontier_list[j]) throw std::runtime_error("Duplicate frontiers found.");
        if (polygon_xy_min_max.size() >= 4) costCalculator_->setPolygon(polygon_xy_min_max);
        const double dmax = std::numeric_limits<double>::max();
        const int32_t n = (int32_t)frontier_list.size();
        std::vector<uint8_t> black(n, 0), ach_in(n, 1);
        std::vector<double> goal(3 * (size_t)n), plen(n, 0.0), phead(n, 0.0), cost(n), au(n), du(n);
        std::vector<int32_t> fsize(n);
        {
            std::lock_guard<std::mutex> lock(blacklist_mutex_);
            for (int32_t i = 0; i < n; ++i) black[i] = frontier_blacklist_.count(frontier_list[i].get()) > 0;
        }
        for (int32_t i = 0; i < n; ++i) {
            auto &f = frontier_list[i];
            const Point &g = f->getGoalPoint();
            goal[3 * i] = g.x; goal[3 * i + 1] = g.y; goal[3 * i + 2] = g.z;
            fsize[i] = f->getSize();
            if (black[i]) continue;
            if (planner_) planner_(start_pose_w, f);
            else costCalculator_->setPlanForFrontierEuclidean(start_pose_w, f);
            ach_in[i] = f->isAchievable();
            if (ach_in[i]) { plen[i] = f->getPathLength(); phead[i] = f->getPathHeading(); }
        }
        std::vector<fs_record> rec(n);
        auto ctx = costCalculator_->context();
        const int rc = fs_get_frontier_costs(ctx->get(), n, goal.data(), fsize.data(), black.data(), ach_in.data(), plen.data(), phead.data(),
                                             alpha_, beta_, max_vx_, max_wx_, /*with_fisher_information=*/0,
                                             rec.data(), cost.data(), au.data(), du.data(), nullptr);
        if (rc == FS_E_RANGE) throw std::runtime_error("Cost out of bounds");   // :148-149,173-174
        ctx->check(rc, "fs_get_frontier_costs");
        for (int32_t i = 0; i < n; ++i) {
            auto &f = frontier_list[i];
            if (black[i]) {                                                       // :77-86
                f->setArrivalInformation(0.0); f->setGoalOrientation(0.0); f->setFisherInformation(0.0);
                f->setPathLength(dmax); f->setPathLengthInM(dmax); f->setWeightedCost(dmax);
            } else {
                f->setArrivalInformation((double)rec[i].arrival);
                f->setGoalOrientation(((double)rec[i].argmax * costCalculator_->deltaTheta()) + costCalculator_->cameraFov() / 2);   // CostCalculator.cpp:119, in double
                if (FS_RECORD_STATUS(rec[i].flags) != FS_STATUS_OK) f->setGoalOrientation(0.0);                                    // :52-54
                const bool achievable = (rec[i].flags & FS_FLAG_ACHIEVABLE) != 0;
                f->setAchievability(achievable);
                if (!achievable) { f->setPathLength(dmax); f->setPathLengthInM(dmax); f->setPathHeading(dmax); f->setFisherInformation(0); }
                costCalculator_->recomputeNormalizationFactors(f);
            }
            f->setWeightedCost(cost[i]);
            f->setCost("arrival_gain_utility", au[i]);
            f->setCost("distance_utility", du[i]);
        }
        return true;
    }

    // :215-222
    void setFrontierBlacklist(std::vector<FrontierPtr> &blacklist)
    {
        std::lock_guard<std::mutex> lock(blacklist_mutex_);
        for (auto &frontier : blacklist) frontier_blacklist_[frontier.get()] = true;
    }

private:
    double alpha_, beta_, max_vx_, max_wx_;
    std::shared_ptr<FrontierCostCalculator> costCalculator_;
    Planner planner_;
    std::mutex blacklist_mutex_;
    std::unordered_map<const Frontier *, bool> frontier_blacklist_;
};

// FrontierRoadMap::isConnectable (DEP/src/planners/FrontierRoadmap.cpp:716-737) for a batch of node pairs: visitor
// (253,254,0,255); connectable iff the trace succeeded, hit no obstacle and saw at most
// 0.3 * RADIUS_TO_DECIDE_EDGES / resolution unknown cells.
inline std::vector<bool> isConnectable(ScoringContext &ctx, const Costmap2D &costmap,
                                       const std::vector<std::pair<FrontierPtr, FrontierPtr>> &pairs,
                                       double max_connection_length, double radius_to_decide_edges)
{
    const int32_t n = (int32_t)pairs.size();
    std::vector<double> s(3 * (size_t)n), e(3 * (size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        const Point &a = pairs[i].first->getGoalPoint(), &b = pairs[i].second->getGoalPoint();
        s[3 * i] = a.x; s[3 * i + 1] = a.y; s[3 * i + 2] = a.z; e[3 * i] = b.x; e[3 * i + 1] = b.y; e[3 * i + 2] = b.z;
    }
    std::vector<uint8_t> ok(n), hit(n);
    std::vector<int32_t> traced(n), unknown(n), all(n);
    const unsigned int max_length = max_connection_length / costmap.getResolution();
    ctx.check(fs_trace_segments(ctx.get(), n, s.data(), e.data(), (double)max_length, 253, 254, 0, 255, ok.data(), traced.data(),
                                hit.data(), unknown.data(), all.data()), "fs_trace_segments");
    std::vector<bool> out(n);
    for (int32_t i = 0; i < n; ++i)
        out[i] = ok[i] && !hit[i] && !(unknown[i] > radius_to_decide_edges / costmap.getResolution() * 0.3);
    return out;
}

// DEP/include/.../Frontier.hpp:151-161
inline size_t generateUID(const FrontierPtr &output)
{
    std::hash<double> hash_fn;
    return hash_fn(output->getGoalPoint().x) ^ (hash_fn(output->getGoalPoint().y) << 1);
}

// DEP/include/.../FrontierSearch.hpp:31-127 — the producer of the candidate list, same signature:
//     std::vector<FrontierPtr> FrontierSearch::searchFrom(geometry_msgs::msg::Point position)   (FrontierSearch.cpp:21)
// What searchFrom computes splits into a part that does not depend on the order of its queues and one that does:
//   * which cells are frontier cells, which 8-connected components of them the search reaches, their sizes — ONE
//     fs_frontier_clusters call on the GPU (stencil + union-find labelling; bit-exact against the reference's search);
//   * per component, the reference's own tail on the host (buildNewFrontier, :98-216): a breadth-first walk from the
//     component's seed cell in nhood8 order that cuts the component into pieces of max_frontier_cluster_size + 1 cells
//     in the order of its queue, and per piece the angular-median goal point (getCentroidOfCells, SortByMedianFunctor,
//     std::sort, middle element) and setSize.  That tail is inherently serial and tiny (it touches frontier cells only).
// The seed of a component and the order of the output list follow from the order of the reference's OUTER search:
//   SeedOrder::Reference  the outer breadth-first search is replayed on the host (map cells < 254 inside the search
//                         radius, nhood4 order, nearestFreeCell start) only to learn in which order, and at which cell, it
//                         meets the components — output identical to the reference's, record for record;
//   SeedOrder::Nearest    (default) no outer search: components in ascending label order, seed = the component's cell
//                         nearest the robot (ties: smaller index).  Same components, same number and sizes of records; a
//                         goal point may be a different cell of its piece than the reference's queue order would pick.
class FrontierSearch
{
public:
    enum class SeedOrder { Nearest, Reference };
    struct Cluster
    {
        int label = -1, size = 0;                       // label: smallest cell index of the component
        Point centroid;                                 // mean of the cell centres (mapToWorld)
    };
    FrontierSearch(std::shared_ptr<ScoringContext> ctx, std::shared_ptr<Costmap2D> costmap, int min_frontier_cluster_size = 1,
                   int max_frontier_cluster_size = 20, double max_frontier_distance = 50.0, int lethal_threshold = 160,
                   SeedOrder seed_order = SeedOrder::Nearest)
        : ctx_(std::move(ctx)), costmap_(std::move(costmap)), min_frontier_cluster_size_(min_frontier_cluster_size),
          max_frontier_cluster_size_(max_frontier_cluster_size), max_frontier_distance_(max_frontier_distance),
          original_search_distance_(max_frontier_distance), lethal_threshold_(lethal_threshold), seed_order_(seed_order) {}
    void reset() { every_frontier_list.clear(); }
    void incrementSearchDistance(double value) { max_frontier_distance_ += value; }
    void resetSearchDistance() { max_frontier_distance_ = original_search_distance_; }
    std::vector<std::vector<double>> getAllFrontiers() { return every_frontier_list; }
    void setSeedOrder(SeedOrder o) { seed_order_ = o; }
    const std::vector<Cluster> &lastClusters() const { return clusters_; }   // the GPU stage's output of the last search

    // FrontierSearch.cpp:21-96.  The grid must have been staged (FrontierCostCalculator::updateCostmap).
    std::vector<FrontierPtr> searchFrom(Point position)
    {
        std::vector<FrontierPtr> frontier_list;
        clusters_.clear();
        unsigned int mx, my;
        if (!costmap_->worldToMap(position.x, position.y, mx, my)) return frontier_list;     // :26-33
        const unsigned int nx = costmap_->getSizeInCellsX(), ny = costmap_->getSizeInCellsY();
        labels_.assign((size_t)nx * ny, -1);
        std::vector<fs_frontier_cluster> raw((size_t)nx * ny / 2 + 1);
        int32_t n = 0;
        int64_t n_cells = 0;
        const double xy[2] = {position.x, position.y};
        ctx_->check(fs_frontier_clusters(ctx_->get(), xy, lethal_threshold_, max_frontier_distance_, max_frontier_cluster_size_,
                                         labels_.data(), (int32_t)raw.size(), raw.data(), &n, &n_cells), "fs_frontier_clusters");
        clusters_.resize((size_t)n);
        for (int32_t k = 0; k < n; ++k) {
            clusters_[k].label = raw[k].label; clusters_[k].size = raw[k].size;
            clusters_[k].centroid.x = raw[k].centroid_x; clusters_[k].centroid.y = raw[k].centroid_y;
        }
        claimed_.assign((size_t)nx * ny, 0);
        const unsigned int pos = costmap_->getIndex(mx, my);
        if (seed_order_ == SeedOrder::Reference) {
            replayOuterSearch(position, pos, frontier_list);
        } else {
            // one pass over the labels: the cell of every component nearest the robot's cell
            std::unordered_map<int, std::pair<long long, unsigned int>> best;
            for (unsigned int idx = 0; idx < nx * ny; ++idx) {
                const int l = labels_[idx];
                if (l < 0) continue;
                const long long dx = (long long)(idx % nx) - (long long)mx, dy = (long long)(idx / nx) - (long long)my;
                const long long d2 = dx * dx + dy * dy;
                auto it = best.find(l);
                if (it == best.end() || d2 < it->second.first) best[l] = {d2, idx};
            }
            for (const Cluster &c : clusters_) emitComponent(best[c.label].second, frontier_list);
        }
        return frontier_list;
    }

private:
    // the neighbourhoods in the order the reference's helpers produce them (DEP/src/Helpers.cpp:185-255): left, right, up,
    // down; then the four diagonals (-1-sx, -1+sx, +1-sx, +1+sx)
    int neighbours(unsigned int idx, bool eight, unsigned int out[8]) const
    {
        const unsigned int sx = costmap_->getSizeInCellsX(), sy = costmap_->getSizeInCellsY();
        if (idx > sx * sy - 1) return 0;
        const bool l = idx % sx > 0, r = idx % sx < sx - 1, u = idx >= sx, d = idx < sx * (sy - 1);
        int k = 0;
        if (l) out[k++] = idx - 1;
        if (r) out[k++] = idx + 1;
        if (u) out[k++] = idx - sx;
        if (d) out[k++] = idx + sx;
        if (eight) {
            if (l && u) out[k++] = idx - 1 - sx;
            if (l && d) out[k++] = idx - 1 + sx;
            if (r && u) out[k++] = idx + 1 - sx;
            if (r && d) out[k++] = idx + 1 + sx;
        }
        return k;
    }
    bool unclaimedFrontierCell(unsigned int idx) const { return labels_[idx] >= 0 && !claimed_[idx]; }   // isNewFrontierCell given the GPU's labels
    std::pair<double, double> cellCentre(unsigned int idx) const
    {
        unsigned int cx, cy;
        double wx, wy;
        costmap_->indexToCells(idx, cx, cy);
        costmap_->mapToWorld(cx, cy, wx, wy);
        return {wx, wy};
    }

    // FrontierSearch.hpp:84-127: mean of the cells, pushed sideways by 2 sqrt2 cells when some cell lies within three cells of
    // it (so that the angular order below is taken about a point off the cluster's spine)
    std::pair<double, double> centroidOfCells(const std::vector<std::pair<double, double>> &cells, double offset) const
    {
        double sx = 0, sy = 0;
        for (const auto &c : cells) { sx += c.first; sy += c.second; }
        double cx = sx / cells.size(), cy = sy / cells.size();
        bool near_spine = false;
        double spread_x = 0, spread_y = 0;
        for (const auto &c : cells) {
            if (std::sqrt(std::pow(c.first - cx, 2) + std::pow(c.second - cy, 2)) < costmap_->getResolution() * 3) near_spine = true;
            spread_x += std::abs(c.first - cx);
            spread_y += std::abs(c.second - cy);
        }
        if (spread_x > spread_y && near_spine) cy -= offset;
        if (spread_x < spread_y && near_spine) cx -= offset;
        return {cx, cy};
    }

    // one Frontier record of a finished piece (FrontierSearch.cpp:156-174 / :191-208): goal = the middle element of the cells
    // sorted by angle about the centroid (SortByMedianFunctor, FrontierSearch.hpp:156-181 — angles in [0, 2 pi), except that a
    // first-quadrant angle never sorts before a fourth-quadrant one; not a strict weak order, so the result is whatever
    // std::sort makes of it: the same library call as the reference)
    FrontierPtr finishPiece(std::vector<std::pair<double, double>> &cells, int size) const
    {
        const auto c = centroidOfCells(cells, costmap_->getResolution() * 1.414 * 2);
        auto angle = [&](const std::pair<double, double> &p) {
            double a = std::atan2(p.second - c.second, p.first - c.first);
            return a < 0 ? a + (2 * M_PI) : a;
        };
        std::sort(cells.begin(), cells.end(), [&](const std::pair<double, double> &a, const std::pair<double, double> &b) {
            const double aa = angle(a), ab = angle(b);
            if (0 <= aa && aa <= M_PI / 2 && 3 * M_PI / 2 <= ab && ab <= 2 * M_PI) return false;
            if (0 <= ab && ab <= M_PI / 2 && 3 * M_PI / 2 <= aa && aa <= 2 * M_PI) return true;
            return aa < ab;
        });
        const auto goal = cells[static_cast<int>(cells.size() / 2)];
        auto out = std::make_shared<Frontier>();
        out->setGoalPoint(goal.first, goal.second);
        out->setSize(size);
        out->setUID(generateUID(out));
        cells.clear();
        return out;
    }

    // buildNewFrontier (FrontierSearch.cpp:98-216) over the component of `seed`, then searchFrom's size filter (:81)
    void emitComponent(unsigned int seed, std::vector<FrontierPtr> &frontier_list)
    {
        int count = 1;
        std::vector<std::pair<double, double>> piece{cellCentre(seed)};
        every_frontier_list.push_back({piece[0].first, piece[0].second});
        claimed_[seed] = 1;
        std::vector<FrontierPtr> made;
        std::queue<unsigned int> walk;
        walk.push(seed);
        unsigned int nb[8];
        while (!walk.empty()) {
            const unsigned int idx = walk.front();
            walk.pop();
            const int k = neighbours(idx, true, nb);
            for (int j = 0; j < k; ++j) {
                if (!unclaimedFrontierCell(nb[j])) continue;
                claimed_[nb[j]] = 1;
                const auto w = cellCentre(nb[j]);
                every_frontier_list.push_back({w.first, w.second});
                piece.push_back(w);
                ++count;
                walk.push(nb[j]);
                if (count > max_frontier_cluster_size_) {              // a full piece: max + 1 cells
                    made.push_back(finishPiece(piece, count));
                    count = 0;
                }
            }
        }
        if (count > min_frontier_cluster_size_) made.push_back(finishPiece(piece, count));
        for (auto &f : made)
            if (f->getSize() > min_frontier_cluster_size_) frontier_list.push_back(f);
    }

    // nearestFreeCell (DEP/src/Helpers.cpp:285-329; its `unsigned char val` parameter truncates the threshold)
    bool nearestFreeCell(unsigned int &result, unsigned int start) const
    {
        const unsigned int total = costmap_->getSizeInCellsX() * costmap_->getSizeInCellsY();
        if (start >= total) return false;
        const unsigned char val = (unsigned char)lethal_threshold_;
        std::vector<bool> seen(total, false);
        std::queue<unsigned int> q;
        q.push(start);
        seen[start] = true;
        unsigned int nb[8];
        while (!q.empty()) {
            const unsigned int idx = q.front();
            q.pop();
            if (costmap_->getCost(idx) < val) { result = idx; return true; }
            const int k = neighbours(idx, true, nb);
            for (int j = 0; j < k; ++j)
                if (!seen[nb[j]]) { seen[nb[j]] = true; q.push(nb[j]); }
        }
        return false;
    }

    // SeedOrder::Reference — the outer search of FrontierSearch.cpp:44-94, replayed for its ORDER only: frontier-cell
    // membership comes from the GPU's labels
    void replayOuterSearch(const Point &position, unsigned int pos, std::vector<FrontierPtr> &frontier_list)
    {
        const unsigned int total = costmap_->getSizeInCellsX() * costmap_->getSizeInCellsY();
        std::vector<bool> visited(total, false);
        std::queue<unsigned int> q;
        unsigned int clear = 0;
        q.push(nearestFreeCell(clear, pos) ? clear : pos);
        visited[q.front()] = true;
        const double reach = max_frontier_distance_ + (max_frontier_cluster_size_ * costmap_->getResolution() * 1.414);
        unsigned int nb[8];
        while (!q.empty()) {
            const unsigned int idx = q.front();
            q.pop();
            const int k = neighbours(idx, false, nb);
            for (int j = 0; j < k; ++j) {
                const unsigned int nbr = nb[j];
                if (costmap_->getCost(nbr) < 254 && !visited[nbr]) {
                    visited[nbr] = true;
                    const auto w = cellCentre(nbr);
                    if (std::sqrt(std::pow(position.x - w.first, 2) + std::pow(position.y - w.second, 2)) < reach) q.push(nbr);
                } else if (unclaimedFrontierCell(nbr)) {
                    emitComponent(nbr, frontier_list);
                }
            }
        }
    }

    std::shared_ptr<ScoringContext> ctx_;
    std::shared_ptr<Costmap2D> costmap_;
    int min_frontier_cluster_size_, max_frontier_cluster_size_;
    double max_frontier_distance_, original_search_distance_;
    int lethal_threshold_;
    SeedOrder seed_order_;
    std::vector<std::vector<double>> every_frontier_list;
    std::vector<int32_t> labels_;
    std::vector<uint8_t> claimed_;
    std::vector<Cluster> clusters_;
};

// One in-process scorer over several GPUs (fs_multi_* of the C ABI): what ProcessFrontierCostsBT calls from the behaviour-tree
// thread (DEP/src/ExplorationBT.cpp:376-410) stays ONE object and ONE call on a multi-GPU node.  Staging is broadcast to every
// device; scoreFrontiers cuts the list into contiguous blocks, runs them side by side and writes the results into the
// frontiers in list order — arrival information, goal orientation, achievability (CostCalculator.cpp:112-119) and the Fisher
// information at the pose (goal, best yaw) (isPoseSafe(Point, Point), FisherInfoManager.cpp:31-37).
class ShardedScorer
{
public:
    explicit ShardedScorer(const std::vector<int> &devices)
    {
        const int rc = fs_multi_create(devices.data(), (int)devices.size(), &m_);
        if (rc != FS_OK) throw std::runtime_error("fs_multi_create failed: no MI355X (gfx950) under one of the device ordinals; there is no CPU fallback");
    }
    ~ShardedScorer() { fs_multi_destroy(m_); }
    ShardedScorer(const ShardedScorer &) = delete;
    ShardedScorer &operator=(const ShardedScorer &) = delete;
    fs_multi *get() const { return m_; }
    int numDevices() const { return fs_multi_num_devices(m_); }
    void check(int rc, const char *what) const
    {
        if (rc != FS_OK) throw std::runtime_error(std::string(what) + ": " + fs_multi_last_error(m_));
    }
    void setRayParams(const fs_ray_params &p) { check(fs_multi_set_ray_params(m_, &p), "fs_multi_set_ray_params"); }
    // the snapshot point: taken under the costmap's mutex, sent to every device
    void updateCostmap(Costmap2D &costmap)
    {
        std::lock_guard<std::mutex> lock(costmap.getMutex());
        const double origin[3] = {costmap.getOriginX(), costmap.getOriginY(), costmap.getOriginZ()};
        check(fs_multi_upload_grid(m_, costmap.getCharMap(), (int32_t)costmap.getSizeInCellsX(), (int32_t)costmap.getSizeInCellsY(),
                                   (int32_t)costmap.getSizeInCellsZ(), origin, costmap.getResolution()), "fs_multi_upload_grid");
    }
    // ... and the window a costmap update cycle rewrote, on every device (fs_multi_update_grid_region)
    void updateCostmapWindow(Costmap2D &costmap, int min_i, int min_j, int max_i, int max_j)
    {
        std::lock_guard<std::mutex> lock(costmap.getMutex());
        const int64_t size_x = (int64_t)costmap.getSizeInCellsX(), size_y = (int64_t)costmap.getSizeInCellsY();
        check(fs_multi_update_grid_region(m_, min_i, min_j, 0, max_i - min_i, max_j - min_j, (int32_t)costmap.getSizeInCellsZ(),
                                          costmap.getCharMap() + (int64_t)min_j * size_x + min_i, size_x, size_x * size_y), "fs_multi_update_grid_region");
    }
    void setLandmarks(const std::vector<float> &xyz) { check(fs_multi_upload_landmarks(m_, xyz.data(), (int32_t)(xyz.size() / 3)), "fs_multi_upload_landmarks"); }
    void generateLookupTable() { check(fs_multi_lookup_generate(m_, nullptr), "fs_multi_lookup_generate"); }
    void loadLookupTable(const std::string &path)
    {
        if (fs_multi_lookup_load(m_, path.c_str()) != FS_OK) throw std::runtime_error("Cannot load lookup table. Does it exist in the path?");
    }
    void setVisibility(double max_dist, double max_angle)
    {
        const fs_fim_params p{max_dist, max_angle};
        check(fs_multi_set_fim_params(m_, &p), "fs_multi_set_fim_params");
    }
    double setMaxArrivalInformation()
    {
        double max_value = 0, max_gt = 0, min_gt = 0;
        check(fs_multi_max_arrival(m_, &max_value, &max_gt, &min_gt), "fs_multi_max_arrival");
        return max_value;
    }
    // order-preserving; `blacklisted` (or empty) as in FrontierCostsManager.cpp:77-86
    std::vector<fs_record> scoreFrontiers(std::vector<FrontierPtr> &frontiers, const std::vector<uint8_t> &blacklisted = {})
    {
        const int32_t n = (int32_t)frontiers.size();
        std::vector<fs_record> rec((size_t)n);
        if (n == 0) return rec;
        std::vector<double> goal(3 * (size_t)n);
        std::vector<int32_t> fsize((size_t)n);
        std::vector<uint8_t> ach_in((size_t)n);
        for (int32_t i = 0; i < n; ++i) {
            const Point &g = frontiers[i]->getGoalPoint();
            goal[3 * i] = g.x; goal[3 * i + 1] = g.y; goal[3 * i + 2] = g.z;
            fsize[i] = frontiers[i]->getSize();
            ach_in[i] = frontiers[i]->isAchievable();
        }
        check(fs_multi_score_candidates(m_, n, goal.data(), fsize.data(), blacklisted.empty() ? nullptr : blacklisted.data(),
                                        ach_in.data(), rec.data()), "fs_multi_score_candidates");
        for (int32_t i = 0; i < n; ++i) {
            frontiers[i]->setArrivalInformation((double)rec[i].arrival);
            frontiers[i]->setGoalOrientation((double)rec[i].yaw);
            frontiers[i]->setAchievability((rec[i].flags & FS_FLAG_ACHIEVABLE) != 0);
            frontiers[i]->setFisherInformation((double)rec[i].info_ref);
        }
        return rec;
    }

    // FrontierCostsManager::assignCosts' device half as ONE call over every device (fs_multi_get_frontier_costs): arrival information
    // on each block's device, the records gathered device to device onto the first GPU, U1 costs there, one transfer back.  The
    // planner has run first (FrontierCostsManager::assignCostsFused explains why that gives the reference's results): path_length /
    // path_heading as it set them, achievable_in = what it left achievable.  Sets the reference's fields on every frontier.
    void assignCosts(std::vector<FrontierPtr> &frontiers, const std::vector<uint8_t> &blacklisted, const std::vector<uint8_t> &achievable_in,
                     const std::vector<double> &path_length, const std::vector<double> &path_heading, double delta_theta, double camera_fov,
                     double alpha = 0.25, double beta = 1.0, double vx_max = 0.5, double wz_max = 0.5)
    {
        const int32_t n = (int32_t)frontiers.size();
        if (n == 0) return;
        const double dmax = std::numeric_limits<double>::max();
        std::vector<double> goal(3 * (size_t)n), cost((size_t)n), au((size_t)n), du((size_t)n);
        std::vector<int32_t> fsize((size_t)n);
        for (int32_t i = 0; i < n; ++i) {
            const Point &g = frontiers[i]->getGoalPoint();
            goal[3 * i] = g.x; goal[3 * i + 1] = g.y; goal[3 * i + 2] = g.z;
            fsize[i] = frontiers[i]->getSize();
        }
        std::vector<fs_record> rec((size_t)n);
        const int rc = fs_multi_get_frontier_costs(m_, n, goal.data(), fsize.data(), blacklisted.empty() ? nullptr : blacklisted.data(),
                                                   achievable_in.empty() ? nullptr : achievable_in.data(), path_length.data(), path_heading.data(),
                                                   alpha, beta, vx_max, wz_max, /*with_fisher_information=*/0, rec.data(), cost.data(), au.data(), du.data(), nullptr);
        if (rc == FS_E_RANGE) throw std::runtime_error("Cost out of bounds");   // FrontierCostsManager.cpp:148-149,173-174
        check(rc, "fs_multi_get_frontier_costs");
        for (int32_t i = 0; i < n; ++i) {
            auto &f = frontiers[i];
            if (!blacklisted.empty() && blacklisted[i]) {                         // :77-86
                f->setArrivalInformation(0.0); f->setGoalOrientation(0.0); f->setFisherInformation(0.0);
                f->setPathLength(dmax); f->setPathLengthInM(dmax); f->setWeightedCost(dmax);
            } else {
                f->setArrivalInformation((double)rec[i].arrival);
                f->setGoalOrientation(FS_RECORD_STATUS(rec[i].flags) == FS_STATUS_OK ? (double)rec[i].argmax * delta_theta + camera_fov / 2 : 0.0);   // CostCalculator.cpp:119 / :52-54
                f->setAchievability((rec[i].flags & FS_FLAG_ACHIEVABLE) != 0);
            }
            f->setWeightedCost(cost[i]);
            f->setCost("arrival_gain_utility", au[i]);
            f->setCost("distance_utility", du[i]);
        }
    }
    // FisherInformationManager::poseInformation over every device (fs_multi_score_fim, info_ref alone)
    void poseInformation(const std::vector<Pose> &poses, std::vector<float> &information)
    {
        std::vector<double> p7(poses.size() * 7);
        for (size_t i = 0; i < poses.size(); ++i) {
            const Pose &p = poses[i];
            const double v[7] = {p.position.x, p.position.y, p.position.z, p.orientation.x, p.orientation.y, p.orientation.z, p.orientation.w};
            for (int k = 0; k < 7; ++k) p7[7 * i + k] = v[k];
        }
        information.assign(poses.size(), 0.0f);
        check(fs_multi_score_fim(m_, (int32_t)poses.size(), p7.data(), information.data(), nullptr, nullptr, nullptr, nullptr, nullptr), "fs_multi_score_fim");
    }
    int gatherMode() const { return fs_multi_gather_mode(m_); }
    void setOption(const char *key, double value) { check(fs_multi_set_option(m_, key, value), "fs_multi_set_option"); }

private:
    fs_multi *m_ = nullptr;
};

// DEP/include/.../CostAssigner.hpp:43-59
struct GetFrontierCostsRequest
{
    PoseStamped start_pose;
    std::vector<FrontierPtr> frontier_list;
    std::vector<std::vector<double>> every_frontier;
    std::vector<FrontierPtr> prohibited_frontiers;
};
struct GetFrontierCostsResponse
{
    bool success = false;
    std::vector<FrontierPtr> frontier_list;
    std::vector<double> frontier_costs, frontier_distances, frontier_arrival_information, frontier_path_information;
};

class CostAssigner
{
public:
    CostAssigner(std::shared_ptr<ScoringContext> ctx, std::shared_ptr<Costmap2D> costmap)
        : costmap_(costmap), frontierCostsManager_(std::make_shared<FrontierCostsManager>(std::move(ctx), std::move(costmap))) {}

    // DEP/src/CostAssigner.cpp:121-167 — polygon -> bbox [minx, miny, maxx, maxy]; empty polygon = whole map (with the
    // reference's quirk of using getSizeInMeters, not origin + size, for the far corner).
    bool updateBoundaryPolygon(const std::vector<Point> &explore_boundary)
    {
        std::vector<Point> pts = explore_boundary;
        if (pts.empty()) {
            Point t;
            t.x = costmap_->getOriginX(); t.y = costmap_->getOriginY(); pts.push_back(t);
            t.y = costmap_->getSizeInMetersY(); pts.push_back(t);
            t.x = costmap_->getSizeInMetersX(); pts.push_back(t);
            t.y = costmap_->getOriginY(); pts.push_back(t);
        }
        double mnx = std::numeric_limits<double>::infinity(), mny = mnx, mxx = -mnx, mxy = -mnx;
        for (const auto &p : pts) {
            // the reference stores the polygon as Point32 (float) before taking min/max
            const double px = (double)(float)p.x, py = (double)(float)p.y;
            mnx = std::min(mnx, px); mny = std::min(mny, py); mxx = std::max(mxx, px); mxy = std::max(mxy, py);
        }
        polygon_xy_min_max_ = {mnx, mny, mxx, mxy};
        return true;
    }

    // DEP/src/CostAssigner.cpp:73-119
    bool getFrontierCosts(std::shared_ptr<GetFrontierCostsRequest> requestData, std::shared_ptr<GetFrontierCostsResponse> resultData)
    {
        frontierCostsManager_->setFrontierBlacklist(requestData->prohibited_frontiers);
        const bool costsResult = fused_ ? frontierCostsManager_->assignCostsFused(requestData->frontier_list, polygon_xy_min_max_, requestData->start_pose.pose)
                                        : frontierCostsManager_->assignCosts(requestData->frontier_list, polygon_xy_min_max_, requestData->start_pose.pose);
        if (costsResult == false) { resultData->success = false; return resultData->success; }
        resultData->success = true;
        resultData->frontier_list.clear(); resultData->frontier_costs.clear();
        resultData->frontier_distances.clear(); resultData->frontier_arrival_information.clear();
        for (auto &frontier : requestData->frontier_list) {
            resultData->frontier_list.push_back(frontier);
            resultData->frontier_costs.push_back(frontier->getWeightedCost());
            resultData->frontier_distances.push_back(frontier->getPathLengthInM());
            resultData->frontier_arrival_information.push_back(frontier->getArrivalInformation());
        }
        if (resultData->frontier_list != requestData->frontier_list) throw std::runtime_error("Lists are not SAME!");
        return resultData->success;
    }
    std::shared_ptr<FrontierCostsManager> getCostManagerPtr() { return frontierCostsManager_; }
    // true: the whole cost assignment as ONE device call (FrontierCostsManager::assignCostsFused); same results
    void setFused(bool on) { fused_ = on; }

private:
    bool fused_ = false;
    std::shared_ptr<Costmap2D> costmap_;
    std::vector<double> polygon_xy_min_max_;
    std::shared_ptr<FrontierCostsManager> frontierCostsManager_;
};
}  // namespace frontier_exploration

// ---- key-frame pose information (dead code in the reference) ------------------------------
namespace slam_msgs_lite
{
// the fields of slam_msgs::msg::MapData the function reads (map_data.graph.poses / poses_id, map_data.nodes[].id / word_pts)
struct KeyFrame { int id = 0; std::vector<frontier_exploration::Point> word_pts; };
struct MapGraph { std::vector<frontier_exploration::PoseStamped> poses; std::vector<int> poses_id; };
struct MapData { MapGraph graph; std::vector<KeyFrame> nodes; };
}  // namespace slam_msgs_lite

namespace frontier_exploration_information_affine
{
// Stages a MapData snapshot: graph pose i is paired with the node of the same id (getNodeDataAndOptTransform,
// deprecated/util.hpp:791-818); a pose without node data contributes no points.
inline void setMapData(frontier_exploration::ScoringContext &ctx, const slam_msgs_lite::MapData &map_data)
{
    std::unordered_map<int, const slam_msgs_lite::KeyFrame *> by_id;
    for (auto it = map_data.nodes.rbegin(); it != map_data.nodes.rend(); ++it) by_id[it->id] = &*it;   // first match wins
    const size_t n = map_data.graph.poses.size();
    std::vector<double> pose(7 * n);
    std::vector<int32_t> off(n + 1, 0);
    std::vector<float> pts;
    for (size_t i = 0; i < n; ++i) {
        const auto &p = map_data.graph.poses[i].pose;
        double *o = &pose[7 * i];
        o[0] = p.position.x; o[1] = p.position.y; o[2] = p.position.z;
        o[3] = p.orientation.x; o[4] = p.orientation.y; o[5] = p.orientation.z; o[6] = p.orientation.w;
        auto it = i < map_data.graph.poses_id.size() ? by_id.find(map_data.graph.poses_id[i]) : by_id.end();
        if (it != by_id.end())
            for (const auto &w : it->second->word_pts) { pts.push_back((float)w.x); pts.push_back((float)w.y); pts.push_back((float)w.z); }
        off[i + 1] = (int32_t)(pts.size() / 3);
    }
    ctx.check(fs_upload_keyframes(ctx.get(), (int32_t)n, pose.data(), off.data(), pts.data()), "fs_upload_keyframes");
}

// computeInformationForPose (deprecated/util.hpp:840-916) for a batch of poses against the staged MapData and costmap;
// `radius` folds in getNodesInRadius (:616-632; 4.5 m at the call site, CostCalculator.cpp:354).  Q = q_diag * I.
inline std::vector<float> computeInformationForPoses(frontier_exploration::ScoringContext &ctx,
                                                     const std::vector<frontier_exploration::Pose> &poses, double max_depth,
                                                     double hfov, double max_depth_error, float q_diag, double radius = 4.5)
{
    const int32_t n = (int32_t)poses.size();
    std::vector<double> p7(7 * (size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        double *o = &p7[7 * (size_t)i];
        o[0] = poses[i].position.x; o[1] = poses[i].position.y; o[2] = poses[i].position.z;
        o[3] = poses[i].orientation.x; o[4] = poses[i].orientation.y; o[5] = poses[i].orientation.z; o[6] = poses[i].orientation.w;
    }
    fs_keyframe_params prm{max_depth, hfov, max_depth_error, q_diag, radius};
    std::vector<float> info(n);
    ctx.check(fs_information_for_pose(ctx.get(), n, p7.data(), &prm, info.data(), nullptr, nullptr), "fs_information_for_pose");
    return info;
}
}  // namespace frontier_exploration_information_affine

namespace roadmap_explorer
{
using frontier_exploration::Point;
using frontier_exploration::Pose;
using frontier_exploration::ScoringContext;

// FIP/include/.../FisherInfoManager.hpp:72-145
class FisherInformationManager
{
public:
    explicit FisherInformationManager(std::shared_ptr<ScoringContext> ctx, double fisher_information_threshold = 550.0)
        : ctx_(std::move(ctx)), threshold_(fisher_information_threshold)
    {
        fs_fim_params p{14.0, 1.0};     // FisherInfoManager.cpp:64 radius; cone per DESIGN.md "Visibility"
        ctx_->check(fs_set_fim_params(ctx_->get(), &p), "fs_set_fim_params");
    }
    // replaces the GetLandmarksInView service round-trip (FisherInfoManager.cpp:52-77): the whole cloud, staged once
    void setLandmarks(const std::vector<float> &xyz) { ctx_->check(fs_upload_landmarks(ctx_->get(), xyz.data(), (int32_t)(xyz.size() / 3)), "fs_upload_landmarks"); }
    void setVisibility(double max_dist, double max_angle)
    {
        fs_fim_params p{max_dist, max_angle};
        ctx_->check(fs_set_fim_params(ctx_->get(), &p), "fs_set_fim_params");
    }
    // FisherInfoManager.hpp:94,96 — generateLookupTable / loadLookupTable (throws like the reference when the file is missing)
    void generateLookupTable(float minX, float maxX, float minY, float maxY, float minZ, float maxZ)
    {
        const float b[6] = {minX, maxX, minY, maxY, minZ, maxZ};
        ctx_->check(fs_lookup_generate(ctx_->get(), b), "fs_lookup_generate");
    }
    void generateLookupTable() { ctx_->check(fs_lookup_generate(ctx_->get(), nullptr), "fs_lookup_generate"); }
    void saveLookupTable(const std::string &path) { ctx_->check(fs_lookup_save(ctx_->get(), path.c_str()), "fs_lookup_save"); }
    void loadLookupTable(const std::string &path)
    {
        if (fs_lookup_load(ctx_->get(), path.c_str()) != FS_OK) throw std::runtime_error("Cannot load lookup table. Does it exist in the path?");
    }
    // FisherInfoManager.cpp:264-285 — plain table value, NaN when the key is absent
    float getInformationFromLookup(const float landmark_camera_frame[3])
    {
        float v = 0;
        ctx_->check(fs_lookup_query(ctx_->get(), landmark_camera_frame, &v), "fs_lookup_query");
        return v;
    }
    // FisherInfoManager.cpp:39-115
    bool isPoseSafe(Pose &given_pose, bool /*exhaustiveSearch*/, float &information)
    {
        const double pose7[7] = {given_pose.position.x, given_pose.position.y, given_pose.position.z,
                                 given_pose.orientation.x, given_pose.orientation.y, given_pose.orientation.z, given_pose.orientation.w};
        float info = 0;
        if (fs_score_fim(ctx_->get(), 1, pose7, &info, nullptr, nullptr, nullptr, nullptr, nullptr) != FS_OK) return false;   // service failure -> false (:52-57,73-77)
        information = info;
        return info > threshold_;                                                                                                 // :112-114
    }
    // FisherInfoManager.cpp:31-37
    bool isPoseSafe(Point point_from, Point point_to, bool exhaustiveSearch)
    {
        Pose relative_pose;
        frontier_exploration::getRelativePoseGivenTwoPoints(point_from, point_to, relative_pose);
        float information;
        return isPoseSafe(relative_pose, exhaustiveSearch, information);
    }
    // batch form used by the scorer: one call for many poses
    void poseInformation(const std::vector<Pose> &poses, std::vector<float> &information)
    {
        std::vector<double> p7(poses.size() * 7);
        for (size_t i = 0; i < poses.size(); ++i) {
            const Pose &p = poses[i];
            const double v[7] = {p.position.x, p.position.y, p.position.z, p.orientation.x, p.orientation.y, p.orientation.z, p.orientation.w};
            for (int k = 0; k < 7; ++k) p7[7 * i + k] = v[k];
        }
        information.assign(poses.size(), 0.0f);
        ctx_->check(fs_score_fim(ctx_->get(), (int32_t)poses.size(), p7.data(), information.data(), nullptr, nullptr, nullptr, nullptr, nullptr), "fs_score_fim");
    }

private:
    std::shared_ptr<ScoringContext> ctx_;
    double threshold_;
};
}  // namespace roadmap_explorer

#endif
