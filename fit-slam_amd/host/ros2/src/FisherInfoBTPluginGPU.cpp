// FisherInfoBTPluginGPU.cpp — the Fisher-information BT plugin with the scoring on an MI355X.
//
// Registers the same two node IDs as roadmap_explorer::FisherInfoBTPlugin
// (FIP/src/fisher_information/FisherInfoBTPlugin.cpp:199-214):
//   "EvaluateFisherInformation"  -> FisherInformationBTGPU below: the tick of FisherInfoBTPlugin.cpp:24-57 with
//                                   FisherInformationManagerGPU::isPoseSafe behind it
//   "MarkLethalFOV"              -> the reference's own node, untouched: it is actuation (a costmap service call), not on
//                                   the scoring path; it comes from the reference plugin — already loaded, or registered from
//                                   here when this plugin is loaded alone (see registerNodes: loading both is fine)
// Blackboard key "latest_robot_pose", input port `exhaustive_landmark_search`, TF map -> base_footprint, the 700 ms
// back-off and the error code on FAILURE are the reference's.
#include "fitslam_frontier_ros2/FisherInfoBTPluginGPU.hpp"

#include <chrono>
#include <stdexcept>

#include <pluginlib/class_list_macros.hpp>
#include <tf2_ros/buffer.h>

#include <fisher_information_plugins/fisher_information/FisherInfoBTPlugin.hpp>   // the reference plugin (MarkLethalFOV)

namespace roadmap_explorer
{
    class FisherInformationBTGPU : public BT::SyncActionNode
    {
    public:
        FisherInformationBTGPU(const std::string &name, const BT::NodeConfiguration &config,
                               std::shared_ptr<nav2_costmap_2d::Costmap2DROS> explore_costmap_ros,
                               std::shared_ptr<nav2_util::LifecycleNode> node, std::shared_ptr<tf2_ros::Buffer> tf_buffer)
            : BT::SyncActionNode(name, config), explore_costmap_ros_(explore_costmap_ros), node_(node), tf_buffer_(tf_buffer)
        {
            // hard-set like the reference (FisherInfoBTPlugin.cpp:20); the YAML value is dead there too
            parameterInstance.setValue<double>("fisherInformation.fisher_information_threshold", 550.0);
            fisher_info_manager_ = std::make_shared<fitslam_frontier_ros2::FisherInformationManagerGPU>(node_);
            LOG_INFO("FisherInformationBTGPU Constructor");
        }

        BT::NodeStatus tick() override
        {
            geometry_msgs::msg::PoseStamped robot_pose_2d;
            if (!config().blackboard->get<geometry_msgs::msg::PoseStamped>("latest_robot_pose", robot_pose_2d))
            {
                LOG_FATAL("Failed to retrieve latest_robot_pose from blackboard.");
                throw std::runtime_error("Failed to retrieve latest_robot_pose from blackboard.");
            }
            // the 3-D camera pose comes from TF, not from the blackboard (FisherInfoBTPlugin.cpp:34-39)
            const geometry_msgs::msg::TransformStamped tf_map_base =
                tf_buffer_->lookupTransform("map", "base_footprint", tf2::TimePointZero);
            geometry_msgs::msg::Pose camera_pose;
            camera_pose.position.x = tf_map_base.transform.translation.x;
            camera_pose.position.y = tf_map_base.transform.translation.y;
            camera_pose.position.z = tf_map_base.transform.translation.z;
            camera_pose.orientation = tf_map_base.transform.rotation;

            // (the reference reads the port into an uninitialised bool, :41-42; initialised here)
            bool exhaustive_landmark_search = false;
            getInput<bool>("exhaustive_landmark_search", exhaustive_landmark_search);
            if (exhaustive_landmark_search)
                LOG_WARN("Exhaustive landmark search is enabled for this iteration");

            float information = 0.0f;
            if (fisher_info_manager_->isPoseSafe(camera_pose, exhaustive_landmark_search, information))
                return BT::NodeStatus::SUCCESS;
            rclcpp::sleep_for(std::chrono::milliseconds(700));
            config().blackboard->set<ExplorationErrorCode>("error_code_id", ExplorationErrorCode::NO_ERROR);
            return BT::NodeStatus::FAILURE;
        }

        static BT::PortsList providedPorts()
        {
            return {BT::InputPort<bool>("exhaustive_landmark_search")};
        }

    private:
        std::shared_ptr<fitslam_frontier_ros2::FisherInformationManagerGPU> fisher_info_manager_;
        std::shared_ptr<nav2_costmap_2d::Costmap2DROS> explore_costmap_ros_;
        std::shared_ptr<nav2_util::LifecycleNode> node_;
        std::shared_ptr<tf2_ros::Buffer> tf_buffer_;
    };

    FisherInfoBTPluginGPU::FisherInfoBTPluginGPU()
    {
    }

    FisherInfoBTPluginGPU::~FisherInfoBTPluginGPU()
    {
    }

    void FisherInfoBTPluginGPU::registerNodes(BT::BehaviorTreeFactory &factory, std::shared_ptr<nav2_util::LifecycleNode> node,
                                              std::shared_ptr<nav2_costmap_2d::Costmap2DROS> explore_costmap_ros,
                                              std::shared_ptr<tf2_ros::Buffer> tf_buffer)
    {
        // This plugin owns ONE node ID, "EvaluateFisherInformation".  "MarkLethalFOV" is the reference's (its class lives in
        // the reference plugin's .cpp, FisherInfoBTPlugin.cpp:73-197, and can only be registered by that plugin).
        //   * loaded NEXT TO the reference plugin (list this one after it in the node's bt_plugins): both IDs are there
        //     already — the reference's CPU builder of "EvaluateFisherInformation" is replaced, nothing is registered twice;
        //   * loaded INSTEAD of it: the factory knows neither ID yet — the reference's registration runs first (builders are
        //     lazy: no CPU manager is constructed by it), then the ID is taken over as above.
        const auto &known = factory.builders();
        const bool have_evaluate = known.count("EvaluateFisherInformation") > 0, have_mark = known.count("MarkLethalFOV") > 0;
        if (!have_evaluate && !have_mark) {
            FisherInfoBTPlugin reference_plugin;
            reference_plugin.registerNodes(factory, node, explore_costmap_ros, tf_buffer);
        }
        if (factory.builders().count("EvaluateFisherInformation") > 0) factory.unregisterBuilder("EvaluateFisherInformation");
        BT::NodeBuilder builder_evaluate =
            [explore_costmap_ros, node, tf_buffer](const std::string &name, const BT::NodeConfiguration &config)
        {
            return std::make_unique<FisherInformationBTGPU>(name, config, explore_costmap_ros, node, tf_buffer);
        };
        factory.registerBuilder<FisherInformationBTGPU>("EvaluateFisherInformation", builder_evaluate);
    }
}

PLUGINLIB_EXPORT_CLASS(
    roadmap_explorer::FisherInfoBTPluginGPU,
    roadmap_explorer::BTPlugin)
