// FisherInfoBTPluginGPU.hpp — pluginlib class with the interface of roadmap_explorer::FisherInfoBTPlugin
// (FIP/include/fisher_information_plugins/fisher_information/FisherInfoBTPlugin.hpp:13-21).
#pragma once

#include <roadmap_explorer/bt_plugins/interface_pluginlib.hpp>
#include "roadmap_explorer/util/Logger.hpp"
#include "roadmap_explorer/util/EventLogger.hpp"
#include "fit_slam2_msgs/srv/mark_lethal.hpp"

#include "fitslam_frontier_ros2/FisherInfoManagerGPU.hpp"

namespace roadmap_explorer
{
    class FisherInfoBTPluginGPU : public BTPlugin
    {
        public:
        FisherInfoBTPluginGPU();

        ~FisherInfoBTPluginGPU();

        void registerNodes(BT::BehaviorTreeFactory & factory, std::shared_ptr<nav2_util::LifecycleNode> node, std::shared_ptr<nav2_costmap_2d::Costmap2DROS> explore_costmap_ros, std::shared_ptr<tf2_ros::Buffer> tf_buffer) override;
    };
};
