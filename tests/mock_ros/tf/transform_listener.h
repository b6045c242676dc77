// mock: see tests/mock_ros/README.md
#pragma once
#include <string>
#include <ros/ros.h>
namespace tf { struct TransformListener { bool waitForTransform(const std::string &, const std::string &, const ros::Time &, const ros::Duration &) { return true; } }; }
