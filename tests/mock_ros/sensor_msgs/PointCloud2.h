// mock: see tests/mock_ros/README.md
#pragma once
#include <ros/ros.h>
#include <std_msgs/String.h>
namespace sensor_msgs { struct PointCloud2 { std_msgs::Header header; }; }
