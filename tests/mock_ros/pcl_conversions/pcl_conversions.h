// mock: see tests/mock_ros/README.md
#pragma once
#include <pcl/point_types.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl { template <class P> void fromROSMsg(const sensor_msgs::PointCloud2 &, PointCloud<P> &) {} }
