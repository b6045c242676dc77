// mock: see tests/mock_ros/README.md
#pragma once
#include <pcl/point_types.h>
#include <tf/transform_listener.h>
namespace pcl_ros { template <class P> bool transformPointCloud(const std::string &, const pcl::PointCloud<P> &, pcl::PointCloud<P> &, const tf::TransformListener &) { return true; } }
