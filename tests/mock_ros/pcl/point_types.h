// mock: see tests/mock_ros/README.md
#pragma once
#include <vector>
namespace pcl {
struct PointXYZ { float x, y, z, pad; };
template <class P> struct PointCloud { std::vector<P> points; };
}
