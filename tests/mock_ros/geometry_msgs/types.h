// mock: see tests/mock_ros/README.md
#pragma once
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
}
