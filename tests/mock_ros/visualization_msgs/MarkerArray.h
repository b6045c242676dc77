// mock: see tests/mock_ros/README.md
#pragma once
#include <string>
#include <vector>
#include <geometry_msgs/types.h>
#include <std_msgs/String.h>
namespace visualization_msgs {
struct Marker {
    enum { CUBE = 1, ADD = 0 };
    std_msgs::Header header;
    std::string ns;
    int id = 0, type = 0, action = 0;
    geometry_msgs::Pose pose;
    geometry_msgs::Vector3 scale;
    struct Color { float r = 0, g = 0, b = 0, a = 0; } color;
};
struct MarkerArray { std::vector<Marker> markers; };
}
