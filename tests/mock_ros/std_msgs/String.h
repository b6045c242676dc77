// mock: see tests/mock_ros/README.md
#pragma once
#include <string>
namespace std_msgs { struct String { std::string data; }; struct Header { ros::Time stamp; std::string frame_id; }; }
