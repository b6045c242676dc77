// mock: see tests/mock_ros/README.md
#pragma once
#include <string>
namespace ros { namespace package { inline std::string getPath(const std::string &) { return "."; } } }
