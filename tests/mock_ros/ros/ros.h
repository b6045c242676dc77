// mock: see tests/mock_ros/README.md
#pragma once
#include <cstdio>
#include <string>
namespace ros {
struct Duration { explicit Duration(double = 0) {} double toSec() const { return 0; } };
struct Time { static Time now() { return Time(); } };
struct Publisher { template <class M> void publish(const M &) const {} };
struct NodeHandle {
    template <class M> Publisher advertise(const std::string &, int) { return Publisher(); }
    template <class T> bool param(const std::string &, T &v, const T &d) { v = d; return false; }
    bool param(const std::string &, std::string &v, const std::string &d) { v = d; return false; }
};
inline void init(int &, char **, const std::string &) {}
inline void spin() {}
inline bool ok() { return true; }
inline void shutdown() {}
namespace this_node { inline std::string getName() { return "node"; } }
}  // namespace ros
#define ROS_FATAL(...) std::fprintf(stderr, __VA_ARGS__)
#define ROS_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#define ROS_WARN(...) std::fprintf(stderr, __VA_ARGS__)
