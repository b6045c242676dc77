// mock: see tests/mock_ros/README.md
#pragma once
#include <functional>
#include <memory>
#include <ros/ros.h>
namespace actionlib {
template <class Action> class SimpleActionServer {
public:
    typedef std::shared_ptr<const typename Action::Goal> GoalConstPtr;
    typedef std::function<void(const GoalConstPtr &)> ExecuteCallback;
    SimpleActionServer(ros::NodeHandle, const std::string &, ExecuteCallback, bool) {}
    void start() {}
    bool isPreemptRequested() { return false; }
    void setPreempted() {}
    void setAborted() {}
    void setSucceeded(const typename Action::Result &) {}
};
}  // namespace actionlib
