// mock of the generated action / message types (action/CalcGraspPointsServer.action, msg/GraspInput.msg, msg/GraspOutput.msg):
// see tests/mock_ros/README.md
#pragma once
#include <memory>
#include <ros/ros.h>
#include <geometry_msgs/types.h>
#include <sensor_msgs/PointCloud2.h>
namespace haf_grasping {
struct GraspInput {
    sensor_msgs::PointCloud2 input_pc;
    std::string goal_frame_id;
    geometry_msgs::Point grasp_area_center;
    float grasp_area_length_x = 0, grasp_area_length_y = 0;
    ros::Duration max_calculation_time;
    bool show_only_best_grasp = false;
    int threshold_grasp_evaluation = 0;
    geometry_msgs::Vector3 approach_vector;
    int gripper_opening_width = 1;
};
struct GraspOutput {
    std_msgs::Header header;
    int eval = 0;
    geometry_msgs::Point graspPoint1, graspPoint2, averagedGraspPoint;
    geometry_msgs::Vector3 approachVector;
    float roll = 0;
};
struct CalcGraspPointsServerGoal { GraspInput graspinput; };
struct CalcGraspPointsServerResult { GraspOutput graspOutput; };
struct CalcGraspPointsServerAction { typedef CalcGraspPointsServerGoal Goal; typedef CalcGraspPointsServerResult Result; };
typedef std::shared_ptr<const CalcGraspPointsServerGoal> CalcGraspPointsServerGoalConstPtr;
}  // namespace haf_grasping
