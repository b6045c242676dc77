// calc_grasppoints_action_server_hip.cpp -- catkin-side adapter: the reference's action (action/CalcGraspPointsServer.action:1-8)
// served by libhafgrasp.so.  NOT built in this repository's image (no ROS / PCL here).  It only moves fields between ROS
// messages and ros_shim/shim_core.h, which holds the whole goal -> result logic and IS compiled and tested here:
//   add_executable(calc_grasppoints_action_server_hip src/calc_grasppoints_action_server_hip.cpp)
//   target_link_libraries(calc_grasppoints_action_server_hip hafgrasp ${catkin_LIBRARIES} ${PCL_LIBRARIES})
// Node name, action name, parameters, TF handling and topics are the reference's (server.cpp:181-228, 250-329).  The per-roll
// grasp grid (901-902, 979-1016) is forwarded as the reference's MarkerArray topic with plain cube markers (one per masked cell at
// the reference's position, height = vote); the tf_help frame and the styling of gp_to_marker / grasp_area_to_marker (1020-1270)
// stay out of scope (rviz).
// tests/test_host_cpu.py compiles this translation unit (-fsyntax-only) against the minimal mock headers of tests/mock_ros/.
#include <ros/ros.h>
#include <ros/package.h>
#include <actionlib/server/simple_action_server.h>
#include <haf_grasping/CalcGraspPointsServerAction.h>
#include <pcl/point_types.h>
#include <pcl_conversions/pcl_conversions.h>
#include <pcl_ros/transforms.h>
#include <std_msgs/String.h>
#include <tf/transform_listener.h>
#include <visualization_msgs/MarkerArray.h>

#include "shim_core.h"

class CalcGrasppointsHip
{
    ros::NodeHandle nh_;
    actionlib::SimpleActionServer<haf_grasping::CalcGraspPointsServerAction> as_;
    ros::Publisher pub_eval_, pub_input_pc_, pub_grid_;
    tf::TransformListener tf_listener_;
    haf_engine *engine_ = nullptr;
    haf_config cfg_;
    std::string feature_file_, range_file_, model_file_;

public:
    explicit CalcGrasppointsHip(const std::string &name)
        : as_(nh_, name, [this](const haf_grasping::CalcGraspPointsServerGoalConstPtr &goal) { this->execute(goal); }, false)
    {
        pub_grid_ = nh_.advertise<visualization_msgs::MarkerArray>("visualization_marker_array", 1);           // server.cpp:187
        pub_eval_ = nh_.advertise<std_msgs::String>("/haf_grasping/grasp_hypothesis_with_eval", 1);          // server.cpp:185
        pub_input_pc_ = nh_.advertise<sensor_msgs::PointCloud2>("/haf_grasping/calc_gp_as_inputpcROS", 1);   // 189
        haf_config_default(&cfg_);
        const std::string pkg = ros::package::getPath("haf_grasping");
        nh_.param("feature_file_path", feature_file_, pkg + "/data/Features.txt");                           // 217-225, 626-628
        nh_.param("range_file_path", range_file_, pkg + "/data/range21062012_allfeatures");                  // 767-769
        nh_.param("svmmodel_file_path", model_file_, pkg + "/data/all_features.txt.scale.model");            // 771-773
        nh_.param("nr_features_without_shaf", cfg_.nr_features_without_shaf, 302);
        bool svm_with_probability = false;                 // the reference passes a literal `false` (383); a parameter here
        nh_.param("svm_with_probability", svm_with_probability, false);
        if (svm_with_probability) cfg_.flags |= HAF_FLAG_PROBABILITY;
        cfg_.feature_file = feature_file_.c_str();
        cfg_.range_file = range_file_.c_str();
        cfg_.model_file = model_file_.c_str();
        cfg_.max_points = 1 << 22;
        if (haf_create(&cfg_, &engine_) != HAF_OK) { ROS_FATAL("hafgrasp: %s", haf_last_error(NULL)); ros::shutdown(); return; }
        as_.start();
    }
    ~CalcGrasppointsHip() { haf_destroy(engine_); }

    void execute(const haf_grasping::CalcGraspPointsServerGoalConstPtr &goal)
    {
        const haf_grasping::GraspInput &gi = goal->graspinput;
        hafshim::GoalFields g;
        g.goal_frame_id = gi.goal_frame_id;
        g.center[0] = gi.grasp_area_center.x; g.center[1] = gi.grasp_area_center.y; g.center[2] = gi.grasp_area_center.z;
        g.length_x = gi.grasp_area_length_x; g.length_y = gi.grasp_area_length_y;
        g.approach_vector[0] = gi.approach_vector.x; g.approach_vector[1] = gi.approach_vector.y; g.approach_vector[2] = gi.approach_vector.z;
        g.max_calculation_time = gi.max_calculation_time.toSec();
        g.show_only_best_grasp = gi.show_only_best_grasp;
        g.threshold_grasp_evaluation = gi.threshold_grasp_evaluation;
        g.gripper_opening_width = gi.gripper_opening_width;
        const std::string frame = hafshim::base_frame(g);
        if (!tf_listener_.waitForTransform(frame, gi.input_pc.header.frame_id, gi.input_pc.header.stamp, ros::Duration(1.0)))
            ROS_WARN("NO TRANSFORM FOR POINT CLOUD FOUND");                                                   // 307-311
        pcl::PointCloud<pcl::PointXYZ> in_old, in_base;
        pcl::fromROSMsg(gi.input_pc, in_old);                                                                 // 314
        pcl_ros::transformPointCloud(frame, in_old, in_base, tf_listener_);                                   // 316
        pub_input_pc_.publish(gi.input_pc);                                                                   // 319
        if (as_.isPreemptRequested() || !ros::ok()) { as_.setPreempted(); return; }                           // 350-357

        haf_cloud cloud = {in_base.points.empty() ? NULL : &in_base.points[0].x, in_base.points.size(),
                           sizeof(pcl::PointXYZ) / sizeof(float), 0};
        hafshim::ResultFields r;
        std::string err;
        auto publish = [this](const std::string &s) { std_msgs::String m; m.data = s; pub_eval_.publish(m); };   // 1419
        // the per-roll grasp grid (901-902 -> 979-1016): one marker per masked cell, where publish_grasp_grid puts it (987-996)
        auto on_grid = [this, &frame](int roll, const std::vector<hafshim::GridCell> &cells) {
            visualization_msgs::MarkerArray ma;
            int id = 1;
            for (const hafshim::GridCell &c : cells) {
                visualization_msgs::Marker mk;
                mk.header.frame_id = frame;
                mk.header.stamp = ros::Time::now();
                mk.ns = "haf_grasp_grid";
                mk.id = roll * 100000 + id++;
                mk.type = visualization_msgs::Marker::CUBE;
                mk.action = visualization_msgs::Marker::ADD;
                mk.pose.position.x = c.x; mk.pose.position.y = c.y; mk.pose.position.z = c.z;
                mk.pose.orientation.w = 1.0;
                mk.scale.x = 0.002; mk.scale.y = 0.002; mk.scale.z = 0.001 * (c.value > 1 ? c.value : 1);
                mk.color.a = 1.0; mk.color.g = c.value > 0 ? 1.0 : 0.0; mk.color.r = c.value > 0 ? 0.0 : 1.0;
                ma.markers.push_back(mk);
            }
            pub_grid_.publish(ma);
        };
        auto preempted = [this]() { return as_.isPreemptRequested() || !ros::ok(); };                           // 350
        const int rc = hafshim::run_goal(engine_, cfg_, g, cloud, publish, &r, NULL, &err, on_grid, preempted);
        if (rc == hafshim::HAF_SHIM_PREEMPTED) { as_.setPreempted(); return; }                                  // 354-356
        if (rc != HAF_OK) {
            ROS_ERROR("hafgrasp: %s", err.c_str());
            as_.setAborted();
            return;
        }
        haf_grasping::CalcGraspPointsServerResult result;
        haf_grasping::GraspOutput &o = result.graspOutput;                                                    // 1386-1401
        o.header.stamp = ros::Time::now();
        o.header.frame_id = r.frame_id;
        o.eval = r.eval;
        o.graspPoint1.x = r.grasp_point1[0]; o.graspPoint1.y = r.grasp_point1[1]; o.graspPoint1.z = r.grasp_point1[2];
        o.graspPoint2.x = r.grasp_point2[0]; o.graspPoint2.y = r.grasp_point2[1]; o.graspPoint2.z = r.grasp_point2[2];
        o.averagedGraspPoint.x = r.averaged_grasp_point[0]; o.averagedGraspPoint.y = r.averaged_grasp_point[1]; o.averagedGraspPoint.z = r.averaged_grasp_point[2];
        o.approachVector.x = r.approach_vector[0]; o.approachVector.y = r.approach_vector[1]; o.approachVector.z = r.approach_vector[2];
        o.roll = r.roll;
        as_.setSucceeded(result);                                                                             // 396-401
    }
};

int main(int argc, char **argv)
{
    ros::init(argc, argv, "calc_grasppoints_svm_action_server");                                              // 1428
    CalcGrasppointsHip server(ros::this_node::getName());
    ros::spin();
    return 0;
}
