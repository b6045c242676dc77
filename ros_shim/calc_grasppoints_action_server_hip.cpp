// calc_grasppoints_action_server_hip.cpp -- catkin-side action server that forwards the reference's action
// (action/CalcGraspPointsServer.action:1-8) to libhafgrasp.so.  NOT built in this repository's image (no ROS / PCL
// here); it is the file a maintainer drops next to src/calc_grasppoints_action_server.cpp and builds with
//   add_executable(calc_grasppoints_action_server_hip src/calc_grasppoints_action_server_hip.cpp)
//   target_link_libraries(calc_grasppoints_action_server_hip hafgrasp ${catkin_LIBRARIES} ${PCL_LIBRARIES})
// It keeps the reference's node name, action name, goal parsing and TF handling (server.cpp:250-329) and replaces
// loop_control() (335-402) by one haf_score() call.  Marker / rviz publishing is out of scope.
#include <ros/ros.h>
#include <ros/package.h>
#include <actionlib/server/simple_action_server.h>
#include <haf_grasping/CalcGraspPointsServerAction.h>
#include <pcl/point_types.h>
#include <pcl_conversions/pcl_conversions.h>
#include <pcl_ros/transforms.h>
#include <std_msgs/String.h>
#include <tf/transform_listener.h>

#include <sstream>
#include <string>

#include <hafgrasp.h>

class CalcGrasppointsHip
{
    ros::NodeHandle nh_;
    actionlib::SimpleActionServer<haf_grasping::CalcGraspPointsServerAction> as_;
    ros::Publisher pub_eval_, pub_input_pc_;
    tf::TransformListener tf_listener_;
    haf_engine *engine_ = nullptr;
    haf_config cfg_;
    std::string feature_file_, range_file_, model_file_, base_frame_id_;

public:
    explicit CalcGrasppointsHip(const std::string &name)
        : as_(nh_, name, boost::bind(&CalcGrasppointsHip::execute, this, _1), false)
    {
        pub_eval_ = nh_.advertise<std_msgs::String>("/haf_grasping/grasp_hypothesis_with_eval", 1);          // server.cpp:185
        pub_input_pc_ = nh_.advertise<sensor_msgs::PointCloud2>("/haf_grasping/calc_gp_as_inputpcROS", 1);   // 189
        haf_config_default(&cfg_);
        const std::string pkg = ros::package::getPath("haf_grasping");
        nh_.param("feature_file_path", feature_file_, std::string(""));                                      // 217-225
        nh_.param("range_file_path", range_file_, std::string(""));
        nh_.param("svmmodel_file_path", model_file_, std::string(""));
        int nshaf = 302;
        nh_.param("nr_features_without_shaf", nshaf, nshaf);
        if (feature_file_.empty()) feature_file_ = pkg + "/data/Features.txt";                               // 626-628
        if (range_file_.empty()) range_file_ = pkg + "/data/range21062012_allfeatures";                      // 767-769
        if (model_file_.empty()) model_file_ = pkg + "/data/all_features.txt.scale.model";                   // 771-773
        cfg_.feature_file = feature_file_.c_str();
        cfg_.range_file = range_file_.c_str();
        cfg_.model_file = model_file_.c_str();
        cfg_.nr_features_without_shaf = nshaf;
        cfg_.max_points = 1 << 22;
        if (haf_create(&cfg_, &engine_) != HAF_OK) {
            ROS_FATAL("hafgrasp: %s", haf_last_error(NULL));
            ros::shutdown();
            return;
        }
        as_.start();
    }
    ~CalcGrasppointsHip() { haf_destroy(engine_); }

    void execute(const haf_grasping::CalcGraspPointsServerGoalConstPtr &goal)
    {
        const haf_grasping::GraspInput &gi = goal->graspinput;
        base_frame_id_ = gi.goal_frame_id.empty() ? "/base_link" : gi.goal_frame_id;                          // 294-301
        if (!tf_listener_.waitForTransform(base_frame_id_, gi.input_pc.header.frame_id, gi.input_pc.header.stamp, ros::Duration(1.0)))
            ROS_WARN("NO TRANSFORM FOR POINT CLOUD FOUND");                                                   // 307-311
        pcl::PointCloud<pcl::PointXYZ> in_old, in_base;
        pcl::fromROSMsg(gi.input_pc, in_old);                                                                 // 314
        pcl_ros::transformPointCloud(base_frame_id_, in_old, in_base, tf_listener_);                          // 316
        pub_input_pc_.publish(gi.input_pc);                                                                   // 319
        if (as_.isPreemptRequested() || !ros::ok()) { as_.setPreempted(); return; }                           // 350-357

        haf_grasp_input in;
        haf_grasp_input_default(&in);
        in.grasp_area_center[0] = gi.grasp_area_center.x;                                                     // 258-260
        in.grasp_area_center[1] = gi.grasp_area_center.y;
        in.grasp_area_center[2] = gi.grasp_area_center.z;
        in.grasp_area_length_x = gi.grasp_area_length_x;                                                      // 266-267 (engine truncates)
        in.grasp_area_length_y = gi.grasp_area_length_y;
        in.approach_vector[0] = gi.approach_vector.x;                                                         // 270-273 (engine normalises)
        in.approach_vector[1] = gi.approach_vector.y;
        in.approach_vector[2] = gi.approach_vector.z;
        in.max_calculation_time = gi.max_calculation_time.toSec();                                            // 277
        in.show_only_best_grasp = gi.show_only_best_grasp;                                                    // 284
        in.threshold_grasp_evaluation = gi.threshold_grasp_evaluation;                                        // never read by the reference
        in.gripper_opening_width = gi.gripper_opening_width;                                                  // 281

        haf_cloud cloud;
        cloud.xyz = in_base.points.empty() ? NULL : &in_base.points[0].x;
        cloud.n_points = in_base.points.size();
        cloud.stride_floats = sizeof(pcl::PointXYZ) / sizeof(float);
        cloud.on_device = 0;
        haf_grasp_output out;
        if (haf_score(engine_, &cloud, &in, &out) != HAF_OK) {
            ROS_ERROR("hafgrasp: %s", haf_last_error(engine_));
            as_.setAborted();
            return;
        }
        haf_grasping::CalcGraspPointsServerResult result;
        haf_grasping::GraspOutput &g = result.graspOutput;                                                    // 1386-1401
        g.header.stamp = ros::Time::now();
        g.header.frame_id = base_frame_id_;
        g.eval = out.eval;
        g.graspPoint1.x = out.grasp_point1[0]; g.graspPoint1.y = out.grasp_point1[1]; g.graspPoint1.z = out.grasp_point1[2];
        g.graspPoint2.x = out.grasp_point2[0]; g.graspPoint2.y = out.grasp_point2[1]; g.graspPoint2.z = out.grasp_point2[2];
        g.averagedGraspPoint.x = out.averaged_grasp_point[0]; g.averagedGraspPoint.y = out.averaged_grasp_point[1]; g.averagedGraspPoint.z = out.averaged_grasp_point[2];
        g.approachVector.x = out.approach_vector[0]; g.approachVector.y = out.approach_vector[1]; g.approachVector.z = out.approach_vector[2];
        g.roll = out.roll;
        std::stringstream ss;                                                                                 // 1384
        ss << out.eval << " " << (float)out.grasp_point1[0] << " " << (float)out.grasp_point1[1] << " " << (float)out.grasp_point1[2] << " "
           << (float)out.grasp_point2[0] << " " << (float)out.grasp_point2[1] << " " << (float)out.grasp_point2[2] << " "
           << (float)out.approach_vector[0] << " " << (float)out.approach_vector[1] << " " << (float)out.approach_vector[2] << " "
           << out.averaged_grasp_point[0] << " " << out.averaged_grasp_point[1] << " " << out.averaged_grasp_point[2] << " "
           << out.best_roll * cfg_.roll_step_deg;
        std_msgs::String msg;
        msg.data = ss.str();
        pub_eval_.publish(msg);                                                                               // 1419
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
