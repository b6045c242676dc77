// shim_core.h -- everything of the action-server shim that does not need ROS: goal fields -> haf_grasp_input, the roll-by-roll
// hypothesis publication rule, GraspOutput fields, and the text of /haf_grasping/grasp_hypothesis_with_eval.  Header-only,
// C++11, depends on include/hafgrasp.h alone, so it is compiled and tested in this repository (haf_grasp_cli uses it;
// tests/test_engine_gpu.py::test_cli_and_server_mirror, tests/test_host_cpu.py) although ROS is not installed here.
// ros_shim/calc_grasppoints_action_server_hip.cpp is the thin adapter that fills GoalFields from the ROS message and
// copies ResultFields back.
//
// Reference lines (src/calc_grasppoints_action_server.cpp): goal parsing 258-301, roll loop 343-386 (preemption 350-357), per-roll
// grasp grid 901-902 / 979-1016, per-roll publication 962-969, result fields and string 1384-1401, final call 390.
#ifndef HAF_SHIM_CORE_H_
#define HAF_SHIM_CORE_H_

#include <hafgrasp.h>

#include <cstdint>
#include <functional>
#include <sstream>
#include <string>
#include <vector>

namespace hafshim {

// msg/GraspInput.msg:3-15 without the cloud (passed separately, already in the base frame: server.cpp:316)
struct GoalFields {
    std::string goal_frame_id;               // "" -> "/base_link" (294-301)
    double center[3] = {0, 0, 0};            // grasp_area_center
    float  length_x = 32, length_y = 44;     // grasp_area_length_x/y: cm incl. the +14 border (client.cpp:183-184)
    double max_calculation_time = 50;        // ros::Duration::toSec()
    bool   show_only_best_grasp = false;
    int    threshold_grasp_evaluation = 0;   // never read by the reference server
    double approach_vector[3] = {0, 0, 1};
    int    gripper_opening_width = 1;
};

// msg/GraspOutput.msg:1-7 (header.stamp is the adapter's business)
struct ResultFields {
    std::string frame_id;
    int    eval = -20;
    double grasp_point1[3] = {0, 0, 0}, grasp_point2[3] = {0, 0, 0}, averaged_grasp_point[3] = {0, 0, 0};
    double approach_vector[3] = {0, 0, 0};
    float  roll = 0;
};

// One cell of the per-roll grasp grid as publish_grasp_grid (979-1016) hands it to gp_to_marker: every cell of
// point_inside_box_grid (the mask), its position in the base frame (987-989, 996) and graspseval[row][col] (865-880).
struct GridCell {
    int row, col;
    float x, y, z;
    float value;
};
typedef std::function<void(int roll, const std::vector<GridCell> &)> GridFn;
typedef std::function<bool()> PreemptFn;             // as_.isPreemptRequested() || !ros::ok() (350)
enum { HAF_SHIM_PREEMPTED = 1 };                      // run_goal: the goal was preempted -> setPreempted() (354-356)

inline std::string base_frame(const GoalFields &g) { return g.goal_frame_id.empty() ? std::string("/base_link") : g.goal_frame_id; }   // 294-301

// read_pc_cb 258-284.  The engine normalises the approach vector (270-273) and truncates the lengths to int (266-267)
// itself, exactly as the server does, so the fields go over unchanged; the duration goes through float like 277.
inline void goal_to_input(const GoalFields &g, haf_grasp_input *in)
{
    haf_grasp_input_default(in);
    for (int k = 0; k < 3; k++) in->grasp_area_center[k] = g.center[k];                    // 258-260
    in->grasp_area_length_x = g.length_x;                                                  // 266-267
    in->grasp_area_length_y = g.length_y;
    for (int k = 0; k < 3; k++) in->approach_vector[k] = g.approach_vector[k];             // 270-273
    in->max_calculation_time = (double)(float)g.max_calculation_time;                      // 277
    in->gripper_opening_width = g.gripper_opening_width;                                   // 281
    in->show_only_best_grasp = g.show_only_best_grasp ? 1 : 0;                             // 284
    in->threshold_grasp_evaluation = g.threshold_grasp_evaluation;
}

// 1386-1401
inline void output_to_result(const haf_grasp_output &o, const std::string &frame_id, ResultFields *r)
{
    r->frame_id = frame_id;
    r->eval = o.eval;
    for (int k = 0; k < 3; k++) {
        r->grasp_point1[k] = o.grasp_point1[k];
        r->grasp_point2[k] = o.grasp_point2[k];
        r->averaged_grasp_point[k] = o.averaged_grasp_point[k];
        r->approach_vector[k] = o.approach_vector[k];
    }
    r->roll = o.roll;
}

// The string of server.cpp:1384: "<eval> <gp1 xyz> <gp2 xyz> <approach xyz> <averaged xyz> <roll in degrees>".  The reference
// streams Eigen floats (gp1_wcs, gp2_wcs, appr_vec) and doubles (the averages, (float + float) / 2.0) into a default
// std::stringstream; the same stream with the same types gives the same text (6 significant digits).
inline std::string hypothesis_string(const haf_grasp_output &o, int roll_step_deg)
{
    std::stringstream ss;
    ss << o.eval << " " << (float)o.grasp_point1[0] << " " << (float)o.grasp_point1[1] << " " << (float)o.grasp_point1[2] << " "
       << (float)o.grasp_point2[0] << " " << (float)o.grasp_point2[1] << " " << (float)o.grasp_point2[2] << " "
       << (float)o.approach_vector[0] << " " << (float)o.approach_vector[1] << " " << (float)o.approach_vector[2] << " "
       << o.averaged_grasp_point[0] << " " << o.averaged_grasp_point[1] << " " << o.averaged_grasp_point[2] << " "
       << o.best_roll * roll_step_deg;
    return ss.str();
}

// One goal, as read_pc_cb + loop_control run it once the cloud is in the base frame: every roll is scored (one call), the
// hypotheses the server would publish roll by roll (962-969: !show_only_best and vote > graspval_th, eval = max(vote-20, 10))
// are handed to `publish` in roll order for the rolls the sequential loop would have executed (early exit 362-365), then
// the overall best (390) -- which the server publishes on the same topic (1419) -- and the result fields.
// Returns the engine's status; on failure *err carries haf_last_error().
// `grid` (optional) receives, for every roll the sequential loop would have executed and in roll order, the cells of that roll's
// grasp grid (901-902): what the reference turns into rviz markers.  `preempted` (optional) is asked where the reference asks
// (350: at the top of a roll -- here all rolls of a goal are scored in ONE call of a few hundred microseconds to milliseconds, so
// that is before the call and again before anything is published); a preempted goal returns HAF_SHIM_PREEMPTED and publishes nothing.
inline int run_goal(haf_engine *engine, const haf_config &cfg, const GoalFields &goal, const haf_cloud &cloud,
                    const std::function<void(const std::string &)> &publish, ResultFields *result, haf_grasp_output *raw,
                    std::string *err, const GridFn &grid = GridFn(), const PreemptFn &preempted = PreemptFn())
{
    haf_grasp_input in;
    goal_to_input(goal, &in);
    if (preempted && preempted()) return HAF_SHIM_PREEMPTED;
    std::vector<haf_roll_record> rec((size_t)cfg.n_rolls);
    int rc = haf_score_rolls(engine, 1, &cloud, &in, 0, cfg.n_rolls, rec.data());
    haf_grasp_output out;
    if (rc == HAF_OK) rc = haf_finalize(engine, &in, rec.data(), &out);
    if (rc != HAF_OK) {
        if (err) *err = haf_last_error(engine);
        return rc;
    }
    if (preempted && preempted()) return HAF_SHIM_PREEMPTED;
    if (grid) {
        const size_t HW = (size_t)cfg.grid_h * cfg.grid_w;
        std::vector<float> ev(HW);
        std::vector<uint8_t> mask(HW);
        std::vector<GridCell> cells;
        const int gw = in.gripper_opening_width;
        // publish_grasp_grid 987-989 (float arithmetic on the double centre, integer division HEIGHT/2/gripperwidth inside 0.01*...)
        const float x0 = (float)(in.grasp_area_center[0] - 0.01 * cfg.grid_h / 2 / gw);
        const float y0 = (float)(in.grasp_area_center[1] - 0.01 * cfg.grid_w / 2);
        const float z0 = (float)(in.grasp_area_center[2] + cfg.z_shift);
        for (int r = 0; r < out.rolls_done; r++) {
            rc = haf_get_roll_grid(engine, 0, r, ev.data(), mask.data());
            if (rc != HAF_OK) {
                if (err) *err = haf_last_error(engine);
                return rc;
            }
            cells.clear();
            for (int row = 0; row < cfg.grid_h; row++)
                for (int col = 0; col < cfg.grid_w; col++)
                    if (mask[(size_t)row * cfg.grid_w + col])                                           // 993-995
                        cells.push_back(GridCell{row, col, (float)(x0 + 0.01 / gw * row), (float)(y0 + 0.01 * col), z0,
                                                 ev[(size_t)row * cfg.grid_w + col]});                  // 996
            grid(r, cells);
        }
    }
    for (int r = 0; r < out.rolls_done; r++) {
        haf_grasp_output ro;
        int32_t pub = 0;
        rc = haf_roll_pose(engine, &in, rec.data(), r, &ro, &pub);
        if (rc != HAF_OK) {
            if (err) *err = haf_last_error(engine);
            return rc;
        }
        if (pub && publish) publish(hypothesis_string(ro, cfg.roll_step_deg));
    }
    if (publish) publish(hypothesis_string(out, cfg.roll_step_deg));                      // 390 -> 1419
    if (result) output_to_result(out, base_frame(goal), result);
    if (raw) *raw = out;
    return HAF_OK;
}

}  // namespace hafshim
#endif  // HAF_SHIM_CORE_H_
