// haf_grasp_cli -- ROS-free stand-in for the demo flow of the reference (README:29-41): the client
// (src/calc_grasppoints_action_client.cpp) loads a .pcd, fills a GraspInput from its parameters and waits for the
// GraspOutput.  This tool does the same against libhafgrasp.so through the C-ABI, in C++ like the reference's host code.
//
// Parameter surface = the client's ROS params/services (client.cpp:79-118, 214-300):
//   --center x y z            grasp_search_center            (default 0 0 0)
//   --search-size x y         grasp_search_size_x/y in cm WITHOUT the border; the client adds 14 (client.cpp:183-184)
//   --approach x y z          gripper_approach_vector        (default 0 0 1)
//   --max-time s              max_calculation_time           (default 50)
//   --show-only-best          show_only_best_grasp
//   --gripper-width w         gripper_width                  (default 1)
// plus the engine's generalisations: --grid N, --rolls N, --roll-step deg.
#include "../../include/hafgrasp.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static void usage()
{
    fprintf(stderr,
            "usage: haf_grasp_cli --features F --range R --model M [options] cloud.pcd [cloud2.pcd ...]\n"
            "  --center x y z  --search-size x y  --approach x y z  --max-time s  --show-only-best  --gripper-width w\n"
            "  --grid N  --rolls N  --roll-step deg  --device d  --per-roll\n");
}

int main(int argc, char **argv)
{
    haf_config cfg;
    haf_config_default(&cfg);
    haf_grasp_input in;
    haf_grasp_input_default(&in);
    double sx = 18, sy = 30;                       // launch defaults (launch/haf_grasping_all.launch:25-65)
    bool per_roll = false;
    std::string features, range, model;
    int first_cloud = argc;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { usage(); exit(2); } };
        if (a == "--features") { need(1); features = argv[++i]; }
        else if (a == "--range") { need(1); range = argv[++i]; }
        else if (a == "--model") { need(1); model = argv[++i]; }
        else if (a == "--center") { need(3); for (int k = 0; k < 3; k++) in.grasp_area_center[k] = atof(argv[++i]); }
        else if (a == "--search-size") { need(2); sx = atof(argv[++i]); sy = atof(argv[++i]); }
        else if (a == "--approach") { need(3); for (int k = 0; k < 3; k++) in.approach_vector[k] = atof(argv[++i]); }
        else if (a == "--max-time") { need(1); in.max_calculation_time = atof(argv[++i]); }
        else if (a == "--show-only-best") in.show_only_best_grasp = 1;
        else if (a == "--gripper-width") { need(1); in.gripper_opening_width = atoi(argv[++i]); }
        else if (a == "--grid") { need(1); cfg.grid_h = cfg.grid_w = atoi(argv[++i]); }
        else if (a == "--rolls") { need(1); cfg.n_rolls = atoi(argv[++i]); }
        else if (a == "--roll-step") { need(1); cfg.roll_step_deg = atoi(argv[++i]); }
        else if (a == "--device") { need(1); cfg.device = atoi(argv[++i]); }
        else if (a == "--per-roll") per_roll = true;
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { first_cloud = i; break; }
    }
    if (features.empty() || range.empty() || model.empty() || first_cloud >= argc) { usage(); return 2; }
    in.grasp_area_length_x = (float)(sx + 14);     // client.cpp:183-184
    in.grasp_area_length_y = (float)(sy + 14);
    cfg.feature_file = features.c_str();
    cfg.range_file = range.c_str();
    cfg.model_file = model.c_str();
    cfg.max_points = 1 << 22;

    haf_engine *eng = nullptr;
    if (haf_create(&cfg, &eng) != HAF_OK) { fprintf(stderr, "haf_create: %s\n", haf_last_error(nullptr)); return 1; }
    int rc = 0;
    for (int i = first_cloud; i < argc; i++) {
        float *xyz = nullptr;
        size_t n = 0;
        char err[256];
        if (haf_pcd_load(argv[i], &xyz, &n, err, sizeof err) != HAF_OK) { fprintf(stderr, "%s: %s\n", argv[i], err); rc = 1; continue; }
        haf_cloud cloud = {xyz, n, 3, 0};
        haf_grasp_output out;
        if (haf_score(eng, &cloud, &in, &out) != HAF_OK) { fprintf(stderr, "%s: %s\n", argv[i], haf_last_error(eng)); rc = 1; haf_free(xyz); continue; }
        // the string the server publishes on /haf_grasping/grasp_hypothesis_with_eval (server.cpp:1384)
        printf("%d %g %g %g %g %g %g %g %g %g %g %g %g %d\n", out.eval, out.grasp_point1[0], out.grasp_point1[1], out.grasp_point1[2],
               out.grasp_point2[0], out.grasp_point2[1], out.grasp_point2[2], out.approach_vector[0], out.approach_vector[1],
               out.approach_vector[2], out.averaged_grasp_point[0], out.averaged_grasp_point[1], out.averaged_grasp_point[2],
               out.best_roll * cfg.roll_step_deg);
        fprintf(stderr, "%s: %zu points, %lld evaluations (%lld re-evaluated in fp64), best vote %d at row %d col %d roll %d\n", argv[i], n,
                (long long)out.n_evals, (long long)out.n_rechecked, out.best_vote, out.best_row, out.best_col, out.best_roll);
        if (per_roll) {
            haf_roll_record *rec = (haf_roll_record *)malloc(sizeof(haf_roll_record) * (size_t)cfg.n_rolls);
            if (haf_score_rolls(eng, 1, &cloud, &in, 0, cfg.n_rolls, rec) == HAF_OK)
                for (int r = 0; r < cfg.n_rolls; r++)
                    fprintf(stderr, "  roll %3d deg: vote %4d at (%d, %d), %d cells\n", r * cfg.roll_step_deg, rec[r].vote, rec[r].row, rec[r].col, rec[r].n_evals);
            free(rec);
        }
        haf_free(xyz);
    }
    haf_destroy(eng);
    return rc;
}
