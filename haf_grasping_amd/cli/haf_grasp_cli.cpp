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
// plus the engine's generalisations: --grid N, --rolls N, --roll-step deg, and
//   --gpus N [--shard rolls|clouds]   N GPUs of this node in ONE process through haf_create_multi: the rolls of every request
//                                     sharded with one RCCL all-gather of the roll records (default), or the clouds given on
//                                     the command line sharded with one RCCL all-reduce(max) electing the best grasp
//   --shards-per-gpu K                K shards on every GPU (they share its RCCL rank)
//   --probability                     svm_with_probability: "svm-predict -b 1" output as show_predicted_gps reads it (model with probA/probB)
//   --hypotheses                      also print the per-roll hypotheses the server publishes when show_only_best is off
//                                     (server.cpp:962-969), in its string format
#include "../../include/hafgrasp.h"

#include "shim_core.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void usage();

// --gpus N: the same requests through the multi-device front end of the C-ABI
static int run_multi(haf_config cfg, const haf_grasp_input &in, int gpus, int shards_per_gpu, const std::string &shard, int argc, char **argv,
                     int first_cloud)
{
    // shards_per_gpu > 1: every GPU carries several shards (they share its RCCL rank): also how a one-GPU machine runs sharded
    std::vector<int32_t> devices;
    for (int k = 0; k < shards_per_gpu; k++)
        for (int g = 0; g < gpus; g++) devices.push_back(g);
    gpus = (int)devices.size();
    const bool by_cloud = shard == "clouds";
    const int n_clouds = argc - first_cloud;
    if (by_cloud) cfg.max_clouds = n_clouds;
    haf_multi *m = nullptr;
    if (haf_create_multi(&cfg, devices.data(), gpus, by_cloud ? HAF_SHARD_CLOUDS : HAF_SHARD_ROLLS, &m) != HAF_OK) {
        fprintf(stderr, "haf_create_multi: %s\n", haf_multi_last_error(nullptr));
        return 1;
    }
    int32_t n_shards = 0, n_ranks = 0, ver = 0;
    haf_multi_info(m, &n_shards, &n_ranks, &ver);
    fprintf(stderr, "%d shards on %d RCCL ranks (RCCL %d), sharding %s\n", n_shards, n_ranks, ver, by_cloud ? "clouds" : "rolls");
    int rc = 0;
    std::vector<float *> xyz((size_t)n_clouds, nullptr);
    std::vector<haf_cloud> clouds((size_t)n_clouds);
    for (int i = 0; i < n_clouds; i++) {
        size_t n = 0;
        char err[256];
        if (haf_pcd_load(argv[first_cloud + i], &xyz[(size_t)i], &n, err, sizeof err) != HAF_OK) { fprintf(stderr, "%s: %s\n", argv[first_cloud + i], err); return 1; }
        clouds[(size_t)i] = haf_cloud{xyz[(size_t)i], n, 3, 0};
    }
    std::vector<haf_grasp_output> out((size_t)n_clouds);
    if (by_cloud) {
        std::vector<haf_grasp_input> ins((size_t)n_clouds, in);
        int32_t best = -1;
        if (haf_score_batch_sharded(m, n_clouds, clouds.data(), ins.data(), out.data(), &best) != HAF_OK) { fprintf(stderr, "%s\n", haf_multi_last_error(m)); rc = 1; }
        else {
            for (int i = 0; i < n_clouds; i++) printf("%s\n", hafshim::hypothesis_string(out[(size_t)i], cfg.roll_step_deg).c_str());
            fprintf(stderr, "best grasp of the batch: cloud %d (%s), vote %d\n", best, argv[first_cloud + best], out[(size_t)best].best_vote);
        }
    } else {
        for (int i = 0; i < n_clouds; i++) {
            if (haf_score_sharded(m, &clouds[(size_t)i], &in, &out[(size_t)i]) != HAF_OK) { fprintf(stderr, "%s: %s\n", argv[first_cloud + i], haf_multi_last_error(m)); rc = 1; continue; }
            printf("%s\n", hafshim::hypothesis_string(out[(size_t)i], cfg.roll_step_deg).c_str());
            fprintf(stderr, "%s: %lld evaluations, best vote %d at row %d col %d roll %d\n", argv[first_cloud + i], (long long)out[(size_t)i].n_evals,
                    out[(size_t)i].best_vote, out[(size_t)i].best_row, out[(size_t)i].best_col, out[(size_t)i].best_roll);
        }
    }
    for (float *p : xyz) haf_free(p);
    haf_destroy_multi(m);
    return rc;
}

static void usage()
{
    fprintf(stderr,
            "usage: haf_grasp_cli --features F --range R --model M [options] cloud.pcd [cloud2.pcd ...]\n"
            "  --center x y z  --search-size x y  --approach x y z  --max-time s  --show-only-best  --gripper-width w\n"
            "  --grid N  --rolls N  --roll-step deg  --device d  --per-roll  --hypotheses  --probability  --grid-out FILE\n"
            "  --gpus N [--shard rolls|clouds] [--shards-per-gpu K]\n");
}

int main(int argc, char **argv)
{
    haf_config cfg;
    haf_config_default(&cfg);
    haf_grasp_input in;
    haf_grasp_input_default(&in);
    double sx = 18, sy = 30;                       // launch defaults (launch/haf_grasping_all.launch:25-65)
    bool per_roll = false, hypotheses = false;
    std::string grid_out;                          // --grid-out FILE: the per-roll grasp grid the shim's callback delivers (979-1016)
    int gpus = 0, shards_per_gpu = 1;
    std::string shard = "rolls";
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
        else if (a == "--hypotheses") hypotheses = true;
        else if (a == "--grid-out") { need(1); grid_out = argv[++i]; }
        else if (a == "--probability") cfg.flags |= HAF_FLAG_PROBABILITY;     // svm_with_probability (server.cpp:383, 791, 831-841)
        else if (a == "--gpus") { need(1); gpus = atoi(argv[++i]); }
        else if (a == "--shard") { need(1); shard = argv[++i]; }
        else if (a == "--shards-per-gpu") { need(1); shards_per_gpu = atoi(argv[++i]); if (shards_per_gpu < 1) shards_per_gpu = 1; }
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

    if (gpus > 0) return run_multi(cfg, in, gpus, shards_per_gpu, shard, argc, argv, first_cloud);

    haf_engine *eng = nullptr;
    if (haf_create(&cfg, &eng) != HAF_OK) { fprintf(stderr, "haf_create: %s\n", haf_last_error(nullptr)); return 1; }
    int rc = 0;
    for (int i = first_cloud; i < argc; i++) {
        float *xyz = nullptr;
        size_t n = 0;
        char err[256];
        if (haf_pcd_load(argv[i], &xyz, &n, err, sizeof err) != HAF_OK) { fprintf(stderr, "%s: %s\n", argv[i], err); rc = 1; continue; }
        haf_cloud cloud = {xyz, n, 3, 0};
        // the goal goes the way the ROS adapter sends it: GoalFields -> hafshim::run_goal (ros_shim/shim_core.h).  stdout gets
        // what the server publishes on /haf_grasping/grasp_hypothesis_with_eval: with --hypotheses every roll's own
        // hypothesis first (server.cpp:962-969), always the overall best last (390 -> 1384, 1419)
        hafshim::GoalFields goal;
        for (int k = 0; k < 3; k++) { goal.center[k] = in.grasp_area_center[k]; goal.approach_vector[k] = in.approach_vector[k]; }
        goal.length_x = in.grasp_area_length_x; goal.length_y = in.grasp_area_length_y;
        goal.max_calculation_time = in.max_calculation_time;
        goal.show_only_best_grasp = in.show_only_best_grasp != 0;
        goal.gripper_opening_width = in.gripper_opening_width;
        std::vector<std::string> lines;
        hafshim::ResultFields res;
        haf_grasp_output out;
        std::string serr;
        FILE *gf = grid_out.empty() ? nullptr : fopen(grid_out.c_str(), i == first_cloud ? "w" : "a");
        hafshim::GridFn on_grid;
        if (gf) on_grid = [&](int roll, const std::vector<hafshim::GridCell> &cells) {
            for (const hafshim::GridCell &c : cells)
                fprintf(gf, "%d %d %d %.9g %.9g %.9g %.9g\n", roll, c.row, c.col, c.x, c.y, c.z, c.value);
        };
        const int grc = hafshim::run_goal(eng, cfg, goal, cloud, [&](const std::string &l) { lines.push_back(l); }, &res, &out, &serr, on_grid);
        if (gf) fclose(gf);
        if (grc != HAF_OK) {
            fprintf(stderr, "%s: %s\n", argv[i], serr.c_str());
            rc = 1;
            haf_free(xyz);
            continue;
        }
        for (size_t l = 0; l + 1 < lines.size(); l++)
            if (hypotheses) printf("hypothesis %s\n", lines[l].c_str());
        printf("%s\n", lines.back().c_str());
        (void)res;
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
