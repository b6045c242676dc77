// host_paths.cpp -- AddressSanitizer / UndefinedBehaviorSanitizer job for the HOST side of libhafgrasp (CPU build only: GPU
// sanitizers are not available on the test pool).  engine.cpp and parsers.cpp are compiled with -fsanitize=address,undefined
// and driven through the same entry points the library exports: the three file parsers and the PCD reader on well-formed,
// truncated, bit-flipped and hostile inputs, the per-roll geometry, the cross-roll rule and both pose functions.  No kernel is
// launched and no device is needed (tests/test_host_cpu.py builds and runs this; any sanitizer report fails the test).
#include "../../include/hafgrasp.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

extern "C" {
int haf_test_feature_table(const char *path, int *n, int *reg, float *w, int cap);
int haf_test_range_table(const char *path, double *lower, double *upper, int *max_index, double *fmin, double *fmax, unsigned char *present, int cap);
int haf_test_model(const char *path, double *gamma, double *rho, int *n_sv, int *dim, int *n_sv_class, int *label, double *coef, double *sv, long cap);
int haf_test_roll_geo(const haf_config *cfg, const haf_grasp_input *in, int roll, float *out22, float *m16, float *m16_pose);
int haf_test_finalize(const haf_config *cfg, const haf_grasp_input *in, const haf_roll_record *rec, haf_grasp_output *out);
int haf_test_roll_pose(const haf_config *cfg, const haf_grasp_input *in, const haf_roll_record *rec, int roll, haf_grasp_output *out, int32_t *published);
}

static std::string slurp(const std::string &p)
{
    std::ifstream in(p, std::ios::binary);
    std::stringstream ss;
    ss << in.rdbuf();
    return ss.str();
}
static void spit(const std::string &p, const std::string &s)
{
    std::ofstream out(p, std::ios::binary);
    out.write(s.data(), (std::streamsize)s.size());
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: host_paths <tests/golden dir> <scratch dir>\n"); return 2; }
    const std::string gold = argv[1], tmp = argv[2];
    const std::string files[4] = {gold + "/data/Features.txt", gold + "/data/range21062012_allfeatures", gold + "/surrogate.model",
                                  gold + "/data/pcd2.pcd"};
    std::vector<int> reg(400 * 16);
    std::vector<float> w(400 * 4);
    std::vector<double> fmin(70000), fmax(70000), coef(1 << 16), sv(1 << 22);
    std::vector<unsigned char> present(70000);
    auto run_parsers = [&](const std::string f[4]) {
        int n = 0, mi = 0, nsv = 0, dim = 0, cls[2], lab[2];
        double lo, up, g, r;
        (void)haf_test_feature_table(f[0].c_str(), &n, reg.data(), w.data(), 400);
        (void)haf_test_range_table(f[1].c_str(), &lo, &up, &mi, fmin.data(), fmax.data(), present.data(), 70000);
        (void)haf_test_model(f[2].c_str(), &g, &r, &nsv, &dim, cls, lab, coef.data(), sv.data(), (long)sv.size());
        float *xyz = nullptr;
        size_t np = 0;
        char err[200];
        if (haf_pcd_load(f[3].c_str(), &xyz, &np, err, sizeof err) == HAF_OK) haf_free(xyz);
    };
    run_parsers(files);                                      // well-formed
    // truncations and random byte flips of every file (seeded): the parsers may reject them, they may not read out of bounds
    std::mt19937 rng(12345);
    int cases = 0;
    for (int which = 0; which < 4; which++) {
        const std::string orig = slurp(files[which]);
        if (orig.empty()) { fprintf(stderr, "cannot read %s\n", files[which].c_str()); return 2; }
        for (int k = 0; k < 40; k++) {
            std::string s = orig;
            if (k < 12) s.resize(orig.size() * (size_t)(k + 1) / 14);            // cut somewhere
            else
                for (int flips = 0; flips < 1 + k % 7; flips++) {
                    const size_t pos = rng() % std::min<size_t>(s.size(), which == 3 ? 400 : s.size());   // PCD: damage the header
                    s[pos] = (char)(rng() & 0xFF);
                }
            std::string f[4] = {files[0], files[1], files[2], files[3]};
            f[which] = tmp + "/fuzz_" + std::to_string(which) + "_" + std::to_string(k);
            spit(f[which], s);
            run_parsers(f);
            cases++;
        }
    }
    // geometry, cross-roll rule and poses on ordinary and degenerate requests
    haf_config cfg;
    haf_config_default(&cfg);
    std::vector<haf_roll_record> rec((size_t)cfg.n_rolls);
    for (int r = 0; r < cfg.n_rolls; r++) rec[(size_t)r] = haf_roll_record{40 + 7 * r, (int16_t)(20 + r), (int16_t)(30 - r), 0.2f + 0.01f * r, 100 + r};
    const double avs[4][3] = {{0, 0, 1}, {0, 0, -1}, {0.3, -0.2, 0.9}, {1, 0, 0}};
    for (int a = 0; a < 4; a++)
        for (int best = 0; best < 2; best++) {
            haf_grasp_input in;
            haf_grasp_input_default(&in);
            for (int k = 0; k < 3; k++) in.approach_vector[k] = avs[a][k];
            in.show_only_best_grasp = best;
            in.gripper_opening_width = 1 + a;
            float out22[22], m16[16], m16p[16];
            for (int r = 0; r < cfg.n_rolls; r++) (void)haf_test_roll_geo(&cfg, &in, r, out22, m16, m16p);
            haf_grasp_output out;
            (void)haf_test_finalize(&cfg, &in, rec.data(), &out);
            for (int r = -1; r <= cfg.n_rolls; r++) {
                int32_t pub = 0;
                (void)haf_test_roll_pose(&cfg, &in, rec.data(), r, &out, &pub);
            }
        }
    haf_grasp_input in0;
    haf_grasp_input_default(&in0);
    in0.gripper_opening_width = 0;                           // singular transform: must come back as an error, not a crash
    haf_grasp_output out0;
    const int rc0 = haf_test_finalize(&cfg, &in0, rec.data(), &out0);
    printf("sanitizer job ok: %d damaged files parsed without a report; singular transform -> %d\n", cases, rc0);
    return rc0 == HAF_E_ARG ? 0 : 1;
}
