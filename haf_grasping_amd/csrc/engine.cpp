// engine.cpp -- host side of libhafgrasp.so: the C-ABI of include/hafgrasp.h over the gfx950 kernels.
//
// What stays on the host, and why: the per-roll 4x4 transform and the rotated-rectangle scalars of pnt_in_box
// (a few dozen fp32 operations per roll that use glibc sinf/cosf/atan2f exactly as the reference does), the
// sequential cross-roll rule, and the final grasp pose (once per goal).  Everything that scales with points, cells
// or support vectors runs in the .hip translation units next to this file.  There is no CPU implementation of those stages in this library.
#include "../../include/hafgrasp.h"
#include "kernels.h"
#include "parsers.h"
#include "decq.h"
#include "engine_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace haf;

namespace {

// Environment switches for experiments and for the tests that force the recheck tiers exist in the TESTING build only
// (libhafgrasp_testing.so, -DHAF_TESTING): the guard bands are what makes the fast tiers give libsvm's labels, and a stray
// variable must not be able to scale them in the library a server links.
#ifdef HAF_TESTING
const char *test_env(const char *name) { return getenv(name); }
#else
const char *test_env(const char *) { return nullptr; }
#endif

constexpr double kPi = 3.141592653;   // server.cpp:94 -- the reference's truncated constant, NOT M_PI

thread_local std::string g_create_error;

struct Mat4 {
    float a[4][4];
    static Mat4 identity()
    {
        Mat4 m;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.a[i][j] = (i == j) ? 1.0f : 0.0f;
        return m;
    }
};

// fp32 product, inner sum in index order, unfused (this TU is built with -ffp-contract=off).  Eigen's evaluation
// order for `A*B*C*D*E*F` is not pinned by the reference; this is the definition of record.
Mat4 operator*(const Mat4 &l, const Mat4 &r)
{
    Mat4 o;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s = l.a[i][0] * r.a[0][j];
            s = s + l.a[i][1] * r.a[1][j];
            s = s + l.a[i][2] * r.a[2][j];
            s = s + l.a[i][3] * r.a[3][j];
            o.a[i][j] = s;
        }
    return o;
}

struct NormalisedInput {
    double av[3];      // approach vector after server.cpp:270-273
    int sx, sy;        // grasp_search_area_size_{x,y}_dir (266-267)
    int width;         // gripper_opening_width (281)
};

NormalisedInput normalise(const haf_grasp_input &in)
{
    NormalisedInput n;
    float len = (float)std::sqrt(in.approach_vector[0] * in.approach_vector[0] + in.approach_vector[1] * in.approach_vector[1] +
                                 in.approach_vector[2] * in.approach_vector[2]);
    for (int k = 0; k < 3; k++) n.av[k] = in.approach_vector[k] / len;
    n.sx = (int)in.grasp_area_length_x;
    n.sy = (int)in.grasp_area_length_y;
    n.width = in.gripper_opening_width;
    return n;
}

// mat_transform of generate_grid (423-483) when from_float_av, of transform_gp_in_wcs_and_publish (1276-1334) otherwise:
// the two differ in whether atan2/sqrt see the float copy of the approach vector or the double message fields.
Mat4 roll_transform(const haf_config &cfg, const haf_grasp_input &in, const NormalisedInput &n, int roll, bool from_float_av,
                    Mat4 *pre_roll = nullptr, float *roll_cs = nullptr)
{
    Mat4 scale = Mat4::identity(), to_orig = Mat4::identity(), rot_z = Mat4::identity(), rot_x = Mat4::identity(),
         from_orig = Mat4::identity(), rot = Mat4::identity();
    scale.a[0][0] = (float)n.width;
    to_orig.a[0][3] = (float)(-in.grasp_area_center[0]);
    to_orig.a[1][3] = (float)(-in.grasp_area_center[1]);
    to_orig.a[2][3] = (float)(-in.grasp_area_center[2]);
    from_orig.a[2][3] = 0 + cfg.z_shift;
    float about_z, about_x = 0;
    if (from_float_av) {
        float x = (float)n.av[0], y = (float)n.av[1], z = (float)n.av[2];
        if (y == 0 && x == 0) {
            about_z = 0;
            about_x = (z >= 0) ? 0.0f : (float)kPi;
        } else {
            about_z = (float)(90 * kPi / 180.0 - std::atan2(y, x));                       // float overloads
            about_x = (float)(90 * kPi / 180.0 - std::atan2(z, std::sqrt(y * y + x * x)));
        }
    } else {
        double x = n.av[0], y = n.av[1], z = n.av[2];
        if (y == 0 && x == 0) {
            about_z = 0;
            about_x = (z >= 0) ? 0.0f : (float)kPi;
        } else {
            about_z = (float)(90 * kPi / 180.0 - std::atan2(y, x));
            about_x = (float)(90 * kPi / 180.0 - std::atan2(z, std::sqrt(y * y + x * x)));
        }
    }
    float angle = (float)(roll * cfg.roll_step_deg * kPi / 180);
    rot.a[0][0] = std::cos(angle); rot.a[0][1] = -std::sin(angle);
    rot.a[1][0] = std::sin(angle); rot.a[1][1] = std::cos(angle);
    rot_z.a[0][0] = std::cos(about_z); rot_z.a[0][1] = -std::sin(about_z);
    rot_z.a[1][0] = std::sin(about_z); rot_z.a[1][1] = std::cos(about_z);
    rot_x.a[1][1] = std::cos(about_x); rot_x.a[1][2] = -std::sin(about_x);
    rot_x.a[2][1] = std::sin(about_x); rot_x.a[2][2] = std::cos(about_x);
    if (pre_roll) *pre_roll = from_orig * rot_x * rot_z * to_orig;   // (only a spatial pre-sort key for the binning kernels)
    if (roll_cs) { roll_cs[0] = std::cos(angle); roll_cs[1] = std::sin(angle); roll_cs[2] = (float)n.width; }
    return scale * rot * from_orig * rot_x * rot_z * to_orig;
}

void fill_roll_geo(const haf_config &cfg, const haf_grasp_input &in, const NormalisedInput &n, int roll, RollGeo &g, float *m0 = nullptr)
{
    Mat4 pre;
    float cs[3];
    Mat4 m = roll_transform(cfg, in, n, roll, true, &pre, cs);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) g.m[i * 4 + j] = m.a[i][j];
    g.rc = cs[0]; g.rs = cs[1]; g.rw = cs[2];
    if (m0) for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) m0[i * 4 + j] = pre.a[i][j];
    // pnt_in_box scalars, server.cpp:679-696, with the reference's float/double mix
    const float boxrot_angle_init = 0.0f;                 // never assigned in the reference; zero in practice
    float alpha_deg = (float)(-roll * cfg.roll_step_deg - boxrot_angle_init * 180 / kPi);
    float alpha = (float)(alpha_deg * kPi / 180);
    float cx = (float)(cfg.grid_h / 2), cy = (float)(cfg.grid_h / 2);
    float boarder = 7.0f;
    float height_r = n.sx / 2 - boarder;
    float width_r = n.sy / 2 - boarder;
    g.sa = std::sin(alpha);
    g.ca = std::cos(alpha);
    g.cx1 = cx - std::sin(alpha) * height_r;
    g.cy1 = cy + std::cos(alpha) * height_r;
    g.cx2 = cx + std::sin(alpha) * height_r;
    g.cy2 = cy - std::cos(alpha) * height_r;
    g.cx3 = (float)(cx - std::sin(alpha + kPi / 2) * width_r);    // double sin/cos here (alpha + PI/2 is a double)
    g.cy3 = (float)(cy + std::cos(alpha + kPi / 2) * width_r);
    g.cx4 = (float)(cx + std::sin(alpha + kPi / 2) * width_r);
    g.cy4 = (float)(cy - std::cos(alpha + kPi / 2) * width_r);
    g.pad = 0;
}

// 4x4 inverse: Gauss-Jordan with partial pivoting in double, rounded to float (Eigen's inverse() order is unpinned)
bool invert(const Mat4 &m, Mat4 &inv)
{
    double w[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { w[i][j] = m.a[i][j]; w[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        for (int r = c + 1; r < 4; r++) if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; j++) std::swap(w[piv][j], w[c][j]);
        double d = w[c][c];
        for (int j = 0; j < 8; j++) w[c][j] /= d;
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            double f = w[r][c];
            if (f != 0.0) for (int j = 0; j < 8; j++) w[r][j] -= f * w[c][j];
        }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) inv.a[i][j] = (float)w[i][4 + j];
    return true;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        n = count;
        if (!count) return hipSuccess;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct haf_engine {
    haf_config cfg{};
    std::string feature_file, range_file, model_file;
    std::vector<FeatureRow> features;
    RangeTable range;
    SvmModel model;
    int nf = 0, kx = 0, n_sv_tiles = 0, n_sv_pad = 0, sv_tile_neg = 0;
    int gv0 = 0, gv1 = 0;
    double sum_abs_coef = 0;
    SvmParams svm{};
    ExactParams exact{};
    std::string error;

    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[HAF_ST_COUNT + 1] = {};
    float stage_ms[HAF_ST_COUNT] = {};

    long max_evals = 0, max_evals_pad = 0;
    int max_rolls = 0;      // rolls per haf_score_rolls call the buffers are sized for (cfg.max_rolls_per_call, default n_rolls)
    // Tier lists hold one entry per evaluation of the largest request (list_cap), so no request can overflow them.  flag_cap
    // is the WINDOW of the fp64 MFMA tier: what its operand image (2.6 KB per evaluation) is sized for.  A request that flags
    // more walks the list window by window (decide(), below): slower, never an error.
    int list_cap = 0;
    int flag_cap = 0;
    int flag0_cap = 0;      // screening pass: evaluations that go on to the three-pass kernel
    bool screen_active = true;   // default mode only: cleared (for good) once more than 60 % of a call's evaluations fell inside
                                 // the screening band even with the measured |w|_2 -- for such a model the single pass is wasted work
    // which form of the screening pass serves this model (kernels.h: SCREEN_*): chosen at creation (calibrate()) and re-chosen
    // by the adaptive rule when a call leaves too much undecided.  PLAIN: |w|_2 through its bound; SUMSQ: |w|_2 measured
    // (ill-conditioned models: large coefficients whose kernel values are small); CR_EXP / CR_POLY: the centred-remainder
    // form (round 4: trained models with a large C, whose decisions are 1e-5..1e-8 of sum|coef|K)
    int screen_variant = SCREEN_PLAIN;
    bool cr_available = false;   // the centred-remainder tables exist (screen_cr, d_svt0_cr, ...)
    bool variant_forced = false; // testing build: HAF_SCREEN_VARIANT pins the variant (no adaptive rule)
    bool variant_settled = false;   // every form has been seen (at calibration or on requests) and the engine has chosen: no more switching
    // "tier 0b": behind the PLAIN / SUMSQ form, the centred-remainder form (SCREEN_CR_EXP) runs once more on the first pass's LIST --
    // a few per cent of the evaluations at the price of a per cent of the first pass -- when calibration saw it decide much more.
    // t1_skip: what the screening passes leave goes straight to the exact tiers (tier 1's band has a worst-case floor since round 4 --
    // 76 u of sum|x s| -- and decides little of what a centred-remainder pass could not: measured at calibration)
    bool use_t0b = false, t1_skip = false;
    DevBuf<int> d_flag0b_list;
    double variant_share[SCREEN_VARIANTS] = {-1.0, -1.0, -1.0, -1.0};   // undecided share of each variant on the calibration scene (-1: not tried)
    ScreenParams screen{};
    ScreenParams screen_cr{};    // the centred-remainder form's constants and descriptor tables
    CrParams crp{};
    // tier 1 (three-pass list kernel) in the centred-remainder form, behind SCREEN_CR_POLY: its own SV images (s - m), the centre /
    // linear-term table of the exact-form feature kernel, L per list slot
    bool t1_cr_available = false;
    CrT1Params crt1{};
    DevBuf<char> d_svt_h_cr;
    DevBuf<double> d_t1_tab, d_t1_L;
    size_t cells_cap = 0;   // B*R*H*W

    // ONE input block per request: [CloudDev x B][RollGeo x B*R][host clouds' points], packed at call time so that a single
    // host-to-device copy carries everything (a small request is bound by the number of stream operations); the
    // pinned staging block h_in has the same layout
    DevBuf<char> d_in;
    char *h_in = nullptr;
    size_t in_hdr_cap = 0;          // bytes reserved for the two header arrays
    // ONE output block: [counters][roll records], fetched with a single device-to-host copy (d_counters / d_rec point into it)
    DevBuf<char> d_out;
    char *h_out = nullptr;
    bool counters_clean = false;    // the counters were zeroed behind the previous request's copy-out (off the next request's critical path)
    DevBuf<float> d_sorted;         // bucket-sorted copy of the clouds (binning of large grids, prestages.hip)
    DevBuf<int> d_bkt;              // 3 x max_clouds x kBktInts bucket counters / offsets / cursors
    int bkt_ints = 0;
    DevBuf<int> d_heights;          // ordered keys during binning, fp32 heights afterwards
    DevBuf<double> d_rowsum;        // integral image: band totals of the parallel form / row sums of the sequential fallback
    DevBuf<int> d_inexact;          // per (cloud, roll): the parallel integral image was not exact -> sequential order (prestages.hip)
    DevBuf<float> d_ii;
    DevBuf<uint8_t> d_mask;
    DevBuf<int> d_rowcount, d_rowoff, d_brcount, d_evalcell, d_flag_list, d_flag2_list;
    struct View { int *p = nullptr; } d_counters;      // inside d_out
    DevBuf<float> d_X, d_ax, d_dec, d_svt;
    DevBuf<char> d_svt_h;            // split-fp16 SV tile images
    DevBuf<char> d_svt0;             // screening-pass SV tile images
    DevBuf<char> d_svt0_cr;          // the same for the centred-remainder form: fp16(w_n - mu), t_n = 0, coefficient b_n
    DevBuf<FeatDesc> d_fd_slot_cr;
    DevBuf<ScrDesc> d_sd_cr;
    DevBuf<ScrDesc3> d_sd3_cr;
    DevBuf<ScrCorr> d_corr_cr;
    DevBuf<float> d_X1, d_ax1, d_gband;   // three-pass operand images / a_x of the screened-out rest; per-evaluation guard band
    DevBuf<int> d_flag0_list;
    DevBuf<unsigned long long> d_flag0_words;   // one bit per evaluation: undecided by the screening pass
    DevBuf<int> d_flag0_wgcount;                // popcounts per 256 words, for the ordered compaction
    DevBuf<int8_t> d_labels;
    DevBuf<double> d_dec_exact, d_dec_exact2, d_sv64, d_coef64, d_x64, d_part64;
    DevBuf<double> d_strict_terms;   // strict tier, spread form: kStrictSlots x n_sv_pad products coef K (launch_recheck_known)
    // tier 2a, the exact-integer tier (exact8.hip): int8 digit images of the support vectors, its hand-over list to the fp64 MFMA
    // tier and that tier's decision values for it (d_dec_exact then holds tier 2a's values, in the order of d_flag_list)
    DevBuf<char> d_sv_i8;
    DevBuf<int> d_flagi_list;
    DevBuf<double> d_dec_exacti;
    I8Params i8{};
    bool i8_active = false;
    int last_flaggedi = 0;          // evaluations that entered the fp64 MFMA tier in the last call
    bool last_i8 = false;           // the last call ran tier 2a (then d_dec_exact holds ITS values and d_dec_exacti the fp64 tier's)
    DevBuf<short> d_ev16;
    DevBuf<float> d_margin;         // HAF_FLAG_KEEP_DEBUG, default mode: |dec^| / band of every evaluation the screening tier decided
    DevBuf<AttrRecord> d_attr;      // HAF_FLAG_KEEP_DEBUG: [max_evals][kKP] attribute records of the exact-form feature kernels
    struct RecView { RollRecordDev *p = nullptr; } d_rec;   // inside d_out, behind the counters
    DevBuf<unsigned long long> d_topkey;
    DevBuf<int> d_rowmax;           // best vote per grid row (k_vote_cells -> k_vote_pick)
    // probability-output mode (HAF_FLAG_PROBABILITY, prob.hip): per-cell value of the cell's own output line, the grid
    // show_predicted_gps builds from them, the fp32 votes, and the two "%g" probabilities per evaluation
    DevBuf<float> d_own, d_gridf, d_evf;
    DevBuf<double> d_ptext;
    ProbParams prob{};
    bool prob_mode = false;
    DevBuf<FeatDesc> d_fd, d_fd_slot;
    DevBuf<ScrDesc> d_sd;
    DevBuf<ScrCorr> d_corr;         // per-slot constants of the centred screening band
    DevBuf<double> d_part1;
    long part1_stride = 0;
    // requests with at least this many evaluation slots take the thread-per-evaluation feature kernel: its floor is one thread's
    // chain of 324 attributes (~0.2 ms), the cooperative kernel costs ~1.3 us per 1000 evaluations (crossover measured at ~3e5)
    long large_evals = 1L << 18;
    DevBuf<ScrDesc3> d_sd3;

    // pinned host staging (views into h_in / h_out; the input views are set per request)
    RollRecordDev *h_rec = nullptr;
    int *h_counters = nullptr;
    // requests whose whole SVM work (evaluations x support vectors) is at most this go straight to tier 2's arithmetic in one
    // launch (k_small_direct): cheaper than a feature kernel, a fast contraction and the rechecks behind it (C2: 3 760 x 172 in
    // 36 us against 21 + 30 + 30 us; measured the other way round at C3's 31 093 x 172: 203 us against 186)
    long direct_work = 1L << 21;
    // strict tier: an evaluation whose libsvm-order decision value is within this of zero is decided on the HOST with glibc's exp
    // (the device's exp may differ from it in the last bit: 2^-52 per kernel value, i.e. at most 2^-52 sum|coef| in the sum)
    double host_exp_thr = 0.0;
    int last_host_resolved = 0;
    bool calibrated = false;        // the screening variant was chosen at creation (calibrate())
    double mfma_kappa = 12.0;       // error of one v_mfma_f32_16x16x32_f16 in units of 2^-24 (|c| + sum|a b|): max(12, 1.5 x probe_mfma_rounding())
    double mfma_kappa16 = 12.0;     // the same for v_mfma_f32_16x16x16f16 (the K tail of the three-pass kernel)
    double mfma_kappa_measured = 0.0, mfma_kappa16_measured = 0.0;
    bool no_bucket_sort = false;    // set (for good) when a tile of the bucket-sorted binning path overflowed its candidate list
    bool no_fused_pre = false;      // testing build: HAF_NO_FUSED_PRE keeps the separate pre-stage kernels on small grids too

    std::vector<std::pair<const char *, size_t>> host_regs;   // haf_register_host_cloud: page-locked caller buffers

    // last call
    int last_B = 0, last_R = 0, last_roll_first = 0;
    int last_evals = 0, last_flagged = 0, last_flagged2 = 0, last_flagged0 = 0, last_inexact = 0;
    bool last_screened = false;     // the last call's labels came through the screening tier (not its three-pass fallback)
    std::vector<haf_grasp_input> last_inputs;
};

namespace {

#define HIPCHK(e, call)                                                                                   \
    do {                                                                                                  \
        hipError_t err__ = (call);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            (e)->error = std::string(#call) + ": " + hipGetErrorString(err__);                            \
            return HAF_E_DEVICE;                                                                          \
        }                                                                                                 \
    } while (0)

// Cost model of the screening pass's forms, in units of the plain kernel's time per evaluation (measured at C5, nSV 4096: plain 14.1 ms,
// SUMSQ 15.8, CR_EXP 15.5, CR_POLY 16.3); an undecided evaluation costs ~8.5 screened ones in the three-pass tier and the exact tiers
// behind it (seed 11 of the bench generator: 5.8 ms for 378 k evaluations against 14.1 ms for 7.9 M)
constexpr double kVariantCost[SCREEN_VARIANTS] = {1.0, 1.12, 1.10, 1.16};
constexpr double kUndecidedCost = 8.5;

constexpr int kStrictSlots = 64;     // evaluations per pass of the strict tier's spread form (a few per request reach it at most)

constexpr size_t kCntBytes = (CNT_COUNT * sizeof(int) + 15) / 16 * 16;      // the counters' share of the output block (d_out)

// contraction mode: default = screening pass + three-pass refinement; HAF_FLAG_SPLIT_F16 = three passes for everything;
// HAF_FLAG_FP32_MFMA = one fp32 MFMA pass for everything
enum { MODE_SCREEN = 0, MODE_SPLIT = 1, MODE_F32 = 2 };
int contraction_mode(const haf_config &c)
{
    if (c.flags & HAF_FLAG_FP32_MFMA) return MODE_F32;
    if (c.flags & HAF_FLAG_SPLIT_F16) return MODE_SPLIT;
    return MODE_SCREEN;
}

int fail(haf_engine *e, int code, const std::string &msg)
{
    e->error = msg;
    return code;
}

// ---- no C++ exception may cross the C-ABI: a corrupt input file or an exhausted host must come back as a status the ROS
// shim can turn into setAborted(), not as std::terminate() of the action server ----
template <class F> int guarded(std::string *err, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        if (err) *err = "out of host memory";
    } catch (const std::exception &ex) {
        if (err) *err = std::string("internal error: ") + ex.what();
    } catch (...) {
        if (err) *err = "internal error (unknown exception)";
    }
    return HAF_E_INTERNAL;
}


// What show_predicted_gps (server.cpp:831-841) makes of the FIRST line of "svm-predict -b 1" output, the header
// "labels a b" (svm-predict.c:60-64), which the getline in front of its loops hands to the first masked cell:
// res = (int)atof("la") = 0, prob = atof(" a b") = a, value res*prob = 0 with the sign of a.
static float header_grid_value(int label0, int label1)
{
    char line[96];
    snprintf(line, sizeof line, "labels %d %d", label0, label1);
    const std::string ln(line);
    const int res = (int)atof(ln.substr(0, 2).c_str());
    int start = (int)ln.find(" ", 0), end = (int)ln.find(" ", (size_t)start + 1);
    if (res > 0) { start = end; end = (int)ln.find(" ", (size_t)start + 1); }
    const float prob = (float)atof(ln.substr((size_t)start, (size_t)end).c_str());
    return res * prob;
}

int label_grid_value(int label)
{
    char buf[32];
    snprintf(buf, sizeof buf, "%g", (double)label);   // what svm-predict prints (svm-predict.c:127)
    buf[2] = 0;                                       // line.substr(0,2) (server.cpp:843)
    return atoi(buf);
}

// Upper bound of the largest singular value of the n x d matrix M (row-major): sigma^2 = lambda_max(M'M) <=
// (trace (M'M)^(2^j))^(1/2^j) with j = 7 squarings, i.e. at most d^(1/128) (4.6 % for d = 324) above the true value.  Each
// squaring is normalised by its trace; the fp64 roundings of the products (~1e-13 relative) are covered by the final 1e-9.
double sigma_upper_bound(const double *M, int n, int d)
{
    std::vector<double> G((size_t)d * d, 0.0), T((size_t)d * d);
    for (int r = 0; r < n; r++) {
        const double *row = M + (size_t)r * d;
        for (int k = 0; k < d; k++) {
            const double rk = row[k];
            if (rk == 0.0) continue;
            double *g = G.data() + (size_t)k * d;
            for (int l = k; l < d; l++) g[l] += rk * row[l];
        }
    }
    for (int k = 0; k < d; k++) for (int l = 0; l < k; l++) G[(size_t)k * d + l] = G[(size_t)l * d + k];
    double log_scale = 0.0, pw = 1.0;
    for (int it = 0; it < 7; it++) {
        double tr = 0.0;
        for (int k = 0; k < d; k++) tr += G[(size_t)k * d + k];
        if (!(tr > 0.0)) return 0.0;
        for (auto &x : G) x /= tr;
        log_scale += std::log(tr) / pw;
        std::fill(T.begin(), T.end(), 0.0);
        for (int i = 0; i < d; i++)
            for (int k = 0; k < d; k++) {
                const double a = G[(size_t)i * d + k];
                if (a == 0.0) continue;
                const double *gk = G.data() + (size_t)k * d;
                double *ti = T.data() + (size_t)i * d;
                for (int j = 0; j < d; j++) ti[j] += a * gk[j];
            }
        G.swap(T);
        pw *= 2.0;
    }
    double tr = 0.0;
    for (int k = 0; k < d; k++) tr += G[(size_t)k * d + k];
    if (!(tr > 0.0)) return 0.0;
    return std::sqrt(std::exp(log_scale + std::log(tr) / pw)) * (1.0 + 1e-9);
}

int build_tables(haf_engine *e)
{
    const haf_config &c = e->cfg;
    const int ld = c.grid_w + 1;
    e->nf = (int)e->features.size();
    if (e->nf > kKP) return fail(e, HAF_E_ARG, "feature file has more than 324 rows; this build's contraction kernel is sized for 324 attributes");
    if (e->model.dim > kKP) return fail(e, HAF_E_ARG, "model attribute dimension exceeds 324");
    e->kx = std::max(e->nf, e->model.dim);

    std::vector<FeatDesc> fd((size_t)e->nf);
    for (int f = 0; f < e->nf; f++) {
        const FeatureRow &r = e->features[(size_t)f];
        FeatDesc &d = fd[(size_t)f];
        memset(&d, 0, sizeof d);
        d.shaf = (f >= c.nr_features_without_shaf) ? 1 : 0;
        for (int k = 0; k < 3; k++) {                 // region 3 carries weight 0 in every feature: never evaluated
            int x1 = r.reg[k * 4], x2 = r.reg[k * 4 + 1], y1 = r.reg[k * 4 + 2], y2 = r.reg[k * 4 + 3];
            float w = r.w[k];
            bool skip = (w == 0.0f) || (x2 < x1) || (y2 < y1) || (x2 == 0 && y2 == 0);   // fv.cpp:155-159
            if (skip) continue;
            if (x1 < 0 || y1 < 0 || x2 > 13 || y2 > 13)
                return fail(e, HAF_E_IO, "feature region outside the 14x14 window in " + e->feature_file);
            d.active |= 1 << k;
            d.w[k] = w;
            d.off[k][0] = (x2 + 1) * ld + (y2 + 1);
            d.off[k][1] = x1 * ld + (y2 + 1);
            d.off[k][2] = (x2 + 1) * ld + y1;
            d.off[k][3] = x1 * ld + y1;
            for (int j = 0; j < 4; j++) d.offw[k][j] = (d.off[k][j] / ld) * 15 + d.off[k][j] % ld;
        }
        const int idx = f + 1;
        if (idx <= e->range.max_index && e->range.present[(size_t)idx]) {
            d.fmin = e->range.fmin[(size_t)idx];
            d.fmax = e->range.fmax[(size_t)idx];
            d.skip = (d.fmin == d.fmax) ? 1 : 0;      // svm-scale.c:336
            d.range = d.fmax - d.fmin;                // svm-scale.c:346 denominator
            d.inv_range = d.skip ? 0.0 : 1.0 / d.range;
        } else {
            // Attribute not listed in the range file: svm-scale would take min/max from the rows of each roll's
            // file (svm-scale.c:165-198).  That is data-independent only for a structurally constant feature
            // (no active region: HAF gives 0, SHAF gives -1 in every row), which svm-scale then drops.
            if (d.active != 0) {
                char msg[160];
                snprintf(msg, sizeof msg, "attribute %d is missing from the range file and is not constant; per-file ranges are not supported", idx);
                return fail(e, HAF_E_ARG, msg);
            }
            d.skip = 1;
        }
    }
    if (hipSuccess != e->d_fd.alloc(fd.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(features)");
    HIPCHK(e, hipMemcpy(e->d_fd.p, fd.data(), fd.size() * sizeof(FeatDesc), hipMemcpyHostToDevice));

    // ---- SVM images ----
    const SvmModel &m = e->model;
    const double log2e = 1.4426950408889634;
    // Tile images carry the support vectors with coefficient >= 0 first (padded to whole tiles), then the negative ones:
    // the fast path's fp32 sum may take any order, and the split lets one accumulator deliver both the decision value
    // and the guard scale sum|coef|K.  The exact recheck keeps libsvm's model order (sv64 below).
    std::vector<int> slot_of((size_t)m.n_sv);
    {
        int npos = 0;
        for (int n = 0; n < m.n_sv; n++) if (m.coef[(size_t)n] >= 0) slot_of[(size_t)n] = npos++;
        const int pos_tiles = (npos + kTile - 1) / kTile;
        int nneg = 0;
        for (int n = 0; n < m.n_sv; n++) if (!(m.coef[(size_t)n] >= 0)) slot_of[(size_t)n] = pos_tiles * kTile + nneg++;
        e->n_sv_tiles = pos_tiles + (nneg + kTile - 1) / kTile;
        e->sv_tile_neg = nneg ? pos_tiles : e->n_sv_tiles;
    }
    e->n_sv_pad = ((m.n_sv + kTile - 1) / kTile) * kTile;
    std::vector<float> svt((size_t)e->n_sv_tiles * kTileFloats, 0.0f);
    e->sum_abs_coef = 0;
    for (int n = 0; n < m.n_sv; n++) {
        const int t = slot_of[(size_t)n] / kTile, j = slot_of[(size_t)n] % kTile;
        float *tile = svt.data() + (size_t)t * kTileFloats;
        double ss = 0;
        for (int k = 0; k < m.dim; k++) {
            float s = (float)m.sv[(size_t)n * m.dim + k];
            tile[k * kTile + j] = s;
            ss += (double)s * (double)s;
        }
        tile[kKP * kTile + j] = (float)(-m.gamma * log2e * ss);
        tile[(kKP + 1) * kTile + j] = (float)m.coef[(size_t)n];
        e->sum_abs_coef += std::fabs(m.coef[(size_t)n]);
    }
    if (hipSuccess != e->d_svt.alloc(svt.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(sv tiles)");
    HIPCHK(e, hipMemcpy(e->d_svt.p, svt.data(), svt.size() * sizeof(float), hipMemcpyHostToDevice));

    if (!(e->cfg.flags & HAF_FLAG_FP32_MFMA)) {
        // split-fp16 images: s = sh + sl (fp16 each), hi image then lo image (h_image_offset), then 32 a_s and 32 coef
        std::vector<char> img((size_t)e->n_sv_tiles * kHSvTileBytes, 0);
        for (int n = 0; n < m.n_sv; n++) {
            const int t = slot_of[(size_t)n] / kTile, j = slot_of[(size_t)n] % kTile;
            char *tile = img.data() + (size_t)t * kHSvTileBytes;
            double ss = 0;
            for (int k = 0; k < m.dim; k++) {
                const float s = (float)m.sv[(size_t)n * m.dim + k];
                const _Float16 h = (_Float16)s;
                const _Float16 l = (_Float16)(s - (float)h);
                const size_t off = (size_t)h_image_offset(j, k);
                memcpy(tile + off, &h, 2);
                memcpy(tile + kHMatBytes + off, &l, 2);
                const double se = (double)((float)h + (float)l);      // what the three passes multiply
                ss += se * se;
            }
            float *tail = reinterpret_cast<float *>(tile + 2 * kHMatBytes);
            tail[j] = (float)(-m.gamma * log2e * ss);
            tail[kTile + j] = (float)m.coef[(size_t)n];
        }
        if (hipSuccess != e->d_svt_h.alloc(img.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(split sv tiles)");
        HIPCHK(e, hipMemcpy(e->d_svt_h.p, img.data(), img.size(), hipMemcpyHostToDevice));
    }

    if (contraction_mode(e->cfg) == MODE_SCREEN) {
        // ---- screening pass: K slots, operand images, and the model-wide bounds of the per-evaluation guard band ----
        ScreenParams &sp = e->screen;
        sp.c = std::sqrt(2.0 * m.gamma * log2e);
        sp.acc_rel = (kS0K / 32) * e->mfma_kappa * std::ldexp(1.0, -24);      // ten accumulating instructions, kappa u each (kernels.h)
        std::vector<FeatDesc> fd2((size_t)e->nf);
        HIPCHK(e, hipMemcpy(fd2.data(), e->d_fd.p, fd2.size() * sizeof(FeatDesc), hipMemcpyDeviceToHost));
        // Slots (kernels.h): attributes that are the same function of the window (same active regions, weights and rule) with
        // the same svm-scale range take the same value in every evaluation -- before and after both text round trips -- so
        // u_a v_a + u_b v_b = u_a (v_a + v_b): they share one K slot whose SV-side operand is the sum of their SV components.
        // Every attribute svm-scale keeps gets a slot, whether or not the model has it (it counts in |u|^2 either way); the
        // ones it drops get none.
        const int n_attr = std::min(e->nf, kKP);
        std::vector<int> slot_of_attr((size_t)kKP, -1), rep;          // rep[s] = first attribute of slot s
        std::vector<int> extra;                                         // attributes in slot s beyond the first
        auto same_feature = [&](const FeatDesc &a, const FeatDesc &b) {
            if (a.active != b.active || a.shaf != b.shaf || a.fmin != b.fmin || a.fmax != b.fmax) return false;
            for (int k = 0; k < 3; k++) {
                if (!(a.active & (1 << k))) continue;
                if (a.w[k] != b.w[k]) return false;
                for (int j = 0; j < 4; j++) if (a.off[k][j] != b.off[k][j]) return false;
            }
            return true;
        };
        for (int f = 0; f < n_attr; f++) {
            if (fd2[(size_t)f].skip) continue;
            int s = -1;
            for (size_t r = 0; r < rep.size() && s < 0; r++) if (same_feature(fd2[(size_t)rep[r]], fd2[(size_t)f])) s = (int)r;
            if (s < 0) { rep.push_back(f); extra.push_back(0); s = (int)rep.size() - 1; }
            else extra[(size_t)s]++;
            slot_of_attr[(size_t)f] = s;
        }
        const int n_slots = (int)rep.size();
        if (n_slots > kS0K) e->screen_active = false;   // more distinct attributes than the ten k-steps hold: three-pass kernel for everything
        std::vector<ScrDesc> sd_keep;
        std::vector<ScrDesc3> sd3_keep;
        std::vector<FeatDesc> fds_keep;
        double ea2_keep = 0.0;
        {
            // screening attribute u' = fma(q4, scr_mul, scr_add), scr_add = c*lower - fmin*scr_mul (feature_device.h: screen_attribute).
            // Against c*x' in exact arithmetic it is off by the 2^-52 of q4 = N * RN(10^-k) (|q4| < 1e4, the decimal path's
            // range) amplified by scr_mul, by the rounding of scr_mul times |q4 - fmin|, by the rounding of scr_add (formed in long
            // double: half an ulp of |c*lower| + |fmin*scr_mul| at most) and by the one rounding of the fma; the norm over the
            // attributes is eta_abs.
            std::vector<ScrDesc> sd((size_t)kS0K);
            memset(sd.data(), 0, sd.size() * sizeof(ScrDesc));
            std::vector<ScrDesc3> sd3((size_t)kS0K);
            memset(sd3.data(), 0, sd3.size() * sizeof(ScrDesc3));
            std::vector<FeatDesc> fds((size_t)kS0K);
            memset(fds.data(), 0, fds.size() * sizeof(FeatDesc));
            for (auto &x : fds) x.skip = 1;                          // unused slots evaluate to exactly 0
            double ea2 = 0.0;
            sp.fast_groups = 0;
            sp.extra_groups = 0;
            for (int f = 0; f < n_attr; f++) {
                FeatDesc &d = fd2[(size_t)f];
                if (d.skip) continue;
                d.scr_mul = sp.c * (e->range.upper - e->range.lower) * d.inv_range;
                d.scr_add = d.scr_mul != 0.0 ? (double)((long double)sp.c * (long double)e->range.lower - (long double)d.fmin * (long double)d.scr_mul) : 0.0;
                // x2: svm-scale's own fp64 roundings of the same expression
                const double ef = 2.0 * (std::fabs(d.scr_mul) * 1.0e-15 * (1e4 + 2.0 * std::fabs(d.fmin)) +
                                         4.5e-16 * std::fabs(sp.c * e->range.lower));
                ea2 += ef * ef;
            }
            for (int g = 0; g < kS0Groups; g++) {
                bool fast = true;
                for (int q = 0; q < 8; q++) {
                    const int sl = g * 8 + q;
                    if (sl >= n_slots || sl >= kS0K) continue;
                    FeatDesc &d = fd2[(size_t)rep[(size_t)sl]];
                    ScrDesc &sdesc = sd[(size_t)sl];
                    if (d.shaf || (d.active & ~3)) fast = false;
                    for (int k = 0; k < 2; k++) {
                        sdesc.w[k] = d.w[k];
                        for (int j = 0; j < 4; j++) sdesc.off[k * 4 + j] = ((d.off[k][j] / ld) * kBandPitch + d.off[k][j] % ld) * 4;
                    }
                    sdesc.scr_mul = d.scr_mul; sdesc.scr_add = d.scr_add;
                    sdesc.extra = (float)extra[(size_t)sl];
                    ScrDesc3 &s3 = sd3[(size_t)sl];
                    for (int k = 0; k < 3; k++) {
                        s3.w[k] = d.w[k];
                        for (int j = 0; j < 4; j++) s3.off[k * 4 + j] = ((d.off[k][j] / ld) * kBandPitch + d.off[k][j] % ld) * 4;
                    }
                    s3.shaf = d.shaf;
                    s3.scr_mul = d.scr_mul; s3.scr_add = d.scr_add;
                    s3.extra = (float)extra[(size_t)sl];
                    fds[(size_t)sl] = d;
                    fds[(size_t)sl].scr_extra = (float)extra[(size_t)sl];
                    if (extra[(size_t)sl]) sp.extra_groups |= 1ull << g;
                }
                if (fast) sp.fast_groups |= 1ull << g;
            }
            if (test_env("HAF_NO_FAST_GROUPS")) sp.fast_groups = 0;        // A/B runs and the generic-path test
            // a degenerate target range or bounds beyond the decimal path's error budget: serve the model without screening
            if (!(e->range.upper > e->range.lower) || std::fabs(e->range.lower) > 1e3 || std::fabs(e->range.upper) > 1e3) e->screen_active = false;
            sp.eta_abs = std::max(std::sqrt(ea2) * 1.01, 1e-12 * 18.0 * sp.c);
            sd_keep = sd; sd3_keep = sd3; fds_keep = fds; ea2_keep = ea2;
            HIPCHK(e, hipMemcpy(e->d_fd.p, fd2.data(), fd2.size() * sizeof(FeatDesc), hipMemcpyHostToDevice));
            if (hipSuccess != e->d_sd.alloc(sd.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(screening descriptors)");
            HIPCHK(e, hipMemcpy(e->d_sd.p, sd.data(), sd.size() * sizeof(ScrDesc), hipMemcpyHostToDevice));
            sp.sd = e->d_sd.p;
            if (hipSuccess != e->d_sd3.alloc(sd3.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(screening descriptors)");
            HIPCHK(e, hipMemcpy(e->d_sd3.p, sd3.data(), sd3.size() * sizeof(ScrDesc3), hipMemcpyHostToDevice));
            sp.sd3 = e->d_sd3.p;
            if (hipSuccess != e->d_fd_slot.alloc(fds.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(screening descriptors)");
            HIPCHK(e, hipMemcpy(e->d_fd_slot.p, fds.data(), fds.size() * sizeof(FeatDesc), hipMemcpyHostToDevice));
            sp.fd_slot = e->d_fd_slot.p;
        }
        // SV side in slot space: w_n[s] = c * sum of s_n[k] over the attributes k of slot s (a model attribute svm-scale drops
        // never reaches svm-predict's x: it multiplies 0 in libsvm too, but its square still counts in |s_n|^2)
        const int S = std::min(n_slots, kS0K);
        std::vector<double> W((size_t)m.n_sv * kS0K, 0.0);              // exact (fp64) slot-space operands
        for (int n = 0; n < m.n_sv; n++)
            for (int k = 0; k < m.dim && k < kKP; k++) {
                const int sl = slot_of_attr[(size_t)k];
                if (sl >= 0 && sl < S) W[(size_t)n * kS0K + sl] += m.sv[(size_t)n * m.dim + k] * sp.c;
            }
        sp.v_max = sp.dv_max = sp.das_max = sp.as_max = 0.0;
        std::vector<char> img((size_t)e->n_sv_tiles * kS0SvTileBytes, 0);
        std::vector<double> Wh((size_t)m.n_sv * kS0K, 0.0), Wd((size_t)m.n_sv * kS0K, 0.0), tns((size_t)m.n_sv, 0.0);
        for (int n = 0; n < m.n_sv; n++) {
            const int t = slot_of[(size_t)n] / kTile, j = slot_of[(size_t)n] % kTile;
            char *tile = img.data() + (size_t)t * kS0SvTileBytes;
            double hh = 0, dd = 0;
            for (int sl = 0; sl < kS0K; sl++) {
                const double v = W[(size_t)n * kS0K + sl];
                _Float16 h = (_Float16)(float)v;
                if (std::fabs((float)h) < kF16MinNormal) h = (_Float16)0.0f;
                memcpy(tile + h_image_offset(j, sl), &h, 2);
                const double hd = (double)(float)h;
                Wh[(size_t)n * kS0K + sl] = hd;
                Wd[(size_t)n * kS0K + sl] = hd - v;
                hh += hd * hd; dd += (hd - v) * (hd - v);
            }
            // |v_n|^2 over ALL attributes of the model (libsvm's x has 0 where svm-scale dropped an attribute, so those
            // products vanish, the SV's own square does not)
            double vv = 0;
            for (int k = 0; k < m.dim; k++) { const double v = m.sv[(size_t)n * m.dim + k] * sp.c; vv += v * v; }
            const double tn = -0.5 * vv;
            tns[(size_t)n] = tn;
            const float tf = (float)tn;
            reinterpret_cast<float *>(tile + kS0MatBytes)[j] = tf;                          // padding columns: t = 0, coef = 0
            reinterpret_cast<float *>(tile + kS0MatBytes)[kTile + j] = (float)m.coef[(size_t)n];
            sp.v_max = std::max(sp.v_max, std::sqrt(hh));
            sp.dv_max = std::max(sp.dv_max, std::sqrt(dd));
            sp.das_max = std::max(sp.das_max, std::fabs((double)tf - tn));
            sp.as_max = std::max(sp.as_max, std::fabs(tn));
        }
        // spectral norms of W^ and dW = W^ - W (n_sv x 320) for the sqrt(S) form of the band: sigma^2 = lambda_max(M'M),
        // bounded from ABOVE by (trace (M'M)^(2^j))^(1/2^j), j = 7 (at most 320^(1/128) = 4.6 % above the true value)
        sp.sigma_v = sigma_upper_bound(Wh.data(), m.n_sv, kS0K);
        sp.sigma_dv = sigma_upper_bound(Wd.data(), m.n_sv, kS0K);
        {
            double cmax = 0.0;
            for (int n = 0; n < m.n_sv; n++) cmax = std::max(cmax, std::fabs(m.coef[(size_t)n]));
            sp.sqrt_cmax = std::sqrt(cmax) * (1.0 + 1e-12);
        }
        // ---- centred form of the bilinear band term (kernels.h: ScreenParams) ----
        // Reference operand ubar = the |c_n| 2^(t_n)-weighted centroid of the fp16 support vectors in slot space (for a model whose
        // support vectors are spread evenly around the origin it is ~0 and kappa_n is the kernel value at u.w_n = 0; for a trained
        // model, whose support vectors are themselves data points, it sits where the data does).  Any ubar gives a rigorous band;
        // this one only has to be a good guess.  All model constants in fp64, rounded UP where they feed the band.
        {
            std::vector<double> ub((size_t)kS0K, 0.0), ckap((size_t)m.n_sv, 0.0), G((size_t)kS0K, 0.0), Hd((size_t)kS0K, 0.0);
            double wsum = 0.0;
            for (int n = 0; n < m.n_sv; n++) {
                const double wgt = std::fabs(m.coef[(size_t)n]) * std::exp2(tns[(size_t)n]);
                wsum += wgt;
                for (int sl = 0; sl < kS0K; sl++) ub[(size_t)sl] += wgt * Wh[(size_t)n * kS0K + sl];
            }
            if (wsum > 0.0) for (auto &x : ub) x /= wsum;
            // the kernel reads ubar as fp32: use exactly those values everywhere
            for (auto &x : ub) x = (double)(float)x;
            sp.ubar2 = 0.0;
            for (double x : ub) sp.ubar2 += x * x;
            sp.ck_max = 0.0;
            std::vector<double> DW((size_t)m.n_sv * kS0K, 0.0);
            for (int n = 0; n < m.n_sv; n++) {
                double mn = 0.0;
                for (int sl = 0; sl < kS0K; sl++) mn += ub[(size_t)sl] * Wh[(size_t)n * kS0K + sl];
                const double ck = m.coef[(size_t)n] * std::exp2(tns[(size_t)n] + mn);
                ckap[(size_t)n] = ck;
                sp.ck_max = std::max(sp.ck_max, std::fabs(ck));
                for (int sl = 0; sl < kS0K; sl++) {
                    G[(size_t)sl] += ck * Wh[(size_t)n * kS0K + sl];
                    Hd[(size_t)sl] += ck * Wd[(size_t)n * kS0K + sl];
                    DW[(size_t)n * kS0K + sl] = ck * Wh[(size_t)n * kS0K + sl];
                }
            }
            sp.sigma_dk = sigma_upper_bound(DW.data(), m.n_sv, kS0K) * (1.0 + 1e-9);
            static_assert(sizeof(ScrCorr2) == 2 * sizeof(ScrCorr) && kS0K % 2 == 0, "pair form behind the per-slot form, one buffer");
            std::vector<ScrCorr> sc((size_t)kS0K * 2);
            ScrCorr2 *sc2 = reinterpret_cast<ScrCorr2 *>(sc.data() + kS0K);
            sp.g_norm = sp.hd_norm = 0.0;
            for (int sl = 0; sl < kS0K; sl++) {
                sc[(size_t)sl].g = (float)G[(size_t)sl];
                sc[(size_t)sl].hd = (float)Hd[(size_t)sl];
                sc[(size_t)sl].ub = (float)ub[(size_t)sl];
                sc[(size_t)sl].pad = 0.0f;
                ScrCorr2 &p2 = sc2[sl >> 1];
                p2.g[sl & 1] = sc[(size_t)sl].g; p2.hd[sl & 1] = sc[(size_t)sl].hd; p2.ub[sl & 1] = sc[(size_t)sl].ub; p2.pad[sl & 1] = 0.0f;
                sp.g_norm += G[(size_t)sl] * G[(size_t)sl];
                sp.hd_norm += Hd[(size_t)sl] * Hd[(size_t)sl];
            }
            sp.g_norm = std::sqrt(sp.g_norm) * (1.0 + 1e-6);       // (also covers the fp32 rounding of the stored constants)
            sp.hd_norm = std::sqrt(sp.hd_norm) * (1.0 + 1e-6);
            sp.ck_max *= 1.0 + 1e-12;
            sp.ubar2 *= 1.0 + 1e-12;
            if (!std::isfinite(sp.sigma_dk) || !std::isfinite(sp.g_norm) || !std::isfinite(sp.hd_norm) || sp.g_norm > 1e30 ||
                test_env("HAF_SCREEN_NO_CENTRE"))
                sp.sigma_dk = INFINITY;                          // centred estimate never chosen (A/B runs; degenerate models)
            if (hipSuccess != e->d_corr.alloc(sc.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(screening corrections)");
            HIPCHK(e, hipMemcpy(e->d_corr.p, sc.data(), sc.size() * sizeof(ScrCorr), hipMemcpyHostToDevice));
            sp.corr = e->d_corr.p;
            sp.corr2 = reinterpret_cast<const ScrCorr2 *>(e->d_corr.p + kS0K);
        }
        // the bounds feed a rigorous band: round them up past their own fp64 rounding
        sp.v_max *= 1.0 + 1e-12; sp.dv_max *= 1.0 + 1e-12; sp.das_max = sp.das_max * (1.0 + 1e-12) + 1e-300;
        sp.as_max *= 1.0 + 1e-12;
        if (!(sp.v_max < 60000.0)) return fail(e, HAF_E_ARG, "support vectors too large for the fp16 screening pass; use HAF_FLAG_SPLIT_F16");
        if (hipSuccess != e->d_svt0.alloc(img.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(screening sv tiles)");
        HIPCHK(e, hipMemcpy(e->d_svt0.p, img.data(), img.size(), hipMemcpyHostToDevice));

        // ---- the centred-remainder form (kernels.h: ScreenParams::cr) ----
        // Centre mu: the |c_n| 2^(t_n)-weighted mean of the support vectors, per slot (attributes that share a slot share a centre: their
        // operands are one number).  Any centre gives the same decision function; this one puts the data near the origin, where
        // psi(z) = 2^z - 1 - z ln2 is small.
        if (e->screen_active && !test_env("HAF_NO_CR")) {
            ScreenParams &cp = e->screen_cr;
            cp = sp;
            cp.cr = 1;
            cp.cr_poly = 0;
            const double ln2 = 0.6931471805599453;
            std::vector<double> mu((size_t)kS0K, 0.0), mult((size_t)kS0K, 0.0);
            {
                double wsum = 0.0;
                std::vector<double> acc((size_t)kS0K, 0.0);
                for (int n = 0; n < m.n_sv; n++) {
                    const double wgt = std::fabs(m.coef[(size_t)n]) * std::exp2(tns[(size_t)n]);
                    wsum += wgt;
                    for (int sl = 0; sl < S; sl++) acc[(size_t)sl] += wgt * W[(size_t)n * kS0K + sl];
                }
                for (int sl = 0; sl < S; sl++) mult[(size_t)sl] = 1.0 + (double)extra[(size_t)sl];
                // W holds the SUM of a slot's attributes: the centre of one attribute is the mean over them
                if (wsum > 0.0) for (int sl = 0; sl < S; sl++) mu[(size_t)sl] = acc[(size_t)sl] / (wsum * mult[(size_t)sl]);
                if (test_env("HAF_CR_NO_CENTRE")) std::fill(mu.begin(), mu.end(), 0.0);
            }
            cp.cr_mu_norm = cp.cr_mu_norm_t = 0.0;
            for (int sl = 0; sl < S; sl++) { cp.cr_mu_norm += mu[(size_t)sl] * mu[(size_t)sl]; cp.cr_mu_norm_t += mult[(size_t)sl] * mu[(size_t)sl] * mu[(size_t)sl]; }
            cp.cr_mu_norm = std::sqrt(cp.cr_mu_norm) * (1.0 + 1e-12); cp.cr_mu_norm_t = std::sqrt(cp.cr_mu_norm_t) * (1.0 + 1e-12);
            // descriptors: the same features, scr_add - mu (formed in long double: its rounding joins eta_abs)
            std::vector<ScrDesc> sdc = sd_keep;
            std::vector<ScrDesc3> sd3c = sd3_keep;
            std::vector<FeatDesc> fdsc = fds_keep;
            double ea2c = ea2_keep;
            for (int sl = 0; sl < S; sl++) {
                const FeatDesc &d0 = fds_keep[(size_t)sl];
                const double add = d0.scr_mul != 0.0 ? (double)((long double)sp.c * (long double)e->range.lower - (long double)d0.fmin * (long double)d0.scr_mul -
                                                               (long double)mu[(size_t)sl]) : 0.0;
                if (d0.scr_mul == 0.0) mu[(size_t)sl] = 0.0;          // a slot whose attribute svm-scale drops stays 0
                sdc[(size_t)sl].scr_add = add; sd3c[(size_t)sl].scr_add = add; fdsc[(size_t)sl].scr_add = add;
                const double ef = 4.5e-16 * std::fabs(mu[(size_t)sl]);
                ea2c += ef * ef;
            }
            cp.eta_abs = std::max(std::sqrt(ea2c) * 1.01, 1e-12 * 18.0 * sp.c);
            // centred support vectors: Q (fp64), Q^ = fp16(Q), b_n = c_n 2^(-|q_n|^2/2) with |q_n|^2 over ALL attributes of the model
            std::vector<double> Q((size_t)m.n_sv * kS0K, 0.0), Qh((size_t)m.n_sv * kS0K, 0.0), b((size_t)m.n_sv, 0.0);
            std::vector<char> imgc((size_t)e->n_sv_tiles * kS0SvTileBytes, 0);
            long double B0 = 0.0L;
            std::vector<long double> gl((size_t)kS0K, 0.0L);
            cp.cr_Ca = cp.cr_Cq1 = cp.cr_Cqq = cp.cr_Babs = cp.cr_qmax = cp.cr_dqmax = 0.0;
            for (int n = 0; n < m.n_sv; n++) {
                double qq = 0.0;
                for (int k = 0; k < m.dim; k++) {
                    const int sl = (k < kKP) ? slot_of_attr[(size_t)k] : -1;
                    const double v = m.sv[(size_t)n * m.dim + k] * sp.c - ((sl >= 0 && sl < S) ? mu[(size_t)sl] : 0.0);
                    qq += v * v;
                }
                const double bn = m.coef[(size_t)n] * std::exp2(-0.5 * qq);
                b[(size_t)n] = bn;
                B0 += (long double)bn;
                const int t = slot_of[(size_t)n] / kTile, j = slot_of[(size_t)n] % kTile;
                char *tile = imgc.data() + (size_t)t * kS0SvTileBytes;
                double q2 = 0.0, h2 = 0.0, d2 = 0.0;
                for (int sl = 0; sl < S; sl++) {
                    const double q = W[(size_t)n * kS0K + sl] - mult[(size_t)sl] * mu[(size_t)sl];
                    _Float16 h = (_Float16)(float)q;
                    if (std::fabs((float)h) < kF16MinNormal) h = (_Float16)0.0f;
                    memcpy(tile + h_image_offset(j, sl), &h, 2);
                    const double hd = (double)(float)h;
                    Q[(size_t)n * kS0K + sl] = q; Qh[(size_t)n * kS0K + sl] = hd;
                    q2 += q * q; h2 += hd * hd; d2 += (hd - q) * (hd - q);
                    gl[(size_t)sl] += (long double)bn * (long double)q;
                }
                reinterpret_cast<float *>(tile + kS0MatBytes)[j] = 0.0f;                       // the chains start from 0
                reinterpret_cast<float *>(tile + kS0MatBytes)[kTile + j] = (float)bn;
                const double qn = std::sqrt(q2), qhn = std::sqrt(h2), dqn = std::sqrt(d2), ab = std::fabs(bn);
                cp.cr_Ca += ab * qhn * qn; cp.cr_Cq1 += ab * qhn; cp.cr_Cqq += ab * h2; cp.cr_Babs += ab;
                cp.cr_qmax = std::max(cp.cr_qmax, qhn); cp.cr_dqmax = std::max(cp.cr_dqmax, dqn);
            }
            // N = Q'BQ^ and M = Q'B(Q^ - Q) (320 x 320, SIGNED: the classes cancel), the unsigned second-order matrices through
            // sigma(diag(sqrt|b|) .)^2; g = sum b_n q_n
            {
                const int K = kS0K;
                std::vector<double> Nm((size_t)K * K, 0.0), Mm((size_t)K * K, 0.0), Rh((size_t)m.n_sv * K), Rd((size_t)m.n_sv * K), Ra((size_t)m.n_sv * K);
                for (int n = 0; n < m.n_sv; n++) {
                    const double *q = Q.data() + (size_t)n * K, *h = Qh.data() + (size_t)n * K;
                    const double sb = std::sqrt(std::fabs(b[(size_t)n]));
                    for (int l = 0; l < K; l++) { Rh[(size_t)n * K + l] = sb * h[l]; Rd[(size_t)n * K + l] = sb * (h[l] - q[l]); Ra[(size_t)n * K + l] = sb * std::fabs(h[l]); }
                    for (int k = 0; k < K; k++) {
                        const double a = b[(size_t)n] * q[k];
                        if (a == 0.0) continue;
                        double *nr = Nm.data() + (size_t)k * K, *mr = Mm.data() + (size_t)k * K;
                        for (int l = 0; l < K; l++) { nr[l] += a * h[l]; mr[l] += a * (h[l] - q[l]); }
                    }
                }
                for (int k = 0; k < K; k++)
                    for (int l = 0; l < k; l++) { const double sy = 0.5 * (Mm[(size_t)k * K + l] + Mm[(size_t)l * K + k]); Mm[(size_t)k * K + l] = Mm[(size_t)l * K + k] = sy; }
                // (the 1e-9 of sigma_upper_bound and the 1e-6 here cover the fp64 roundings of the accumulations above)
                cp.cr_nN = sigma_upper_bound(Nm.data(), K, K) * (1.0 + 1e-6);
                cp.cr_nM = sigma_upper_bound(Mm.data(), K, K) * (1.0 + 1e-6);
                const double sh = sigma_upper_bound(Rh.data(), m.n_sv, K), sdq = sigma_upper_bound(Rd.data(), m.n_sv, K);
                cp.cr_nHabs = sh * sh * (1.0 + 1e-6);
                cp.cr_nDabs = sdq * sdq * (1.0 + 1e-6);
                const double sa = sigma_upper_bound(Ra.data(), m.n_sv, K);
                cp.cr_nHaa = sa * sa * (1.0 + 1e-6);
            }
            double gn = 0.0;
            std::vector<ScrCorr> scc((size_t)kS0K * 2);
            ScrCorr2 *sc2 = reinterpret_cast<ScrCorr2 *>(scc.data() + kS0K);
            for (int sl = 0; sl < kS0K; sl++) {
                const double gs = (double)gl[(size_t)sl];
                gn += gs * gs;
                scc[(size_t)sl].g = 0.0f; scc[(size_t)sl].hd = (float)(ln2 * gs); scc[(size_t)sl].ub = 0.0f; scc[(size_t)sl].pad = 0.0f;
                ScrCorr2 &p2 = sc2[sl >> 1];
                p2.g[sl & 1] = 0.0f; p2.hd[sl & 1] = scc[(size_t)sl].hd; p2.ub[sl & 1] = 0.0f; p2.pad[sl & 1] = 0.0f;
            }
            cp.cr_gnorm = std::sqrt(gn) * (1.0 + 1e-6);
            for (double *x : {&cp.cr_Ca, &cp.cr_Cq1, &cp.cr_Cqq, &cp.cr_Babs, &cp.cr_qmax, &cp.cr_dqmax}) *x *= 1.0 + 1e-9;
            e->crp.B0 = (double)B0;
            e->crp.rho = m.rho;
            const bool finite = std::isfinite(cp.cr_nN) && std::isfinite(cp.cr_nM) && std::isfinite(cp.cr_nHabs) && std::isfinite(cp.cr_Babs) &&
                                std::isfinite(e->crp.B0) && cp.cr_Babs < 1e30 && cp.cr_qmax < 60000.0;
            if (finite) {
                bool ok = hipSuccess == e->d_svt0_cr.alloc(imgc.size()) && hipSuccess == e->d_sd_cr.alloc(sdc.size()) &&
                          hipSuccess == e->d_sd3_cr.alloc(sd3c.size()) && hipSuccess == e->d_fd_slot_cr.alloc(fdsc.size()) &&
                          hipSuccess == e->d_corr_cr.alloc(scc.size());
                if (!ok) return fail(e, HAF_E_DEVICE, "hipMalloc(centred-remainder tables)");
                HIPCHK(e, hipMemcpy(e->d_svt0_cr.p, imgc.data(), imgc.size(), hipMemcpyHostToDevice));
                HIPCHK(e, hipMemcpy(e->d_sd_cr.p, sdc.data(), sdc.size() * sizeof(ScrDesc), hipMemcpyHostToDevice));
                HIPCHK(e, hipMemcpy(e->d_sd3_cr.p, sd3c.data(), sd3c.size() * sizeof(ScrDesc3), hipMemcpyHostToDevice));
                HIPCHK(e, hipMemcpy(e->d_fd_slot_cr.p, fdsc.data(), fdsc.size() * sizeof(FeatDesc), hipMemcpyHostToDevice));
                HIPCHK(e, hipMemcpy(e->d_corr_cr.p, scc.data(), scc.size() * sizeof(ScrCorr), hipMemcpyHostToDevice));
                cp.sd = e->d_sd_cr.p; cp.sd3 = e->d_sd3_cr.p; cp.fd_slot = e->d_fd_slot_cr.p;
                cp.corr = e->d_corr_cr.p;
                cp.corr2 = reinterpret_cast<const ScrCorr2 *>(e->d_corr_cr.p + kS0K);
                e->cr_available = true;
                // ---- tier 1 in the same form (kernels.h: CrT1Params): hi/lo fp16 images of fl32(s - m) in raw attribute units, the
                // centre and the linear term's constants per attribute for the exact-form feature kernel ----
                if (!test_env("HAF_NO_CR_T1")) {
                    std::vector<double> tab((size_t)2 * kKP, 0.0);
                    std::vector<long double> Gr((size_t)kKP, 0.0L);
                    for (int k = 0; k < m.dim && k < kKP; k++) {
                        const int sl = slot_of_attr[(size_t)k];
                        tab[(size_t)k] = (sl >= 0 && sl < S) ? mu[(size_t)sl] / sp.c : 0.0;
                    }
                    std::vector<char> imgh((size_t)e->n_sv_tiles * kHSvTileBytes, 0);
                    std::vector<double> Ra1((size_t)m.n_sv * kKP, 0.0);          // sqrt|b_n| c |q~_nk|: the attribute-space |Q~| of cr_nHaa
                    double qmax1 = 0.0, dqmax1 = 0.0, Ca1 = 0.0, Cqq1 = 0.0, Dabs1 = 0.0;
                    for (int n = 0; n < m.n_sv; n++) {
                        const int t = slot_of[(size_t)n] / kTile, j = slot_of[(size_t)n] % kTile;
                        char *tile = imgh.data() + (size_t)t * kHSvTileBytes;
                        double q2 = 0.0, h2 = 0.0, d2 = 0.0;
                        for (int k = 0; k < m.dim; k++) {
                            const double sc = m.sv[(size_t)n * m.dim + k] - (k < kKP ? tab[(size_t)k] : 0.0);     // s - m, raw units
                            const float sf = (float)sc;
                            const _Float16 h = (_Float16)sf;
                            const _Float16 l = (_Float16)(sf - (float)h);
                            const size_t off = (size_t)h_image_offset(j, k);
                            memcpy(tile + off, &h, 2);
                            memcpy(tile + kHMatBytes + off, &l, 2);
                            const double se = (double)(float)h + (double)(float)l;
                            q2 += sc * sc; h2 += se * se; d2 += (se - sc) * (se - sc);
                            if (k < kKP) Gr[(size_t)k] += (long double)b[(size_t)n] * (long double)sc;
                            if (k < kKP) Ra1[(size_t)n * kKP + k] = std::sqrt(std::fabs(b[(size_t)n])) * sp.c * std::fabs(se);
                        }
                        float *tail = reinterpret_cast<float *>(tile + 2 * kHMatBytes);
                        tail[j] = 0.0f;
                        tail[kTile + j] = (float)b[(size_t)n];
                        const double qn = sp.c * std::sqrt(q2), qhn = sp.c * std::sqrt(h2), dqn = sp.c * std::sqrt(d2), ab = std::fabs(b[(size_t)n]);
                        qmax1 = std::max(qmax1, qhn); dqmax1 = std::max(dqmax1, dqn);
                        Ca1 += ab * qhn * qn; Cqq1 += ab * h2 * sp.c * sp.c; Dabs1 += ab * dqn * dqn;
                    }
                    for (int k = 0; k < kKP; k++) tab[(size_t)kKP + k] = ln2 * 2.0 * m.gamma * log2e * (double)Gr[(size_t)k];
                    CrT1Params &t1 = e->crt1;
                    t1.B0 = e->crp.B0; t1.rho = m.rho; t1.c = sp.c;
                    // Q~ against Q^: the signed matrices move by at most sigma(sqrt|b| Q) sigma(sqrt|b| (Q~ - Q^)); slot sums of two
                    // attributes at most double a rounding error's norm (the 2 in front of sqrt(Dabs1)); Frobenius for the spectral norm
                    const double sH = std::sqrt(cp.cr_nHabs), sD = std::sqrt(cp.cr_nDabs), sD1 = 2.0 * std::sqrt(Dabs1);
                    t1.nN = (cp.cr_nN + (sH + sD) * (sD + sD1)) * (1.0 + 1e-9);
                    t1.nM = (sH + sD) * sD1 * (1.0 + 1e-9) + 1e-300;
                    t1.nHabs = (sH + sD + sD1) * (sH + sD + sD1) * (1.0 + 1e-9);
                    t1.nDabs = sD1 * sD1 * (1.0 + 1e-9) + 1e-300;
                    { const double sa1 = sigma_upper_bound(Ra1.data(), m.n_sv, kKP); t1.nHaa = sa1 * sa1 * (1.0 + 1e-6); }
                    // (Ca, Cqq, qmax, dqmax bound sums over ATTRIBUTES -- the three passes multiply attribute by attribute -- so the
                    // attribute-space norms computed above are the right ones as they are)
                    t1.Ca = Ca1 * (1.0 + 1e-9); t1.Cqq = Cqq1 * (1.0 + 1e-9); t1.Babs = cp.cr_Babs;
                    t1.qmax = qmax1 * (1.0 + 1e-9); t1.dqmax = dqmax1 * (1.0 + 1e-9) + 1e-300;
                    // (the same worst-case floor as the plain form of this tier: guard_dot_p, below)
                    t1.acc_rel = ((test_env("HAF_KAPPA_T1_MEASURED") ? std::max(e->mfma_kappa, e->mfma_kappa16)
                                                                     : std::max(64.0, std::max(e->mfma_kappa, e->mfma_kappa16))) + 14.0) * std::ldexp(1.0, -24);
                    t1.dp_rel = (std::ldexp(1.0, -22) + std::ldexp(1.0, -24)) * 1.01;
                    t1.dp_abs = sp.c * std::sqrt((double)kKP) * std::ldexp(1.0, -25) * 1.01;
                    t1.sum_rel = (2.0 + 1.0 + 0.1 + 6.0 + 10.0 + 1.0) * std::ldexp(1.0, -24) * (1.0 + 1e-5);
                    t1.scale = 1.001;
                    if (hipSuccess != e->d_svt_h_cr.alloc(imgh.size()) || hipSuccess != e->d_t1_tab.alloc(tab.size()))
                        return fail(e, HAF_E_DEVICE, "hipMalloc(centred-remainder tier-1 tables)");
                    HIPCHK(e, hipMemcpy(e->d_svt_h_cr.p, imgh.data(), imgh.size(), hipMemcpyHostToDevice));
                    HIPCHK(e, hipMemcpy(e->d_t1_tab.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
                    e->t1_cr_available = std::isfinite(t1.nN) && std::isfinite(t1.nHabs) && std::isfinite(t1.Ca);
                }
            }
        }
    }

    // fp64 image for both recheck tiers, SVs in MODEL order: rows 0..323 attributes, row 324 |s|^2, row 325 coef
    std::vector<double> sv64((size_t)kM64Rows * e->n_sv_pad, 0.0), coef64((size_t)e->n_sv_pad, 0.0);
    double ss_max = 0;
    for (int n = 0; n < m.n_sv; n++) {
        double ss = 0;
        for (int k = 0; k < m.dim; k++) {
            const double v = m.sv[(size_t)n * m.dim + k];
            sv64[(size_t)k * e->n_sv_pad + n] = v;
            ss += v * v;
        }
        sv64[(size_t)kKP * e->n_sv_pad + n] = ss;
        sv64[(size_t)(kKP + 1) * e->n_sv_pad + n] = m.coef[(size_t)n];
        coef64[(size_t)n] = m.coef[(size_t)n];
        ss_max = std::max(ss_max, ss);
    }
    if (hipSuccess != e->d_sv64.alloc(sv64.size()) || hipSuccess != e->d_coef64.alloc(coef64.size()))
        return fail(e, HAF_E_DEVICE, "hipMalloc(fp64 model)");
    HIPCHK(e, hipMemcpy(e->d_sv64.p, sv64.data(), sv64.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->d_coef64.p, coef64.data(), coef64.size() * sizeof(double), hipMemcpyHostToDevice));
    // ---- tier 2a: support vectors as four int8 digit planes in the B-operand layout of v_mfma_i32_16x16x64_i8 (kernels.h) ----
    e->i8_active = !(c.flags & HAF_FLAG_PROBABILITY) && !test_env("HAF_NO_I8") && e->kx <= 64 * kI8Steps;
    if (e->i8_active) {
        const int n_tiles16 = e->n_sv_pad / 16;
        std::vector<char> img((size_t)n_tiles16 * kI8SvTileBytes, 0);
        double s_max2 = 0.0;
        // q_s: as many fractional bits as the largest SV component leaves room for in four digits (the attributes keep kI8Q: svm-scale
        // does not clamp, kernels.h); the quantisation of the SVs is then a small part of delta
        double sv_abs_max = 0.0;
        for (int n = 0; n < m.n_sv; n++)
            for (int k = 0; k < m.dim; k++) sv_abs_max = std::max(sv_abs_max, std::fabs(m.sv[(size_t)n * m.dim + k]));
        int qs = kI8Q;
        while (qs < 30 && (sv_abs_max * std::ldexp(1.0, qs + 1) + 1.0) <= (double)kI8Max) qs++;
        if (!(sv_abs_max < 1e30)) e->i8_active = false;
        for (int n = 0; n < m.n_sv && e->i8_active; n++) {
            char *tile = img.data() + (size_t)(n / 16) * kI8SvTileBytes;
            const int col = n % 16;
            long long ssq = 0;
            for (int k = 0; k < m.dim; k++) {
                const double sc = std::nearbyint(std::ldexp(m.sv[(size_t)n * m.dim + k], qs));
                if (!(std::fabs(sc) <= (double)kI8Max)) { e->i8_active = false; break; }      // a support vector beyond +-15.87: no tier 2a
                int t = (int)sc;
                ssq += (long long)t * t;
                int dg[4];
                dg[3] = ((t + 64) & 127) - 64; t = (t - dg[3]) >> 7;
                dg[2] = ((t + 64) & 127) - 64; t = (t - dg[2]) >> 7;
                dg[1] = ((t + 64) & 127) - 64; t = (t - dg[1]) >> 7;
                dg[0] = t;
                const int ks = k / 64, blk = (k % 64) / 16, jj = k % 16;
                for (int j = 0; j < kI8Slices; j++) tile[(size_t)(j * kI8Steps + ks) * 1024 + (blk * 16 + col) * 16 + jj] = (char)dg[j];
            }
            double *cst = reinterpret_cast<double *>(tile + kI8GroupBytes);
            cst[col] = std::ldexp((double)ssq, -2 * qs);
            cst[16 + col] = m.coef[(size_t)n];
            s_max2 = std::max(s_max2, cst[col]);
        }
        if (e->i8_active) {
            if (hipSuccess != e->d_sv_i8.alloc(img.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(int8 sv tiles)");
            HIPCHK(e, hipMemcpy(e->d_sv_i8.p, img.data(), img.size(), hipMemcpyHostToDevice));
            e->i8.gamma = m.gamma; e->i8.rho = m.rho;
            e->i8.gamma2 = m.gamma * log2e;
            // 2 x.s enters d^2: 2 * 324 attributes * (2 * 128 + 1) * 64 * 64 * 2^-(kI8Q + q_s)
            e->i8.drop = 2.0 * (double)kKP * 257.0 * 4096.0 * std::ldexp(1.0, -(kI8Q + qs)) * (1.0 + 1e-12);
            e->i8.delta = std::sqrt((double)kKP) * (std::ldexp(1.0, -(kI8Q + 1)) + std::ldexp(1.0, -(qs + 1))) * (1.0 + 1e-12);
            e->i8.dq_scale = -2.0 * std::ldexp(1.0, 14 - kI8Q - qs);
            e->i8.s_max = std::sqrt(s_max2) * (1.0 + 1e-12);
            e->i8.guard_scale = 1.0;
            if (const char *g = test_env("HAF_GUARD_I8_REL")) e->i8.guard_scale = atof(g);
            else if (test_env("HAF_GUARD2_REL")) e->i8.guard_scale = 1e30;      // a test that forces the fp64 / strict tiers means all of them
            e->i8.n_sv_pad = e->n_sv_pad;
        }
    }
    e->exact.gamma2 = m.gamma * log2e;
    e->exact.as_max1 = 1.0 + m.gamma * log2e * ss_max;
    // fp64 GEMM-form tier: worst-case error ~ 324 * 2^-53 per unit of (a_x + a_s) * sum|coef|K, i.e. < 2^-44; 2^-40 leaves 16x
    e->exact.guard2 = std::ldexp(1.0, -40);
    if (const char *g = test_env("HAF_GUARD2_REL")) e->exact.guard2 = atof(g);

    e->gv0 = label_grid_value(m.label[0]);
    e->gv1 = label_grid_value(m.label[1]);
    if (e->gv0 < -128 || e->gv0 > 127 || e->gv1 < -128 || e->gv1 > 127) return fail(e, HAF_E_ARG, "model labels out of range");
    e->prob_mode = (c.flags & HAF_FLAG_PROBABILITY) != 0;
    if (e->prob_mode) {
        // svm-predict -b 1 on a model without probA/probB exits ("Model does not support probabiliy estimates",
        // svm-predict.c:219-224) and the reference then votes on a stale or missing file; the engine says so instead
        if (!m.has_prob) return fail(e, HAF_E_ARG, "HAF_FLAG_PROBABILITY needs a model with probA and probB (svm-train -b 1)");
        e->prob.A = m.probA; e->prob.B = m.probB;
        e->prob.gv0 = e->gv0; e->prob.gv1 = e->gv1;       // (int)atof(two characters) == atoi(two characters) for "%g" of an int
        e->prob.hdr = header_grid_value(m.label[0], m.label[1]);
        e->prob.host_all = test_env("HAF_PROB_HOST_ALL") ? 1 : 0;
    }

    e->svm.two_gamma2 = (float)(2.0 * m.gamma * log2e);
    e->svm.neg_gamma2 = (float)(-m.gamma * log2e);
    e->svm.rho = (float)m.rho;
    // Guard band: a fast decision is trusted when
    //     |dec| > (guard_acc + guard_dot * (a_x + max a_s)) * sum_n |coef_n| K_n + guard_abs,   a = gamma'*|.|^2.
    // Both constants are WORST-CASE fp32 error bounds per unit of sum|coef|K:
    //   guard_dot: the 324-term fp32 fma chain of x.s, bounded through Cauchy-Schwarz (324 * 2^-24 * ln2 in K), the fp32
    //              rounding of the attributes (2 * 2^-24; 2^-22 for the fp16 hi+lo split) and the roundings of the argument;
    //   guard_acc: the fp32 part of the sum of coef*K (below), v_exp_f32 and the coefficient product (3 * 2^-23).
    // tests/diag_guard.py measures the actual error with the band disabled: 20-30x smaller.  HAF_GUARD_REL scales the band.
    double guard_scale = 1.0;
    if (const char *g = test_env("HAF_GUARD_REL")) guard_scale = atof(g);
    const double u = std::ldexp(1.0, -24);
    e->svm.guard_dot = (float)(guard_scale * (0.6932 * 324.0 * u + 8.0 * u));
    // PRECISE form of the three-pass kernel (k_svm_rbf_h<true>): every instruction of the main pass starts from zero and is off by
    // at most kappa u of its sum|products| (mfma_kappa: measured at creation, with its margin), 11 VALU adds join the instructions'
    // results (one rounding each, of at most the whole sum|x_i s_i|), one more for the small-pass chain (whose own roundings are
    // 2^-10 of that): kappa + 12 instead of 324 (kappa: the larger of the two shapes' -- the K tail is a 16-wide instruction)
    // the measured kappa may only WIDEN this tier's band.  Its floor is the worst any adder could do with 33 terms -- 32
    // additions that each lose up to an ulp (2 u: the probe shows truncating alignment, not round-to-nearest) = 64 u -- so that tier 1,
    // whose flagged evaluations are cheap since tier 2a exists, never rests on the probe's seven families alone.  (The screening tier
    // keeps the measured constant: ten instructions at 64 u would leave nothing for it to decide, and what it decides wrongly would
    // have to be wrong by 8x the largest error any of 114 688 adversarial sums showed)
    const double kappa_t1 = test_env("HAF_KAPPA_T1_MEASURED") ? std::max(e->mfma_kappa, e->mfma_kappa16) : std::max(64.0, std::max(e->mfma_kappa, e->mfma_kappa16));
    e->svm.guard_dot_p = (float)(guard_scale * (0.6932 * (kappa_t1 + 12.0) * u + 8.0 * u));
    // coefficient sum: sequential over the tiles (fp32 kernel: one fma per tile and sum register) or two-level (split-fp16
    // kernel: an inner sum takes the 2 column blocks of 8 tiles, 16 fmas, then one add per 8 tiles); +2 for the class split
    // (P and N are reduced separately), +4/5 lane-reduction steps, +6 for exp2 and the product.  All terms of a class sum have
    // one sign, so n roundings cost at most n u of it.
    const bool split_mode = !(e->cfg.flags & HAF_FLAG_FP32_MFMA);
    const double acc_adds = split_mode ? (16.0 + e->n_sv_tiles / 8.0 + 2.0 + 4.0) : (e->n_sv_tiles + 5.0);
    e->svm.guard_acc = (float)(guard_scale * ((acc_adds + 6.0) * u));
    // PRECISE form of the three-pass kernel (the list mode behind the screening pass): the fp32 chain is the two fmas of one tile,
    // and from there on everything is fp64 -- fold, lane reduction, class sums, the ranges of the list mode (k_svm_h_combine) --
    // whose roundings (2^-53 each, a few hundred of them) are far inside the 0.1 u added for them; one rounding back to fp32 at the
    // end, +6 as above.  (sum|coef|K itself is measured with the same relative error, a few 1e-6: the factor behind the bracket.)
    e->svm.guard_acc_l = (float)(guard_scale * ((2.0 + 1.0 + 0.1 + 6.0) * u) * (1.0 + 1e-5));
    // screening pass: one sequential fp32 sum per lane over two column blocks per tile, the 4-step lane reduction, the
    // class split, v_exp_f32 and the coefficient product; the band is ~3e-4, so nothing is gained by a two-level sum.
    // HAF_GUARD0_REL scales the whole screening band (this term and the per-evaluation one) for experiments.
    double guard0_scale = 1.0;
    if (const char *g = test_env("HAF_GUARD0_REL")) guard0_scale = atof(g);
    // (plain variant: two levels -- a term passes through at most 16 fmas of the lower level, one fold, and the folds of its
    // sweep, <= tiles/8 + 1; then the final fma and add, the 4-step lane reduction, the class split, exp2 + product, the two
    // products with the common factor.  SUMSQ variant: one level, 2 fmas per tile.)
    // (a sweep covers the tiles of ONE class -- the kernel restarts its sums at the class boundary -- so "tiles" is the larger class's)
    const double sweep_tiles = (double)std::max(e->sv_tile_neg, e->n_sv_tiles - e->sv_tile_neg);
    e->svm.guard_acc0 = (float)(guard0_scale * ((16.0 + 1.0 + (sweep_tiles / 8.0 + 1.0) + 2.0 + 4.0 + 2.0 + 6.0 + 2.0) * u));
    e->svm.guard_acc0_s = (float)(guard0_scale * ((2.0 * sweep_tiles + 4.0 + 2.0 + 6.0 + 2.0) * u));
    e->screen.scale = 1.001 * guard0_scale;
    e->screen_cr.scale = 1.001 * guard0_scale;
    e->crt1.scale = 1.001 * guard_scale;
    e->crt1.guard_abs = e->svm.guard_abs;
    e->crt1.gv0 = e->gv0; e->crt1.gv1 = e->gv1;
    e->svm.guard_abs = (float)(std::fabs(m.rho) * 1.2e-7 + 1e-30);
    {
        double as_max = 0;
        for (int t = 0; t < e->n_sv_tiles; t++)
            for (int j = 0; j < kTile; j++) as_max = std::max(as_max, (double)std::fabs(svt[(size_t)t * kTileFloats + kKP * kTile + j]));
        e->svm.as_max = (float)as_max;
    }
    e->svm.gv0 = e->gv0; e->svm.gv1 = e->gv1;
    e->i8.gv0 = e->gv0; e->i8.gv1 = e->gv1;
    e->svm.sqrt_cmax = (float)(e->screen.sqrt_cmax * (1.0 + 1e-7));
    e->host_exp_thr = std::ldexp(e->sum_abs_coef, -44);        // 256 x the largest difference a last-bit exp error can make
    e->prob.dec_slack = std::ldexp(e->sum_abs_coef, -50);       // probability mode: 4 x what the two libsvm-order sums can differ by
    if (test_env("HAF_HOST_EXP_ALL")) e->host_exp_thr = INFINITY;   // tests: every strict-tier evaluation through the host path
    e->exact.gamma = m.gamma; e->exact.rho = m.rho;
    e->exact.lower = e->range.lower; e->exact.upper = e->range.upper;
    e->exact.n_sv = m.n_sv; e->exact.n_sv_pad = e->n_sv_pad; e->exact.kx = e->kx;
    e->exact.gv0 = e->gv0; e->exact.gv1 = e->gv1;
    return HAF_OK;
}

int alloc_buffers(haf_engine *e)
{
    const haf_config &c = e->cfg;
    e->max_rolls = (c.max_rolls_per_call > 0) ? std::min(c.max_rolls_per_call, c.n_rolls) : c.n_rolls;
    const size_t B = (size_t)c.max_clouds, R = (size_t)e->max_rolls, H = (size_t)c.grid_h, W = (size_t)c.grid_w;
    e->cells_cap = B * R * H * W;
    e->max_evals = (long)(B * R * (H - 14) * (W - 14));
    e->max_evals_pad = (e->max_evals + kS0BlockEvals - 1) / kS0BlockEvals * kS0BlockEvals;
    e->list_cap = (int)e->max_evals_pad;             // (< 2^31: cells are 32-bit ids, checked in haf_create)
    e->flag_cap = (int)std::min<long>(std::max<long>(4096, e->max_evals / 4), 1L << 22);
    if (const char *v = test_env("HAF_FLAG_WINDOW")) e->flag_cap = std::max(64, atoi(v));     // tests: many small windows
    const int mode = contraction_mode(c);
    // screening pass: up to half of the evaluations may go on to the three-pass kernel; a model that sends more is served by
    // the three-pass kernel alone from then on (haf_score_rolls)
    e->flag0_cap = mode == MODE_SCREEN ? (int)std::min<long>(std::max<long>(4096, (e->max_evals / 2 + 255) / 256 * 256), 1L << 23) : 0;
    bool ok = true;
    e->in_hdr_cap = (B * sizeof(CloudDev) + 15) / 16 * 16 + (B * R * sizeof(RollGeo) + 15) / 16 * 16;
    ok &= hipSuccess == e->d_in.alloc(e->in_hdr_cap + (size_t)c.max_points * 3 * sizeof(float));
    ok &= hipSuccess == e->d_out.alloc(kCntBytes + B * R * sizeof(RollRecordDev));
    if (ok) {
        e->d_counters.p = reinterpret_cast<int *>(e->d_out.p);
        e->d_rec.p = reinterpret_cast<RollRecordDev *>(e->d_out.p + kCntBytes);
    }
    {
        const int nb = bin_bucket_grid(c.grid_h, nullptr);
        e->bkt_ints = nb * nb + 1;
        if ((size_t)c.grid_h * c.grid_w > 16384) {           // grids k_bin_lds cannot hold: the bucket-sorted binning path
            ok &= hipSuccess == e->d_sorted.alloc((size_t)c.max_points * 3);
            ok &= hipSuccess == e->d_bkt.alloc((size_t)3 * B * e->bkt_ints);
        }
    }
    ok &= hipSuccess == e->d_heights.alloc(e->cells_cap);
    ok &= hipSuccess == e->d_rowsum.alloc(e->cells_cap);
    ok &= hipSuccess == e->d_inexact.alloc(B * R);
    ok &= hipSuccess == e->d_ii.alloc(B * R * (H + 1) * (W + 1));
    ok &= hipSuccess == e->d_mask.alloc(e->cells_cap);
    ok &= hipSuccess == e->d_rowcount.alloc(B * R * H);
    ok &= hipSuccess == e->d_rowoff.alloc(2 * (B * R * H + 1));      // whole-chunk and left-over starts (k_scan)
    ok &= hipSuccess == e->d_brcount.alloc(B * R);
    ok &= hipSuccess == e->d_evalcell.alloc((size_t)e->max_evals_pad);
    ok &= hipSuccess == e->d_flag_list.alloc((size_t)e->list_cap);
    if (mode == MODE_SCREEN) {
        // sized for the three-pass form as well: a model whose decisions crowd inside the screening band is served by
        // the three-pass kernel alone (screen_active)
        ok &= hipSuccess == e->d_X.alloc((size_t)(e->max_evals_pad / kTile) * (size_t)(kHXTileBytes / 4));
        const size_t slots = ((size_t)e->flag0_cap + kSvmBlockEvals - 1) / kSvmBlockEvals * kSvmBlockEvals;
        ok &= hipSuccess == e->d_X1.alloc(slots / kTile * (size_t)(kHXTileBytes / 4));
        ok &= hipSuccess == e->d_ax1.alloc(slots);
        e->part1_stride = (long)slots;
        ok &= hipSuccess == e->d_part1.alloc(slots * 2 * kHListParts);      // class sums per SV tile range (k_svm_h_combine)
        if (e->t1_cr_available) ok &= hipSuccess == e->d_t1_L.alloc(slots);
        ok &= hipSuccess == e->d_gband.alloc((size_t)e->max_evals_pad * kBandFloats);
        ok &= hipSuccess == e->d_flag0_list.alloc((size_t)e->flag0_cap);
        if (e->cr_available) ok &= hipSuccess == e->d_flag0b_list.alloc((size_t)e->flag0_cap);
        ok &= hipSuccess == e->d_flag0_words.alloc((size_t)e->max_evals_pad / 64);
        ok &= hipSuccess == e->d_flag0_wgcount.alloc((size_t)e->max_evals_pad / 64 / 256 + 1);
    } else {
        ok &= hipSuccess == e->d_X.alloc((size_t)(e->max_evals_pad / kTile) * (size_t)std::max<int>(kTileFloats, kHXTileBytes / 4));
    }
    ok &= hipSuccess == e->d_ax.alloc((size_t)e->max_evals_pad);
    ok &= hipSuccess == e->d_dec.alloc((size_t)e->max_evals_pad);
    ok &= hipSuccess == e->d_labels.alloc(e->cells_cap);
    ok &= hipSuccess == e->d_dec_exact.alloc((size_t)e->list_cap);
    ok &= hipSuccess == e->d_part64.alloc((size_t)e->flag_cap * kRecheckPartRows);
    // k_recheck_mfma reads whole workgroups of 64 evaluations (4 groups of 16): round the image up accordingly
    ok &= hipSuccess == e->d_x64.alloc(((size_t)e->flag_cap + 63) / 64 * 64 * kKP);
    ok &= hipSuccess == e->d_flag2_list.alloc((size_t)e->list_cap);
    if (e->i8_active) {
        ok &= hipSuccess == e->d_flagi_list.alloc((size_t)e->list_cap);
        ok &= hipSuccess == e->d_dec_exacti.alloc((size_t)e->list_cap);
    }
    ok &= hipSuccess == e->d_dec_exact2.alloc((size_t)e->list_cap);
    if (!e->prob_mode) ok &= hipSuccess == e->d_strict_terms.alloc((size_t)kStrictSlots * e->n_sv_pad);
    if ((c.flags & HAF_FLAG_KEEP_DEBUG) && mode == MODE_SCREEN) ok &= hipSuccess == e->d_margin.alloc((size_t)e->max_evals_pad);
    if (c.flags & HAF_FLAG_KEEP_DEBUG) {
        // attribute records of the exact-form feature kernels (haf_debug_fetch_attr): 7.6 KB per evaluation, so only for
        // engines of reference size (up to 2 GiB); a larger debug engine runs without them and the fetch says so
        const size_t bytes = (size_t)e->max_evals * kKP * sizeof(AttrRecord);
        if (bytes <= (2ull << 30)) ok &= hipSuccess == e->d_attr.alloc((size_t)e->max_evals * kKP);
    }
    ok &= hipSuccess == e->d_ev16.alloc(e->cells_cap);
    if (e->prob_mode) {
        ok &= hipSuccess == e->d_own.alloc(e->cells_cap);
        ok &= hipSuccess == e->d_gridf.alloc(e->cells_cap);
        ok &= hipSuccess == e->d_evf.alloc(e->cells_cap);
        ok &= hipSuccess == e->d_ptext.alloc(2 * (size_t)e->list_cap);
    }
    ok &= hipSuccess == e->d_rowmax.alloc(B * R * H);
    ok &= hipSuccess == e->d_topkey.alloc(3 * B * R);          // top vote key, longest-run key, completion counter (k_vote_*)
    if (!ok) return fail(e, HAF_E_DEVICE, std::string("hipMalloc of working buffers failed: ") + hipGetErrorString(hipGetLastError()));
    HIPCHK(e, hipHostMalloc((void **)&e->h_in, e->d_in.n));
    HIPCHK(e, hipHostMalloc((void **)&e->h_out, e->d_out.n));
    e->h_counters = reinterpret_cast<int *>(e->h_out);
    e->h_rec = reinterpret_cast<RollRecordDev *>(e->h_out + kCntBytes);
    HIPCHK(e, hipMemsetAsync(e->d_counters.p, 0, CNT_COUNT * sizeof(int), e->stream));
    e->counters_clean = true;
    return HAF_OK;
}

void mark(haf_engine *e, int idx)
{
    if (e->cfg.flags & HAF_FLAG_PROFILE) (void)hipEventRecord(e->ev[idx], e->stream);
}

}  // namespace

// ---- engine_internal.h ----
namespace haf {
static_assert(sizeof(RollRecordDev) == sizeof(haf_roll_record) && sizeof(haf_roll_record) == 16, "roll records travel as 16 bytes");
static_assert(sizeof(AttrRecord) == sizeof(haf_attr_record) && sizeof(haf_attr_record) == 24, "attribute debug record layout");
const void *engine_records_dev(const haf_engine *e) { return e->d_rec.p; }
hipStream_t engine_stream(const haf_engine *e) { return e->stream; }
const haf_config *engine_config(const haf_engine *e) { return &e->cfg; }
void engine_set_error(haf_engine *e, const char *msg) { e->error = msg; }
}  // namespace haf

// ---------------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------------
extern "C" {

int haf_abi_version(void) { return HAF_ABI_VERSION; }

void haf_config_default(haf_config *c)
{
    memset(c, 0, sizeof *c);
    c->nr_features_without_shaf = 302;
    c->grid_h = 56; c->grid_w = 56;
    c->n_rolls = 190 / 15;
    c->roll_step_deg = 15;
    c->z_shift = 0.15f;
    c->graspval_top = 119;
    c->graspval_th = 70;
    c->max_rolls_per_call = 0;
    c->device = 0;
    c->max_clouds = 1;
    c->max_points = 1 << 20;
    c->flags = 0;
}

void haf_grasp_input_default(haf_grasp_input *in)
{
    memset(in, 0, sizeof *in);
    in->grasp_area_length_x = 32;
    in->grasp_area_length_y = 44;
    in->approach_vector[2] = 1.0;
    in->max_calculation_time = 50.0;
    in->gripper_opening_width = 1;
}

const char *haf_last_error(const haf_engine *e) { return e ? e->error.c_str() : g_create_error.c_str(); }

void haf_destroy(haf_engine *e)
{
    if (!e) return;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto &r : e->host_regs) (void)hipHostUnregister((void *)r.first);
    e->host_regs.clear();
    e->d_in.release(); e->d_out.release(); e->d_sorted.release(); e->d_bkt.release(); e->d_heights.release(); e->d_rowsum.release(); e->d_inexact.release();
    e->d_ii.release(); e->d_mask.release(); e->d_rowcount.release(); e->d_rowoff.release(); e->d_brcount.release();
    e->d_evalcell.release(); e->d_flag_list.release(); e->d_X.release(); e->d_ax.release();
    e->d_dec.release(); e->d_svt.release(); e->d_svt_h.release(); e->d_labels.release(); e->d_dec_exact.release(); e->d_strict_terms.release(); e->d_part64.release(); e->d_dec_exact2.release(); e->d_flag2_list.release(); e->d_x64.release(); e->d_sv64.release();
    e->d_svt0.release(); e->d_X1.release(); e->d_ax1.release(); e->d_gband.release(); e->d_flag0_list.release(); e->d_flag0_words.release(); e->d_flag0_wgcount.release();
    e->d_own.release(); e->d_gridf.release(); e->d_evf.release(); e->d_ptext.release();
    e->d_sv_i8.release(); e->d_flagi_list.release(); e->d_dec_exacti.release();
    e->d_coef64.release(); e->d_ev16.release(); e->d_attr.release(); e->d_margin.release(); e->d_topkey.release(); e->d_rowmax.release(); e->d_fd.release();
    e->d_sd.release(); e->d_corr.release(); e->d_sd3.release(); e->d_fd_slot.release(); e->d_part1.release();
    e->d_svt_h_cr.release(); e->d_t1_tab.release(); e->d_t1_L.release(); e->d_flag0b_list.release();
    e->d_svt0_cr.release(); e->d_fd_slot_cr.release(); e->d_sd_cr.release(); e->d_sd3_cr.release(); e->d_corr_cr.release();
    if (e->h_in) (void)hipHostFree(e->h_in);
    if (e->h_out) (void)hipHostFree(e->h_out);
    for (auto &ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

static int score_rolls_impl(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, int32_t roll_first,
                            int32_t roll_count, haf_roll_record *records);

// Default mode: which screening variant serves this MODEL is found out here, at creation, not on the first goals of a fresh
// action server: up to three requests on a synthetic table scene (a plane with a few dozen boxes and domes of 2-12 cm, a point
// per grid cell) run through the same adaptive rule as every later call (plain variant -> the variant that measures |w|_2 when
// more than a quarter of the evaluations stay undecided -> no screening pass when that still leaves more than 60 %).  On the
// committed surrogate the scene classifies like the real clouds do (plain 94 % undecided, measuring variant 66 %: screening
// off); a model the scene misjudges is still re-classified by the rule on its first real requests, as before.
static int calibrate(haf_engine *e)
{
    const haf_config &c = e->cfg;
    if (contraction_mode(c) != MODE_SCREEN || !e->screen_active || e->prob_mode) return HAF_OK;
    const int G = c.grid_h;
    const long cells = (long)G * G;
    const long stride = std::max<long>(1, (cells + c.max_points - 1) / c.max_points);
    std::vector<float> xyz;
    xyz.reserve((size_t)(cells / stride + 1) * 3);
    uint32_t lcg = 0x9E3779B9u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((lcg >> 8) & 0xFFFFFF) / 16777216.0f; };
    struct Obj { float cx, cy, sx, sy, h, ct, st; int dome; };
    std::vector<Obj> objs((size_t)std::max<long>(4, cells / 400));
    const float half = 0.005f * (float)G;
    for (size_t k = 0; k < objs.size(); k++) {
        Obj &o = objs[k];
        o.cx = (2.0f * rnd() - 1.0f) * half; o.cy = (2.0f * rnd() - 1.0f) * half;
        o.sx = 0.015f + 0.045f * rnd(); o.sy = 0.015f + 0.045f * rnd();
        o.h = 0.02f + 0.10f * rnd();
        const float th = 3.14159265f * rnd();
        o.ct = std::cos(th); o.st = std::sin(th);
        o.dome = (k % 3) == 0;
    }
    // (objects only reach a few cells: a coarse bucket grid keeps the scene of a 512 x 512 engine cheap to build)
    const int nbk = std::max(1, G / 16);
    std::vector<std::vector<int>> bk((size_t)nbk * nbk);
    for (size_t k = 0; k < objs.size(); k++) {
        const float r = 1.5f * std::max(objs[k].sx, objs[k].sy);
        const int i0 = std::max(0, (int)((objs[k].cx - r + half) / (2 * half) * nbk)), i1 = std::min(nbk - 1, (int)((objs[k].cx + r + half) / (2 * half) * nbk));
        const int j0 = std::max(0, (int)((objs[k].cy - r + half) / (2 * half) * nbk)), j1 = std::min(nbk - 1, (int)((objs[k].cy + r + half) / (2 * half) * nbk));
        for (int i = i0; i <= i1; i++) for (int j = j0; j <= j1; j++) bk[(size_t)i * nbk + j].push_back((int)k);
    }
    for (long cell = 0; cell < cells; cell += stride) {
        const int i = (int)(cell / G), j = (int)(cell % G);
        const float x = ((float)i + 0.5f) * 0.01f - half, y = ((float)j + 0.5f) * 0.01f - half;
        float z = 0.0f;
        for (int k : bk[(size_t)std::min(nbk - 1, i * nbk / G) * nbk + std::min(nbk - 1, j * nbk / G)]) {
            const Obj &o = objs[(size_t)k];
            const float u = (x - o.cx) * o.ct + (y - o.cy) * o.st, v = -(x - o.cx) * o.st + (y - o.cy) * o.ct;
            if (o.dome) {
                const float q = u * u / (o.sx * o.sx) + v * v / (o.sy * o.sy);
                if (q < 1.0f) z = std::max(z, o.h * std::sqrt(std::max(0.0f, 1.0f - 0.5f * q)));
            } else if (std::fabs(u) < o.sx && std::fabs(v) < o.sy) {
                z = std::max(z, o.h);
            }
        }
        xyz.push_back(x); xyz.push_back(y); xyz.push_back(z + 0.002f * rnd());
    }
    haf_cloud cl{};
    cl.xyz = xyz.data(); cl.n_points = xyz.size() / 3; cl.stride_floats = 3; cl.on_device = 0;
    haf_grasp_input in;
    haf_grasp_input_default(&in);
    in.grasp_area_length_x = (float)G; in.grasp_area_length_y = (float)G;
    const int R = std::min(e->max_rolls, 2);
    std::vector<haf_roll_record> rec((size_t)R);
    const long keep_direct = e->direct_work;
    e->direct_work = 0;                                   // the tiers themselves, also on a small grid
    // Every form the engine has is tried on the scene (pinned, so that the adaptive rule does not interfere), cheapest kernel first;
    // a form that leaves next to nothing undecided ends the search.  The choice minimises kernel cost + what the undecided
    // evaluations cost behind it, in units of the plain kernel's time per evaluation (SUMSQ and CR_EXP: three VALU instructions
    // behind the exp instead of one, measured 1.085; CR_POLY: six and no exp, ~1.2; an undecided evaluation costs ~8.5 screened
    // ones in the three-pass tier and the exact tiers behind it: seed 11 of the bench generator).
    int rc = HAF_OK;
    const bool forced0 = e->variant_forced;
    const int variant0 = e->screen_variant;
    if (!forced0) {
        static const int order[SCREEN_VARIANTS] = {SCREEN_PLAIN, SCREEN_CR_EXP, SCREEN_SUMSQ, SCREEN_CR_POLY};   // cheapest kernel first
        double best_cost = 1e30;
        int best = -1;
        e->variant_forced = true;
        for (int oi = 0; oi < SCREEN_VARIANTS && rc == HAF_OK; oi++) {
            const int v = order[oi];
            if (v >= SCREEN_CR_EXP && !e->cr_available) continue;
            e->screen_variant = v;
            e->screen_active = true;
            rc = score_rolls_impl(e, 1, &cl, &in, 0, R, rec.data());
            if (rc != HAF_OK) break;
            const double ne = (double)std::max(1, e->last_evals);
            const double share = e->last_screened ? (double)e->last_flagged0 / ne : 1.0;
            e->variant_share[v] = share;
            if (share < 0.001) break;                    // nothing a later form could win back
        }
        // The choice: kernel cost + what the undecided evaluations cost behind it.  A PLAIN / SUMSQ first pass may have the
        // centred-remainder form as a SECOND pass on its list (tier 0b: ~1.6 screened evaluations per listed one -- its own feature
        // kernel and a contraction launch at a fraction of the chip), after which only what that form leaves is undecided: for a
        // well-conditioned model (a few per cent after the first pass) cheaper than the centred-remainder form over everything.
        for (int v = 0; v < SCREEN_VARIANTS && rc == HAF_OK; v++) {
            const double share = e->variant_share[v];
            if (share < 0.0) continue;
            double cost = kVariantCost[v] + kUndecidedCost * share;
            const double s_cr = e->variant_share[SCREEN_CR_EXP];
            if ((v == SCREEN_PLAIN || v == SCREEN_SUMSQ) && e->cr_available && s_cr >= 0.0 && s_cr < share)
                cost = std::min(cost, kVariantCost[v] + 1.6 * share + kUndecidedCost * s_cr);
            if (cost < best_cost) { best_cost = cost; best = v; }
        }
        e->variant_forced = false;
        if (rc == HAF_OK) {
            e->screen_variant = best >= 0 ? best : variant0;
            e->screen_active = best >= 0 && e->variant_share[best] <= 0.6;
        }
    } else {
        rc = score_rolls_impl(e, 1, &cl, &in, 0, R, rec.data());
    }
    if (rc == HAF_OK && e->screen_active) {
        // tier 0b behind a PLAIN / SUMSQ first pass: when the centred-remainder form left less than half as much undecided on the scene
        const int v = e->screen_variant;
        const char *f0b = test_env("HAF_T0B"), *fsk = test_env("HAF_T1_SKIP");
        // (the scene is small and tame -- the C5 bench leaves five times the share of the 56 x 56 scene undecided -- so the rule is
        // "whenever it decided more there", not a threshold)
        e->use_t0b = e->cr_available && (v == SCREEN_PLAIN || v == SCREEN_SUMSQ) && e->variant_share[SCREEN_CR_EXP] >= 0.0 &&
                     e->variant_share[SCREEN_CR_EXP] < e->variant_share[v];
        if (f0b) e->use_t0b = atoi(f0b) != 0 && e->cr_available;
        // is tier 1 of use behind the screening passes of this model?  One request in the final configuration: how much of what they
        // left did the three-pass kernel decide
        const bool forced1 = e->variant_forced;
        e->variant_forced = true;
        e->t1_skip = false;
        rc = score_rolls_impl(e, 1, &cl, &in, 0, R, rec.data());
        e->variant_forced = forced1;
        if (rc == HAF_OK && e->last_screened && e->last_flagged0 >= 32)
            e->t1_skip = (double)(e->last_flagged0 - e->last_flagged) < 0.35 * (double)e->last_flagged0;
        if (fsk) e->t1_skip = atoi(fsk) != 0;
    }
    e->direct_work = keep_direct;
    e->calibrated = true;
    // (the calibration requests are not a "last scored batch")
    e->last_B = e->last_R = e->last_roll_first = 0;
    e->last_evals = e->last_flagged = e->last_flagged2 = e->last_flagged0 = e->last_inexact = e->last_host_resolved = 0;
    e->last_flaggedi = 0;
    e->last_i8 = false;
    e->last_screened = false;
    e->last_inputs.clear();
    return rc;
}

static int create_impl(const haf_config *cfg, haf_engine **out)
{
    if (out) *out = nullptr;
    if (!cfg || !out) { g_create_error = "haf_create: null argument"; return HAF_E_ARG; }
    haf_engine *e = new haf_engine();
    struct Guard { haf_engine *e; ~Guard() { if (e) haf_destroy(e); } } guard{e};      // an exception below must not leak the engine
    auto bail = [&](int code) { g_create_error = e->error; return code; };
    e->cfg = *cfg;
    if (!cfg->feature_file || !cfg->range_file || !cfg->model_file) { e->error = "feature_file, range_file and model_file are required"; return bail(HAF_E_ARG); }
    e->feature_file = cfg->feature_file; e->range_file = cfg->range_file; e->model_file = cfg->model_file;
    e->cfg.feature_file = e->feature_file.c_str(); e->cfg.range_file = e->range_file.c_str(); e->cfg.model_file = e->model_file.c_str();
    if (cfg->grid_h != cfg->grid_w) { e->error = "grid_h must equal grid_w (the reference's mask geometry assumes a square grid, server.cpp:681-682, 705)"; return bail(HAF_E_ARG); }
    if (cfg->grid_h < 15 || cfg->grid_h > 4096) { e->error = "grid size must be in [15, 4096]"; return bail(HAF_E_ARG); }
    if (cfg->n_rolls < 1 || cfg->n_rolls > 4096 || cfg->max_clouds < 1 || cfg->max_points < 1) { e->error = "n_rolls, max_clouds and max_points must be positive"; return bail(HAF_E_ARG); }
    // integral images are read through a buffer descriptor (32-bit byte offsets): all of them must fit 4 GiB
    if (cfg->max_rolls_per_call < 0) { e->error = "max_rolls_per_call must be >= 0"; return bail(HAF_E_ARG); }
    {
        const int rolls_cap = (cfg->max_rolls_per_call > 0) ? std::min(cfg->max_rolls_per_call, cfg->n_rolls) : cfg->n_rolls;
        if ((double)cfg->max_clouds * rolls_cap * (cfg->grid_h + 1) * (cfg->grid_w + 1) * 4.0 >= 4294967296.0) { e->error = "max_clouds*rolls per call*grid cells exceeds 2^30"; return bail(HAF_E_CAPACITY); }
    }

    if (!load_features(e->feature_file, e->features, e->error)) return bail(HAF_E_IO);
    if (!load_range(e->range_file, e->range, e->error)) return bail(HAF_E_IO);
    if (!load_model(e->model_file, e->model, e->error)) return bail(HAF_E_IO);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { e->error = "no HIP device available: this engine has no CPU fallback"; return bail(HAF_E_DEVICE); }
    if (cfg->device < 0 || cfg->device >= ndev) { e->error = "device ordinal out of range"; return bail(HAF_E_DEVICE); }
    if (hipSetDevice(cfg->device) != hipSuccess) { e->error = "hipSetDevice failed"; return bail(HAF_E_DEVICE); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) { e->error = "hipGetDeviceProperties failed"; return bail(HAF_E_DEVICE); }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) { e->error = std::string("device is ") + prop.gcnArchName + ", the kernels are built for gfx950 only"; return bail(HAF_E_DEVICE); }
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { e->error = "hipStreamCreate failed"; return bail(HAF_E_DEVICE); }
    e->own_stream = true;
    for (auto &ev : e->ev) if (hipEventCreate(&ev) != hipSuccess) { e->error = "hipEventCreate failed"; return bail(HAF_E_DEVICE); }

#ifndef HAF_FLUSH_F16_SUBNORMALS
    if (contraction_mode(e->cfg) == MODE_SCREEN) {
        // the screening operands keep fp16 subnormals (features.hip: screen_operand): make sure the matrix core does too
        static std::mutex probe_mutex;                // one probe per DEVICE and process, whichever thread gets there first
        static int probed[64];                        // 0: not yet; else result + 2
        int keeps;
        {
            std::lock_guard<std::mutex> lock(probe_mutex);
            const int slot = cfg->device & 63;
            if (cfg->device >= 64 || probed[slot] == 0) {
                keeps = probe_f16_subnormal_mfma(e->stream);
                if (cfg->device < 64 && keeps >= 0) probed[slot] = keeps + 2;
            } else {
                keeps = probed[slot] - 2;
            }
        }
        if (keeps < 0) { e->error = "fp16 subnormal probe failed to run"; return bail(HAF_E_DEVICE); }
        if (keeps == 0) { e->error = "this device flushes fp16 subnormal MFMA operands: rebuild with -DHAF_FLUSH_F16_SUBNORMALS"; return bail(HAF_E_DEVICE); }
    }
#endif
    if (contraction_mode(e->cfg) != MODE_F32) {
        // the guard bands of the fp16 tiers carry the rounding of the matrix core as a MEASURED constant (screen.hip:
        // probe_mfma_rounding): once per device and process
        static std::mutex kappa_mutex;
        static double kappa_of[64], kappa16_of[64];   // 0: not yet
        double meas, meas16 = 0.0;
        {
            std::lock_guard<std::mutex> lock(kappa_mutex);
            const int slot = cfg->device & 63;
            if (cfg->device >= 64 || kappa_of[slot] == 0.0) {
                meas = probe_mfma_rounding(e->stream, &meas16);
                if (cfg->device < 64 && meas > 0.0) { kappa_of[slot] = meas; kappa16_of[slot] = meas16; }
            } else {
                meas = kappa_of[slot];
                meas16 = kappa16_of[slot];
            }
        }
        if (!(meas > 0.0) || !(meas16 > 0.0)) { e->error = "matrix-core rounding probe failed to run"; return bail(HAF_E_DEVICE); }
        e->mfma_kappa_measured = meas;
        e->mfma_kappa16_measured = meas16;
        // floor 12 (round 4; 8 before): the largest value seen by any search so far is 9.1 -- one product of order 1 over 31 products
        // with 22-bit mantissas, tests/test_engine_gpu.py::test_matrix_core_rounding_adversarial_search -- and the reading that fits
        // every measurement (terms aligned to the largest exponent, two guard bits, one final rounding) allows 33 x 0.25 + 1 = 9.25
        e->mfma_kappa = std::max(12.0, 1.5 * meas);
        e->mfma_kappa16 = std::max(12.0, 1.5 * meas16);
        // testing build: a matrix core that rounds worse than any seen, injected -- the bands must widen with it, the labels must not
        // move, and from 64 on the engine must refuse the device (tests/test_engine_gpu.py)
        if (const char *k = test_env("HAF_KAPPA")) e->mfma_kappa = e->mfma_kappa16 = atof(k);
        if (!(e->mfma_kappa < 64.0) || !(e->mfma_kappa16 < 64.0)) { e->error = "this device's fp16 MFMA rounds far worse than the guard bands allow for (probe_mfma_rounding)"; return bail(HAF_E_DEVICE); }
    }
    if (const char *v = test_env("HAF_LARGE_EVALS")) e->large_evals = atol(v);      // experiments
    // the tests that scale a guard band or force a tier mean the tiers themselves, also on a tiny request
    if (test_env("HAF_NO_DIRECT") || test_env("HAF_GUARD_REL") || test_env("HAF_GUARD0_REL") || test_env("HAF_GUARD2_REL") ||
        test_env("HAF_LARGE_EVALS") || test_env("HAF_NO_FAST_GROUPS") || test_env("HAF_SCREEN_NO_CENTRE") || test_env("HAF_FLAG_WINDOW") ||
        test_env("HAF_HOST_EXP_ALL") || test_env("HAF_NO_I8") || test_env("HAF_GUARD_I8_REL"))
        e->direct_work = 0;
    if (test_env("HAF_NO_FUSED_PRE")) e->no_fused_pre = true;
    if (const char *v = test_env("HAF_SCREEN_VARIANT")) { e->screen_variant = std::max(0, std::min(SCREEN_VARIANTS - 1, atoi(v))); e->variant_forced = true; e->direct_work = 0; }
    int rc = build_tables(e);
    if (rc != HAF_OK) return bail(rc);
    if (e->screen_variant >= SCREEN_CR_EXP && !e->cr_available) e->screen_variant = SCREEN_PLAIN;   // (a pinned form the model has no tables for)
    rc = alloc_buffers(e);
    if (rc != HAF_OK) return bail(rc);
    // (a test that scales the screening band wants the tiers and the adaptive rule as they are, not a model classified under that band)
    if (!test_env("HAF_NO_CALIBRATE") && !test_env("HAF_GUARD0_REL")) {
        rc = calibrate(e);
        if (rc != HAF_OK) return bail(rc);
    }
    guard.e = nullptr;
    *out = e;
    return HAF_OK;
}

int haf_model_info(const haf_engine *e, int32_t *n_sv, int32_t *dim, int32_t *n_features)
{
    if (!e) return HAF_E_ARG;
    if (n_sv) *n_sv = e->model.n_sv;
    if (dim) *dim = e->model.dim;
    if (n_features) *n_features = e->nf;
    return HAF_OK;
}

int haf_register_host_cloud(haf_engine *e, const void *ptr, size_t bytes)
{
    if (!e) return HAF_E_ARG;
    if (!ptr || !bytes) return fail(e, HAF_E_ARG, "haf_register_host_cloud: null or empty buffer");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    for (auto &r : e->host_regs) if (r.first == ptr) return fail(e, HAF_E_ARG, "haf_register_host_cloud: buffer is registered already");
    HIPCHK(e, hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault));
    e->host_regs.emplace_back(reinterpret_cast<const char *>(ptr), bytes);
    return HAF_OK;
}

int haf_unregister_host_cloud(haf_engine *e, const void *ptr)
{
    if (!e) return HAF_E_ARG;
    for (size_t i = 0; i < e->host_regs.size(); i++)
        if (e->host_regs[i].first == ptr) {
            if (e->stream) (void)hipStreamSynchronize(e->stream);
            HIPCHK(e, hipHostUnregister(const_cast<void *>(ptr)));
            e->host_regs.erase(e->host_regs.begin() + (long)i);
            return HAF_OK;
        }
    return fail(e, HAF_E_ARG, "haf_unregister_host_cloud: buffer was not registered");
}

int haf_screen_form(const haf_engine *e, int32_t *form, int32_t *active)
{
    if (!e) return HAF_E_ARG;
    const bool on = contraction_mode(e->cfg) == MODE_SCREEN && e->screen_active && !e->prob_mode;
    if (form) *form = e->screen_variant;
    if (active) *active = on ? 1 : 0;
    return HAF_OK;
}

int haf_last_counts(const haf_engine *e, int64_t *n_evals, int64_t *n_rechecked, int64_t *n_strict)
{
    if (!e) return HAF_E_ARG;
    if (n_evals) *n_evals = e->last_evals;
    if (n_rechecked) *n_rechecked = e->last_flagged;
    if (n_strict) *n_strict = e->last_flagged2;
    return HAF_OK;
}

int haf_last_tiers(const haf_engine *e, int64_t *n_evals, int64_t *n_refined, int64_t *n_rechecked, int64_t *n_strict)
{
    if (!e) return HAF_E_ARG;
    if (n_evals) *n_evals = e->last_evals;
    if (n_refined) *n_refined = e->last_flagged0;
    if (n_rechecked) *n_rechecked = e->last_flagged;
    if (n_strict) *n_strict = e->last_flagged2;
    return HAF_OK;
}

int haf_last_exact_tiers(const haf_engine *e, int64_t *n_integer, int64_t *n_fp64)
{
    if (!e) return HAF_E_ARG;
    if (n_integer) *n_integer = e->last_i8 ? e->last_flagged : 0;
    if (n_fp64) *n_fp64 = e->last_flaggedi;
    return HAF_OK;
}

int haf_last_strict_host(const haf_engine *e, int64_t *n_host)
{
    if (!e) return HAF_E_ARG;
    if (n_host) *n_host = e->last_host_resolved;
    return HAF_OK;
}

int haf_last_prestage(const haf_engine *e, int64_t *n_inexact_grids)
{
    if (!e) return HAF_E_ARG;
    if (n_inexact_grids) *n_inexact_grids = e->last_inexact;
    return HAF_OK;
}

int haf_set_stream(haf_engine *e, void *s)
{
    if (!e) return HAF_E_ARG;
    if (e->stream) (void)hipStreamSynchronize(e->stream);          // (the counters of the next request are zeroed on the old stream)
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    e->stream = (hipStream_t)s;
    e->own_stream = false;
    return HAF_OK;
}

void *haf_get_stream(haf_engine *e) { return e ? (void *)e->stream : nullptr; }

// The strict tier (k_recheck) restates libsvm's summation order operation for operation, but its exp() is the device's, not
// glibc's.  Both are within an ulp of the true value, so the two sums differ by at most 2^-52 sum|coef|; a strict-tier decision
// value closer to zero than host_exp_thr (256 x that) is therefore evaluated once more HERE, on the host, with the C library's
// exp -- the very function the reference's svm-predict calls (svm.cpp:325-365, 2478-2532) -- from the attributes the device
// computed (the decimal round trips are bit-pinned to glibc, tests/).  Nothing has come this far in any run; the path exists so
// that "the labels are libsvm's" has no residual.  Returns the number of evaluations decided here; *changed = a label moved.
static int host_resolve_strict(haf_engine *e, const Dims &d, hipStream_t s, bool *changed)
{
    *changed = false;
    e->last_host_resolved = 0;
    const int n2 = std::min(std::min(e->h_counters[CNT_FLAGGED2], e->list_cap), e->flag_cap);     // (one window of the attribute image)
    if (n2 <= 0 || e->prob_mode) return HAF_OK;
    std::vector<double> dec((size_t)n2);
    std::vector<int> ev((size_t)n2);
    HIPCHK(e, hipMemcpyAsync(dec.data(), e->d_dec_exact2.p, (size_t)n2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipMemcpyAsync(ev.data(), e->d_flag2_list.p, (size_t)n2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
    std::vector<int> cand;
    for (int i = 0; i < n2; i++) if (!(std::fabs(dec[(size_t)i]) > e->host_exp_thr)) cand.push_back(i);
    if (cand.empty()) return HAF_OK;
    // the fp64 attribute image of the strict tier's list ([group of 16][324][16]) through the feature kernel, then to the host
    launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, reinterpret_cast<float *>(e->d_x64.p), nullptr, d, e->range.lower,
                    e->range.upper, 0.0f, n2, XMODE_F64, ScreenParams{}, e->d_flag2_list.p, CNT_FLAGGED2, n2, false, n2, nullptr, nullptr, s, 0);
    const size_t groups = ((size_t)n2 + 15) / 16;
    std::vector<double> x64(groups * kKP * 16);
    HIPCHK(e, hipMemcpyAsync(x64.data(), e->d_x64.p, x64.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
    const SvmModel &m = e->model;
    const int kx = e->kx;
    for (int i : cand) {
        const double *xg = x64.data() + (size_t)(i >> 4) * kKP * 16 + (i & 15);
        double sum = 0.0;
        for (int n = 0; n < m.n_sv; n++) {                       // svm.cpp:2509-2512: both classes' terms in model order
            double d2 = 0.0;
            for (int k = 0; k < kx; k++) {                       // svm.cpp:333-347: index order, a missing entry is 0
                const double sv = k < m.dim ? m.sv[(size_t)n * m.dim + k] : 0.0;
                const double dd = xg[(size_t)k * 16] - sv;
                d2 += dd * dd;
            }
            sum += m.coef[(size_t)n] * std::exp(-m.gamma * d2);  // svm.cpp:364: glibc's exp
        }
        const double dv = sum - m.rho;                           // 2513
        const int8_t lab = (int8_t)(dv > 0.0 ? e->gv0 : e->gv1);
        int cell = 0;
        int8_t old = 0;
        HIPCHK(e, hipMemcpy(&cell, e->d_evalcell.p + ev[(size_t)i], sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(e, hipMemcpy(&old, e->d_labels.p + cell, 1, hipMemcpyDeviceToHost));
        if (old != lab) *changed = true;
        HIPCHK(e, hipMemcpy(e->d_labels.p + cell, &lab, 1, hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->d_dec_exact2.p + i, &dv, sizeof(double), hipMemcpyHostToDevice));
        e->last_host_resolved++;
    }
    return HAF_OK;
}

// svm_predict_probability for two classes on the HOST, operation for operation as svm.cpp:2550-2587 (sigmoid_predict 1818-1826 with
// the C library's exp, the [1e-7, 1 - 1e-7] clamp, multiclass_probability 1829-1888 for k = 2); this TU is built with -ffp-contract=off
static int host_probability(double dec, double A, double B, double p[2])
{
    const double fApB = dec * A + B;
    double s = fApB >= 0.0 ? std::exp(-fApB) / (1.0 + std::exp(-fApB)) : 1.0 / (1.0 + std::exp(fApB));
    const double min_prob = 1e-7;
    s = std::min(std::max(s, min_prob), 1.0 - min_prob);
    const int k = 2;
    double r[2][2] = {{0.0, s}, {1.0 - s, 0.0}}, Q[2][2], Qp[2], pQp;
    const double eps = 0.005 / k;
    for (int t = 0; t < k; t++) {
        p[t] = 1.0 / k;
        Q[t][t] = 0.0;
        for (int j = 0; j < t; j++) { Q[t][t] += r[j][t] * r[j][t]; Q[t][j] = Q[j][t]; }
        for (int j = t + 1; j < k; j++) { Q[t][t] += r[j][t] * r[j][t]; Q[t][j] = -r[j][t] * r[t][j]; }
    }
    for (int iter = 0; iter < 100; iter++) {
        pQp = 0.0;
        for (int t = 0; t < k; t++) {
            Qp[t] = 0.0;
            for (int j = 0; j < k; j++) Qp[t] += Q[t][j] * p[j];
            pQp += p[t] * Qp[t];
        }
        double max_error = 0.0;
        for (int t = 0; t < k; t++) max_error = std::max(max_error, std::fabs(Qp[t] - pQp));
        if (max_error < eps) break;
        for (int t = 0; t < k; t++) {
            const double diff = (-Qp[t] + pQp) / Q[t][t];
            p[t] += diff;
            pQp = (pQp + diff * (diff * Q[t][t] + 2.0 * Qp[t])) / (1.0 + diff) / (1.0 + diff);
            for (int j = 0; j < k; j++) { Qp[j] = (Qp[j] + diff * Q[t][j]) / (1.0 + diff); p[j] /= (1.0 + diff); }
        }
    }
    return p[1] > p[0] ? 1 : 0;
}

// Probability mode: the estimates k_prob_eval could not vouch for (a last-bit difference between the device's exp and glibc's could
// move their label or a printed digit; CNT_FLAGGED / d_flag_list) are finished HERE: the libsvm-order decision value with the C
// library's exp from the device's attributes (as host_resolve_strict does), svm_predict_probability with the C library's exp, the
// "%g" forms by the host build of decq (pinned to glibc's printf + strtod).  Writes what k_prob_eval writes.
static int host_resolve_probability(haf_engine *e, const Dims &d, hipStream_t s)
{
    e->last_host_resolved = 0;
    const int n = std::min(e->h_counters[CNT_FLAGGED], e->list_cap);
    if (n <= 0) return HAF_OK;
    const SvmModel &m = e->model;
    const int kx = e->kx;
    std::vector<int> ev((size_t)n);
    HIPCHK(e, hipMemcpyAsync(ev.data(), e->d_flag_list.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
    for (int off = 0; off < n; off += e->flag_cap) {
        const int nw = std::min(e->flag_cap, n - off);
        launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, reinterpret_cast<float *>(e->d_x64.p), nullptr, d, e->range.lower,
                        e->range.upper, 0.0f, nw, XMODE_F64, ScreenParams{}, e->d_flag_list.p + off, CNT_FLAGGED, nw, false, nw, nullptr, nullptr, s, off);
        const size_t groups = ((size_t)nw + 15) / 16;
        std::vector<double> x64(groups * kKP * 16);
        HIPCHK(e, hipMemcpyAsync(x64.data(), e->d_x64.p, x64.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(e, hipStreamSynchronize(s));
        for (int i = 0; i < nw; i++) {
            const double *xg = x64.data() + (size_t)(i >> 4) * kKP * 16 + (i & 15);
            double sum = 0.0;
            for (int nn = 0; nn < m.n_sv; nn++) {                    // svm.cpp:2509-2512
                double d2 = 0.0;
                for (int k = 0; k < kx; k++) {
                    const double sv = k < m.dim ? m.sv[(size_t)nn * m.dim + k] : 0.0;
                    const double dd = xg[(size_t)k * 16] - sv;
                    d2 += dd * dd;
                }
                sum += m.coef[(size_t)nn] * std::exp(-m.gamma * d2);
            }
            const double dv = sum - m.rho;
            double p[2];
            const int idx = host_probability(dv, e->prob.A, e->prob.B, p);
            const double q[2] = {hafq::decq(p[0], 6), hafq::decq(p[1], 6)};
            const int res = idx ? e->prob.gv1 : e->prob.gv0;
            const float own = (float)res * (float)(res > 0 ? q[1] : q[0]);
            const int8_t lab = (int8_t)res;
            const int evi = ev[(size_t)(off + i)];
            int cell = 0;
            HIPCHK(e, hipMemcpy(&cell, e->d_evalcell.p + evi, sizeof(int), hipMemcpyDeviceToHost));
            HIPCHK(e, hipMemcpy(e->d_own.p + cell, &own, sizeof(float), hipMemcpyHostToDevice));
            HIPCHK(e, hipMemcpy(e->d_labels.p + cell, &lab, 1, hipMemcpyHostToDevice));
            HIPCHK(e, hipMemcpy(e->d_ptext.p + 2 * (size_t)evi, q, 2 * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(e, hipMemcpy(e->d_dec_exact2.p + evi, &dv, sizeof(double), hipMemcpyHostToDevice));
            e->last_host_resolved++;
        }
    }
    return HAF_OK;
}

static int score_rolls_impl(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, int32_t roll_first,
                            int32_t roll_count, haf_roll_record *records)
{
    if (!e) return HAF_E_ARG;
    if (!clouds || !in || !records || n_clouds < 1) return fail(e, HAF_E_ARG, "haf_score_rolls: null or empty argument");
    const haf_config &c = e->cfg;
    if (n_clouds > c.max_clouds) return fail(e, HAF_E_CAPACITY, "more clouds than max_clouds");
    if (roll_first < 0 || roll_count < 1 || roll_first + roll_count > c.n_rolls) return fail(e, HAF_E_ARG, "roll range outside [0, n_rolls)");
    if (roll_count > e->max_rolls) return fail(e, HAF_E_CAPACITY, "more rolls in one call than max_rolls_per_call");
    HIPCHK(e, hipSetDevice(c.device));
    const int B = n_clouds, R = roll_count, H = c.grid_h, W = c.grid_w;

    // ---- host preparation ----
    size_t host_pts = 0;
    int max_n = 0;
    for (int b = 0; b < B; b++) {
        if (clouds[b].n_points && !clouds[b].xyz) return fail(e, HAF_E_ARG, "cloud with null xyz");
        if (clouds[b].stride_floats < 3) return fail(e, HAF_E_ARG, "stride_floats must be >= 3");
        if (clouds[b].n_points > (size_t)INT32_MAX) return fail(e, HAF_E_CAPACITY, "cloud too large");
        if (clouds[b].on_device == 2) {
            const char *p0 = reinterpret_cast<const char *>(clouds[b].xyz), *p1 = p0 + clouds[b].n_points * 12;
            bool inside = false;
            for (auto &r : e->host_regs) inside = inside || (p0 >= r.first && p1 <= r.first + r.second);
            if (clouds[b].stride_floats != 3 || (clouds[b].n_points && !inside))
                return fail(e, HAF_E_ARG, "on_device = 2 needs a packed xyz cloud inside a buffer registered with haf_register_host_cloud");
        }
        if (clouds[b].on_device != 1) host_pts += clouds[b].n_points;
        max_n = std::max(max_n, (int)clouds[b].n_points);
    }
    if (host_pts > (size_t)c.max_points) return fail(e, HAF_E_CAPACITY, "more host points than max_points");
    // the request's input block (d_in / h_in): [CloudDev x B][RollGeo x B*R][points of the host clouds], one copy
    const size_t geo_off = ((size_t)B * sizeof(CloudDev) + 15) / 16 * 16;
    const size_t pts_off = geo_off + ((size_t)B * R * sizeof(RollGeo) + 15) / 16 * 16;
    CloudDev *h_clouds = reinterpret_cast<CloudDev *>(e->h_in);
    RollGeo *h_geo = reinterpret_cast<RollGeo *>(e->h_in + geo_off);
    float *h_points = reinterpret_cast<float *>(e->h_in + pts_off);
    const CloudDev *d_clouds = reinterpret_cast<const CloudDev *>(e->d_in.p);
    const RollGeo *d_geo = reinterpret_cast<const RollGeo *>(e->d_in.p + geo_off);
    const float *d_points = reinterpret_cast<const float *>(e->d_in.p + pts_off);
    size_t off = 0;
    long total_n = 0;
    bool bucket_ok = true;
    for (int b = 0; b < B; b++) {
        NormalisedInput n = normalise(in[b]);
        CloudDev &cd = h_clouds[b];
        for (int r = 0; r < R; r++) fill_roll_geo(c, in[b], n, roll_first + r, h_geo[b * R + r], r == 0 ? cd.m0 : nullptr);
        if (n.width == 0) bucket_ok = false;             // x-scale 0: every point lands in row H/2, whatever its distance
        cd.sorted_off = (int)total_n;
        cd.bucket_off = b * e->bkt_ints;
        total_n += (long)clouds[b].n_points;
        cd.n = (int)clouds[b].n_points;
        if (clouds[b].on_device == 1) {
            cd.xyz = clouds[b].xyz;
            cd.stride = (int)clouds[b].stride_floats;
        } else {
            cd.xyz = d_points + off * 3;
            cd.stride = 3;
            off += clouds[b].n_points;
        }
    }
    hipStream_t s = e->stream;
    mark(e, 0);
    // Host clouds go through the pinned block in pieces: while the DMA engine moves one piece the host packs the next (a 1.2 MB
    // cloud -- C3 -- costs ~100 us of host memcpy; its transfer hides behind that).  The first copy carries the two header arrays.
    {
        constexpr size_t kPiece = 256 * 1024;                     // bytes of packed points per copy
        size_t staged = 0, sent = 0;                              // bytes of the points area packed / handed to the DMA engine
        bool header_sent = false;
        auto flush = [&](bool last) -> int {
            if (!header_sent) {
                HIPCHK(e, hipMemcpyAsync(e->d_in.p, e->h_in, pts_off + staged, hipMemcpyHostToDevice, s));
                header_sent = true;
            } else if (staged > sent) {
                HIPCHK(e, hipMemcpyAsync(e->d_in.p + pts_off + sent, e->h_in + pts_off + sent, staged - sent, hipMemcpyHostToDevice, s));
            }
            sent = staged;
            (void)last;
            return HAF_OK;
        };
        for (int b = 0; b < B; b++) {
            if (clouds[b].on_device == 1 || clouds[b].n_points == 0) continue;
            if (clouds[b].on_device == 2 && clouds[b].n_points * 12 >= kPiece) {
                // (a small cloud is cheaper packed into the one staged copy than as a DMA transfer of its own: ~10 us each)
                // page-locked caller memory: whatever has been packed so far goes out, then the DMA engine takes this cloud from
                // where it lies (the staging block keeps the same layout, its share of it stays unused)
                const int rc = flush(false);
                if (rc != HAF_OK) return rc;
                const size_t bytes = clouds[b].n_points * 12;
                HIPCHK(e, hipMemcpyAsync(e->d_in.p + pts_off + staged, clouds[b].xyz, bytes, hipMemcpyHostToDevice, s));
                staged += bytes;
                sent = staged;
                continue;
            }
            const float *src = clouds[b].xyz;
            const size_t st = clouds[b].stride_floats, n = clouds[b].n_points;
            for (size_t i0 = 0; i0 < n;) {
                const size_t room = std::max<size_t>(1, (kPiece - (staged - sent)) / 12);
                const size_t cnt = std::min(n - i0, room);
                float *dst = reinterpret_cast<float *>(e->h_in + pts_off + staged);
                if (st == 3) memcpy(dst, src + i0 * 3, cnt * 12);
                else for (size_t i = 0; i < cnt; i++) { dst[i * 3] = src[(i0 + i) * st]; dst[i * 3 + 1] = src[(i0 + i) * st + 1]; dst[i * 3 + 2] = src[(i0 + i) * st + 2]; }
                staged += cnt * 12;
                i0 += cnt;
                if (staged - sent >= kPiece) { const int rc = flush(false); if (rc != HAF_OK) return rc; }
            }
        }
        const int rc = flush(true);
        if (rc != HAF_OK) return rc;
        (void)h_points;
    }
    // (the counters were zeroed behind the previous request's copy-out; after an error they may not have been)
    if (!e->counters_clean) HIPCHK(e, hipMemsetAsync(e->d_counters.p, 0, CNT_COUNT * sizeof(int), s));
    e->counters_clean = false;
    const size_t cells = (size_t)B * R * H * W;
    if (e->d_attr.p) HIPCHK(e, hipMemsetAsync(e->d_attr.p, 0xFF, e->d_attr.n * sizeof(AttrRecord), s));   // debug: "not computed"

    Dims d;
    d.H = H; d.W = W; d.R = R; d.B = B; d.nf = e->nf; d.n_sv = e->model.n_sv; d.n_sv_tiles = e->n_sv_tiles; d.sv_tile_neg = e->sv_tile_neg;
    const float r_row = (float)((0.5 * (float)H) / 100.0), r_col = (float)((0.5 * (float)W) / 100.0);   // server.cpp:410-411
    const long evals_cap = (long)B * R * (H - 14) * (W - 14);
    // For choosing between the feature kernels only: the masked cells of a roll lie inside the rotated search rectangle of
    // half sizes sx/2 - 7, sy/2 - 7 (pnt_in_box 687-688), at most (a + 2)(b + 2) lattice points for sides a, b -- usually far
    // fewer than the grid could hold (the client's default 32 x 44 area on the 56 x 56 grid: a third).
    long evals_sel = 0;
    for (int b = 0; b < B; b++) {
        const long a2 = std::max(0, 2 * ((int)in[b].grasp_area_length_x / 2 - 7)) + 2, b2 = std::max(0, 2 * ((int)in[b].grasp_area_length_y / 2 - 7)) + 2;
        evals_sel += (long)R * std::min<long>((long)(H - 14) * (W - 14), a2 * b2);
    }
    // A request whose whole SVM work is tiny goes straight to the fp64 MFMA tier (every evaluation enters its list): same
    // labels by construction -- the tier decides outside its own band and hands the rest to the strict tier -- and three
    // launches instead of a feature kernel, a contraction kernel and the rechecks behind them.
    const bool direct = !e->prob_mode && e->direct_work > 0 && evals_sel * (long)e->n_sv_pad <= e->direct_work;
    const bool short_request = evals_sel * (long)e->n_sv_pad <= (1L << 26) && total_n <= (1L << 20);
    // A small engine with a small model behind one of the fast contractions: what that contraction flags goes through the SAME one-launch
    // kernel in list mode (exact attributes + fp64 MFMA decision, tier 2's arithmetic) instead of tier 2a's three launches and tier 2's
    // three -- at a few thousand evaluations x a few hundred SVs the six launches and two more attribute kernels cost more than the
    // exact work (C3: 81 -> 30 us; the kernel costs ~9 ns per listed evaluation at 192 SVs, so it wins up to ~8 000 of them: a request
    // of up to 2^25 evaluation x SV pairs, of which a trained model flags around a tenth).  Decided from the request's search areas and
    // the model's size, so identical calls take identical paths.
    const bool small_exact = !direct && !e->prob_mode && e->direct_work > 0 && evals_sel * (long)e->n_sv_pad <= 16 * e->direct_work;
    // small grids: a1 (tail) + a2 + a3 + a4 in ONE launch (k_small_pre); the probability branch needs k_scan's row-major order
    bool fused_pre = false;
    mark(e, HAF_ST_BIN);
    if (!e->prob_mode && !e->no_fused_pre)
        fused_pre = launch_small_pre(d_clouds, d_geo, max_n, e->d_heights.p, e->d_ii.p, e->d_mask.p, e->d_rowcount.p, e->d_brcount.p,
                                     e->d_labels.p, e->d_evalcell.p, e->d_counters.p, e->d_flag_list.p, direct, d, r_row, r_col, s);
    if (fused_pre) {
        mark(e, HAF_ST_INTEGRAL);
        mark(e, HAF_ST_MASK);
    } else {
        HIPCHK(e, hipMemsetAsync(e->d_labels.p, 0xFF, cells, s));        // -1: no feature vector for this cell (server.cpp:828-829)
        BinScratch bs{};
        bs.sorted = e->d_sorted.p; bs.sorted_cap = e->d_sorted.p ? (long)c.max_points : 0;
        bs.bkt_count = e->d_bkt.p; bs.bkt_off = e->d_bkt.p ? e->d_bkt.p + (size_t)c.max_clouds * e->bkt_ints : nullptr;
        bs.bkt_cursor = e->d_bkt.p ? e->d_bkt.p + (size_t)2 * c.max_clouds * e->bkt_ints : nullptr;
        bs.bkt_cap = e->d_bkt.p ? c.max_clouds * e->bkt_ints : 0;
        launch_bin(d_clouds, h_clouds, max_n, total_n, d_geo, e->d_heights.p, d, r_row, r_col, bucket_ok && !e->no_bucket_sort, bs, e->d_counters.p, s);
        mark(e, HAF_ST_INTEGRAL);
        launch_integral(e->d_heights.p, e->d_rowsum.p, e->d_ii.p, e->d_inexact.p, e->d_counters.p, d, s);
        mark(e, HAF_ST_MASK);
        launch_mask_count(e->d_ii.p, d_geo, e->d_mask.p, e->d_rowcount.p, d, s);
        launch_scan(e->d_rowcount.p, e->d_rowoff.p, e->d_brcount.p, e->d_counters.p, d, s);
        launch_compact(e->d_mask.p, e->d_rowcount.p, e->d_rowoff.p, e->d_evalcell.p, d, s);
        if (direct) launch_prob_list(e->d_counters.p, CNT_FLAGGED, e->d_flag_list.p, e->list_cap, s);
    }
    // features -> decision tiers -> vote -> records on the host, for one contraction mode
    bool i8_used = false;                                    // the exact-integer tier ran in the last decide()
    bool t0b_used = false;                                   // tier 0b ran in the last decide()
    auto decide = [&](int mode, bool reuse_operands) -> int {
        t0b_used = false;
        mark(e, HAF_ST_FEATURES);
        const bool large = evals_sel >= e->large_evals;      // enough evaluations to fill the chip with one thread each
        if (direct) {
            // tiny request: exact attributes, fp64 MFMA decision and label of EVERY evaluation in one launch (k_small_direct: tier 2's
            // arithmetic); every evaluation counts as rechecked (k_small_pre / k_prob_list have put them on that tier's list)
            launch_small_direct(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_sv64.p, e->exact, d, std::min<long>(evals_cap, e->list_cap),
                                e->d_dec_exact.p, e->d_labels.p, e->d_flag2_list.p, e->list_cap, e->d_attr.p, s);
            mark(e, HAF_ST_SVM);
            mark(e, HAF_ST_REFINE);
        } else if (mode == MODE_SCREEN) {
            // tier 0: single-pass fp16 screening of every evaluation; tier 1: the three-pass kernel on what it could not decide
            // (the centred-remainder variants have their own operand images: translated attributes, centred support vectors)
            const bool cr = e->screen_variant == SCREEN_CR_EXP || e->screen_variant == SCREEN_CR_POLY;
            ScreenParams sp_now = cr ? e->screen_cr : e->screen;
            sp_now.cr_poly = e->screen_variant == SCREEN_CR_POLY;
            if (!reuse_operands)
                launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X.p, e->d_gband.p, d, e->range.lower,
                                e->range.upper, e->svm.neg_gamma2, evals_cap, XMODE_SCREEN, sp_now, nullptr, 0, 0, large, evals_sel, nullptr, e->d_ax.p, s);
            mark(e, HAF_ST_SVM);
            // A small request with a small model (small_exact): what the screening pass leaves goes STRAIGHT to the one-launch exact kernel
            // (k_small_direct in list mode: exact attributes + fp64 MFMA decision, 9 ns per listed evaluation at 192 SVs) -- the list
            // is written where that kernel reads it.  Tier 1 in between was a feature kernel and a contraction launch at their latency
            // floors (C3: 44 + 36 us for 4 072 evaluations, of which it decided nine tenths) in front of the same exact kernel.
            // The same hand-over when calibration found tier 1 of little use behind the screening passes (t1_skip).
            const bool t0b = e->use_t0b && e->cr_available && !cr && !small_exact;
            const bool straight = small_exact || (e->t1_skip && !t0b);
            launch_svm_screen(e->d_X.p, e->d_gband.p, e->d_ax.p, cr ? e->d_svt0_cr.p : e->d_svt0.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                              e->d_flag0_words.p, e->d_flag0_wgcount.p, straight ? e->d_flag_list.p : e->d_flag0_list.p, e->flag0_cap, e->d_counters.p, d, evals_cap, e->d_margin.p,
                              e->screen_variant, e->crp, s, straight ? CNT_FLAGGED : -1);
            mark(e, HAF_ST_REFINE);
            const long list_cap = std::min<long>(e->flag0_cap, evals_cap);
            const int *t1_list = e->d_flag0_list.p;
            int t1_counter = CNT_FLAGGED0;
            bool t1_run = !straight;
            if (t0b) {
                // tier 0b: the centred-remainder form on the LIST of the first pass (its own operand images: translated attributes,
                // centred support vectors; band, common factor and images indexed by list slot)
                ScreenParams sp_b = e->screen_cr;
                sp_b.cr_poly = 0;
                launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X1.p, e->d_gband.p, d, e->range.lower,
                                e->range.upper, e->svm.neg_gamma2, list_cap, XMODE_SCREEN, sp_b, e->d_flag0_list.p, CNT_FLAGGED0, e->flag0_cap,
                                false, list_cap, nullptr, e->d_ax.p, s);
                const bool skip1 = e->t1_skip;
                launch_svm_screen(e->d_X1.p, e->d_gband.p, e->d_ax.p, e->d_svt0_cr.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                                  e->d_flag0_words.p, e->d_flag0_wgcount.p, skip1 ? e->d_flag_list.p : e->d_flag0b_list.p, e->flag0_cap, e->d_counters.p, d,
                                  list_cap, e->d_margin.p, SCREEN_CR_EXP, e->crp, s, skip1 ? CNT_FLAGGED : -1, e->d_flag0_list.p, CNT_FLAGGED0, CNT_FLAGGED0B);
                t1_list = e->d_flag0b_list.p;
                t1_counter = CNT_FLAGGED0B;
                t1_run = !skip1;
                t0b_used = true;
            }
            if (t1_run) {
            // (the list is short whenever screening is worth its while: always the group-parallel feature kernel, whose
            // workgroups beyond the list's end exit at once)
            // behind the polynomial centred-remainder form (a model whose decisions are 1e-7 of sum|coef|K) tier 1 runs in that form too:
            // the plain three-pass kernel's band is relative to sum|coef|K and could decide nothing there
            const bool t1cr = e->screen_variant == SCREEN_CR_POLY && e->t1_cr_available;
            ScreenParams sp_t1 = e->screen;
            if (t1cr) { sp_t1.cr_t1_tab = e->d_t1_tab.p; sp_t1.cr_t1_L = e->d_t1_L.p; }
            launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X1.p, e->d_ax1.p, d, e->range.lower,
                            e->range.upper, e->svm.neg_gamma2, list_cap, XMODE_SPLIT, sp_t1, t1_list, t1_counter,
                            e->flag0_cap, false, list_cap, e->d_attr.p, nullptr, s);
            launch_svm_h(e->d_X1.p, e->d_ax1.p, t1cr ? e->d_svt_h_cr.p : e->d_svt_h.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                         e->d_flag_list.p, e->list_cap, e->d_counters.p, d, list_cap, t1_list, t1_counter, e->flag0_cap,
                         e->d_part1.p, e->part1_stride, s, t1cr ? &e->crt1 : nullptr, t1cr ? e->d_t1_L.p : nullptr);
            }
        } else if (mode == MODE_SPLIT) {
            launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X.p, e->d_ax.p, d, e->range.lower,
                            e->range.upper, e->svm.neg_gamma2, evals_cap, XMODE_SPLIT, e->screen, nullptr, 0, 0, large, evals_sel, e->d_attr.p, nullptr, s);
            mark(e, HAF_ST_SVM);
            launch_svm_h(e->d_X.p, e->d_ax.p, e->d_svt_h.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                         e->d_flag_list.p, e->list_cap, e->d_counters.p, d, evals_cap, nullptr, 0, 0, nullptr, 0, s);
            mark(e, HAF_ST_REFINE);
        } else {
            launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X.p, e->d_ax.p, d, e->range.lower,
                            e->range.upper, e->svm.neg_gamma2, evals_cap, XMODE_F32, e->screen, nullptr, 0, 0, large, evals_sel, e->d_attr.p, nullptr, s);
            mark(e, HAF_ST_SVM);
            launch_svm(e->d_X.p, e->d_ax.p, e->d_svt.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                       e->d_flag_list.p, e->list_cap, e->d_counters.p, d, evals_cap, s);
            mark(e, HAF_ST_REFINE);
        }
        mark(e, HAF_ST_RECHECK);
        // tier 2: fp64 MFMA (GEMM form) for the guard band of the fast contraction; tier 3: libsvm's strict order for what
        // is still within 2^-40 of zero (practically nothing).  Window 0 of the tier-2 list goes with every request; the
        // strict tier is launched only when the counters that come back with the roll records say it has work (never so far).
        // tier 2a in front of it (exact8.hip): the same evaluations on EXACT integer dot products (int8 digit planes); what it
        // cannot decide either -- |dec| inside the operands' quantisation, ~1e-7 S -- is the fp64 MFMA tier's list
        // (behind the centred-remainder form of tier 1 the exact-integer tier has nothing to add: its band is the quantisation of the
        // operands relative to sum|coef|K -- 8e-9 S for the trained model, 0.14 -- and tier 1's is relative to S_psi, 0.02: measured,
        // it decided 8 of 9984 evaluations in 4.6 ms.  What tier 1 leaves goes straight to the fp64 MFMA tier)
        const bool i8 = e->i8_active && !direct && !small_exact && !(mode == MODE_SCREEN && e->screen_variant == SCREEN_CR_POLY && e->t1_cr_available);
        i8_used = i8;
        auto fp64_window = [&](int off) {
            if (small_exact)
                launch_small_direct(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_sv64.p, e->exact, d, e->flag_cap, e->d_dec_exact.p,
                                    e->d_labels.p, e->d_flag2_list.p, e->list_cap, nullptr, s, e->d_flag_list.p, CNT_FLAGGED, off);
            else if (i8)
                launch_recheck_mfma(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv64.p, e->exact, e->d_flagi_list.p, e->flag_cap, off, e->d_counters.p,
                                    e->d_x64.p, e->d_part64.p, e->d_dec_exacti.p, e->d_labels.p, e->d_flag2_list.p, e->list_cap, d, s, nullptr, false,
                                    CNT_FLAGGEDI);
            else
                launch_recheck_mfma(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv64.p, e->exact, e->d_flag_list.p, e->flag_cap, off, e->d_counters.p,
                                    e->d_x64.p, e->d_part64.p, e->d_dec_exact.p, e->d_labels.p, e->d_flag2_list.p, e->list_cap, d, s);
        };
        auto i8_window = [&](int off) {
            launch_recheck_i8(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv_i8.p, e->i8, e->range.lower, e->range.upper, e->d_flag_list.p, e->flag_cap,
                              off, e->d_counters.p, e->d_x64.p, e->d_part64.p, e->d_dec_exact.p, e->d_labels.p, e->d_flagi_list.p, e->list_cap, d, s);
        };
        if (!direct) {
            if (i8) i8_window(0);
            fp64_window(0);
        }
        // the counters come back with the roll records: a second window costs nothing unless it is needed
        auto vote = [&]() -> int {
            mark(e, HAF_ST_VOTE);
            launch_vote(e->d_labels.p, reinterpret_cast<const float *>(e->d_heights.p), e->d_brcount.p, e->d_ev16.p, e->d_topkey.p, e->d_rowmax.p, e->d_rec.p, d, s);
            mark(e, HAF_ST_DOWNLOAD);
            HIPCHK(e, hipMemcpyAsync(e->h_out, e->d_out.p, kCntBytes + (size_t)B * R * sizeof(RollRecordDev), hipMemcpyDeviceToHost, s));   // counters + records
            mark(e, HAF_ST_COUNT);
            // a short request (tens to hundreds of microseconds on the device) is waited for by polling: the wake-up of a
            // blocked host thread costs more than the request's last kernels
            if (short_request) {
                hipError_t q;
                while ((q = hipStreamQuery(s)) == hipErrorNotReady) __builtin_ia32_pause();   // (spin politely: the sibling hyper-thread may be the driver's)
                HIPCHK(e, q);
            } else {
                HIPCHK(e, hipStreamSynchronize(s));
            }
            HIPCHK(e, hipGetLastError());
            return HAF_OK;
        };
        int rc = vote();
        if (rc != HAF_OK) return rc;
        bool strict_ran = false;
        e->last_host_resolved = 0;
        const int flagged = e->h_counters[CNT_FLAGGED];
        const bool lists_valid = !direct && !(mode == MODE_SCREEN && e->h_counters[CNT_FLAGGED0] > e->flag0_cap);
        const bool more_i8 = lists_valid && i8 && flagged > e->flag_cap;
        const bool more_fp64 = lists_valid && (i8 ? e->h_counters[CNT_FLAGGEDI] > e->flag_cap : flagged > e->flag_cap);
        if (more_i8 || more_fp64) {
            // More evaluations inside a guard band than one window of an exact tier holds (an ill-conditioned model): the
            // reference never fails a goal on this path (server.cpp:778-796), so neither does the engine -- the remaining
            // windows of the lists go through the same kernels one after the other, then the strict tier over its whole list,
            // then the vote again.  Slower, same labels.
            int done_fp64 = e->flag_cap;                  // entries of its list the fp64 tier has seen (window 0)
            if (more_i8) {
                for (int off = e->flag_cap; off < flagged; off += e->flag_cap) i8_window(off);
                // the fp64 tier's list has grown behind its first window: all of it again from the start (its results and the
                // strict tier's list are rebuilt; both are idempotent)
                HIPCHK(e, hipMemsetAsync(e->d_counters.p + CNT_FLAGGED2, 0, sizeof(int), s));
                HIPCHK(e, hipMemcpyAsync(e->h_out, e->d_out.p, kCntBytes, hipMemcpyDeviceToHost, s));
                HIPCHK(e, hipStreamSynchronize(s));
                done_fp64 = 0;
            }
            const int n_fp64 = i8 ? e->h_counters[CNT_FLAGGEDI] : flagged;
            for (int off = done_fp64; off < n_fp64; off += e->flag_cap) fp64_window(off);
            launch_recheck(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv64.p, e->d_coef64.p, e->exact, e->d_flag2_list.p, e->list_cap,
                           e->d_counters.p, CNT_FLAGGED2, e->d_dec_exact2.p, e->d_labels.p, d, s);
            rc = vote();
            if (rc != HAF_OK) return rc;
            strict_ran = e->h_counters[CNT_FLAGGED2] > 0;
        } else if (e->h_counters[CNT_FLAGGED2] > 0) {
            // (the host knows the list's length here: the spread form of the tier, recheck.hip)
            launch_recheck_known(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv64.p, e->d_coef64.p, e->exact, e->d_flag2_list.p,
                                 std::min(e->h_counters[CNT_FLAGGED2], e->list_cap), e->d_strict_terms.p, kStrictSlots, e->d_dec_exact2.p, e->d_labels.p, d, s);
            rc = vote();
            if (rc != HAF_OK) return rc;
            strict_ran = true;
        }
        if (strict_ran) {
            // what the strict tier left within a last-bit exp error of zero: glibc's exp on the host, then the vote once more
            bool changed = false;
            rc = host_resolve_strict(e, d, s, &changed);
            if (rc != HAF_OK) return rc;
            if (changed) { rc = vote(); if (rc != HAF_OK) return rc; }
        }
        return HAF_OK;
    };
    // probability-output mode: every evaluation through the strict tier (libsvm's own order), then svm_predict_probability,
    // the output lines as show_predicted_gps reads them, the fp32 vote (prob.hip)
    auto decide_probability = [&]() -> int {
        mark(e, HAF_ST_FEATURES); mark(e, HAF_ST_SVM); mark(e, HAF_ST_REFINE); mark(e, HAF_ST_RECHECK);
        launch_prob_list(e->d_counters.p, CNT_FLAGGED2, e->d_flag2_list.p, e->list_cap, s);
        launch_recheck(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv64.p, e->d_coef64.p, e->exact, e->d_flag2_list.p, e->list_cap,
                       e->d_counters.p, CNT_FLAGGED2, e->d_dec_exact2.p, e->d_labels.p, d, s);
        // the estimates; those a last-bit exp difference could move come back as a list and are finished on the host (round 4)
        launch_probability_eval(e->d_dec_exact2.p, e->d_evalcell.p, e->d_counters.p, e->prob, e->d_labels.p, e->d_own.p, e->d_ptext.p,
                                e->d_flag_list.p, e->list_cap, e->d_counters.p, evals_cap, s);
        HIPCHK(e, hipMemcpyAsync(e->h_out, e->d_out.p, kCntBytes, hipMemcpyDeviceToHost, s));
        HIPCHK(e, hipStreamSynchronize(s));
        if (e->h_counters[CNT_FLAGGED] > 0) {
            const int rc = host_resolve_probability(e, d, s);
            if (rc != HAF_OK) return rc;
        } else {
            e->last_host_resolved = 0;
        }
        mark(e, HAF_ST_VOTE);
        launch_probability(e->d_dec_exact2.p, e->d_evalcell.p, e->d_counters.p, e->prob, e->d_labels.p, e->d_mask.p, e->d_rowcount.p,
                           e->d_brcount.p, reinterpret_cast<const float *>(e->d_heights.p), e->d_own.p, e->d_ptext.p, e->d_gridf.p,
                           e->d_evf.p, e->d_rec.p, evals_cap, d, s);
        mark(e, HAF_ST_DOWNLOAD);
        HIPCHK(e, hipMemcpyAsync(e->h_out, e->d_out.p, kCntBytes + (size_t)B * R * sizeof(RollRecordDev), hipMemcpyDeviceToHost, s));
        mark(e, HAF_ST_COUNT);
        HIPCHK(e, hipStreamSynchronize(s));
        HIPCHK(e, hipGetLastError());
        return HAF_OK;
    };
    int mode = contraction_mode(c);
    if (mode == MODE_SCREEN && !e->screen_active) mode = MODE_SPLIT;
    int rc = e->prob_mode ? decide_probability() : decide(mode, false);
    if (rc != HAF_OK) return rc;
    if (e->h_counters[CNT_ERROR] != 0 && !e->no_bucket_sort) {
        // a tile of k_bin_tiles had more candidate buckets than its list holds (never observed; the bound is geometric): the
        // height grids of this call may miss points.  Serve the request -- and this engine from now on -- with k_bin instead.
        e->no_bucket_sort = true;
        e->counters_clean = false;
        return score_rolls_impl(e, n_clouds, clouds, in, roll_first, roll_count, records);
    }
    const int inexact_grids = e->h_counters[CNT_INEXACT];     // (a redo of the decision stage below resets the counters)
    if (mode == MODE_SCREEN && !e->prob_mode && !direct) {
        auto undecided = [&]() { return e->h_counters[CNT_FLAGGED0]; };
        const int ne = e->h_counters[CNT_EVALS];
        // the next form of the screening pass to try when the one in use leaves too much undecided: PLAIN -> SUMSQ (the same operand
        // images: only the decision stage is redone) -> CR_EXP -> CR_POLY (their own images) -> none
        auto next_variant = [&](int v) {
            if (v == SCREEN_PLAIN) return (int)SCREEN_SUMSQ;
            if (v == SCREEN_SUMSQ && e->cr_available) return (int)SCREEN_CR_EXP;
            if (v == SCREEN_CR_EXP) return (int)SCREEN_CR_POLY;
            return -1;
        };
        // More undecided evaluations than the refinement list holds: this pass's labels are incomplete.  Remedy: the next form, and
        // stay with it; when none is left (or the variant is pinned by a test), the three-pass kernel for every evaluation of
        // this call (same labels by construction) -- and, unless pinned, no screening pass for this model from now on.
        while (undecided() > e->flag0_cap && !e->variant_forced && next_variant(e->screen_variant) >= 0) {
            const bool reuse = e->screen_variant == SCREEN_PLAIN && !t0b_used;     // (tier 0b writes its bands where the first pass's were)
            t0b_used = false;
            e->screen_variant = next_variant(e->screen_variant);
            HIPCHK(e, hipMemsetAsync(e->d_counters.p + 1, 0, (CNT_COUNT - 1) * sizeof(int), s));
            rc = decide(MODE_SCREEN, reuse);
            if (rc != HAF_OK) return rc;
        }
        if (undecided() > e->flag0_cap) {
            if (!e->variant_forced) e->screen_active = false;
            mode = MODE_SPLIT;
            HIPCHK(e, hipMemsetAsync(e->d_counters.p + 1, 0, (CNT_COUNT - 1) * sizeof(int), s));
            rc = decide(MODE_SPLIT, false);
            if (rc != HAF_OK) return rc;
        } else if (ne >= 256 && !e->variant_forced && !e->variant_settled) {
            // Adaptive rule on real requests (an engine that was not calibrated, or whose calibration scene misjudged the model): a form
            // that leaves more than a quarter undecided makes room for the next untried one; when all have been seen the engine
            // settles on the one with the lowest cost -- or on none, if even that one leaves more than 60 %.
            const double share = (double)undecided() / (double)ne;
            e->variant_share[e->screen_variant] = share;
            if (share > 0.25) {
                int nv = next_variant(e->screen_variant);
                while (nv >= 0 && e->variant_share[nv] >= 0.0) nv = next_variant(nv);       // (already seen: at calibration or on a request)
                if (nv >= 0) {
                    e->screen_variant = nv;
                } else {
                    int best = e->screen_variant;
                    double best_cost = 1e30;
                    for (int v = 0; v < SCREEN_VARIANTS; v++) {
                        if (e->variant_share[v] < 0.0) continue;
                        const double cost = kVariantCost[v] + kUndecidedCost * e->variant_share[v];
                        if (cost < best_cost) { best_cost = cost; best = v; }
                    }
                    e->screen_variant = best;
                    e->variant_settled = true;
                    if (e->variant_share[best] > 0.6) e->screen_active = false;
                }
            }
        }
    }

    if (c.flags & HAF_FLAG_PROFILE)
        for (int i = 0; i < HAF_ST_COUNT; i++) (void)hipEventElapsedTime(&e->stage_ms[i], e->ev[i], e->ev[i + 1]);

    e->last_B = B; e->last_R = R; e->last_roll_first = roll_first;
    e->last_evals = e->h_counters[CNT_EVALS];
    e->last_flagged = e->prob_mode ? 0 : e->h_counters[CNT_FLAGGED];      // (probability mode: the counter holds the estimates the host finished)
    e->last_flagged2 = e->h_counters[CNT_FLAGGED2];
    e->last_flagged0 = t0b_used ? std::min(e->h_counters[CNT_FLAGGED0B], e->h_counters[CNT_FLAGGED0]) : e->h_counters[CNT_FLAGGED0];   // what leaves the screening passes
    e->last_flaggedi = i8_used ? e->h_counters[CNT_FLAGGEDI] : e->h_counters[CNT_FLAGGED];
    e->last_inexact = inexact_grids;
    e->last_screened = (mode == MODE_SCREEN) && !e->prob_mode && !direct && e->h_counters[CNT_FLAGGED0] <= e->flag0_cap;
    // zero the counters for the next request now, behind this one's copy-out: off that request's critical path
    if (hipMemsetAsync(e->d_counters.p, 0, CNT_COUNT * sizeof(int), s) == hipSuccess) e->counters_clean = true;
    e->last_inputs.assign(in, in + B);
    // (the tier lists hold every evaluation of a request: list_cap >= last_evals >= last_flagged >= last_flagged2)
    if (e->last_flagged > e->list_cap || e->last_flagged2 > e->list_cap || e->last_flaggedi > e->list_cap) return fail(e, HAF_E_INTERNAL, "recheck list counters exceed the number of evaluations");
    e->last_i8 = i8_used;
    for (int i = 0; i < B * R; i++) {
        records[i].vote = e->h_rec[i].vote;
        records[i].row = e->h_rec[i].row;
        records[i].col = e->h_rec[i].col;
        records[i].h_locmax = e->h_rec[i].h_locmax;
        records[i].n_evals = e->h_rec[i].n_evals;
    }
    return HAF_OK;
}

// grasp pose of (row, col) found at `roll` (transform_gp_in_wcs_and_publish, server.cpp:1274-1401) into out; `av_roll` is the
// roll whose matrix the reference's av_trans_mat holds at that moment (the last one generate_grid ran, 484)
static int pose_impl(const haf_config &c, const haf_grasp_input *in, const haf_roll_record &rec, int roll, int av_roll,
                     haf_grasp_output *out, std::string &error)
{
    NormalisedInput n = normalise(*in);
    Mat4 m = roll_transform(c, *in, n, roll, false), inv;
    float x_gp_roll = -((float)(c.grid_h / 2 - rec.row)) / 100;                // 1339
    float y_gp_roll = -((float)(c.grid_w / 2 - rec.col)) / 100;                // 1340
    float h_locmax = rec.h_locmax;                                             // 1342-1351 (device, k_vote)
    h_locmax = (float)(h_locmax - 0.01);                                       // 1354
    const float x_gp_dis = 0.03f;                                              // 1360
    const float gp[2][4] = {{x_gp_roll - x_gp_dis, y_gp_roll, h_locmax, 1.0f}, {x_gp_roll + x_gp_dis, y_gp_roll, h_locmax, 1.0f}};
    if (!invert(m, inv)) { error = "transform is singular (gripper_opening_width 0?)"; return HAF_E_ARG; }
    float w[2][3];
    for (int p = 0; p < 2; p++)
        for (int i = 0; i < 3; i++) {                                          // 1367-1368
            float s = inv.a[i][0] * gp[p][0];
            s = s + inv.a[i][1] * gp[p][1];
            s = s + inv.a[i][2] * gp[p][2];
            s = s + inv.a[i][3] * gp[p][3];
            w[p][i] = s;
        }
    for (int i = 0; i < 3; i++) {
        out->grasp_point1[i] = w[0][i];
        out->grasp_point2[i] = w[1][i];
        out->averaged_grasp_point[i] = (w[0][i] + w[1][i]) / 2.0;             // 1395-1397
    }
    // av_trans_mat is the matrix of the LAST roll generate_grid ran (484); its third row does not depend on the roll
    Mat4 last = roll_transform(c, *in, n, av_roll, true);
    out->approach_vector[0] = last.a[2][0];                                    // 1370-1374
    out->approach_vector[1] = last.a[2][1];
    out->approach_vector[2] = last.a[2][2];
    out->roll = (float)((roll * c.roll_step_deg * kPi) / 180);                 // 1401
    return HAF_OK;
}

// cross-roll rule + pose; pure host arithmetic on the configuration, so it is also reachable without a device
static int finalize_impl(const haf_config &c, const haf_grasp_input *in, const haf_roll_record *rec, haf_grasp_output *out,
                         std::string &error)
{
    memset(out, 0, sizeof *out);
    // loop_control + show_predicted_gps bookkeeping: server.cpp:322-326, 362-365, 953-960
    int o_row = -1, o_col = -1, o_roll = -1, o_top = -1000, done = 0;
    int64_t evals = 0;
    // A negative budget (337: truncated to int) stops the reference's loop before roll 0 (367-374: 0 s elapsed > budget); the goal
    // still SUCCEEDS with the untouched overall best (322-326): eval -1000 - 20, roll -1.  (Its pose is then computed from row/col
    // -1, reading the height grid out of bounds at 1343-1347; the engine returns zero points instead.)
    const int n_run = ((int)in->max_calculation_time < 0) ? 0 : c.n_rolls;
    for (int r = 0; r < n_run; r++) {
        if (in->show_only_best_grasp && o_top >= c.graspval_top) break;
        if (rec[r].vote > o_top) { o_top = rec[r].vote; o_row = rec[r].row; o_col = rec[r].col; o_roll = r; }
        evals += rec[r].n_evals;
        done++;
    }
    out->best_row = o_row; out->best_col = o_col; out->best_roll = o_roll; out->best_vote = o_top;
    out->rolls_done = done;
    out->n_evals = evals;
    out->eval = o_top - 20;                                                   // 390
    if (o_roll < 0) return HAF_OK;
    return pose_impl(c, in, rec[o_roll], o_roll, std::max(0, done - 1), out, error);
}

// one roll's own hypothesis (show_predicted_gps, server.cpp:962-969)
static int roll_pose_impl(const haf_config &c, const haf_grasp_input *in, const haf_roll_record *rec, int roll, haf_grasp_output *out,
                          int32_t *published, std::string &error)
{
    memset(out, 0, sizeof *out);
    if (roll < 0 || roll >= c.n_rolls) { error = "haf_roll_pose: roll outside [0, n_rolls)"; return HAF_E_ARG; }
    const haf_roll_record &r = rec[roll];
    int scaled = r.vote - 20;                                                  // 965
    if (scaled < 10) scaled = 10;                                              // 966
    out->eval = scaled;
    out->best_row = r.row; out->best_col = r.col; out->best_roll = roll; out->best_vote = r.vote;
    out->rolls_done = roll + 1;
    out->n_evals = r.n_evals;
    if (published) *published = (!in->show_only_best_grasp && r.vote > c.graspval_th) ? 1 : 0;   // 960-962
    return pose_impl(c, in, r, roll, roll, out, error);
}

int haf_finalize(haf_engine *e, const haf_grasp_input *in, const haf_roll_record *rec, haf_grasp_output *out)
{
    if (!e) return HAF_E_ARG;
    if (!in || !rec || !out) return fail(e, HAF_E_ARG, "haf_finalize: null argument");
    return finalize_impl(e->cfg, in, rec, out, e->error);
}

static int score_batch_impl(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, haf_grasp_output *out)
{
    if (!e) return HAF_E_ARG;
    if (!out) return fail(e, HAF_E_ARG, "haf_score_batch: null output");
    std::vector<haf_roll_record> rec((size_t)std::max(1, n_clouds) * e->cfg.n_rolls);
    // A request whose every budget is negative runs no roll in the reference (server.cpp:367-374: the loop breaks before roll 0 and the
    // goal still succeeds with the untouched overall best): nothing for the device to do -- empty records, finalised below
    bool none_runs = in != nullptr && clouds != nullptr && n_clouds >= 1 && n_clouds <= e->cfg.max_clouds;
    for (int b = 0; none_runs && b < n_clouds; b++) none_runs = (int)in[b].max_calculation_time < 0;
    int rc = HAF_OK;
    if (none_runs) {
        e->last_B = e->last_R = e->last_roll_first = 0;
        e->last_evals = e->last_flagged = e->last_flagged2 = e->last_flagged0 = e->last_flaggedi = e->last_inexact = e->last_host_resolved = 0;
        e->last_i8 = e->last_screened = false;
    } else {
        rc = haf_score_rolls(e, n_clouds, clouds, in, 0, e->cfg.n_rolls, rec.data());
    }
    if (rc != HAF_OK) return rc;
    for (int b = 0; b < n_clouds; b++) {
        rc = haf_finalize(e, &in[b], rec.data() + (size_t)b * e->cfg.n_rolls, &out[b]);
        if (rc != HAF_OK) return rc;
    }
    // rechecks are counted per batch; attribute them to the first cloud's output and leave the others at 0
    out[0].n_rechecked = e->last_flagged;
    return HAF_OK;
}


static int get_roll_grid_impl(haf_engine *e, int32_t cloud, int32_t roll, float *eval_grid, uint8_t *mask)
{
    if (!e) return HAF_E_ARG;
    const int rl = roll - e->last_roll_first;
    if (cloud < 0 || cloud >= e->last_B || rl < 0 || rl >= e->last_R) return fail(e, HAF_E_ARG, "haf_get_roll_grid: (cloud, roll) not in the last scored batch");
    const size_t HW = (size_t)e->cfg.grid_h * e->cfg.grid_w, base = ((size_t)cloud * e->last_R + rl) * HW;
    if (eval_grid && e->prob_mode) {
        HIPCHK(e, hipMemcpy(eval_grid, e->d_evf.p + base, HW * sizeof(float), hipMemcpyDeviceToHost));
    } else if (eval_grid) {
        std::vector<short> tmp(HW);
        HIPCHK(e, hipMemcpy(tmp.data(), e->d_ev16.p + base, HW * sizeof(short), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < HW; i++) eval_grid[i] = (float)tmp[i];
    }
    if (mask) HIPCHK(e, hipMemcpy(mask, e->d_mask.p + base, HW, hipMemcpyDeviceToHost));
    return HAF_OK;
}

static int debug_fetch_impl(haf_engine *e, int32_t what, int32_t cloud, int32_t roll, void *dst, size_t dst_bytes)
{
    if (!e) return HAF_E_ARG;
    if (!dst) return fail(e, HAF_E_ARG, "haf_debug_fetch: null dst");
    if (!(e->cfg.flags & HAF_FLAG_KEEP_DEBUG)) return fail(e, HAF_E_ARG, "haf_debug_fetch: engine was created without HAF_FLAG_KEEP_DEBUG");
    const int rl = roll - e->last_roll_first;
    if (cloud < 0 || cloud >= e->last_B || rl < 0 || rl >= e->last_R) return fail(e, HAF_E_ARG, "haf_debug_fetch: (cloud, roll) not in the last scored batch");
    const size_t H = (size_t)e->cfg.grid_h, W = (size_t)e->cfg.grid_w, HW = H * W;
    const size_t br = (size_t)cloud * e->last_R + rl;
    auto need = [&](size_t n) { return dst_bytes >= n; };
    switch (what) {
        case HAF_DBG_HEIGHTS:
            if (!need(HW * 4)) break;
            HIPCHK(e, hipMemcpy(dst, e->d_heights.p + br * HW, HW * 4, hipMemcpyDeviceToHost));
            return HAF_OK;
        case HAF_DBG_INTEGRAL:
            if (!need((H + 1) * (W + 1) * 4)) break;
            HIPCHK(e, hipMemcpy(dst, e->d_ii.p + br * (H + 1) * (W + 1), (H + 1) * (W + 1) * 4, hipMemcpyDeviceToHost));
            return HAF_OK;
        case HAF_DBG_MASK:
            if (!need(HW)) break;
            HIPCHK(e, hipMemcpy(dst, e->d_mask.p + br * HW, HW, hipMemcpyDeviceToHost));
            return HAF_OK;
        case HAF_DBG_LABELS:
            if (!need(HW)) break;
            HIPCHK(e, hipMemcpy(dst, e->d_labels.p + br * HW, HW, hipMemcpyDeviceToHost));
            return HAF_OK;
        case HAF_DBG_TRANSFORM: {
            if (!need(16 * 4)) break;
            NormalisedInput n = normalise(e->last_inputs[(size_t)cloud]);
            Mat4 m = roll_transform(e->cfg, e->last_inputs[(size_t)cloud], n, roll, true);
            memcpy(dst, m.a, 16 * 4);
            return HAF_OK;
        }
        case HAF_DBG_DECISION: {
            if (!need(HW * 8)) break;
            double *g = (double *)dst;
            for (size_t i = 0; i < HW; i++) g[i] = NAN;
            const size_t ne = (size_t)e->last_evals;
            if (!ne) return HAF_OK;
            std::vector<int> cell(ne);
            std::vector<float> dec(ne);
            HIPCHK(e, hipMemcpy(cell.data(), e->d_evalcell.p, ne * 4, hipMemcpyDeviceToHost));
            HIPCHK(e, hipMemcpy(dec.data(), e->d_dec.p, ne * 4, hipMemcpyDeviceToHost));
            const size_t nfl = (size_t)std::min(e->last_flagged, e->list_cap);
            std::vector<int> fl(nfl);
            std::vector<double> ex(nfl);
            if (nfl) {
                HIPCHK(e, hipMemcpy(fl.data(), e->d_flag_list.p, nfl * 4, hipMemcpyDeviceToHost));
                HIPCHK(e, hipMemcpy(ex.data(), e->d_dec_exact.p, nfl * 8, hipMemcpyDeviceToHost));
            }
            std::vector<double> d64(dec.begin(), dec.end());
            for (size_t k = 0; k < nfl; k++) d64[(size_t)fl[k]] = ex[k];
            if (e->last_i8) {                              // behind tier 2a the fp64 tier has its own list and values
                const size_t nfi = (size_t)std::min(e->last_flaggedi, e->list_cap);
                if (nfi) {
                    std::vector<int> fli(nfi);
                    std::vector<double> exi(nfi);
                    HIPCHK(e, hipMemcpy(fli.data(), e->d_flagi_list.p, nfi * 4, hipMemcpyDeviceToHost));
                    HIPCHK(e, hipMemcpy(exi.data(), e->d_dec_exacti.p, nfi * 8, hipMemcpyDeviceToHost));
                    for (size_t k = 0; k < nfi; k++) d64[(size_t)fli[k]] = exi[k];
                }
            }
            const size_t nf2 = (size_t)std::min(e->last_flagged2, e->list_cap);
            if (nf2) {
                std::vector<int> fl2(nf2);
                std::vector<double> ex2(nf2);
                HIPCHK(e, hipMemcpy(fl2.data(), e->d_flag2_list.p, nf2 * 4, hipMemcpyDeviceToHost));
                HIPCHK(e, hipMemcpy(ex2.data(), e->d_dec_exact2.p, nf2 * 8, hipMemcpyDeviceToHost));
                for (size_t k = 0; k < nf2; k++) d64[(size_t)fl2[k]] = ex2[k];
            }
            for (size_t k = 0; k < ne; k++) {
                size_t cb = (size_t)cell[k] / HW;
                if (cb == br) g[(size_t)cell[k] - cb * HW] = d64[k];
            }
            return HAF_OK;
        }
        case HAF_DBG_GRASPSGRID:
            if (!e->prob_mode) return fail(e, HAF_E_ARG, "haf_debug_fetch: HAF_DBG_GRASPSGRID needs HAF_FLAG_PROBABILITY");
            if (!need(HW * 4)) break;
            HIPCHK(e, hipMemcpy(dst, e->d_gridf.p + br * HW, HW * 4, hipMemcpyDeviceToHost));
            return HAF_OK;
        case HAF_DBG_PROBABILITY: {
            if (!e->prob_mode) return fail(e, HAF_E_ARG, "haf_debug_fetch: HAF_DBG_PROBABILITY needs HAF_FLAG_PROBABILITY");
            if (!need(HW * 16)) break;
            double *g = (double *)dst;
            for (size_t i = 0; i < 2 * HW; i++) g[i] = NAN;
            const size_t ne = (size_t)e->last_evals;
            if (!ne) return HAF_OK;
            std::vector<int> cell(ne);
            std::vector<double> pt(2 * ne);
            HIPCHK(e, hipMemcpy(cell.data(), e->d_evalcell.p, ne * 4, hipMemcpyDeviceToHost));
            HIPCHK(e, hipMemcpy(pt.data(), e->d_ptext.p, 2 * ne * 8, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < ne; k++) {
                size_t cb = (size_t)cell[k] / HW;
                if (cb == br) { g[2 * ((size_t)cell[k] - cb * HW)] = pt[2 * k]; g[2 * ((size_t)cell[k] - cb * HW) + 1] = pt[2 * k + 1]; }
            }
            return HAF_OK;
        }
        case HAF_DBG_SCREEN_MARGIN: {
            if (!need(HW * 4)) break;
            float *g = (float *)dst;
            for (size_t i = 0; i < HW; i++) g[i] = NAN;
            const size_t ne = (size_t)e->last_evals;
            if (!ne || !e->d_margin.p || !e->last_screened) return HAF_OK;
            std::vector<int> cell(ne);
            std::vector<float> mg(ne);
            HIPCHK(e, hipMemcpy(cell.data(), e->d_evalcell.p, ne * 4, hipMemcpyDeviceToHost));
            HIPCHK(e, hipMemcpy(mg.data(), e->d_margin.p, ne * 4, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < ne; k++) {
                size_t cb = (size_t)cell[k] / HW;
                if (cb == br) g[(size_t)cell[k] - cb * HW] = mg[k];
            }
            return HAF_OK;
        }
        default:
            return fail(e, HAF_E_ARG, "haf_debug_fetch: unknown item");
    }
    return fail(e, HAF_E_ARG, "haf_debug_fetch: dst too small");
}

static int debug_fetch_attr_impl(haf_engine *e, int32_t cloud, int32_t roll, int32_t max_cells, int32_t *cells, haf_attr_record *attr,
                                 uint8_t *computed, int32_t *n_cells)
{
    if (!e) return HAF_E_ARG;
    if (!n_cells || max_cells < 0) return fail(e, HAF_E_ARG, "haf_debug_fetch_attr: bad argument");
    if (!(e->cfg.flags & HAF_FLAG_KEEP_DEBUG)) return fail(e, HAF_E_ARG, "haf_debug_fetch_attr: engine was created without HAF_FLAG_KEEP_DEBUG");
    if (!e->d_attr.p) return fail(e, HAF_E_CAPACITY, "haf_debug_fetch_attr: attribute records are kept for engines of up to 2 GiB of them only");
    const int rl = roll - e->last_roll_first;
    if (cloud < 0 || cloud >= e->last_B || rl < 0 || rl >= e->last_R) return fail(e, HAF_E_ARG, "haf_debug_fetch_attr: (cloud, roll) not in the last scored batch");
    const size_t H = (size_t)e->cfg.grid_h, W = (size_t)e->cfg.grid_w, HW = H * W;
    const size_t br = (size_t)cloud * e->last_R + rl;
    const size_t ne = (size_t)e->last_evals;
    std::vector<int> cell(ne);
    if (ne) HIPCHK(e, hipMemcpy(cell.data(), e->d_evalcell.p, ne * 4, hipMemcpyDeviceToHost));
    std::vector<int> eval_of(HW, -1);
    for (size_t k = 0; k < ne; k++)
        if ((size_t)cell[k] / HW == br) eval_of[(size_t)cell[k] - br * HW] = (int)k;
    int n = 0;
    std::vector<haf_attr_record> row((size_t)kKP);
    for (size_t idx = 0; idx < HW; idx++) {                     // row-major = the reference's line order
        if (eval_of[idx] < 0) continue;
        if (n < max_cells) {
            if (cells) { cells[2 * n] = (int)(idx / W); cells[2 * n + 1] = (int)(idx % W); }
            if (attr || computed) {
                HIPCHK(e, hipMemcpy(row.data(), e->d_attr.p + (size_t)eval_of[idx] * kKP, (size_t)kKP * sizeof(AttrRecord), hipMemcpyDeviceToHost));
                uint32_t bits;
                memcpy(&bits, &row[0].feature, 4);
                if (computed) computed[n] = bits != 0xFFFFFFFFu;
                if (attr) memcpy(attr + (size_t)n * kKP, row.data(), (size_t)kKP * sizeof(haf_attr_record));
            }
        }
        n++;
    }
    *n_cells = n;
    return HAF_OK;
}

int haf_debug_fetch_attr(haf_engine *e, int32_t cloud, int32_t roll, int32_t max_cells, int32_t *cells, haf_attr_record *attr,
                         uint8_t *computed, int32_t *n_cells)
{
    return guarded(e ? &e->error : nullptr, [&] { return debug_fetch_attr_impl(e, cloud, roll, max_cells, cells, attr, computed, n_cells); });
}

int haf_get_stage_ms(haf_engine *e, float *ms)
{
    if (!e || !ms) return HAF_E_ARG;
    if (!(e->cfg.flags & HAF_FLAG_PROFILE)) return fail(e, HAF_E_ARG, "engine was created without HAF_FLAG_PROFILE");
    memcpy(ms, e->stage_ms, sizeof e->stage_ms);
    return HAF_OK;
}

static int pcd_load_impl(const char *path, float **xyz, size_t *n_points, char *err, size_t err_cap)
{
    if (!path || !xyz || !n_points) return HAF_E_ARG;
    std::vector<float> v;
    std::string msg;
    if (!load_pcd(path, v, msg)) {
        if (err && err_cap) snprintf(err, err_cap, "%s", msg.c_str());
        return HAF_E_IO;
    }
    *n_points = v.size() / 3;
    *xyz = (float *)malloc(std::max<size_t>(1, v.size()) * sizeof(float));
    if (!*xyz) return HAF_E_INTERNAL;
    memcpy(*xyz, v.data(), v.size() * sizeof(float));
    return HAF_OK;
}

void haf_free(void *p) { free(p); }

int haf_create(const haf_config *cfg, haf_engine **out)
{
    return guarded(&g_create_error, [&] { return create_impl(cfg, out); });
}

int haf_score_rolls(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, int32_t roll_first,
                    int32_t roll_count, haf_roll_record *records)
{
    return guarded(e ? &e->error : nullptr, [&] { return score_rolls_impl(e, n_clouds, clouds, in, roll_first, roll_count, records); });
}

int haf_roll_pose(haf_engine *e, const haf_grasp_input *in, const haf_roll_record *rec, int32_t roll, haf_grasp_output *out,
                  int32_t *published)
{
    if (!e) return HAF_E_ARG;
    if (!in || !rec || !out) return fail(e, HAF_E_ARG, "haf_roll_pose: null argument");
    return roll_pose_impl(e->cfg, in, rec, roll, out, published, e->error);
}

int haf_score_batch(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, haf_grasp_output *out)
{
    return guarded(e ? &e->error : nullptr, [&] { return score_batch_impl(e, n_clouds, clouds, in, out); });
}

int haf_score(haf_engine *e, const haf_cloud *cloud, const haf_grasp_input *in, haf_grasp_output *out)
{
    return haf_score_batch(e, 1, cloud, in, out);
}

int haf_get_roll_grid(haf_engine *e, int32_t cloud, int32_t roll, float *eval_grid, uint8_t *mask)
{
    return guarded(e ? &e->error : nullptr, [&] { return get_roll_grid_impl(e, cloud, roll, eval_grid, mask); });
}

int haf_debug_fetch(haf_engine *e, int32_t what, int32_t cloud, int32_t roll, void *dst, size_t dst_bytes)
{
    return guarded(e ? &e->error : nullptr, [&] { return debug_fetch_impl(e, what, cloud, roll, dst, dst_bytes); });
}

int haf_pcd_load(const char *path, float **xyz, size_t *n_points, char *err, size_t err_cap)
{
    std::string msg;
    const int rc = guarded(&msg, [&] { return pcd_load_impl(path, xyz, n_points, err, err_cap); });
    if (rc == HAF_E_INTERNAL && !msg.empty() && err && err_cap) snprintf(err, err_cap, "%s", msg.c_str());
    return rc;
}


#ifdef HAF_TESTING
// ---- the hooks below exist in libhafgrasp_testing.so only (-DHAF_TESTING); the product library does not export them ----
// host-only hooks: parsers, per-roll geometry and the cross-roll rule/pose, none of which touches a device
int haf_test_feature_table(const char *path, int *n, int *reg /* cap*16 */, float *w /* cap*4 */, int cap)
{
    std::vector<FeatureRow> rows;
    std::string err;
    if (!load_features(path, rows, err)) return HAF_E_IO;
    *n = (int)rows.size();
    for (int i = 0; i < *n && i < cap; i++) {
        memcpy(reg + i * 16, rows[(size_t)i].reg, sizeof rows[0].reg);
        memcpy(w + i * 4, rows[(size_t)i].w, sizeof rows[0].w);
    }
    return HAF_OK;
}

int haf_test_range_table(const char *path, double *lower, double *upper, int *max_index, double *fmin, double *fmax,
                         unsigned char *present, int cap)
{
    RangeTable rt;
    std::string err;
    if (!load_range(path, rt, err)) return HAF_E_IO;
    *lower = rt.lower; *upper = rt.upper; *max_index = rt.max_index;
    for (int i = 0; i <= rt.max_index && i < cap; i++) { fmin[i] = rt.fmin[(size_t)i]; fmax[i] = rt.fmax[(size_t)i]; present[i] = rt.present[(size_t)i]; }
    return HAF_OK;
}

int haf_test_model(const char *path, double *gamma, double *rho, int *n_sv, int *dim, int *n_sv_class, int *label, double *coef,
                   double *sv, long cap_sv_values)
{
    SvmModel m;
    std::string err;
    if (!load_model(path, m, err)) return HAF_E_IO;
    *gamma = m.gamma; *rho = m.rho; *n_sv = m.n_sv; *dim = m.dim;
    n_sv_class[0] = m.n_sv_class[0]; n_sv_class[1] = m.n_sv_class[1];
    label[0] = m.label[0]; label[1] = m.label[1];
    if (coef && sv && (long)m.sv.size() <= cap_sv_values) {
        memcpy(coef, m.coef.data(), m.coef.size() * sizeof(double));
        memcpy(sv, m.sv.data(), m.sv.size() * sizeof(double));
    }
    return HAF_OK;
}

// out: 12 transform floats, then sa, ca, cx1, cy1, cx2, cy2, cx3, cy3, cx4, cy4; full 4x4 (generate_grid form) in m16
int haf_test_roll_geo(const haf_config *cfg, const haf_grasp_input *in, int roll, float *out22, float *m16, float *m16_pose)
{
    NormalisedInput n = normalise(*in);
    RollGeo g;
    fill_roll_geo(*cfg, *in, n, roll, g);
    memcpy(out22, g.m, 12 * 4);
    const float tail[10] = {g.sa, g.ca, g.cx1, g.cy1, g.cx2, g.cy2, g.cx3, g.cy3, g.cx4, g.cy4};
    memcpy(out22 + 12, tail, sizeof tail);
    if (m16) { Mat4 m = roll_transform(*cfg, *in, n, roll, true); memcpy(m16, m.a, 64); }
    if (m16_pose) { Mat4 m = roll_transform(*cfg, *in, n, roll, false); memcpy(m16_pose, m.a, 64); }
    return HAF_OK;
}

int haf_test_finalize(const haf_config *cfg, const haf_grasp_input *in, const haf_roll_record *rec, haf_grasp_output *out)
{
    std::string err;
    return finalize_impl(*cfg, in, rec, out, err);
}

int haf_test_roll_pose(const haf_config *cfg, const haf_grasp_input *in, const haf_roll_record *rec, int roll, haf_grasp_output *out,
                       int32_t *published)
{
    std::string err;
    return roll_pose_impl(*cfg, in, rec, roll, out, published, err);
}

// ---- test hooks (host and device builds of the decimal round-trip arithmetic; see tests/) ----
double haf_test_decq_host(double x, int digits) { return digits == 40 ? hafq::decq4_float((float)x) : hafq::decq(x, digits); }
double haf_test_scale_host(double q4, double fmin, double fmax, double lower, double upper)
{
    const double range = fmax - fmin;
    return hafq::scale_q6(q4, fmin, fmax, range, 1.0 / range, lower, upper);
}

// host-side pieces of the screening band (tests/test_host_cpu.py)
double haf_test_sigma_upper(const double *M, int n, int d) { return sigma_upper_bound(M, n, d); }
double haf_test_split3(double a, float *parts)
{
    _Float16 h[3];
    const double rep = split3_f16(a, h);
    for (int i = 0; i < 3; i++) parts[i] = (float)h[i];
    return rep;
}

double haf_test_decq4_scr(float v)
{
    static unsigned long long tab[hafq::kScrTabWords];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < hafq::kScrTabWords; i++) tab[i] = hafq::scr_tab_word(i);
        init = true;
    }
    hafq::ScrTabs st;
    st.w = tab;
    return hafq::decq4_float_scr(v, st);
}

// runs the screening kernel's MFMA chain on host-chosen data (testkernels.hip); a, b: fp16 bit patterns
int haf_test_mfma_accum(const uint16_t *a, const uint16_t *b, const float *c0, float *out, int trials)
{
    void *da = nullptr, *db = nullptr;
    float *dc = nullptr, *dout = nullptr;
    const size_t na = (size_t)trials * 16 * 320 * 2, nc = (size_t)trials * 16 * 4, no = (size_t)trials * 256 * 4;
    if (hipMalloc(&da, na) != hipSuccess || hipMalloc(&db, na) != hipSuccess || hipMalloc((void **)&dc, nc) != hipSuccess ||
        hipMalloc((void **)&dout, no) != hipSuccess) return HAF_E_DEVICE;
    (void)hipMemcpy(da, a, na, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b, na, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, c0, nc, hipMemcpyHostToDevice);
    haf::launch_mfma_accum_test(da, db, dc, dout, trials, nullptr);
    const hipError_t rc = hipMemcpy(out, dout, no, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dout);
    return rc == hipSuccess ? HAF_OK : HAF_E_DEVICE;
}

// bare v_mfma_f32_16x16x32_f16 loop on `device` for about `iters` * 0.55 us: executed TFLOP/s by HIP events (bench.py context)
int haf_test_mfma_rate(int device, int iters, double *tflops)      // iters < 0: v_mfma_i32_16x16x64_i8 (TOP/s), else v_mfma_f32_16x16x32_f16
{
    if (!tflops || iters == 0) return HAF_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return HAF_E_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HAF_E_DEVICE;
    const int blocks = 2 * prop.multiProcessorCount;
    std::vector<uint16_t> h(65536 * 8);
    uint32_t x = 12345u;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3000u | ((x >> 9) & 0x83FFu)); }   // +-[0.125, 0.25): random mantissas and signs
    void *din = nullptr;
    float *dout = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = HAF_E_DEVICE;
    float ms = 0.0f;
    if (hipMalloc(&din, h.size() * 2) == hipSuccess && hipMalloc((void **)&dout, (size_t)blocks * 256 * 4) == hipSuccess &&
        hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
        hipEventCreate(&e1) == hipSuccess) {
        haf::launch_mfma_rate_test(din, dout, blocks, iters < 0 ? -64 : 64, nullptr);    // warm the code path
        (void)hipEventRecord(e0, nullptr);
        haf::launch_mfma_rate_test(din, dout, blocks, iters, nullptr);
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.0f) {
            *tflops = (double)blocks * 4.0 * std::abs(iters) * 32.0 * (iters < 0 ? 32768.0 : 16384.0) / (ms * 1e-3) / 1e12;
            rc = HAF_OK;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(din); (void)hipFree(dout);
    return rc;
}

// timing model of the screening kernel's inner loop with mb = 4 or 8 row blocks per wave (testkernels.hip): executed TFLOP/s
int haf_test_mfma_model(int device, int mb, int tiles, double *tflops)
{
    if (!tflops || tiles < 1 || (mb != 4 && mb != 5 && mb != 8 && mb != 9)) return HAF_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return HAF_E_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HAF_E_DEVICE;
    const int blocks = (mb >= 5 ? 1 : 2) * prop.multiProcessorCount * 8;          // eight rounds of workgroups
    const int mbe = mb == 9 ? 8 : (mb == 5 ? 8 : mb);                              // (9 = the hand-placed form of 8; 5 = 4 row blocks x 8 waves: same flop per workgroup as 8)
    std::vector<uint16_t> h(65536 * 8);
    uint32_t x = 777u;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x2800u | ((x >> 9) & 0x83FFu)); }
    void *din = nullptr;
    float *dout = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = HAF_E_DEVICE;
    float ms = 0.0f;
    // (out holds one float per thread of the widest form: 512 threads per workgroup)
    if (hipMalloc(&din, h.size() * 2) == hipSuccess && hipMalloc((void **)&dout, (size_t)blocks * 512 * 4) == hipSuccess &&
        hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
        hipEventCreate(&e1) == hipSuccess) {
        haf::launch_mfma_model_test(din, dout, mb, blocks, 2, nullptr);
        (void)hipEventRecord(e0, nullptr);
        haf::launch_mfma_model_test(din, dout, mb, blocks, tiles, nullptr);
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.0f) {
            *tflops = (double)blocks * 4.0 * tiles * 20.0 * mbe * 16384.0 / (ms * 1e-3) / 1e12;
            rc = HAF_OK;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(din); (void)hipFree(dout);
    return rc;
}

// v_mfma_i32_16x16x64_i8 on host-chosen int8 data: a [16][64], b [64][16] row-major -> c [16][16] (testkernels.hip)
int haf_test_i8_mfma(const signed char *a, const signed char *b, int *c)
{
    void *da = nullptr, *db = nullptr;
    int *dc = nullptr;
    if (hipMalloc(&da, 1024) != hipSuccess || hipMalloc(&db, 1024) != hipSuccess || hipMalloc((void **)&dc, 1024) != hipSuccess) return HAF_E_DEVICE;
    (void)hipMemcpy(da, a, 1024, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b, 1024, hipMemcpyHostToDevice);
    haf::launch_i8_layout_probe(da, db, dc, nullptr);
    const hipError_t rc = hipMemcpy(c, dc, 1024, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc);
    return rc == hipSuccess ? HAF_OK : HAF_E_DEVICE;
}

// which form of the screening pass serves the model, whether the pass is on, and the undecided shares calibrate() saw per form
int haf_test_screen_state(haf_engine *e, int *variant, int *active, double *shares /* [4] */)
{
    if (!e) return HAF_E_ARG;
    if (variant) *variant = e->screen_variant | (e->use_t0b ? 16 : 0) | (e->t1_skip ? 32 : 0);
    if (active) *active = e->screen_active ? 1 : 0;
    if (shares) for (int i = 0; i < SCREEN_VARIANTS; i++) shares[i] = e->variant_share[i];
    return HAF_OK;
}

// the engine's matrix-core rounding constant: what the probe measured and what the bands use
int haf_test_mfma_kappa(haf_engine *e, double *measured, double *used)      // [0]: 16x16x32, [1]: 16x16x16
{
    if (!e) return HAF_E_ARG;
    measured[0] = e->mfma_kappa_measured; used[0] = e->mfma_kappa;
    measured[1] = e->mfma_kappa16_measured; used[1] = e->mfma_kappa16;
    return HAF_OK;
}

// v_mfma_f32_16x16x32_f16 on host-chosen data (testkernels.hip: k_f16_mfma_probe)
int haf_test_f16_mfma(const unsigned short *a, const unsigned short *b, const float *c, float *d, int trials, int chain)
{
    void *da = nullptr, *db = nullptr;
    float *dc = nullptr, *dd = nullptr;
    const size_t na = (size_t)trials * 1024, nc = (size_t)trials * 1024;
    if (hipMalloc(&da, na) != hipSuccess || hipMalloc(&db, na) != hipSuccess || hipMalloc((void **)&dc, nc) != hipSuccess ||
        hipMalloc((void **)&dd, nc) != hipSuccess) return HAF_E_DEVICE;
    (void)hipMemcpy(da, a, na, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b, na, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, c, nc, hipMemcpyHostToDevice);
    haf::launch_f16_mfma_probe(da, db, dc, dd, trials, chain, nullptr);
    const hipError_t rc = hipMemcpy(d, dd, nc, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dd);
    return rc == hipSuccess ? HAF_OK : HAF_E_DEVICE;
}

int haf_test_decq_device(const double *in, double *out, int n, int digits)
{
    double *di = nullptr, *dout = nullptr;
    if (hipMalloc((void **)&di, (size_t)n * 8) != hipSuccess || hipMalloc((void **)&dout, (size_t)n * 8) != hipSuccess) return HAF_E_DEVICE;
    (void)hipMemcpy(di, in, (size_t)n * 8, hipMemcpyHostToDevice);
    launch_decq_test(di, dout, n, digits, nullptr);
    hipError_t rc = hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(di); (void)hipFree(dout);
    return rc == hipSuccess ? HAF_OK : HAF_E_DEVICE;
}

int haf_test_scale_device(const double *q4, const double *fmin, const double *fmax, double lower, double upper, double *out, int n)
{
    double *d[4] = {nullptr, nullptr, nullptr, nullptr};
    for (auto &p : d) if (hipMalloc((void **)&p, (size_t)n * 8) != hipSuccess) return HAF_E_DEVICE;
    (void)hipMemcpy(d[0], q4, (size_t)n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d[1], fmin, (size_t)n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d[2], fmax, (size_t)n * 8, hipMemcpyHostToDevice);
    launch_scale_test(d[0], d[1], d[2], lower, upper, d[3], n, nullptr);
    hipError_t rc = hipMemcpy(out, d[3], (size_t)n * 8, hipMemcpyDeviceToHost);
    for (auto &p : d) (void)hipFree(p);
    return rc == hipSuccess ? HAF_OK : HAF_E_DEVICE;
}
#endif  // HAF_TESTING

}  // extern "C"
