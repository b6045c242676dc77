// engine_state.h -- the engine object and what the host-side translation units of libhafgrasp.so share:
//   engine.cpp           create / destroy, calibration of the screening pass at creation, the C-ABI entry points
//   engine_tables.cpp    model, range and feature tables -> device tables and the constants of every guard band; buffers
//   engine_request.cpp   one request: stage launches, decision tiers, host resolution of the residual cases, the batch wrapper
//   engine_geometry.cpp  per-roll transforms, the rotated-rectangle scalars, the final grasp pose (host fp32, glibc)
//   engine_debug.cpp     haf_get_roll_grid / haf_debug_fetch* (intermediate stages for the parity tests)
//   engine_testing.cpp   haf_test_* hooks (libhafgrasp_testing.so only)
// Private to csrc/: not installed, nothing here is part of the ABI (include/hafgrasp.h).  Every translation unit above is
// compiled twice, without and with -DHAF_TESTING (test_env below), for the product and the testing library.
#pragma once
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

namespace haf_host {

// variable must not be able to scale them in the library a server links.
#ifdef HAF_TESTING
inline const char *test_env(const char *name) { return getenv(name); }
#else
inline const char *test_env(const char *) { return nullptr; }
#endif

constexpr double kPi = 3.141592653;   // server.cpp:94 -- the reference's truncated constant, NOT M_PI

struct Mat4 {
    float a[4][4];
    static Mat4 identity()
    {
        Mat4 m;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.a[i][j] = (i == j) ? 1.0f : 0.0f;
        return m;
    }
};

struct NormalisedInput {
    double av[3];      // approach vector after server.cpp:270-273
    int sx, sy;        // grasp_search_area_size_{x,y}_dir (266-267)
    int width;         // gripper_opening_width (281)
};

// engine_geometry.cpp
Mat4 operator*(const Mat4 &l, const Mat4 &r);
NormalisedInput normalise(const haf_grasp_input &in);
Mat4 roll_transform(const haf_config &cfg, const haf_grasp_input &in, const NormalisedInput &n, int roll, bool from_float_av,
                    Mat4 *pre_roll = nullptr, float *roll_cs = nullptr);
void fill_roll_geo(const haf_config &cfg, const haf_grasp_input &in, const NormalisedInput &n, int roll, RollGeo &g, float *m0 = nullptr);
bool invert(const Mat4 &m, Mat4 &inv);

// Testing build: every device buffer lies between two guard zones filled with kCanaryByte -- kCanaryGuard bytes in front, and from
// the buffer's last byte to the next multiple of kCanaryGuard plus kCanaryGuard behind -- and is registered with the source line that
// allocated it (engine_testing.cpp: canary_check).  A kernel that writes one element past a list, an operand image or a flag-word
// array changes a guard byte; read as a list entry the pattern is a NEGATIVE evaluation id (0xA5A5A5A5), which no consumer may follow.
// The tests check the zones after every request (HAF_CANARY_CHECK=1 in tests/conftest.py; haf_test_check_canaries).  The product
// library allocates exactly what is asked for.
#ifdef HAF_TESTING
constexpr size_t kCanaryGuard = 256;
constexpr int kCanaryByte = 0xA5;
void canary_register(void *user, size_t bytes, const char *file, int line);
void canary_unregister(void *user);
int canary_check(std::string *report);          // number of buffers with a damaged guard zone (engine_testing.cpp)
#endif

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
#ifdef HAF_TESTING
    hipError_t alloc(size_t count, const char *file = __builtin_FILE(), int line = __builtin_LINE())
    {
        n = count;
        if (!count) return hipSuccess;
        const size_t bytes = count * sizeof(T), padded = (bytes + kCanaryGuard - 1) / kCanaryGuard * kCanaryGuard;
        char *raw = nullptr;
        hipError_t rc = hipMalloc((void **)&raw, kCanaryGuard + padded + kCanaryGuard);
        if (rc != hipSuccess) return rc;
        rc = hipMemset(raw, kCanaryByte, kCanaryGuard);
        if (rc == hipSuccess) rc = hipMemset(raw + kCanaryGuard + bytes, kCanaryByte, padded - bytes + kCanaryGuard);
        if (rc != hipSuccess) { (void)hipFree(raw); return rc; }
        p = reinterpret_cast<T *>(raw + kCanaryGuard);
        canary_register(p, bytes, file, line);
        return hipSuccess;
    }
    void release()
    {
        if (p) { canary_unregister(p); (void)hipFree(reinterpret_cast<char *>(p) - kCanaryGuard); }
        p = nullptr; n = 0;
    }
#else
    hipError_t alloc(size_t count)
    {
        n = count;
        if (!count) return hipSuccess;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
#endif
};

}  // namespace haf_host

using namespace haf_host;

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
    bool generic_kernel = false; // the model's kernel is not RBF: every evaluation through the libsvm-order tier (engine.cpp, engine_request.cpp)
    bool screen_active = true;   // default mode only: cleared once more than 60 % of a call's evaluations fell inside the band of every
                                 // form of the screening pass -- for such a model the single pass is wasted work.  Not for good: every
                                 // reprobe_every-th full-size request afterwards tries the pass again (the judgement may have come from
                                 // the synthetic calibration scene or from one unusual cloud) and switches it back on when it pays
    int inactive_calls = 0;      // full-size requests served without the screening pass since it was switched off / last re-tried
    int reprobe_every = 64;      // (testing build: HAF_REPROBE_EVERY)
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
    // partial class sums of the screening kernel's PART form (requests that do not fill the chip are split over SV ranges: screen.hip)
    DevBuf<char> d_screen_part;
    int screen_parts = 0;        // 0: by the live evaluation count; testing build (HAF_SCREEN_PARTS): 1 = never, n = forced
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
    DevBuf<unsigned char> d_t1_flags;          // tier 1: one "undecided" byte per entry (list slot, or evaluation in the all-evaluations modes)
    DevBuf<unsigned long long> d_tier_words;   // exact tiers: one "undecided" bit per entry of a window, for the ordered hand-over lists
    DevBuf<unsigned long long> d_brslot;   // k_small_pre: per (cloud, roll) {request epoch, evaluations} in one word (ordered evaluation list)
    unsigned pre_epoch = 0;
    struct View { int *p = nullptr; } d_counters;      // inside d_out
    DevBuf<float> d_X, d_ax, d_dec, d_svt;
    DevBuf<char> d_svt_h;            // split-fp16 SV tile images
    DevBuf<char> d_svt0;             // screening-pass SV tile images
    DevBuf<char> d_svt0_cr;          // the same for the centred-remainder form: fp16(w_n - mu), t_n = 0, coefficient b_n
    // low-rank form of the centred-remainder pass (kernels.h: kLrK; large requests only): projection tiles (rows of B^'), 6-step SV
    // tile images of q~_n, the band's constants, per (cloud, roll) "a height is negative" flags (prestages.hip)
    bool lr_available = false;
    bool lr_enabled = true;          // testing build: HAF_NO_LR switches it off; HAF_LR_ALWAYS lifts the request-size rule
    bool lr_always = false;
    int lr_rank = 0;                 // dimension of the HAF slots' linear span (158 for the reference's Features.txt)
    DevBuf<char> d_lr_btiles, d_svt_lr;
    ScreenParams screen_lrp{};       // the feature kernel's constants for the PLAIN epilogue in the low-rank form (centred descriptors, own correction vectors)
    DevBuf<ScrCorr> d_corr_lrp;
    bool lr_plain_available = false;
    DevBuf<char> d_lr_btiles_in;     // the projection matrix by input k-step (fused form: the projection is the sweep's prologue)
    bool lr_fused = true;            // testing build: HAF_LR_UNFUSED = k_project + sweep as two launches
    DevBuf<unsigned long long> d_iiabs;   // per (cloud, roll): sum of |height| in units of 2^-20 m (k_integral_totals)
    LrBand lr_band{};
    bool last_lr = false;            // the last request's screening pass ran in the low-rank form
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
    int last_bypass = 0;            // ... of them through the short-list gate (engine_request.cpp), around tier 1 and the exact-integer tier
    bool short_gate = true;         // testing build: HAF_NO_SHORT_GATE switches the gate off
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

    // how often a request met a list smaller than what it had to hold (the overflow campaigns read them: haf_test_overflow_stats)
    long stat_flag0_overflows = 0;  // the screening passes left more undecided than their list holds: decision stage redone
    long stat_extra_windows = 0;    // windows of the exact tiers' lists beyond the first
    // last call
    int last_B = 0, last_R = 0, last_roll_first = 0;
    int last_evals = 0, last_flagged = 0, last_flagged2 = 0, last_flagged0 = 0, last_inexact = 0;
    bool last_screened = false;     // the last call's labels came through the screening tier (not its three-pass fallback)
    std::vector<haf_grasp_input> last_inputs;
};

namespace haf_host {

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
// with the low-rank form (kernels.h: kLrK) serving the engine's full-size requests, in units of the TEN-step plain kernel (13.8 ms at C5,
// 4096 SVs; the feature kernel's 0.8 ms of noise bounds included): plain epilogue 10.3 + 0.8, CR_EXP 11.6 + 0.8, CR_POLY 24.3 + 0.8
// against 28.9 ms (plain equivalent at 8964 SVs); SUMSQ has no low-rank form
constexpr double kVariantCostLr[SCREEN_VARIANTS] = {0.80, 1.12, 0.90, 0.87};

constexpr int kShortListGate = 256;  // the short-list gate (engine_request.cpp): lists of at most this many entries in front of tier 1 ...
constexpr int kShortGateMinSv = 2048; // ... of a model with at least this many support vectors go straight to the fp64 MFMA tier
constexpr int kStrictSlots = 64;     // evaluations per pass of the strict tier's spread form (a few per request reach it at most)

constexpr size_t kCntBytes = (CNT_COUNT * sizeof(int) + 15) / 16 * 16;      // the counters' share of the output block (d_out)

// contraction mode: default = screening pass + three-pass refinement; HAF_FLAG_SPLIT_F16 = three passes for everything;
// HAF_FLAG_FP32_MFMA = one fp32 MFMA pass for everything
// does the low-rank form serve this engine's full-size requests?  (engine_request.cpp applies it per request: whole requests of at
// least large_evals evaluations on grids that go through the parallel integral image)
inline bool lr_typical(const haf_engine *e)
{
    const haf_config &c = e->cfg;
    return e->lr_available && e->lr_enabled && (long)c.grid_h * c.grid_w > 8192 &&
           (long)(c.grid_h - 14) * (c.grid_w - 14) * e->max_rolls >= e->large_evals;
}
inline double variant_cost(const haf_engine *e, int v)
{
    if (!lr_typical(e) || (v == SCREEN_PLAIN && !e->lr_plain_available)) return kVariantCost[v];
    return kVariantCostLr[v];
}
enum { MODE_SCREEN = 0, MODE_SPLIT = 1, MODE_F32 = 2 };
inline int contraction_mode(const haf_config &c)
{
    if (c.flags & HAF_FLAG_FP32_MFMA) return MODE_F32;
    if (c.flags & HAF_FLAG_SPLIT_F16) return MODE_SPLIT;
    return MODE_SCREEN;
}

inline int fail(haf_engine *e, int code, const std::string &msg)
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

inline void mark(haf_engine *e, int idx)
{
    if (e->cfg.flags & HAF_FLAG_PROFILE) (void)hipEventRecord(e->ev[idx], e->stream);
}

// engine_tables.cpp
int label_grid_value(int label);
double sigma_upper_bound(const double *M, int n, int d);
int build_tables(haf_engine *e);
int alloc_buffers(haf_engine *e);
// engine_request.cpp
int score_rolls_impl(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, int32_t roll_first,
                     int32_t roll_count, haf_roll_record *records);
int score_batch_impl(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, haf_grasp_output *out);
// engine_geometry.cpp
int finalize_impl(const haf_config &c, const haf_grasp_input *in, const haf_roll_record *rec, haf_grasp_output *out, std::string &error);
int roll_pose_impl(const haf_config &c, const haf_grasp_input *in, const haf_roll_record *rec, int roll, haf_grasp_output *out,
                   int32_t *published, std::string &error);

}  // namespace haf_host
