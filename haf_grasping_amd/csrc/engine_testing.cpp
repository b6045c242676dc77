// engine_testing.cpp -- the haf_test_* hooks of libhafgrasp_testing.so (tests/ only; the product library does not contain this file).
#include "engine_state.h"

#ifndef HAF_TESTING
#error "engine_testing.cpp belongs to the testing build (-DHAF_TESTING)"
#endif

// ---- guard zones around every device buffer (engine_state.h: DevBuf) ----
namespace haf_host {
namespace {
struct CanaryRec { char *user; size_t bytes; const char *file; int line; };
std::mutex g_canary_mu;
std::vector<CanaryRec> g_canary;
}
void canary_register(void *user, size_t bytes, const char *file, int line)
{
    std::lock_guard<std::mutex> lk(g_canary_mu);
    g_canary.push_back(CanaryRec{static_cast<char *>(user), bytes, file, line});
}
void canary_unregister(void *user)
{
    std::lock_guard<std::mutex> lk(g_canary_mu);
    for (size_t i = 0; i < g_canary.size(); i++)
        if (g_canary[i].user == user) { g_canary[i] = g_canary.back(); g_canary.pop_back(); return; }
}
// Copies both zones of every registered buffer back and compares them with the pattern; the report names each damaged buffer by the
// source line that allocated it, the side, the first damaged byte's offset from the buffer's end (or start) and the 4 bytes found there.
// Synchronises the device first: every kernel of the requests so far has finished.
int canary_check(std::string *report)
{
    std::lock_guard<std::mutex> lk(g_canary_mu);
    if (hipDeviceSynchronize() != hipSuccess) { if (report) *report += "hipDeviceSynchronize failed; "; return -1; }
    int bad = 0;
    std::vector<unsigned char> h;
    for (const CanaryRec &r : g_canary) {
        const size_t padded = (r.bytes + kCanaryGuard - 1) / kCanaryGuard * kCanaryGuard, back = padded - r.bytes + kCanaryGuard;
        bool hit = false;
        for (int side = 0; side < 2; side++) {
            const size_t len = side ? back : kCanaryGuard;
            const char *src = side ? r.user + r.bytes : r.user - kCanaryGuard;
            h.resize(len);
            if (hipMemcpy(h.data(), src, len, hipMemcpyDeviceToHost) != hipSuccess) { if (report) *report += "hipMemcpy of a guard zone failed; "; return -1; }
            for (size_t i = 0; i < len; i++)
                if (h[i] != (unsigned char)kCanaryByte) {
                    hit = true;
                    if (report) {
                        char buf[256];
                        const size_t w = i / 4 * 4;
                        unsigned word = 0;
                        memcpy(&word, h.data() + w, std::min<size_t>(4, len - w));
                        snprintf(buf, sizeof buf, "%s:%d (%zu bytes): %s guard damaged at %s%zu, word there 0x%08x; ", r.file, r.line, r.bytes,
                                 side ? "back" : "front", side ? "end+" : "start-", side ? i : kCanaryGuard - i, word);
                        *report += buf;
                    }
                    break;
                }
        }
        bad += hit ? 1 : 0;
    }
    return bad;
}
}  // namespace haf_host

extern "C" {

// number of device buffers whose guard zones were written (0 = intact, < 0: HIP failure); msg (cap bytes) names them.  Covers every
// engine of the process.
int haf_test_check_canaries(char *msg, int cap)
{
    std::string rep;
    const int bad = canary_check(&rep);
    if (msg && cap > 0) { strncpy(msg, rep.c_str(), (size_t)cap - 1); msg[cap - 1] = 0; }
    return bad;
}
// how many buffers are registered (the test checks that the hook sees the engine's buffers at all)
int haf_test_canary_buffers()
{
    std::lock_guard<std::mutex> lk(g_canary_mu);
    return (int)g_canary.size();
}
// the device lists of the last request as they lie in memory: 0 the evaluation list (cell ids), 1 the exact tiers' input list, 2 the
// exact-integer tier's hand-over list, 3 the strict tier's list, 4 the screening pass's list.  *n = entries the list holds.
int haf_test_fetch_list(haf_engine *e, int which, int *out, int cap, int *n)
{
    if (!e || !out || !n || cap < 0) return HAF_E_ARG;
    const int *src = nullptr;
    int cnt = 0;
    switch (which) {
        case 0: src = e->d_evalcell.p; cnt = e->last_evals; break;
        case 1: src = e->d_flag_list.p; cnt = std::min(e->last_flagged, e->list_cap); break;
        case 2: src = e->d_flagi_list.p; cnt = e->last_i8 ? std::min(e->last_flaggedi, e->list_cap) : 0; break;
        case 3: src = e->d_flag2_list.p; cnt = std::min(e->last_flagged2, e->list_cap); break;
        case 4: src = e->d_flag0_list.p; cnt = std::min(e->last_flagged0, e->flag0_cap); break;
        default: return HAF_E_ARG;
    }
    *n = cnt;
    if (!src || cnt <= 0) { *n = 0; return HAF_OK; }
    if (hipDeviceSynchronize() != hipSuccess) return HAF_E_DEVICE;
    return hipMemcpy(out, src, (size_t)std::min(cnt, cap) * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess ? HAF_OK : HAF_E_DEVICE;
}
int haf_test_overflow_stats(haf_engine *e, long long *out2)
{
    if (!e || !out2) return HAF_E_ARG;
    out2[0] = e->stat_flag0_overflows; out2[1] = e->stat_extra_windows;
    return HAF_OK;
}
// writes `count` ints of value `v` at int offset `at` relative to the END of the d_flag0_list buffer (at >= 0) of this engine -- the
// canary test's own "bug": what a producer that ignores its capacity does
int haf_test_poke_flag0_list(haf_engine *e, int at, int count, int v)
{
    if (!e || !e->d_flag0_list.p) return HAF_E_ARG;
    std::vector<int> h((size_t)count, v);
    return hipMemcpy(e->d_flag0_list.p + e->d_flag0_list.n + at, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess ? HAF_OK : HAF_E_DEVICE;
}

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

int haf_test_model_kernel(const char *path, int *kernel_type, int *degree, double *coef0, double *gamma)
{
    SvmModel m;
    std::string err;
    if (!load_model(path, m, err)) return HAF_E_IO;
    *kernel_type = m.kernel_type; *degree = m.degree; *coef0 = m.coef0; *gamma = m.gamma;
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
    if (variant) *variant = e->screen_variant | (e->use_t0b ? 16 : 0) | (e->t1_skip ? 32 : 0) | (e->last_lr ? 64 : 0);
    if (active) *active = e->screen_active ? 1 : 0;
    if (shares) for (int i = 0; i < SCREEN_VARIANTS; i++) shares[i] = e->variant_share[i];
    return HAF_OK;
}

// switches the screening pass off as the adaptive rule would after a request that every form of the pass failed on (tests of the
// periodic re-try)
int haf_test_set_screen_inactive(haf_engine *e)
{
    if (!e) return HAF_E_ARG;
    e->screen_active = false;
    e->inactive_calls = 0;
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

}  // extern "C"
