// engine_request.cpp -- one request through the engine: stage launches on the engine's stream, the decision tiers (screening pass,
// second screening pass, three-pass tier, exact-integer tier, fp64 MFMA tier, strict order), the host resolution of the residual
// cases with glibc's exp, the adaptive choice of the screening form, and the batch wrapper (loop_control, server.cpp:335-402).
#include "engine_state.h"

namespace haf_host {

// One kernel value K(x, s_n) on the HOST exactly as svm-predict forms it (Kernel::k_function, svm.cpp:318-371) with the C library's
// exp / tanh: x = the device's attributes (stride 16 doubles: one column of the fp64 image), s_n in model order.  -ffp-contract=off.
static double host_kernel_value(const SvmModel &m, const double *xg, int n, int kx)
{
    double acc = 0.0;
    if (m.kernel_type == HAF_KERNEL_RBF) {
        for (int k = 0; k < kx; k++) {                           // svm.cpp:333-347: index order, a missing entry is 0
            const double sv = k < m.dim ? m.sv[(size_t)n * m.dim + k] : 0.0;
            const double dd = xg[(size_t)k * 16] - sv;
            acc += dd * dd;
        }
        return std::exp(-m.gamma * acc);                         // svm.cpp:364: glibc's exp
    }
    for (int k = 0; k < kx; k++) {                               // Kernel::dot, svm.cpp:299-316 (a zero on either side adds +-0)
        const double sv = k < m.dim ? m.sv[(size_t)n * m.dim + k] : 0.0;
        acc += xg[(size_t)k * 16] * sv;
    }
    if (m.kernel_type == HAF_KERNEL_POLY) {                      // powi, svm.cpp:26-36
        double tmp = m.gamma * acc + m.coef0, ret = 1.0;
        for (int t = m.degree; t > 0; t /= 2) { if (t % 2 == 1) ret *= tmp; tmp = tmp * tmp; }
        return ret;
    }
    if (m.kernel_type == HAF_KERNEL_SIGMOID) return std::tanh(m.gamma * acc + m.coef0);   // svm.cpp:367: glibc's tanh
    return acc;                                                  // LINEAR
}

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
            const double kv = host_kernel_value(m, xg, n, kx);
            sum += m.coef[(size_t)n] * kv;
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
            for (int nn = 0; nn < m.n_sv; nn++) sum += m.coef[(size_t)nn] * host_kernel_value(m, xg, nn, kx);   // svm.cpp:2509-2512
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

int score_rolls_impl(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, int32_t roll_first,
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
    const bool direct = !e->prob_mode && !e->generic_kernel && e->direct_work > 0 && evals_sel * (long)e->n_sv_pad <= e->direct_work;
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
                                     e->d_labels.p, e->d_evalcell.p, e->d_counters.p, e->d_flag_list.p, direct, d, r_row, r_col, s,
                                     e->d_brslot.p, ++e->pre_epoch);
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
        launch_integral(e->d_heights.p, e->d_rowsum.p, e->d_ii.p, e->d_inexact.p, e->d_counters.p, d, s, e->lr_available ? e->d_iiabs.p : nullptr);
        mark(e, HAF_ST_MASK);
        launch_mask_count(e->d_ii.p, d_geo, e->d_mask.p, e->d_rowcount.p, d, s);
        launch_scan(e->d_rowcount.p, e->d_rowoff.p, e->d_brcount.p, e->d_counters.p, d, s);
        launch_compact(e->d_mask.p, e->d_rowcount.p, e->d_rowoff.p, e->d_evalcell.p, d, s);
        if (direct) launch_prob_list(e->d_counters.p, CNT_FLAGGED, e->d_flag_list.p, e->list_cap, s);
    }
    // features -> decision tiers -> vote -> records on the host, for one contraction mode
    bool i8_used = false;                                    // the exact-integer tier ran in the last decide()
    bool t0b_used = false;                                   // tier 0b ran in the last decide()
    bool lr_used = false;                                    // the screening pass of the last decide() ran in the low-rank form
    auto decide = [&](int mode, bool reuse_operands) -> int {
        t0b_used = false;
        mark(e, HAF_ST_FEATURES);
        const bool large = evals_sel >= e->large_evals;      // enough evaluations to fill the chip with one thread each
        if (e->generic_kernel) {
            // a model whose kernel is not RBF: every evaluation on the strict tier's list, libsvm's own arithmetic for all of them
            launch_prob_list(e->d_counters.p, CNT_FLAGGED2, e->d_flag2_list.p, e->list_cap, s);
            launch_recheck(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv64.p, e->d_coef64.p, e->exact, e->d_flag2_list.p, e->list_cap,
                           e->d_counters.p, CNT_FLAGGED2, e->d_dec_exact2.p, e->d_labels.p, d, s);
            mark(e, HAF_ST_SVM);
            mark(e, HAF_ST_REFINE);
        } else if (direct) {
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
            // Low-rank form of the centred-remainder pass (kernels.h: kLrK): whole requests large enough for the thread-per-evaluation
            // feature kernel on grids that went through the parallel integral image (whose pass records negative heights), when the
            // centred-remainder form serves the model; the 10-step images go through k_project, the sweep runs on 6-step images.
            const bool lr_plain = e->screen_variant == SCREEN_PLAIN && e->lr_plain_available;      // the plain epilogue on the projected (centred) operands
            const bool lr = (cr || lr_plain) && large && e->lr_available && e->lr_enabled && !fused_pre && (long)H * W > 8192 && !reuse_operands;
            lr_used = lr;
            if (lr && lr_plain) sp_now = e->screen_lrp;
            if (lr) { sp_now.lr = 1; sp_now.lr_negflags = e->d_inexact.p; sp_now.lr_iiabs = e->d_iiabs.p; }
            if (!reuse_operands)
                launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X.p, e->d_gband.p, d, e->range.lower,
                                e->range.upper, e->svm.neg_gamma2, evals_cap, XMODE_SCREEN, sp_now, nullptr, 0, 0, large, evals_sel, nullptr, e->d_ax.p, s);
            char *y_img = reinterpret_cast<char *>(e->d_X.p) + (size_t)(e->max_evals_pad / kTile) * (size_t)kS0MatBytes;   // behind the 10-step images (the buffer holds the 42 KiB three-pass form)
            mark(e, HAF_ST_SVM);
            if (lr && !e->lr_fused) launch_project(e->d_X.p, e->d_lr_btiles.p, y_img, e->d_gband.p, e->d_counters.p, evals_cap, s);   // (counted with the sweep it feeds)
            // A small request with a small model (small_exact): what the screening pass leaves goes STRAIGHT to the one-launch exact kernel
            // (k_small_direct in list mode: exact attributes + fp64 MFMA decision, 9 ns per listed evaluation at 192 SVs) -- the list
            // is written where that kernel reads it.  Tier 1 in between was a feature kernel and a contraction launch at their latency
            // floors (C3: 44 + 36 us for 4 072 evaluations, of which it decided nine tenths) in front of the same exact kernel.
            // The same hand-over when calibration found tier 1 of little use behind the screening passes (t1_skip).
            // (behind the low-rank pass the full-rank centred-remainder form runs once more on its list: the waves it cannot bound --
            // the borders of the grid, waves that are not a run of neighbours -- and what its slightly wider band leaves)
            const bool t0b = e->use_t0b && e->cr_available && !small_exact && (!cr || (lr && e->screen_variant == SCREEN_CR_EXP));
            const bool straight = small_exact || (e->t1_skip && !t0b);
            if (lr)
                launch_svm_screen_lr(e->lr_fused ? reinterpret_cast<char *>(e->d_X.p) : y_img, e->d_gband.p, e->d_ax.p, e->d_svt_lr.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                                     e->d_flag0_words.p, e->d_flag0_wgcount.p, straight ? e->d_flag_list.p : e->d_flag0_list.p, e->flag0_cap, e->d_counters.p, d,
                                     evals_cap, e->d_margin.p, e->screen_variant, e->crp, e->lr_band, s, straight ? CNT_FLAGGED : -1,
                                     e->lr_fused ? e->d_lr_btiles_in.p : nullptr);
            else
            launch_svm_screen(e->d_X.p, e->d_gband.p, e->d_ax.p, cr ? e->d_svt0_cr.p : e->d_svt0.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                              e->d_flag0_words.p, e->d_flag0_wgcount.p, straight ? e->d_flag_list.p : e->d_flag0_list.p, e->flag0_cap, e->d_counters.p, d, evals_cap, e->d_margin.p,
                              e->screen_variant, e->crp, s, straight ? CNT_FLAGGED : -1, nullptr, CNT_EVALS, CNT_FLAGGED0, e->d_screen_part.p, e->screen_parts);
            mark(e, HAF_ST_REFINE);
            const long list_cap = std::min<long>(e->flag0_cap, evals_cap);
            const int *t1_list = e->d_flag0_list.p;
            int t1_counter = CNT_FLAGGED0;
            bool t1_run = !straight;
            const bool skip1 = e->t1_skip;
            // round 5: behind a LOW-RANK first pass with the plain epilogue tier 0b is the low-rank sweep itself with the centred-remainder
            // epilogue in its GATHER form -- the first pass's operand images and raw sums by evaluation id, no second feature kernel
            const bool t0b_gather = t0b && lr && lr_plain && e->lr_fused && !test_env("HAF_T0B_NO_GATHER");
            if (t0b_gather) {
                launch_svm_screen_lr(reinterpret_cast<char *>(e->d_X.p), e->d_gband.p, e->d_ax.p, e->d_svt_lr.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p,
                                     e->d_labels.p, e->d_flag0_words.p, e->d_flag0_wgcount.p, skip1 ? e->d_flag_list.p : e->d_flag0b_list.p, e->flag0_cap,
                                     e->d_counters.p, d, list_cap, e->d_margin.p, SCREEN_CR_EXP, e->crp, e->lr_band, s, skip1 ? CNT_FLAGGED : -1,
                                     e->d_lr_btiles_in.p, e->d_flag0_list.p, CNT_FLAGGED0, CNT_FLAGGED0B);
                t1_list = e->d_flag0b_list.p;
                t1_counter = CNT_FLAGGED0B;
                t1_run = !skip1;
                t0b_used = true;
            } else if (t0b) {
                // tier 0b: the centred-remainder form on the LIST of the first pass (its own operand images: translated attributes,
                // centred support vectors; band, common factor and images indexed by list slot)
                ScreenParams sp_b = e->screen_cr;
                sp_b.cr_poly = 0;
                // (the group-parallel feature kernel even for lists of 10^5: one evaluation per lane on SCATTERED cells -- k_features_serial in
                // list mode -- was measured at 7.7 ns per evaluation against 2.1 ns, its 2 400 corner loads per lane being 64 separate
                // L1 accesses each)
                launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X1.p, e->d_gband.p, d, e->range.lower,
                                e->range.upper, e->svm.neg_gamma2, list_cap, XMODE_SCREEN, sp_b, e->d_flag0_list.p, CNT_FLAGGED0, e->flag0_cap,
                                false, list_cap, nullptr, e->d_ax.p, s);
                launch_svm_screen(e->d_X1.p, e->d_gband.p, e->d_ax.p, e->d_svt0_cr.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                                  e->d_flag0_words.p, e->d_flag0_wgcount.p, skip1 ? e->d_flag_list.p : e->d_flag0b_list.p, e->flag0_cap, e->d_counters.p, d,
                                  list_cap, e->d_margin.p, SCREEN_CR_EXP, e->crp, s, skip1 ? CNT_FLAGGED : -1, e->d_flag0_list.p, CNT_FLAGGED0, CNT_FLAGGED0B,
                                  e->d_screen_part.p, e->screen_parts);
                t1_list = e->d_flag0b_list.p;
                t1_counter = CNT_FLAGGED0B;
                t1_run = !skip1;
                t0b_used = true;
            }
            // Round 5, the short-list gate: a SMALL request against a BIG model leaves the screening passes a handful of evaluations (C3
            // against the headline's 4 096-SV model: 31), and tier 1 and the exact-integer tier behind it are then ~120 us of launches and
            // minimum chains -- feature kernels, sweeps over thousands of SVs by a few workgroups -- for nothing the fp64 tier would
            // not do in the one pass it makes anyway.  At most kShortListGate entries go straight to its list; tier 1 reads its list's
            // length from CNT_T1_N (0 then), the exact-integer tier finds an empty input.  Not while the engine calibrates (the
            // calibration measures what tier 1 decides).
            if (t1_run && e->calibrated && e->short_gate && d.n_sv >= kShortGateMinSv && evals_cap <= 65536 && !e->generic_kernel) {
                const bool i8_next = e->i8_active && !(e->screen_variant == SCREEN_CR_POLY && e->t1_cr_available);
                launch_short_list_gate(e->d_counters.p, t1_counter, t1_list, e->flag0_cap, i8_next ? e->d_flagi_list.p : e->d_flag_list.p,
                                       CNT_FLAGGEDI, i8_next, std::min(kShortListGate, e->flag_cap), s);
                t1_counter = CNT_T1_N;
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
                         e->d_part1.p, e->part1_stride, s, t1cr ? &e->crt1 : nullptr, t1cr ? e->d_t1_L.p : nullptr, e->d_t1_flags.p);
            }
        } else if (mode == MODE_SPLIT) {
            launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X.p, e->d_ax.p, d, e->range.lower,
                            e->range.upper, e->svm.neg_gamma2, evals_cap, XMODE_SPLIT, e->screen, nullptr, 0, 0, large, evals_sel, e->d_attr.p, nullptr, s);
            mark(e, HAF_ST_SVM);
            launch_svm_h(e->d_X.p, e->d_ax.p, e->d_svt_h.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                         e->d_flag_list.p, e->list_cap, e->d_counters.p, d, evals_cap, nullptr, 0, 0, nullptr, 0, s, nullptr, nullptr, e->d_t1_flags.p);
            mark(e, HAF_ST_REFINE);
        } else {
            launch_features(e->d_ii.p, e->d_evalcell.p, e->d_counters.p, e->d_fd.p, e->d_X.p, e->d_ax.p, d, e->range.lower,
                            e->range.upper, e->svm.neg_gamma2, evals_cap, XMODE_F32, e->screen, nullptr, 0, 0, large, evals_sel, e->d_attr.p, nullptr, s);
            mark(e, HAF_ST_SVM);
            launch_svm(e->d_X.p, e->d_ax.p, e->d_svt.p, e->d_evalcell.p, e->d_counters.p, e->svm, e->d_dec.p, e->d_labels.p,
                       e->d_flag_list.p, e->list_cap, e->d_counters.p, d, evals_cap, s, e->d_t1_flags.p);
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
                                    CNT_FLAGGEDI, e->d_tier_words.p);
            else
                launch_recheck_mfma(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv64.p, e->exact, e->d_flag_list.p, e->flag_cap, off, e->d_counters.p,
                                    e->d_x64.p, e->d_part64.p, e->d_dec_exact.p, e->d_labels.p, e->d_flag2_list.p, e->list_cap, d, s, nullptr, false,
                                    CNT_FLAGGED, e->d_tier_words.p);
        };
        auto i8_window = [&](int off) {
            launch_recheck_i8(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv_i8.p, e->i8, e->range.lower, e->range.upper, e->d_flag_list.p, e->flag_cap,
                              off, e->d_counters.p, e->d_x64.p, e->d_part64.p, e->d_dec_exact.p, e->d_labels.p, e->d_flagi_list.p, e->list_cap, d, s,
                              e->d_tier_words.p);
        };
        if (!direct && !e->generic_kernel) {
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
                for (int off = e->flag_cap; off < flagged; off += e->flag_cap) { i8_window(off); e->stat_extra_windows++; }
                // the fp64 tier's list has grown behind its first window: all of it again from the start (its results and the
                // strict tier's list are rebuilt; both are idempotent)
                HIPCHK(e, hipMemsetAsync(e->d_counters.p + CNT_FLAGGED2, 0, sizeof(int), s));
                HIPCHK(e, hipMemcpyAsync(e->h_out, e->d_out.p, kCntBytes, hipMemcpyDeviceToHost, s));
                HIPCHK(e, hipStreamSynchronize(s));
                done_fp64 = 0;
            }
            const int n_fp64 = i8 ? e->h_counters[CNT_FLAGGEDI] : flagged;
            for (int off = done_fp64; off < n_fp64; off += e->flag_cap) { fp64_window(off); e->stat_extra_windows++; }
            launch_recheck(e->d_ii.p, e->d_evalcell.p, e->d_fd.p, e->d_sv64.p, e->d_coef64.p, e->exact, e->d_flag2_list.p, e->list_cap,
                           e->d_counters.p, CNT_FLAGGED2, e->d_dec_exact2.p, e->d_labels.p, d, s);
            rc = vote();
            if (rc != HAF_OK) return rc;
            strict_ran = e->h_counters[CNT_FLAGGED2] > 0;
        } else if (e->generic_kernel) {
            strict_ran = true;                            // (it has run, on everything; what is left is the C library's tanh near zero)
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
    bool reprobe = false;
    if (mode == MODE_SCREEN && !e->screen_active) {
        // (the switch-off used to be for the engine's lifetime) every reprobe_every-th request that is large enough to judge by
        // runs the screening pass again; it costs that request one wasted pass at worst
        if (!e->variant_forced && !e->prob_mode && !direct && evals_sel >= 4096 && ++e->inactive_calls >= e->reprobe_every) {
            e->inactive_calls = 0;
            reprobe = true;
        } else {
            mode = MODE_SPLIT;
        }
    }
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
        if (undecided() > e->flag0_cap) e->stat_flag0_overflows++;
        while (undecided() > e->flag0_cap && !e->variant_forced && next_variant(e->screen_variant) >= 0) {
            const bool reuse = e->screen_variant == SCREEN_PLAIN && !t0b_used && !lr_used;     // (tier 0b writes its bands where the first pass's were; the low-rank form's images are the centred ones)
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
        } else if (reprobe) {
            const double share = ne > 0 ? (double)undecided() / (double)ne : 1.0;
            e->variant_share[e->screen_variant] = share;
            if (ne >= 256 && share <= 0.6) e->screen_active = true;
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
                        const double cost = variant_cost(e, v) + kUndecidedCost * e->variant_share[v];
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
    e->last_lr = lr_used;
    e->last_flagged0 = t0b_used ? std::min(e->h_counters[CNT_FLAGGED0B], e->h_counters[CNT_FLAGGED0]) : e->h_counters[CNT_FLAGGED0];   // what leaves the screening passes
    e->last_flaggedi = i8_used ? e->h_counters[CNT_FLAGGEDI] : e->h_counters[CNT_FLAGGED];
    e->last_bypass = e->h_counters[CNT_BYPASS];                            // (the short-list gate: entries that went around tier 1 and the exact-integer tier)
    e->last_inexact = inexact_grids;
    e->last_screened = (mode == MODE_SCREEN) && !e->prob_mode && !direct && e->h_counters[CNT_FLAGGED0] <= e->flag0_cap;
    // zero the counters for the next request now, behind this one's copy-out: off that request's critical path
    if (hipMemsetAsync(e->d_counters.p, 0, CNT_COUNT * sizeof(int), s) == hipSuccess) e->counters_clean = true;
    e->last_inputs.assign(in, in + B);
    // (the tier lists hold every evaluation of a request: list_cap >= last_evals >= last_flagged >= last_flagged2)
    if (e->last_flagged > e->list_cap || e->last_flagged2 > e->list_cap || e->last_flaggedi > e->list_cap) return fail(e, HAF_E_INTERNAL, "recheck list counters exceed the number of evaluations");
    e->last_i8 = i8_used;
#ifdef HAF_TESTING
    // testing build, HAF_CANARY_CHECK set (tests/conftest.py): the guard zones around every device buffer after EVERY request
    if (test_env("HAF_CANARY_CHECK")) {
        std::string rep;
        const int bad = canary_check(&rep);
        if (bad != 0) return fail(e, HAF_E_INTERNAL, "device buffer guard zones damaged (" + std::to_string(bad) + "): " + rep);
    }
#endif
    for (int i = 0; i < B * R; i++) {
        records[i].vote = e->h_rec[i].vote;
        records[i].row = e->h_rec[i].row;
        records[i].col = e->h_rec[i].col;
        records[i].h_locmax = e->h_rec[i].h_locmax;
        records[i].n_evals = e->h_rec[i].n_evals;
    }
    return HAF_OK;
}

int score_batch_impl(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in, haf_grasp_output *out)
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
        e->last_bypass = 0;
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
    out[0].n_rechecked = e->last_flagged + (e->last_i8 ? e->last_bypass : 0);
    return HAF_OK;
}

}  // namespace haf_host
