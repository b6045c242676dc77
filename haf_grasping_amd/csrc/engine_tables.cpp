// engine_tables.cpp -- haf_create's table work: the parsed feature / range / model files -> the device tables of every kernel and the
// constants of every guard band (build_tables), and the engine's device buffers (alloc_buffers).
#include "engine_state.h"

namespace haf_host {

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
            double ea2 = 0.0, ea2_f32 = 0.0;
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
                // round 5: the scaling runs in fp32 -- fl32(q4), fl32(scr_mul), fl32(scr_add) and the fma's rounding: at most
                // 3 u (|q4 scr_mul| + |scr_add|) <= 3 u (|u'| + 2 |scr_add|) per attribute; the |u'| part is in kScreenEtaRel
                const double e32 = 3.6e-7 * std::fabs(d.scr_add);
                ea2_f32 += e32 * e32;
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
                    sdesc.scr_mul = (float)d.scr_mul; sdesc.scr_add = (float)d.scr_add;
                    sdesc.extra = (float)extra[(size_t)sl];
                    // low-rank form: |scr_mul| rounded up for a slot that is a linear functional of the window (HAF), 0 for a SHAF slot
                    const float mul_up = d.shaf ? 0.0f : std::nextafterf((float)std::fabs(d.scr_mul), INFINITY);
                    sdesc.pad = mul_up;
                    ScrDesc3 &s3 = sd3[(size_t)sl];
                    for (int k = 0; k < 3; k++) {
                        s3.w[k] = d.w[k];
                        for (int j = 0; j < 4; j++) s3.off[k * 4 + j] = ((d.off[k][j] / ld) * kBandPitch + d.off[k][j] % ld) * 4;
                    }
                    s3.shaf = d.shaf;
                    s3.scr_mul = (float)d.scr_mul; s3.scr_add = (float)d.scr_add;
                    s3.extra = (float)extra[(size_t)sl];
                    s3.pad[0] = mul_up;
                    fds[(size_t)sl] = d;
                    fds[(size_t)sl].scr_extra = (float)extra[(size_t)sl];
                    fds[(size_t)sl].pad2 = mul_up;
                    if (extra[(size_t)sl]) sp.extra_groups |= 1ull << g;
                }
                if (fast) sp.fast_groups |= 1ull << g;
            }
            if (test_env("HAF_NO_FAST_GROUPS")) sp.fast_groups = 0;        // A/B runs and the generic-path test
            // a degenerate target range or bounds beyond the decimal path's error budget: serve the model without screening
            if (!(e->range.upper > e->range.lower) || std::fabs(e->range.lower) > 1e3 || std::fabs(e->range.upper) > 1e3) e->screen_active = false;
            sp.eta_abs = std::max(std::sqrt(ea2 + ea2_f32) * 1.01, 1e-12 * 18.0 * sp.c);
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
            // ---- low-rank form (kernels.h: kLrK): an orthonormal basis B of the span of the HAF slots' linear functionals ----
            // Slot sl (HAF) is u'_sl = scr_mul_sl * feature_sl + scr_add_sl with feature_sl = sum_k w_k (A - B - C + D)_k a LINEAR functional of
            // the 225 window corners up to the "%.4g" round trip and fp32 roundings: the columns of Ms (slots x 225) span what the exactly
            // linear part of the operand can move in.  Modified Gram-Schmidt with pivoting and re-orthogonalisation in long double; the
            // SHAF slots (not linear: fv.cpp:187-191) get identity columns.  The centre mu is then moved INTO the affine subspace
            // {scr_add + range(Ms)} (any centre gives the same decision function), so that the centred linear part lies in range(B).
            const int KL = kLrK;
            std::vector<double> Bm((size_t)kS0K * KL, 0.0);                  // B, row-major [slot][k]
            int lr_cols = 0, lr_rank = 0;
            double lr_rho = 0.0;
            bool lr_basis = false;
            // (only for engines whose grid can carry a request the form applies to: engine_request.cpp; a 56 x 56 engine skips the seconds of table work)
            if (!test_env("HAF_NO_LR") && !test_env("HAF_CR_NO_CENTRE") && (long)e->cfg.grid_h * e->cfg.grid_w > 8192) {
                std::vector<int> lin((size_t)kS0K, 0), pass((size_t)kS0K, 0);
                std::vector<long double> Ms((size_t)225 * kS0K, 0.0L);         // column j of the map = Ms[j * kS0K ...]
                static const long double sgn[4] = {1.0L, -1.0L, -1.0L, 1.0L};
                for (int sl = 0; sl < S; sl++) {
                    const FeatDesc &d0 = fds_keep[(size_t)sl];
                    if (d0.skip || d0.scr_mul == 0.0) continue;
                    if (d0.shaf) { pass[(size_t)sl] = 1; continue; }
                    lin[(size_t)sl] = 1;
                    for (int k = 0; k < 3; k++) {
                        if (!(d0.active & (1 << k))) continue;
                        for (int j = 0; j < 4; j++) Ms[(size_t)d0.offw[k][j] * kS0K + sl] += sgn[j] * (long double)d0.w[k] * (long double)d0.scr_mul;
                    }
                }
                std::vector<long double> Rs = Ms;                             // residual columns
                std::vector<std::vector<long double>> basis;
                auto nrm2 = [&](const long double *v) { long double t = 0; for (int sl = 0; sl < kS0K; sl++) t += v[sl] * v[sl]; return t; };
                long double first = 0.0L;
                for (int it = 0; it < 225; it++) {
                    int best = -1; long double bn = 0.0L;
                    for (int j = 0; j < 225; j++) { const long double t = nrm2(Rs.data() + (size_t)j * kS0K); if (t > bn) { bn = t; best = j; } }
                    if (it == 0) first = bn;
                    if (best < 0 || !(bn > 1e-24L * first)) break;
                    std::vector<long double> v(Rs.begin() + (size_t)best * kS0K, Rs.begin() + (size_t)(best + 1) * kS0K);
                    for (int pass2 = 0; pass2 < 2; pass2++)                    // re-orthogonalise against the basis so far
                        for (const auto &bv : basis) { long double dp = 0; for (int sl = 0; sl < kS0K; sl++) dp += bv[(size_t)sl] * v[(size_t)sl]; for (int sl = 0; sl < kS0K; sl++) v[(size_t)sl] -= dp * bv[(size_t)sl]; }
                    const long double nv = std::sqrt(nrm2(v.data()));
                    if (!(nv > 0.0L)) break;
                    for (auto &x : v) x /= nv;
                    for (int j = 0; j < 225; j++) {
                        long double *rc = Rs.data() + (size_t)j * kS0K;
                        long double dp = 0; for (int sl = 0; sl < kS0K; sl++) dp += v[(size_t)sl] * rc[sl];
                        for (int sl = 0; sl < kS0K; sl++) rc[sl] -= dp * v[(size_t)sl];
                    }
                    basis.push_back(v);
                }
                lr_rank = (int)basis.size();
                int n_pass = 0;
                for (int sl = 0; sl < S; sl++) n_pass += pass[(size_t)sl];
                if (lr_rank > 0 && lr_rank + n_pass <= KL) {
                    for (int k = 0; k < lr_rank; k++) for (int sl = 0; sl < kS0K; sl++) Bm[(size_t)sl * KL + k] = (double)basis[(size_t)k][(size_t)sl];
                    int kp = lr_rank;
                    for (int sl = 0; sl < S; sl++) if (pass[(size_t)sl]) Bm[(size_t)sl * KL + kp++] = 1.0;
                    lr_cols = kp;
                    // what the ORIGINAL columns leave outside range(B) once B is rounded to fp64: sum of the residual norms (per unit of the
                    // largest corner: |(I - BB')Ms w| <= sum_j |res_j| |w_j|)
                    long double rs = 0.0L;
                    for (int j = 0; j < 225; j++) {
                        std::vector<long double> c0(Ms.begin() + (size_t)j * kS0K, Ms.begin() + (size_t)(j + 1) * kS0K);
                        for (int k = 0; k < lr_rank; k++) {
                            long double dp = 0; for (int sl = 0; sl < kS0K; sl++) dp += (long double)Bm[(size_t)sl * KL + k] * Ms[(size_t)j * kS0K + sl];
                            for (int sl = 0; sl < kS0K; sl++) c0[(size_t)sl] -= dp * (long double)Bm[(size_t)sl * KL + k];
                        }
                        rs += std::sqrt(nrm2(c0.data()));
                    }
                    lr_rho = (double)rs * 1.01 + 1e-300;
                    // centre into the affine subspace: (scr_add - mu) restricted to the linear slots must lie in range(B)
                    std::vector<long double> am((size_t)kS0K, 0.0L);
                    for (int sl = 0; sl < S; sl++) if (lin[(size_t)sl]) am[(size_t)sl] = (long double)fds_keep[(size_t)sl].scr_add - (long double)mu[(size_t)sl];
                    std::vector<long double> pr((size_t)kS0K, 0.0L);
                    for (int k = 0; k < lr_rank; k++) {
                        long double dp = 0; for (int sl = 0; sl < kS0K; sl++) dp += (long double)Bm[(size_t)sl * KL + k] * am[(size_t)sl];
                        for (int sl = 0; sl < kS0K; sl++) pr[(size_t)sl] += dp * (long double)Bm[(size_t)sl * KL + k];
                    }
                    for (int sl = 0; sl < S; sl++) if (lin[(size_t)sl]) mu[(size_t)sl] = (double)((long double)fds_keep[(size_t)sl].scr_add - pr[(size_t)sl]);
                    lr_basis = true;
                }
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
                sdc[(size_t)sl].scr_add = (float)add; sd3c[(size_t)sl].scr_add = (float)add; fdsc[(size_t)sl].scr_add = add;
                const double ef = 4.5e-16 * std::fabs(mu[(size_t)sl]);
                ea2c += ef * ef;
                const double e32 = 3.6e-7 * std::fabs(add) * (1.0 + extra[(size_t)sl]);       // (the fp32 scaling of THIS instance's constants; per slot, its attributes counted)
                ea2c += e32 * e32;
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
                // ---- low-rank form: projection tiles, images of q~_n = (B'B^)^-1 B'q_n, r_n = (I - BB')q_n - (B^ - B)q~_n, the band's constants ----
                if (lr_basis) {
                    const int K = kS0K, KL2 = kLrK, kc = lr_cols;
                    std::vector<double> Bh((size_t)K * KL2, 0.0);
                    for (size_t i = 0; i < Bh.size(); i++) {
                        _Float16 h = (_Float16)(float)Bm[i];
                        if (std::fabs((float)h) < kF16MinNormal) h = (_Float16)0.0f;
                        Bh[i] = (double)(float)h;
                    }
                    // G = B'B^ (kc x kc), inverted by Gauss-Jordan with partial pivoting in long double (G = I + O(2^-11))
                    std::vector<long double> Gm((size_t)kc * kc, 0.0L), Gi((size_t)kc * kc, 0.0L);
                    for (int a2 = 0; a2 < kc; a2++)
                        for (int b2 = 0; b2 < kc; b2++) {
                            long double t = 0.0L;
                            for (int sl = 0; sl < K; sl++) t += (long double)Bm[(size_t)sl * KL2 + a2] * (long double)Bh[(size_t)sl * KL2 + b2];
                            Gm[(size_t)a2 * kc + b2] = t;
                        }
                    for (int a2 = 0; a2 < kc; a2++) Gi[(size_t)a2 * kc + a2] = 1.0L;
                    bool inv_ok = true;
                    for (int col = 0; col < kc && inv_ok; col++) {
                        int piv = col;
                        for (int r2 = col + 1; r2 < kc; r2++) if (fabsl(Gm[(size_t)r2 * kc + col]) > fabsl(Gm[(size_t)piv * kc + col])) piv = r2;
                        if (!(fabsl(Gm[(size_t)piv * kc + col]) > 0.25L)) { inv_ok = false; break; }
                        if (piv != col) for (int c2 = 0; c2 < kc; c2++) { std::swap(Gm[(size_t)piv * kc + c2], Gm[(size_t)col * kc + c2]); std::swap(Gi[(size_t)piv * kc + c2], Gi[(size_t)col * kc + c2]); }
                        const long double iv = 1.0L / Gm[(size_t)col * kc + col];
                        for (int c2 = 0; c2 < kc; c2++) { Gm[(size_t)col * kc + c2] *= iv; Gi[(size_t)col * kc + c2] *= iv; }
                        for (int r2 = 0; r2 < kc; r2++) {
                            if (r2 == col) continue;
                            const long double f = Gm[(size_t)r2 * kc + col];
                            if (f == 0.0L) continue;
                            for (int c2 = 0; c2 < kc; c2++) { Gm[(size_t)r2 * kc + c2] -= f * Gm[(size_t)col * kc + c2]; Gi[(size_t)r2 * kc + c2] -= f * Gi[(size_t)col * kc + c2]; }
                        }
                    }
                    LrBand lb{};
                    std::vector<double> lr_corr_g, lr_corr_h;            // plain epilogue: the feature kernel's two correction vectors B^ Q~^'b, B^ dQ~'b
                    std::vector<double> Qt((size_t)m.n_sv * KL2, 0.0), Qth((size_t)m.n_sv * KL2, 0.0), dQt((size_t)m.n_sv * KL2, 0.0), Rm((size_t)m.n_sv * K, 0.0);
                    std::vector<char> imgl((size_t)e->n_sv_tiles * kLrSvTileBytes, 0);
                    if (inv_ok) {
                        // (fp64 from here: q~_n and r_n are DEFINED by these computed values -- z_n = y_e.q~_n + p_perp.r_n holds for any q~_n once
                        // r_n = q_n - B q~'_n ... is formed from the same numbers; what fp64 rounding leaves in r_n is 1e-16 |q_n|, far inside rmax's margin)
                        std::vector<double> Gid((size_t)kc * kc), bq((size_t)kc), qt((size_t)kc), Bt((size_t)KL2 * K), Bd((size_t)KL2 * K);
                        for (size_t i = 0; i < Gid.size(); i++) Gid[i] = (double)Gi[i];
                        for (int sl = 0; sl < K; sl++) for (int a2 = 0; a2 < KL2; a2++) { Bt[(size_t)a2 * K + sl] = Bm[(size_t)sl * KL2 + a2]; Bd[(size_t)a2 * K + sl] = Bh[(size_t)sl * KL2 + a2] - Bm[(size_t)sl * KL2 + a2]; }
                        std::vector<double> rrow((size_t)K);
                        for (int n = 0; n < m.n_sv; n++) {
                            const double *q = Q.data() + (size_t)n * K;
                            for (int a2 = 0; a2 < kc; a2++) { const double *bt = Bt.data() + (size_t)a2 * K; double t = 0.0; for (int sl = 0; sl < K; sl++) t += bt[sl] * q[sl]; bq[(size_t)a2] = t; }
                            for (int a2 = 0; a2 < kc; a2++) { const double *gr = Gid.data() + (size_t)a2 * kc; double t = 0.0; for (int b2 = 0; b2 < kc; b2++) t += gr[b2] * bq[(size_t)b2]; qt[(size_t)a2] = t; }
                            const int t = slot_of[(size_t)n] / kTile, j = slot_of[(size_t)n] % kTile;
                            char *tile = imgl.data() + (size_t)t * kLrSvTileBytes;
                            double h2 = 0.0, d2 = 0.0, q2 = 0.0, r2n = 0.0;
                            for (int a2 = 0; a2 < kc; a2++) {
                                const double v = (double)qt[(size_t)a2];
                                _Float16 h = (_Float16)(float)v;
                                if (std::fabs((float)h) < kF16MinNormal) h = (_Float16)0.0f;
                                memcpy(tile + h_image_offset(j, a2), &h, 2);
                                const double hd = (double)(float)h;
                                Qt[(size_t)n * KL2 + a2] = v; Qth[(size_t)n * KL2 + a2] = hd; dQt[(size_t)n * KL2 + a2] = hd - v;
                                h2 += hd * hd; d2 += (hd - v) * (hd - v);
                            }
                            // r_n = q_n - B (B'q_n) - (B^ - B) q~_n
                            for (int sl = 0; sl < K; sl++) rrow[(size_t)sl] = q[sl];
                            for (int a2 = 0; a2 < kc; a2++) {
                                const double c1 = bq[(size_t)a2], c2 = qt[(size_t)a2];
                                const double *bt = Bt.data() + (size_t)a2 * K, *bd = Bd.data() + (size_t)a2 * K;
                                for (int sl = 0; sl < K; sl++) rrow[(size_t)sl] -= bt[sl] * c1 + bd[sl] * c2;
                            }
                            for (int sl = 0; sl < K; sl++) {
                                const double t = rrow[(size_t)sl];
                                Rm[(size_t)n * K + sl] = t;
                                r2n += t * t;
                                q2 += q[sl] * q[sl];
                            }
                            reinterpret_cast<float *>(tile + kLrMatBytes)[j] = 0.0f;
                            reinterpret_cast<float *>(tile + kLrMatBytes)[kTile + j] = (float)b[(size_t)n];
                            const double ab = std::fabs(b[(size_t)n]), qhn = std::sqrt(h2), qn = std::sqrt(q2);
                            lb.Ca += ab * qhn * qn; lb.Cq1 += ab * qhn; lb.Cqq += ab * h2; lb.Babs += ab;
                            lb.qmax = std::max(lb.qmax, qhn); lb.dqmax = std::max(lb.dqmax, std::sqrt(d2)); lb.rmax = std::max(lb.rmax, std::sqrt(r2n));
                        }
                        // signed first-order matrices and the unsigned second-order ones
                        std::vector<double> N1((size_t)K * KL2, 0.0), M1((size_t)K * KL2, 0.0), N2((size_t)K * K, 0.0);
                        std::vector<double> Rh((size_t)m.n_sv * KL2), Rd((size_t)m.n_sv * KL2), Ra((size_t)m.n_sv * KL2), Rq((size_t)m.n_sv * K), Rr((size_t)m.n_sv * K);
                        for (int n = 0; n < m.n_sv; n++) {
                            const double *q = Q.data() + (size_t)n * K, *h = Qth.data() + (size_t)n * KL2, *dq = dQt.data() + (size_t)n * KL2, *rr = Rm.data() + (size_t)n * K;
                            const double sb = std::sqrt(std::fabs(b[(size_t)n]));
                            for (int l = 0; l < KL2; l++) { Rh[(size_t)n * KL2 + l] = sb * h[l]; Rd[(size_t)n * KL2 + l] = sb * dq[l]; Ra[(size_t)n * KL2 + l] = sb * std::fabs(h[l]); }
                            for (int l = 0; l < K; l++) { Rq[(size_t)n * K + l] = sb * q[l]; Rr[(size_t)n * K + l] = sb * rr[l]; }
                            for (int k = 0; k < K; k++) {
                                const double a = b[(size_t)n] * q[k];
                                if (a == 0.0) continue;
                                double *n1 = N1.data() + (size_t)k * KL2, *m1 = M1.data() + (size_t)k * KL2, *n2 = N2.data() + (size_t)k * K;
                                for (int l = 0; l < KL2; l++) { n1[l] += a * h[l]; m1[l] += a * dq[l]; }
                                for (int l = 0; l < K; l++) n2[l] += a * rr[l];
                            }
                        }
                        // sym(M1 B^') (K x K)
                        std::vector<double> MB((size_t)K * K, 0.0);
                        for (int k = 0; k < K; k++)
                            for (int l = 0; l < K; l++) {
                                double t = 0.0;
                                for (int a2 = 0; a2 < kc; a2++) t += M1[(size_t)k * KL2 + a2] * Bh[(size_t)l * KL2 + a2];
                                MB[(size_t)k * K + l] = t;
                            }
                        for (int k = 0; k < K; k++)
                            for (int l = 0; l < k; l++) { const double sy = 0.5 * (MB[(size_t)k * K + l] + MB[(size_t)l * K + k]); MB[(size_t)k * K + l] = MB[(size_t)l * K + k] = sy; }
                        std::vector<double> Bab((size_t)K * KL2);
                        for (size_t i = 0; i < Bab.size(); i++) Bab[i] = std::fabs(Bh[i]);
                        const double up = 1.0 + 1e-6;
                        lb.nN1 = sigma_upper_bound(N1.data(), K, KL2) * up;
                        lb.nM1 = sigma_upper_bound(MB.data(), K, K) * up;
                        lb.nN2 = sigma_upper_bound(N2.data(), K, K) * up;
                        { const double x = sigma_upper_bound(Rh.data(), m.n_sv, KL2); lb.nHabs = x * x * up; }
                        { const double x = sigma_upper_bound(Rd.data(), m.n_sv, KL2); lb.nDabs = x * x * up; }
                        { const double x = sigma_upper_bound(Rr.data(), m.n_sv, K); lb.nRabs = x * x * up; }
                        lb.sQb = sigma_upper_bound(Rq.data(), m.n_sv, K) * up;
                        lb.sQtaa = sigma_upper_bound(Ra.data(), m.n_sv, KL2) * up;
                        // ---- the plain epilogue on the same operands (LrBand: sig_q ...; screen_band.h: lr_finish_band_plain) ----
                        std::vector<double> gt((size_t)KL2, 0.0), ht((size_t)KL2, 0.0), rhoR((size_t)K, 0.0), Bq((size_t)m.n_sv * KL2), Br((size_t)m.n_sv * K);
                        for (int n = 0; n < m.n_sv; n++) {
                            const double bn = b[(size_t)n];
                            for (int l = 0; l < KL2; l++) {
                                gt[(size_t)l] += bn * Qth[(size_t)n * KL2 + l];
                                ht[(size_t)l] += bn * dQt[(size_t)n * KL2 + l];
                                Bq[(size_t)n * KL2 + l] = bn * Qt[(size_t)n * KL2 + l];
                            }
                            for (int l = 0; l < K; l++) { rhoR[(size_t)l] += bn * Rm[(size_t)n * K + l]; Br[(size_t)n * K + l] = bn * Rm[(size_t)n * K + l]; }
                        }
                        lb.sig_q = sigma_upper_bound(Qth.data(), m.n_sv, KL2) * up;
                        lb.sig_dq = sigma_upper_bound(dQt.data(), m.n_sv, KL2) * up;
                        lb.sig_r = sigma_upper_bound(Rm.data(), m.n_sv, K) * up;
                        lb.sbq = sigma_upper_bound(Bq.data(), m.n_sv, KL2) * up;
                        lb.sbr = sigma_upper_bound(Br.data(), m.n_sv, K) * up;
                        {
                            double r2 = 0.0, g2 = 0.0, bg2 = 0.0, bh2 = 0.0, ga2 = 0.0;
                            for (int l = 0; l < K; l++) r2 += rhoR[(size_t)l] * rhoR[(size_t)l];
                            for (int l = 0; l < KL2; l++) g2 += gt[(size_t)l] * gt[(size_t)l];
                            lr_corr_g.assign((size_t)K, 0.0); lr_corr_h.assign((size_t)K, 0.0);
                            for (int sl = 0; sl < K; sl++) {
                                double tg = 0.0, th = 0.0, ta = 0.0;
                                for (int a2 = 0; a2 < kc; a2++) {
                                    const double bv = Bh[(size_t)sl * KL2 + a2];
                                    tg += bv * gt[(size_t)a2]; th += bv * ht[(size_t)a2]; ta += std::fabs(bv) * std::fabs(gt[(size_t)a2]);
                                }
                                lr_corr_g[(size_t)sl] = tg; lr_corr_h[(size_t)sl] = th;
                                bg2 += tg * tg; bh2 += th * th; ga2 += ta * ta;
                            }
                            lb.rho_norm = std::sqrt(r2) * up; lb.gt_norm = std::sqrt(g2) * up;
                            lb.bg_norm = std::sqrt(bg2) * up; lb.bh_norm = std::sqrt(bh2) * up; lb.gabsB = std::sqrt(ga2) * up;
                        }
                        lb.sigB = sigma_upper_bound(Bh.data(), K, KL2) * up;
                        lb.sigAbsB = sigma_upper_bound(Bab.data(), K, KL2) * up;
                        for (double *x : {&lb.Ca, &lb.Cq1, &lb.Cqq, &lb.Babs, &lb.qmax, &lb.dqmax, &lb.rmax}) *x *= 1.0 + 1e-9;
                        // (B is orthonormal and q~_n solves G q~_n = B'q_n only to fp64 roundings: what that leaves -- y*.(B'q_n - G q~_n) and
                        // p_perp'B(B'q_n - q~_n) -- is 1e-15 of |p||q_n|: charged to the two per-SV bounds)
                        lb.dqmax += 1e-13 * (1.0 + lb.qmax);
                        lb.rmax += 1e-13 * (1.0 + lb.qmax) + 1e-300;
                        lb.acc10 = cp.acc_rel;
                        lb.acc6 = cp.acc_rel * (double)kLrSteps / (double)(kS0K / 32);
                        lb.gnorm = cp.cr_gnorm;
                        lb.mu_norm = cp.cr_mu_norm; lb.mu_norm_t = cp.cr_mu_norm_t;
                        lb.eta_abs = cp.eta_abs + 1e-14 * (1.0 + cp.cr_mu_norm);     // + the roundings of the centre's projection into the affine subspace
                        lb.scale = cp.scale;
                        // projection tiles: output slot k -> tile k / 32, row 16 n + 4 g + r with w = k % 32, g = w / 8, n = (w % 8) / 4, r = w % 4 (screen.hip: k_project)
                        std::vector<char> bt((size_t)kLrSteps * kLrProjTileBytes, 0);
                        for (int k = 0; k < KL2; k++) {
                            const int w = k & 31, g = w >> 3, nn = (w & 7) >> 2, r = w & 3, j = 16 * nn + 4 * g + r;
                            char *tile = bt.data() + (size_t)(k >> 5) * kLrProjTileBytes;
                            for (int sl = 0; sl < K; sl++) {
                                const _Float16 h = (_Float16)(float)Bh[(size_t)sl * KL2 + k];
                                memcpy(tile + h_image_offset(j, sl), &h, 2);
                            }
                        }
                        // the same matrix by INPUT k-step for the fused form (k_svm_screen_lr<., true>): tile s = the twelve 16-row blocks of output
                        // slots (row i of block rb = output slot 32 (rb / 2) + 8 (i / 4) + 4 (rb % 2) + i % 4) for input slots 32 s .. 32 s + 31, lane
                        // 16 kg + i holding the 8 inputs 32 s + 8 kg .. + 7: [rb][lane][8 halves] = 12 KiB, the SV tiles' piece layout
                        std::vector<char> bti((size_t)kHFull * kLrMatBytes, 0);
                        for (int st = 0; st < kHFull; st++)
                            for (int rb = 0; rb < 2 * kLrSteps; rb++)
                                for (int ln = 0; ln < 64; ln++) {
                                    const int i = ln & 15, kg = ln >> 4;
                                    const int ko = 32 * (rb >> 1) + 8 * (i >> 2) + 4 * (rb & 1) + (i & 3);
                                    for (int jj = 0; jj < 8; jj++) {
                                        const _Float16 h = (_Float16)(float)Bh[(size_t)(32 * st + 8 * kg + jj) * KL2 + ko];
                                        memcpy(bti.data() + (size_t)st * kLrMatBytes + (size_t)rb * 1024 + (size_t)ln * 16 + (size_t)jj * 2, &h, 2);
                                    }
                                }
                        const bool fin2 = std::isfinite(lb.nN1) && std::isfinite(lb.nM1) && std::isfinite(lb.nN2) && std::isfinite(lb.nHabs) && std::isfinite(lb.nRabs) &&
                                          std::isfinite(lb.sigB) && lb.qmax < 60000.0 && lb.sigB < 1.5;
                        if (fin2) {
                            if (hipSuccess != e->d_lr_btiles.alloc(bt.size()) || hipSuccess != e->d_svt_lr.alloc(imgl.size()) || hipSuccess != e->d_lr_btiles_in.alloc(bti.size()))
                                return fail(e, HAF_E_DEVICE, "hipMalloc(low-rank tables)");
                            HIPCHK(e, hipMemcpy(e->d_lr_btiles.p, bt.data(), bt.size(), hipMemcpyHostToDevice));
                            HIPCHK(e, hipMemcpy(e->d_lr_btiles_in.p, bti.data(), bti.size(), hipMemcpyHostToDevice));
                            e->lr_fused = !test_env("HAF_LR_UNFUSED");
                            HIPCHK(e, hipMemcpy(e->d_svt_lr.p, imgl.data(), imgl.size(), hipMemcpyHostToDevice));
                            e->lr_band = lb;
                            e->lr_rank = lr_rank;
                            cp.lr_rho = lr_rho;
                            e->lr_available = true;
                            // the plain epilogue's instance of the feature kernel's constants: the centred descriptors, its own correction vectors
                            {
                                std::vector<ScrCorr> scp((size_t)kS0K * 2);
                                ScrCorr2 *sp2 = reinterpret_cast<ScrCorr2 *>(scp.data() + kS0K);
                                for (int sl = 0; sl < kS0K; sl++) {
                                    scp[(size_t)sl].g = (float)lr_corr_g[(size_t)sl]; scp[(size_t)sl].hd = (float)lr_corr_h[(size_t)sl];
                                    // round 5: the feature kernel's fifth sum (unused by this form's band) carries L = ln2 p.g of the
                                    // centred-remainder form -- the very constants and the very fp32 sum of screen_cr's `cr` -- so that
                                    // tier 0b can run on this pass's images and raw sums (k_svm_screen_lr<CR_EXP, ., GATHER>: raw[6])
                                    scp[(size_t)sl].ub = scc[(size_t)sl].hd; scp[(size_t)sl].pad = 0.0f;
                                    ScrCorr2 &p2 = sp2[sl >> 1];
                                    p2.g[sl & 1] = scp[(size_t)sl].g; p2.hd[sl & 1] = scp[(size_t)sl].hd; p2.ub[sl & 1] = scp[(size_t)sl].ub; p2.pad[sl & 1] = 0.0f;
                                }
                                if (hipSuccess != e->d_corr_lrp.alloc(scp.size())) return fail(e, HAF_E_DEVICE, "hipMalloc(low-rank tables)");
                                HIPCHK(e, hipMemcpy(e->d_corr_lrp.p, scp.data(), scp.size() * sizeof(ScrCorr), hipMemcpyHostToDevice));
                                e->screen_lrp = cp;
                                e->screen_lrp.corr = e->d_corr_lrp.p;
                                e->screen_lrp.corr2 = reinterpret_cast<const ScrCorr2 *>(e->d_corr_lrp.p + kS0K);
                                const bool finp = std::isfinite(lb.sig_q) && std::isfinite(lb.sig_r) && std::isfinite(lb.sbq) && std::isfinite(lb.sbr) &&
                                                  std::isfinite(lb.rho_norm) && std::isfinite(lb.bg_norm) && std::isfinite(lb.gabsB);
                                e->lr_plain_available = finp && !test_env("HAF_NO_LR_PLAIN");
                            }
                        }
                    }
                }
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
    e->i8_active = !(c.flags & HAF_FLAG_PROBABILITY) && !test_env("HAF_NO_I8") && e->kx <= 64 * kI8Steps && !e->generic_kernel;
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
    e->screen_lrp.scale = 1.001 * guard0_scale;
    e->lr_band.scale = 1.001 * guard0_scale;
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
    e->exact.kernel_type = m.kernel_type; e->exact.degree = m.degree; e->exact.coef0 = m.coef0;
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
    // a whole number of 64-entry blocks: the exact tiers' feature kernels finish whole blocks, padding entries included, and a FULL window
    // whose length was not a multiple of 64 had them write up to 63 doubles past the last row of d_part64 (found by the guard zones of
    // the testing build, round 5: every request of the GPU suite under HAF_CANARY_CHECK)
    e->flag_cap = std::max(64, e->flag_cap / 64 * 64);
    const int mode = contraction_mode(c);
    // screening pass: up to half of the evaluations may go on to the three-pass kernel; a model that sends more is served by
    // the three-pass kernel alone from then on (haf_score_rolls)
    e->flag0_cap = mode == MODE_SCREEN ? (int)std::min<long>(std::max<long>(4096, (e->max_evals / 2 + 255) / 256 * 256), 1L << 23) : 0;
    // testing build, the overflow campaigns (tools/fuzz_parity.py --overflow): a screening list far smaller than a request, so that every
    // producer of it runs into its capacity and every consumer meets a counter beyond it (the engine's answer to such an overflow: the
    // next form, then the three-pass kernel for everything).  The windows of the exact tiers shrink with HAF_FLAG_WINDOW above.  The
    // TIER lists (list_cap) are not shrunk: they hold one entry per evaluation of the largest request, no producer can overrun them,
    // and their window-by-window consumers rely on exactly that (a counter never exceeds the list).
    if (const char *v = test_env("HAF_FLAG0_CAP")) e->flag0_cap = mode == MODE_SCREEN ? (int)std::min<long>(e->list_cap, std::max(256, atoi(v) / 256 * 256)) : 0;
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
    ok &= hipSuccess == e->d_iiabs.alloc(B * R);
    ok &= hipSuccess == e->d_ii.alloc(B * R * (H + 1) * (W + 1));
    ok &= hipSuccess == e->d_mask.alloc(e->cells_cap);
    ok &= hipSuccess == e->d_rowcount.alloc(B * R * H);
    ok &= hipSuccess == e->d_rowoff.alloc(2 * (B * R * H + 1));      // whole-chunk and left-over starts (k_scan)
    ok &= hipSuccess == e->d_brcount.alloc(B * R);
    ok &= hipSuccess == e->d_brslot.alloc(B * R);
    if (ok) ok &= hipSuccess == hipMemsetAsync(e->d_brslot.p, 0, B * R * sizeof(unsigned long long), e->stream);   // epoch 0: no request yet (on the engine's stream: in front of its first kernel)
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
        ok &= hipSuccess == e->d_screen_part.alloc(screen_part_bytes());
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
    ok &= hipSuccess == e->d_tier_words.alloc(((size_t)e->flag_cap + 63) / 64 + 4);
    ok &= hipSuccess == e->d_t1_flags.alloc(t1_flag_bytes(e->max_evals_pad));    // (a default-mode engine can fall back to three passes for every evaluation)
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

}  // namespace haf_host
