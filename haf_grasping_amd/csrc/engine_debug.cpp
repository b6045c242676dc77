// engine_debug.cpp -- haf_get_roll_grid, haf_debug_fetch and haf_debug_fetch_attr: intermediate stages of the last request for the
// parity tests (height grid, integral image, mask, features, attributes, decision values, vote grids).
#include "engine_state.h"

extern "C" {

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

int haf_get_roll_grid(haf_engine *e, int32_t cloud, int32_t roll, float *eval_grid, uint8_t *mask)
{
    return guarded(e ? &e->error : nullptr, [&] { return get_roll_grid_impl(e, cloud, roll, eval_grid, mask); });
}

int haf_debug_fetch(haf_engine *e, int32_t what, int32_t cloud, int32_t roll, void *dst, size_t dst_bytes)
{
    return guarded(e ? &e->error : nullptr, [&] { return debug_fetch_impl(e, what, cloud, roll, dst, dst_bytes); });
}

}  // extern "C"
