// engine_geometry.cpp -- what stays on the host of the geometry: the per-roll 4x4 transform and the rotated-rectangle scalars of
// pnt_in_box (a few dozen fp32 operations per roll that use glibc sinf/cosf/atan2f exactly as the reference does), the sequential
// cross-roll rule and the final grasp pose (once per goal).  Built with -ffp-contract=off.
#include "engine_state.h"

namespace haf_host {

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
                    Mat4 *pre_roll, float *roll_cs)
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

void fill_roll_geo(const haf_config &cfg, const haf_grasp_input &in, const NormalisedInput &n, int roll, RollGeo &g, float *m0)
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
int finalize_impl(const haf_config &c, const haf_grasp_input *in, const haf_roll_record *rec, haf_grasp_output *out,
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
int roll_pose_impl(const haf_config &c, const haf_grasp_input *in, const haf_roll_record *rec, int roll, haf_grasp_output *out,
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

}  // namespace haf_host
