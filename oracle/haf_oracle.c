/*
 * haf_oracle.c -- TEST INFRASTRUCTURE ONLY (see haf_oracle.h for the rules and
 * the parity-pinning status).  Plain C11, scalar, no dependencies but libc/libm.
 *
 * Every function cites the reference lines it restates:
 *   server.cpp = /root/reference/src/calc_grasppoints_action_server.cpp
 *   fv.cpp     = /root/reference/src/CIntImage_to_Featurevec.cpp
 *   libsvm     = /root/reference/libsvm-3.12/{svm.cpp,svm-scale.c,svm-predict.c}
 */
#define _GNU_SOURCE
#include "haf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define HAFO_PI 3.141592653 /* server.cpp:94 (truncated on purpose) */

/* ------------------------------------------------------------------ */
/* file helpers                                                        */
/* ------------------------------------------------------------------ */
static char *slurp(const char *path, size_t *len)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (!buf) { fclose(fp); return NULL; }
    size_t got = fread(buf, 1, (size_t)n, fp);
    fclose(fp);
    buf[got] = 0;
    *len = got;
    return buf;
}

/* ------------------------------------------------------------------ */
/* Features.txt  (fv.cpp:47-84, CHaarFeature.cpp:54-78)                */
/* ------------------------------------------------------------------ */

/* std::string::find("\t", start) with the result stored in an `int` (fv.cpp:63,68) */
static int str_find_tab(const char *s, int len, int start)
{
    if (start < 0 || start > len) return -1;
    for (int i = start; i < len; i++)
        if (s[i] == '\t') return i;
    return -1; /* npos truncated to int */
}

/* line.substr(start, end-start) -> NUL-terminated copy in tmp */
static void str_substr(const char *s, int len, int start, int end, char *tmp, size_t cap)
{
    size_t count = (size_t)(end - start); /* huge when end == -1 : "rest of line" */
    size_t avail = (start <= len) ? (size_t)(len - start) : 0;
    if (count > avail) count = avail;
    if (count >= cap) count = cap - 1;
    if (start <= len) memcpy(tmp, s + start, count);
    tmp[count] = 0;
}

hafo_features *hafo_features_load(const char *path)
{
    size_t size;
    char *buf = slurp(path, &size);
    if (!buf) return NULL;
    hafo_features *f = (hafo_features *)calloc(1, sizeof(*f));
    int cap = 0;
    size_t pos = 0;
    /* emulate: getline(file,line); while (file.good()) { parse; getline; }   (fv.cpp:60-82) */
    for (;;) {
        if (pos >= size) break;                       /* getline extracts nothing -> failbit */
        const char *nl = (const char *)memchr(buf + pos, '\n', size - pos);
        if (!nl) break;                               /* last line without '\n': eofbit -> !good(), NOT parsed */
        const char *line = buf + pos;
        int len = (int)(nl - line);
        pos = (size_t)(nl - buf) + 1;

        if (f->n == cap) {
            cap = cap ? cap * 2 : 512;
            f->reg = (int *)realloc(f->reg, sizeof(int) * 16 * (size_t)cap);
            f->w = (float *)realloc(f->w, sizeof(float) * 4 * (size_t)cap);
        }
        int *reg = f->reg + 16 * f->n;
        float *w = f->w + 4 * f->n;
        char tmp[256];
        int start = 0, end = 0;
        for (int i = 0; i < 16; i++) {                /* fv.cpp:67-71 */
            end = str_find_tab(line, len, start);
            str_substr(line, len, start, end, tmp, sizeof tmp);
            reg[i] = atoi(tmp);
            start = end + 1;
        }
        float reg_w[4];
        for (int j = 0; j < 4; j++) {                 /* fv.cpp:72-76: atof -> float */
            end = str_find_tab(line, len, start);
            str_substr(line, len, start, end, tmp, sizeof tmp);
            reg_w[j] = (float)atof(tmp);
            start = end + 1;
        }
        /* 4-region constructor stores weights 0..2 only; weights[3] stays 0 (CHaarFeature.cpp:56-60) */
        w[0] = (float)(double)reg_w[0];
        w[1] = (float)(double)reg_w[1];
        w[2] = (float)(double)reg_w[2];
        w[3] = 0.0f;
        f->n++;
    }
    free(buf);
    return f;
}

void hafo_features_free(hafo_features *f)
{
    if (!f) return;
    free(f->reg); free(f->w); free(f);
}

/* ------------------------------------------------------------------ */
/* range file  (svm-scale.c:108-132, 204-231)                          */
/* ------------------------------------------------------------------ */
hafo_range *hafo_range_load(const char *path)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return NULL;
    hafo_range *r = (hafo_range *)calloc(1, sizeof(*r));
    r->lower = -1.0; r->upper = 1.0;           /* svm-scale.c:23 */
    int c = fgetc(fp);
    if (c == 'y') {                            /* svm-scale.c:210-215: y-scaling lines, irrelevant for x */
        double a, b;
        if (fscanf(fp, "%lf %lf\n", &a, &b) != 2) { /* ignore */ }
        if (fscanf(fp, "%lf %lf\n", &a, &b) != 2) { /* ignore */ }
    } else {
        ungetc(c, fp);
    }
    int cap = 0;
    if (fgetc(fp) == 'x') {                    /* svm-scale.c:219-229 */
        if (fscanf(fp, "%lf %lf\n", &r->lower, &r->upper) != 2) { /* keep defaults */ }
        int idx; double fmin, fmax;
        while (fscanf(fp, "%d %lf %lf\n", &idx, &fmin, &fmax) == 3) {
            if (idx < 0) continue;
            if (idx >= cap) {
                int ncap = cap ? cap : 512;
                while (ncap <= idx) ncap *= 2;
                r->fmin = (double *)realloc(r->fmin, sizeof(double) * (size_t)ncap);
                r->fmax = (double *)realloc(r->fmax, sizeof(double) * (size_t)ncap);
                r->present = (unsigned char *)realloc(r->present, (size_t)ncap);
                for (int i = cap; i < ncap; i++) { r->fmin[i] = 0; r->fmax[i] = 0; r->present[i] = 0; }
                cap = ncap;
            }
            r->fmin[idx] = fmin; r->fmax[idx] = fmax; r->present[idx] = 1;
            if (idx > r->max_index) r->max_index = idx;
        }
    }
    fclose(fp);
    if (cap == 0) {
        r->fmin = (double *)calloc(1, sizeof(double));
        r->fmax = (double *)calloc(1, sizeof(double));
        r->present = (unsigned char *)calloc(1, 1);
    }
    return r;
}

void hafo_range_free(hafo_range *r)
{
    if (!r) return;
    free(r->fmin); free(r->fmax); free(r->present); free(r);
}

/* ------------------------------------------------------------------ */
/* libsvm model (svm.cpp:2714-2927)                                    */
/* ------------------------------------------------------------------ */
static const char *k_svm_types[] = {"c_svc", "nu_svc", "one_class", "epsilon_svr", "nu_svr", NULL};
static const char *k_kernel_types[] = {"linear", "polynomial", "rbf", "sigmoid", "precomputed", NULL};

hafo_model *hafo_model_load(const char *path)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    hafo_model *m = (hafo_model *)calloc(1, sizeof(*m));
    m->svm_type = -1; m->kernel_type = -1;
    char cmd[81];
    int ok = 1, have_sv = 0, have_a = 0, have_b = 0;
    while (ok) {
        if (fscanf(fp, "%80s", cmd) != 1) { ok = 0; break; }
        if (!strcmp(cmd, "svm_type")) {
            if (fscanf(fp, "%80s", cmd) != 1) { ok = 0; break; }
            for (int i = 0; k_svm_types[i]; i++) if (!strcmp(cmd, k_svm_types[i])) m->svm_type = i;
            if (m->svm_type < 0) ok = 0;
        } else if (!strcmp(cmd, "kernel_type")) {
            if (fscanf(fp, "%80s", cmd) != 1) { ok = 0; break; }
            for (int i = 0; k_kernel_types[i]; i++) if (!strcmp(cmd, k_kernel_types[i])) m->kernel_type = i;
            if (m->kernel_type < 0) ok = 0;
        } else if (!strcmp(cmd, "degree")) ok = fscanf(fp, "%d", &m->degree) == 1;
        else if (!strcmp(cmd, "gamma")) ok = fscanf(fp, "%lf", &m->gamma) == 1;
        else if (!strcmp(cmd, "coef0")) ok = fscanf(fp, "%lf", &m->coef0) == 1;
        else if (!strcmp(cmd, "nr_class")) ok = fscanf(fp, "%d", &m->nr_class) == 1;
        else if (!strcmp(cmd, "total_sv")) ok = fscanf(fp, "%d", &m->l) == 1;
        else if (!strcmp(cmd, "rho")) {
            if (m->nr_class != 2) { ok = 0; break; }
            ok = fscanf(fp, "%lf", &m->rho) == 1;
        } else if (!strcmp(cmd, "label")) {
            if (m->nr_class != 2) { ok = 0; break; }
            ok = fscanf(fp, "%d %d", &m->label[0], &m->label[1]) == 2;
        } else if (!strcmp(cmd, "probA")) { ok = m->nr_class == 2 && fscanf(fp, "%lf", &m->probA) == 1; have_a = 1; }   /* 2811-2817 */
        else if (!strcmp(cmd, "probB")) { ok = m->nr_class == 2 && fscanf(fp, "%lf", &m->probB) == 1; have_b = 1; }     /* 2818-2824 */
        else if (!strcmp(cmd, "nr_sv")) {
            if (m->nr_class != 2) { ok = 0; break; }
            ok = fscanf(fp, "%d %d", &m->nSV[0], &m->nSV[1]) == 2;
        } else if (!strcmp(cmd, "SV")) {
            for (;;) { int c = getc(fp); if (c == EOF || c == '\n') break; }   /* svm.cpp:2834-2838 */
            have_sv = 1;
            break;
        } else ok = 0;                                                        /* svm.cpp:2841-2852 */
    }
    /* what svm-predict serves on the server's path: a 2-class C-SVC / nu-SVC with any of libsvm's four vector kernels (round 5; the
       reference's own model is RBF); precomputed kernels have no attribute vectors */
    if (!ok || !have_sv || m->nr_class != 2 || m->kernel_type > 3 || m->svm_type > 1 || m->l <= 0) {
        fclose(fp); free(m); return NULL;
    }
    m->has_prob = have_a && have_b;
    long pos = ftell(fp);
    fseek(fp, 0, SEEK_END);
    long end = ftell(fp);
    fseek(fp, pos, SEEK_SET);
    char *body = (char *)malloc((size_t)(end - pos) + 1);
    size_t got = fread(body, 1, (size_t)(end - pos), fp);
    body[got] = 0;
    fclose(fp);

    /* pass 1: max index */
    int D = 0;
    for (char *p = body; *p; p++) {
        if (*p == ':') {
            char *q = p;
            while (q > body && q[-1] >= '0' && q[-1] <= '9') q--;
            int idx = atoi(q);
            if (idx > D) D = idx;
        }
    }
    if (D <= 0) { free(body); free(m); return NULL; }
    m->D = D;
    m->coef = (double *)calloc((size_t)m->l, sizeof(double));
    m->sv = (double *)calloc((size_t)m->l * (size_t)D, sizeof(double));
    char *p = body;
    for (int i = 0; i < m->l; i++) {                                 /* svm.cpp:2890-2916 */
        char *eol = strchr(p, '\n');
        if (eol) *eol = 0;
        char *endp;
        m->coef[i] = strtod(p, &endp);
        p = endp;
        for (;;) {
            while (*p == ' ' || *p == '\t' || *p == '\r') p++;
            if (!*p) break;
            long idx = strtol(p, &endp, 10);
            if (endp == p || *endp != ':') break;
            p = endp + 1;
            double val = strtod(p, &endp);
            p = endp;
            if (idx >= 1 && idx <= D) m->sv[(size_t)i * D + (idx - 1)] = val;
        }
        if (!eol) { if (i != m->l - 1) { hafo_model_free(m); free(body); return NULL; } break; }
        p = eol + 1;
    }
    free(body);
    return m;
}

void hafo_model_free(hafo_model *m)
{
    if (!m) return;
    free(m->coef); free(m->sv); free(m);
}

/* ------------------------------------------------------------------ */
/* a1: transform matrix (server.cpp:423-483 and 1276-1334)             */
/* ------------------------------------------------------------------ */
static void mat4_identity(float *A) { memset(A, 0, 16 * sizeof(float)); A[0] = A[5] = A[10] = A[15] = 1.0f; }

/* Third-party arithmetic the reference does not pin (Eigen 4x4 products, pcl::transformPointCloud, cv::integral):
 * variant 0 is the definition of record -- what the product's kernels reproduce bit for bit.  The other variants restate the
 * plausible alternative evaluation orders of those libraries so that tests/test_oracle.py can MEASURE how much the choice
 * matters (how many height bins, mask cells, labels and winners change over every golden cloud x configuration).  Test
 * infrastructure only; not thread-safe (set it, run, reset). */
static int g_variant = 0;
void hafo_set_variant(int flags) { g_variant = flags; }
int hafo_get_variant(void) { return g_variant; }

/* C = A*B, fp32, no FMA.  Variant 0: inner sum left to right (definition of record).  HAFO_V_EIGEN_TREE: Eigen's vectorised
 * 4-term reduction (a0b0 + a1b1) + (a2b2 + a3b3). */
static void mat4_mul(const float *A, const float *B, float *C)
{
    float T[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s;
            if (g_variant & HAFO_V_EIGEN_TREE) {
                const float lo = A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j];
                const float hi = A[i * 4 + 2] * B[2 * 4 + j] + A[i * 4 + 3] * B[3 * 4 + j];
                s = lo + hi;
            } else {
                s = A[i * 4 + 0] * B[0 * 4 + j];
                s = s + A[i * 4 + 1] * B[1 * 4 + j];
                s = s + A[i * 4 + 2] * B[2 * 4 + j];
                s = s + A[i * 4 + 3] * B[3 * 4 + j];
            }
            T[i * 4 + j] = s;
        }
    memcpy(C, T, sizeof T);
}

void hafo_transform(const hafo_cfg *cfg, const hafo_input *in, int roll, int use_double_atan2, float M[16])
{
    /* approach-vector normalisation, server.cpp:270-273 */
    float vector_length = (float)sqrt(in->approach[0] * in->approach[0] + in->approach[1] * in->approach[1] +
                                      in->approach[2] * in->approach[2]);
    double avx = in->approach[0] / vector_length;
    double avy = in->approach[1] / vector_length;
    double avz = in->approach[2] / vector_length;

    float S[16], To[16], Rz[16], Rx[16], Tf[16], R[16];
    mat4_identity(S); mat4_identity(To); mat4_identity(Rz); mat4_identity(Rx); mat4_identity(Tf); mat4_identity(R);
    S[0] = (float)in->gripper_width;                 /* 433 / 1290 */
    To[3] = (float)(-in->center[0]);                 /* 435-437 / 1292-1294 */
    To[7] = (float)(-in->center[1]);
    To[11] = (float)(-in->center[2]);
    Tf[11] = 0 + cfg->z_shift;                       /* 441 / 1298 */

    float rot_about_z, rot_about_x = 0;
    if (!use_double_atan2) {                         /* generate_grid: float av (418-420, 444-454) */
        float fx = (float)avx, fy = (float)avy, fz = (float)avz;
        if (fy == 0 && fx == 0) {
            rot_about_z = 0;
            rot_about_x = (fz >= 0) ? 0 : (float)HAFO_PI;
        } else {
            rot_about_z = (float)(90 * HAFO_PI / 180.0 - atan2f(fy, fx));
            rot_about_x = (float)(90 * HAFO_PI / 180.0 - atan2f(fz, sqrtf(fy * fy + fx * fx)));
        }
    } else {                                         /* transform_gp: double approach_vector (1300-1310) */
        if (avy == 0 && avx == 0) {
            rot_about_z = 0;
            rot_about_x = (avz >= 0) ? 0 : (float)HAFO_PI;
        } else {
            rot_about_z = (float)(90 * HAFO_PI / 180.0 - atan2(avy, avx));
            rot_about_x = (float)(90 * HAFO_PI / 180.0 - atan2(avz, sqrt(avy * avy + avx * avx)));
        }
    }
    float angle = (float)(roll * cfg->roll_step_deg * HAFO_PI / 180);   /* 462 / 1315 */
    R[0] = cosf(angle); R[1] = -sinf(angle); R[4] = sinf(angle); R[5] = cosf(angle);
    Rz[0] = cosf(rot_about_z); Rz[1] = -sinf(rot_about_z); Rz[4] = sinf(rot_about_z); Rz[5] = cosf(rot_about_z);
    Rx[5] = cosf(rot_about_x); Rx[6] = -sinf(rot_about_x); Rx[9] = sinf(rot_about_x); Rx[10] = cosf(rot_about_x);

    /* mat_scale_x_dir * mat_rot * mat_sh_from_orig * mat_rot_x_axis * mat_rot_z_axis * mat_sh_to_orig (483 / 1334) */
    float T[16];
    if (g_variant & HAFO_V_CHAIN_RTL) {              /* the product chain associated from the right: S (R (Tf (Rx (Rz To)))) */
        mat4_mul(Rz, To, T);
        mat4_mul(Rx, T, T);
        mat4_mul(Tf, T, T);
        mat4_mul(R, T, T);
        mat4_mul(S, T, M);
        return;
    }
    mat4_mul(S, R, T);
    mat4_mul(T, Tf, T);
    mat4_mul(T, Rx, T);
    mat4_mul(T, Rz, T);
    mat4_mul(T, To, M);
}

/* ------------------------------------------------------------------ */
/* a1: height grid (server.cpp:487-528)                                */
/* ------------------------------------------------------------------ */
void hafo_height_grid(const hafo_cfg *cfg, const float *xyz, size_t n, size_t stride, const float M[16], float *h)
{
    const int H = cfg->H, W = cfg->W;
    const float r_col_m = (float)((0.5 * (float)W) / 100.0);   /* 410 */
    const float r_row_m = (float)((0.5 * (float)H) / 100.0);   /* 411 */
    for (int i = 0; i < H * W; i++) h[i] = -1.0f;               /* 499-501 */
    for (size_t i = 0; i < n; i++) {
        const float x = xyz[i * stride + 0], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
        /* pcl::transformPointCloud (488): fp32, left to right, no FMA (definition of record) */
        float px, py, pz;
        if (g_variant & HAFO_V_PCL_SSE) {            /* PCL >= 1.8 SSE path: (m0 x + m1 y) + (m2 z + m3) */
            px = (M[0] * x + M[1] * y) + (M[2] * z + M[3]);
            py = (M[4] * x + M[5] * y) + (M[6] * z + M[7]);
            pz = (M[8] * x + M[9] * y) + (M[10] * z + M[11]);
        } else if (g_variant & HAFO_V_FMA) {         /* the left-to-right expression with the compiler contracting a*b + c */
            px = fmaf(M[2], z, fmaf(M[1], y, M[0] * x)) + M[3];
            py = fmaf(M[6], z, fmaf(M[5], y, M[4] * x)) + M[7];
            pz = fmaf(M[10], z, fmaf(M[9], y, M[8] * x)) + M[11];
        } else {
            px = M[0] * x; px = px + M[1] * y; px = px + M[2] * z; px = px + M[3];
            py = M[4] * x; py = py + M[5] * y; py = py + M[6] * z; py = py + M[7];
            pz = M[8] * x; pz = pz + M[9] * y; pz = pz + M[10] * z; pz = pz + M[11];
        }
        if ((px > -r_row_m) && (px < r_row_m) && (py > -r_col_m) && (py < r_col_m)) {    /* 510-511 */
            int idx_x = (int)floorf(100 * (px - (-r_row_m)));                             /* 513 */
            int idx_y = (int)floorf(100 * (py - (-r_col_m)));                             /* 514 */
            if (idx_x < 0 || idx_x >= H || idx_y < 0 || idx_y >= W) continue; /* reference would write out of bounds */
            if (h[idx_x * W + idx_y] < pz) h[idx_x * W + idx_y] = pz;                     /* 515-518 */
        }
    }
    for (int i = 0; i < H * W; i++)
        if (h[i] < -0.99) h[i] = 0;                                                       /* 522-528 (double compare) */
}

/* ------------------------------------------------------------------ */
/* a2: integral image (server.cpp:577-613; cv::integral CV_64F)        */
/* ------------------------------------------------------------------ */
void hafo_integral(const hafo_cfg *cfg, const float *h, float *ii)
{
    const int H = cfg->H, W = cfg->W, W1 = W + 1;
    if (g_variant & HAFO_V_INTEGRAL_COLFIRST) {       /* running COLUMN sums, then accumulate along the row */
        double *col = (double *)calloc((size_t)W, sizeof(double));
        for (int c = 0; c < W1; c++) ii[c] = 0.0f;
        for (int r = 0; r < H; r++) {
            double acc = 0.0;
            ii[(r + 1) * W1] = 0.0f;
            for (int c = 0; c < W; c++) {
                col[c] += (double)h[r * W + c];
                acc += col[c];
                ii[(r + 1) * W1 + c + 1] = (float)acc;
            }
        }
        free(col);
        return;
    }
    double *prev = (double *)calloc((size_t)W1, sizeof(double));
    double *cur = (double *)calloc((size_t)W1, sizeof(double));
    for (int c = 0; c < W1; c++) ii[c] = 0.0f;
    for (int r = 0; r < H; r++) {
        double s = 0.0;                       /* running row sum, then add the row above: OpenCV's integral_ order */
        cur[0] = 0.0;
        for (int c = 0; c < W; c++) {
            s += (double)h[r * W + c];        /* 589: widened to double before the integral */
            cur[c + 1] = prev[c + 1] + s;
        }
        for (int c = 0; c < W1; c++) ii[(r + 1) * W1 + c] = (float)cur[c];   /* 599-601: narrowed to float */
        double *t = prev; prev = cur; cur = t;
    }
    free(prev); free(cur);
}

/* ------------------------------------------------------------------ */
/* a3: mask (server.cpp:666-749)                                       */
/* ------------------------------------------------------------------ */
void hafo_mask(const hafo_cfg *cfg, const hafo_input *in, int roll, const float *ii, unsigned char *mask)
{
    const int H = cfg->H, W1 = cfg->W + 1;
    const float boxrot_angle_init = 0.0f;    /* never assigned; zero pages in practice */
    float alpha_deg = (float)(-roll * cfg->roll_step_deg - boxrot_angle_init * 180 / HAFO_PI);   /* 679 */
    float alpha = (float)(alpha_deg * HAFO_PI / 180);                                            /* 680 */
    float cx = (float)(H / 2), cy = (float)(H / 2);                                               /* 681-682 */
    float boarder = 7.0f;
    int sx = (int)in->length_x, sy = (int)in->length_y;                                           /* 266-267 */
    float height_r = sx / 2 - boarder;                                                            /* 687 */
    float width_r = sy / 2 - boarder;                                                             /* 688 */
    float cx1 = cx - sinf(alpha) * height_r;                                                      /* 689-696 */
    float cy1 = cy + cosf(alpha) * height_r;
    float cx2 = cx + sinf(alpha) * height_r;
    float cy2 = cy - cosf(alpha) * height_r;
    float cx3 = (float)(cx - sin(alpha + HAFO_PI / 2) * width_r);
    float cy3 = (float)(cy + cos(alpha + HAFO_PI / 2) * width_r);
    float cx4 = (float)(cx + sin(alpha + HAFO_PI / 2) * width_r);
    float cy4 = (float)(cy - cos(alpha + HAFO_PI / 2) * width_r);
    const int th = 4;                      /* th_empty_r, 709 */
    const float ii_th_in_r = 0.03f;        /* 710 */
    const float sa = sinf(alpha), ca = cosf(alpha);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < H; j++) {
            int in_box = 0;
            if (i > 6 && i < H - 7 && j > 6 && j < H - 7) {                                       /* 713 */
                float box = ii[(i + th) * W1 + (j + th)] - ii[(i - th - 1) * W1 + (j + th)];
                box = box - ii[(i + th) * W1 + (j - th - 1)];
                box = box + ii[(i - th - 1) * W1 + (j - th - 1)];                                 /* 714-717 */
                if (box > ii_th_in_r) {
                    float t1 = -sa * (-cx1 + j) + ca * (-cy1 + i);                                /* 718-721 */
                    float t2 = -sa * (-cx2 + j) + ca * (-cy2 + i);
                    float t3 = ca * (-cx3 + j) + sa * (-cy3 + i);
                    float t4 = ca * (-cx4 + j) + sa * (-cy4 + i);
                    if ((double)t1 < 0.00001 && (double)t2 > -0.00001 && (double)t3 > -0.00001 &&
                        (double)t4 < 0.00001)
                        in_box = 1;
                }
            }
            mask[i * cfg->W + j] = (unsigned char)in_box;
        }
}

/* ------------------------------------------------------------------ */
/* a5/a6: feature values (fv.cpp:141-199)                              */
/* ------------------------------------------------------------------ */
void hafo_feature_values(const hafo_features *ft, int nshaf, const float *win, int ld, float *out)
{
    for (int f = 0; f < ft->n; f++) {
        const int *reg = ft->reg + 16 * f;
        const float *w = ft->w + 4 * f;
        float returnval = 0;
        if (f < nshaf) {
            for (int k = 0; k < 4; k++) {
                int x1 = reg[k * 4], x2 = reg[k * 4 + 1], y1 = reg[k * 4 + 2], y2 = reg[k * 4 + 3];
                float wgt = w[k];
                if (wgt == 0.0f || x2 < x1 || y2 < y1 || (x2 == 0 && y2 == 0)) continue;   /* 155-159 */
                float s = win[(x2 + 1) * ld + (y2 + 1)] - win[x1 * ld + (y2 + 1)];
                s = s - win[(x2 + 1) * ld + y1];
                s = s + win[x1 * ld + y1];
                returnval = returnval + wgt * s;                                             /* 161-162 */
            }
        } else {
            float r[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++) {
                int x1 = reg[k * 4], x2 = reg[k * 4 + 1], y1 = reg[k * 4 + 2], y2 = reg[k * 4 + 3];
                float wgt = w[k];
                if (wgt == 0.0f || x2 < x1 || y2 < y1 || (x2 == 0 && y2 == 0)) continue;   /* 177-181 */
                float s = win[(x2 + 1) * ld + (y2 + 1)] - win[x1 * ld + (y2 + 1)];
                s = s - win[(x2 + 1) * ld + y1];
                s = s + win[x1 * ld + y1];
                r[k] = wgt * s;                                                              /* 183-184 */
            }
            if (r[1] > r[0] && r[1] > r[2]) {                                                /* 187-191 */
                float a = r[1] - r[0], b = r[1] - r[2];
                returnval = (b < a) ? b : a;       /* std::min(a,b) */
            } else {
                returnval = -1.0f;
            }
        }
        out[f] = returnval;
    }
}

int hafo_feature_line(const float *vals, int n, char *buf, size_t cap)
{
    /* fv.cpp:125-135: "+1"/"-1" (uninitialised goodgps; the sign is never consumed), " k:%.4g" ..., "\n" */
    size_t o = 0;
    o += (size_t)snprintf(buf + o, cap - o, "-1");
    for (int k = 0; k < n && o < cap; k++)
        o += (size_t)snprintf(buf + o, cap - o, " %d:%.4g", k + 1, (double)vals[k]);
    if (o < cap) o += (size_t)snprintf(buf + o, cap - o, "\n");
    return (int)o;
}

double hafo_q4(float v)
{
    char b[64];
    snprintf(b, sizeof b, "%.4g", (double)v);   /* ostream << setprecision(4) << float (fv.cpp:133) */
    return strtod(b, NULL);                     /* sscanf("%lf") (svm-scale.c:270) */
}

double hafo_q6(double v)
{
    char b[64];
    snprintf(b, sizeof b, "%g", v);             /* svm-scale.c:350 */
    return strtod(b, NULL);                     /* svm-predict.c:108 */
}

/* svm-scale.c output() 333-353 on one row.  fmin/fmax are the EFFECTIVE tables after pass 2 / 2.5 */
static void scale_row(double lower, double upper, const double *fmin, const double *fmax, const double *q4, int n,
                      int skip_text, double *xs, int nx)
{
    for (int k = 0; k < nx; k++) xs[k] = 0.0;
    for (int idx = 1; idx <= n; idx++) {
        if (fmax[idx] == fmin[idx]) continue;                 /* 336-337 */
        double value = q4[idx - 1];
        if (value == fmin[idx]) value = lower;                /* 339-346 */
        else if (value == fmax[idx]) value = upper;
        else value = lower + (upper - lower) * (value - fmin[idx]) / (fmax[idx] - fmin[idx]);
        if (value != 0) xs[idx - 1] = skip_text ? value : hafo_q6(value);   /* 348-352 */
    }
}

void hafo_scale_row(const hafo_range *rg, const unsigned char *skip, const double *q4, int n, int skip_text,
                    double *xs, int D)
{
    /* convenience form: attributes listed in the range file use it, others must be flagged in skip[] */
    double *fmin = (double *)calloc((size_t)n + 1, sizeof(double));
    double *fmax = (double *)calloc((size_t)n + 1, sizeof(double));
    for (int k = 1; k <= n; k++) {
        if (k <= rg->max_index && rg->present[k] && !(skip && skip[k])) { fmin[k] = rg->fmin[k]; fmax[k] = rg->fmax[k]; }
        else { fmin[k] = fmax[k] = 0; }
    }
    scale_row(rg->lower, rg->upper, fmin, fmax, q4, n, skip_text, xs, D);
    free(fmin); free(fmax);
}

/* ------------------------------------------------------------------ */
/* a8: decision function (Kernel::k_function svm.cpp:318-371, svm_predict_values 2478-2532) */
/* ------------------------------------------------------------------ */
static double powi_(double base, int times)           /* svm.cpp:26-36 */
{
    double tmp = base, ret = 1.0;
    for (int t = times; t > 0; t /= 2) {
        if (t % 2 == 1) ret *= tmp;
        tmp = tmp * tmp;
    }
    return ret;
}

static double decision_nx2(const hafo_model *m, const double *xs, int nx, double *sabs_out)
{
    const int D = m->D;
    const int K = nx > D ? nx : D;
    double dec = 0, sabs = 0;
    for (int i = 0; i < m->l; i++) {
        const double *sv = m->sv + (size_t)i * D;
        double sum = 0, kv;
        if (m->kernel_type == 2) {             /* RBF, 325-365 */
            for (int k = 0; k < K; k++) {      /* dense form of the sparse merge: missing attribute = 0 */
                double xv = k < nx ? xs[k] : 0.0, yv = k < D ? sv[k] : 0.0;
                double d = xv - yv;
                sum += d * d;
            }
            kv = exp(-m->gamma * sum);
        } else {
            /* Kernel::dot (299-316): products of the attributes BOTH vectors carry, in index order.  A text row omits zeros
               (svm-scale.c:348, svm.cpp:2677), so "carried" = non-zero here; a product with a zero adds +-0 and changes nothing */
            for (int k = 0; k < K; k++) {
                double xv = k < nx ? xs[k] : 0.0, yv = k < D ? sv[k] : 0.0;
                if (xv != 0.0 && yv != 0.0) sum += xv * yv;
            }
            if (m->kernel_type == 0) kv = sum;                                          /* LINEAR 321-322 */
            else if (m->kernel_type == 1) kv = powi_(m->gamma * sum + m->coef0, m->degree);   /* POLY 323-324 */
            else kv = tanh(m->gamma * sum + m->coef0);                                  /* SIGMOID 366-367 */
        }
        dec += m->coef[i] * kv;                     /* 2509-2512 */
        sabs += fabs(m->coef[i] * kv);              /* not part of the reference: error scale for the tests (|coef| K for the RBF kernel) */
    }
    dec -= m->rho;                                  /* 2513 */
    if (sabs_out) *sabs_out = sabs;
    return dec;
}

static double decision_nx(const hafo_model *m, const double *xs, int nx) { return decision_nx2(m, xs, nx, NULL); }

double hafo_decision(const hafo_model *m, const double *xs) { return decision_nx(m, xs, m->D); }

void hafo_decision_rows(const hafo_model *m, const double *xs, long rows, double *dec)
{
    for (long r = 0; r < rows; r++) dec[r] = decision_nx(m, xs + (size_t)r * m->D, m->D);
}

int hafo_label_gridval(int label)
{
    char b[32];
    snprintf(b, sizeof b, "%g", (double)label);   /* svm-predict.c:127 */
    b[2] = 0;                                     /* line.substr(0,2)  server.cpp:843 */
    return atoi(b);
}

/* ------------------------------------------------------------------ */
/* f4: probability output (svm.cpp:1818-1888, 2550-2587)               */
/* ------------------------------------------------------------------ */
static double sigmoid_predict(double decision_value, double A, double B)     /* svm.cpp:1818-1826 */
{
    double fApB = decision_value * A + B;
    if (fApB >= 0) return exp(-fApB) / (1.0 + exp(-fApB));
    else return 1.0 / (1 + exp(fApB));
}

/* multiclass_probability (svm.cpp:1829-1888), k = 2, same operations in the same order */
static void multiclass_probability2(double r[2][2], double p[2])
{
    const int k = 2;
    int t, j, iter = 0, max_iter = 100;                                       /* max(100, k) */
    double Q[2][2], Qp[2], pQp, eps = 0.005 / k;
    for (t = 0; t < k; t++) {
        p[t] = 1.0 / k;
        Q[t][t] = 0;
        for (j = 0; j < t; j++) { Q[t][t] += r[j][t] * r[j][t]; Q[t][j] = Q[j][t]; }
        for (j = t + 1; j < k; j++) { Q[t][t] += r[j][t] * r[j][t]; Q[t][j] = -r[j][t] * r[t][j]; }
    }
    for (iter = 0; iter < max_iter; iter++) {
        pQp = 0;
        for (t = 0; t < k; t++) {
            Qp[t] = 0;
            for (j = 0; j < k; j++) Qp[t] += Q[t][j] * p[j];
            pQp += p[t] * Qp[t];
        }
        double max_error = 0;
        for (t = 0; t < k; t++) {
            double error = fabs(Qp[t] - pQp);
            if (error > max_error) max_error = error;
        }
        if (max_error < eps) break;
        for (t = 0; t < k; t++) {
            double diff = (-Qp[t] + pQp) / Q[t][t];
            p[t] += diff;
            pQp = (pQp + diff * (diff * Q[t][t] + 2 * Qp[t])) / (1 + diff) / (1 + diff);
            for (j = 0; j < k; j++) {
                Qp[j] = (Qp[j] + diff * Q[t][j]) / (1 + diff);
                p[j] /= (1 + diff);
            }
        }
    }
}

int hafo_probability(const hafo_model *m, double dec, double prob[2])            /* svm.cpp:2550-2587 */
{
    if (!m->has_prob) return 0;
    const double min_prob = 1e-7;
    double r[2][2] = {{0, 0}, {0, 0}};
    double s = sigmoid_predict(dec, m->probA, m->probB);
    s = s > min_prob ? s : min_prob;                                              /* max(.., min_prob) */
    s = s < 1 - min_prob ? s : 1 - min_prob;                                      /* min(.., 1 - min_prob) */
    r[0][1] = s;
    r[1][0] = 1 - r[0][1];
    multiclass_probability2(r, prob);
    return prob[1] > prob[0] ? m->label[1] : m->label[0];                         /* first maximum */
}

/* server.cpp:831-841 on one line of the output file: int res = atof(line.substr(0,2)); the first or (res > 0) second number
 * behind the label; float prob = atof(...); the cell gets res*prob.  std::string::find's npos becomes -1 in the reference's
 * `int` variables and a huge length in substr(pos, len): "to the end of the line". */
float hafo_probability_gridval(const char *line)
{
    const int len = (int)strlen(line);
    char two[3] = {0, 0, 0};
    for (int i = 0; i < 2 && i < len; i++) two[i] = line[i];
    int res = (int)atof(two);
    int start = -1, end = -1;
    for (int i = 0; i < len; i++) if (line[i] == ' ') { start = i; break; }
    if (start >= 0) for (int i = start + 1; i < len; i++) if (line[i] == ' ') { end = i; break; }
    if (res > 0) {
        start = end;
        end = -1;
        if (start >= 0) for (int i = start + 1; i < len; i++) if (line[i] == ' ') { end = i; break; }
    }
    if (start < 0 || start > len) return NAN;               /* the reference would throw std::out_of_range here */
    char tmp[128];
    int n = end < 0 ? len - start : end;                     /* substr(pos, LEN): `end` is used as a length */
    if (n > len - start) n = len - start;
    if (n > (int)sizeof tmp - 1) n = (int)sizeof tmp - 1;
    memcpy(tmp, line + start, (size_t)n);
    tmp[n] = 0;
    float prob = (float)atof(tmp);
    return res * prob;
}

/* server.cpp:865-932 with the float cell values of the probability branch: the 29 products and their sum in fp32, left to
 * right; topval_gp is an int, so every assignment truncates and every comparison converts it back */
void hafo_vote_f(const hafo_cfg *cfg, const float *g, float *ev, int best[3])
{
    const int H = cfg->H, W = cfg->W;
    const int w1 = 1, w2 = 2, w3 = 3, w4 = 4, w5 = 55;
    int topval = -1000, id_row = -1, id_col = -1;
#define G(r, c) (g[(r) * W + (c)])
    for (int row = 0; row < H; row++)
        for (int col = 0; col < W; col++) {
            float v;
            if (G(row, col) < 0 || row < 2 || row >= H - 2 || col < 4 || col >= W - 4) {
                v = 0;
            } else {
                v = w1 * G(row - 2, col - 2) + w2 * G(row - 2, col - 1) + w3 * G(row - 2, col) + w2 * G(row - 2, col + 1) + w1 * G(row - 2, col + 2) +
                    w2 * G(row - 1, col - 2) + w3 * G(row - 1, col - 1) + w4 * G(row - 1, col) + w3 * G(row - 1, col + 1) + w2 * G(row - 1, col + 2) +
                    w2 * G(row, col - 4) + w2 * G(row, col - 3) + w3 * G(row, col - 2) + w4 * G(row, col - 1) + w5 * G(row, col) + w4 * G(row, col + 1) + w3 * G(row, col + 2) + w2 * G(row, col + 3) + w2 * G(row, col + 4) +
                    w2 * G(row + 1, col - 2) + w3 * G(row + 1, col - 1) + w4 * G(row + 1, col) + w3 * G(row + 1, col + 1) + w2 * G(row + 1, col + 2) +
                    w1 * G(row + 2, col - 2) + w2 * G(row + 2, col - 1) + w3 * G(row + 2, col) + w2 * G(row + 2, col + 1) + w1 * G(row + 2, col + 2);
            }
            ev[row * W + col] = v;
            if (v > topval) { topval = (int)v; id_row = row; id_col = col; }   /* 882-885 */
        }
#undef G
    int longest = 0;                                                           /* 904-932 */
    for (int row = 0; row < H; row++) {
        int cur = 0;
        for (int col = 0; col < W; col++) {
            if (ev[row * W + col] == topval) {
                cur++;
                if (cur > longest) { longest = cur; id_row = row; id_col = col - cur / 2; }
            } else cur = 0;
        }
    }
    best[0] = id_row; best[1] = id_col; best[2] = topval;
}

/* ------------------------------------------------------------------ */
/* a10: vote, argmax, run centring (server.cpp:865-932)                */
/* ------------------------------------------------------------------ */
void hafo_vote(const hafo_cfg *cfg, const signed char *g, float *ev, int best[3])
{
    const int H = cfg->H, W = cfg->W;
    const int w1 = 1, w2 = 2, w3 = 3, w4 = 4, w5 = 55;
    int topval = -1000, id_row = -1, id_col = -1;
#define G(r, c) ((int)g[(r) * W + (c)])
    for (int row = 0; row < H; row++)
        for (int col = 0; col < W; col++) {
            float v;
            if (G(row, col) < 0 || row < 2 || row >= H - 2 || col < 4 || col >= W - 4) {
                v = 0;   /* 870-871; the border guard never triggers for masked cells (mask needs 6 < i < H-7) */
            } else {
                int s = w1 * G(row - 2, col - 2) + w2 * G(row - 2, col - 1) + w3 * G(row - 2, col) + w2 * G(row - 2, col + 1) + w1 * G(row - 2, col + 2) +
                        w2 * G(row - 1, col - 2) + w3 * G(row - 1, col - 1) + w4 * G(row - 1, col) + w3 * G(row - 1, col + 1) + w2 * G(row - 1, col + 2) +
                        w2 * G(row, col - 4) + w2 * G(row, col - 3) + w3 * G(row, col - 2) + w4 * G(row, col - 1) + w5 * G(row, col) + w4 * G(row, col + 1) + w3 * G(row, col + 2) + w2 * G(row, col + 3) + w2 * G(row, col + 4) +
                        w2 * G(row + 1, col - 2) + w3 * G(row + 1, col - 1) + w4 * G(row + 1, col) + w3 * G(row + 1, col + 1) + w2 * G(row + 1, col + 2) +
                        w1 * G(row + 2, col - 2) + w2 * G(row + 2, col - 1) + w3 * G(row + 2, col) + w2 * G(row + 2, col + 1) + w1 * G(row + 2, col + 2);
                v = (float)s;                                                   /* 873-878 */
            }
            ev[row * W + col] = v;
            if (v > topval) { topval = (int)v; id_row = row; id_col = col; }   /* 882-885 first wins */
        }
#undef G
    int longest = 0;                                                           /* 904-932 */
    for (int row = 0; row < H; row++) {
        int cur = 0;
        for (int col = 0; col < W; col++) {
            if (ev[row * W + col] == topval) {
                cur++;
                if (cur > longest) { longest = cur; id_row = row; id_col = col - cur / 2; }
            } else cur = 0;
        }
    }
    best[0] = id_row; best[1] = id_col; best[2] = topval;
}

/* ------------------------------------------------------------------ */
/* whole request (loop_control 335-402, transform_gp 1274-1421)        */
/* ------------------------------------------------------------------ */
static int hafo_threads(void)
{
    const char *t = getenv("HAFO_THREADS");
    int n = t ? atoi(t) : 1;
    return n < 1 ? 1 : n;
}

/* 4x4 inverse: Gauss-Jordan with partial pivoting in double, rounded to float (Eigen's order is unpinned) */
static int mat4_inverse(const float *M, float *inv)
{
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { a[i][j] = M[i * 4 + j]; a[i][4 + j] = (i == j); }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) return -1;
        if (p != c) for (int j = 0; j < 8; j++) { double t = a[p][j]; a[p][j] = a[c][j]; a[c][j] = t; }
        double d = a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] /= d;
        for (int r = 0; r < 4; r++) if (r != c) {
            double f = a[r][c];
            if (f != 0.0) for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) inv[i * 4 + j] = (float)a[i][4 + j];
    return 0;
}

static int hafo_threads(void);

static void roll_features_q4(const hafo_cfg *cfg, const hafo_features *ft, const float *ii, const unsigned char *mask,
                             double **q4_out, long *rows_out, int **cells_out)
{
    const int H = cfg->H, W = cfg->W, W1 = W + 1, nf = ft->n;
    long rows = 0;
    int *cells = (int *)malloc(sizeof(int) * (size_t)(H > 14 ? (H - 14) * (W - 14) : 1));
    for (int row = 0; row < H - 14; row++)                         /* server.cpp:637-643: row-major line order */
        for (int col = 0; col < W - 14; col++)
            if (mask[(row + 7) * W + (col + 7)]) cells[rows++] = (row + 7) * W + (col + 7);
    double *q4 = (double *)malloc(sizeof(double) * (size_t)(rows ? rows : 1) * (size_t)nf);
    /* one feature line per masked cell (645-653); lines are independent, so they may be produced in parallel */
#pragma omp parallel num_threads(hafo_threads())
    {
        float *vals = (float *)malloc(sizeof(float) * (size_t)nf);
#pragma omp for schedule(dynamic, 16)
        for (long r = 0; r < rows; r++) {
            const int row = cells[r] / W - 7, col = cells[r] % W - 7;
            hafo_feature_values(ft, cfg->nshaf, ii + row * W1 + col, W1, vals);
            for (int k = 0; k < nf; k++)
                q4[(size_t)r * nf + k] = cfg->skip_text ? (double)vals[k] : hafo_q4(vals[k]);
        }
        free(vals);
    }
    *q4_out = q4; *rows_out = rows; *cells_out = cells;
}

/* svm-scale passes 1-2.5 for one roll's file: effective min/max tables (svm-scale.c:104-231) */
static void effective_ranges(const hafo_range *rg, const double *q4, long rows, int nf, double **fmin_o, double **fmax_o,
                             int *max_index_o)
{
    int max_index = rg->max_index;
    if (rows > 0 && nf > max_index) max_index = nf;
    double *fmin = (double *)malloc(sizeof(double) * ((size_t)max_index + 1));
    double *fmax = (double *)malloc(sizeof(double) * ((size_t)max_index + 1));
    for (int i = 0; i <= max_index; i++) { fmax[i] = -DBL_MAX; fmin[i] = DBL_MAX; }
    for (long r = 0; r < rows; r++) {
        for (int k = 1; k <= nf; k++) {                /* every row lists all nf attributes (fv.cpp:131-134) */
            double v = q4[(size_t)r * nf + (k - 1)];
            if (v > fmax[k]) fmax[k] = v;              /* max()/min() macros: 186-187 */
            if (v < fmin[k]) fmin[k] = v;
        }
        for (int i = nf + 1; i <= max_index; i++) {    /* 193-197 */
            if (0 > fmax[i]) fmax[i] = 0;
            if (0 < fmin[i]) fmin[i] = 0;
        }
    }
    for (int idx = 0; idx <= rg->max_index; idx++)     /* 221-228 */
        if (rg->present[idx] && idx <= max_index) { fmin[idx] = rg->fmin[idx]; fmax[idx] = rg->fmax[idx]; }
    *fmin_o = fmin; *fmax_o = fmax; *max_index_o = max_index;
}

/* test hook (haf_oracle.h): the roll loop of hafo_run starts here instead of at 0 */
static int g_roll_first = 0;
void hafo_set_roll_first(int roll) { g_roll_first = roll < 0 ? 0 : roll; }

int hafo_run(const hafo_cfg *cfg, const hafo_features *ft, const hafo_range *rg, const hafo_model *m,
             const float *xyz, size_t n, size_t stride, const hafo_input *in, hafo_output *out, hafo_debug *dbg)
{
    const int H = cfg->H, W = cfg->W, W1 = W + 1, nf = ft->n;
    if (H != W || H < 15 || cfg->n_rolls < 1) return -1;
    float *heights_all = (float *)malloc(sizeof(float) * (size_t)cfg->n_rolls * H * W);
    float *ii = (float *)malloc(sizeof(float) * (size_t)(H + 1) * W1);
    unsigned char *mask = (unsigned char *)malloc((size_t)H * W);
    signed char *grid = (signed char *)malloc((size_t)H * W);
    float *ev = (float *)malloc(sizeof(float) * (size_t)H * W);
    const int nx = nf > m->D ? nf : m->D;
    double *xs = (double *)malloc(sizeof(double) * (size_t)nx);
    const int gv0 = hafo_label_gridval(m->label[0]), gv1 = hafo_label_gridval(m->label[1]);
    if (cfg->probability && !m->has_prob) return -2;
    float *gridf = cfg->probability ? (float *)malloc(sizeof(float) * (size_t)H * W) : NULL;

    int o_row = -1, o_col = -1, o_roll = -1, o_top = -1000;       /* 322-326 */
    long n_evals = 0;
    int rolls_done = 0;
    float M_last[16]; mat4_identity(M_last);

    for (int roll = g_roll_first; roll < cfg->n_rolls; roll++) {   /* 345 (the reference always starts at 0: g_roll_first is a test hook) */
        if (in->show_only_best && o_top >= cfg->graspval_top) break;   /* 362-365 */
        float M[16];
        hafo_transform(cfg, in, roll, 0, M);
        memcpy(M_last, M, sizeof M);                               /* av_trans_mat, 484 */
        float *h = heights_all + (size_t)roll * H * W;
        hafo_height_grid(cfg, xyz, n, stride, M, h);
        hafo_integral(cfg, h, ii);
        hafo_mask(cfg, in, roll, ii, mask);

        double *q4; long rows; int *cells;
        roll_features_q4(cfg, ft, ii, mask, &q4, &rows, &cells);
        double *fmin, *fmax; int max_index;
        effective_ranges(rg, q4, rows, nf, &fmin, &fmax, &max_index);

        for (int i = 0; i < H * W; i++) grid[i] = -1;              /* 828-829 */
        if (dbg && dbg->dec) for (int i = 0; i < H * W; i++) dbg->dec[(size_t)roll * H * W + i] = NAN;
        double *row_prob = cfg->probability ? (double *)malloc(sizeof(double) * 2 * (size_t)(rows > 0 ? rows : 1)) : NULL;
        int *row_label = cfg->probability ? (int *)malloc(sizeof(int) * (size_t)(rows > 0 ? rows : 1)) : NULL;
        /* rows are independent: the only parallel region of the oracle (HAFO_THREADS, default 1), used by the
         * all-cores CPU baseline of bench.py; per-row arithmetic and its order are untouched */
#pragma omp parallel num_threads(hafo_threads())
        {
            double *xs_t = (double *)malloc(sizeof(double) * (size_t)nx);
#pragma omp for schedule(dynamic, 16)
            for (long r = 0; r < rows; r++) {
                scale_row(rg->lower, rg->upper, fmin, fmax, q4 + (size_t)r * nf, nf, cfg->skip_text, xs_t, nx);
                double sabs = 0;
                double dec = decision_nx2(m, xs_t, nx, &sabs);
                if (dbg && dbg->sabs) dbg->sabs[(size_t)roll * H * W + cells[r]] = sabs;
                int label = dec > 0 ? m->label[0] : m->label[1];       /* svm.cpp:2516-2531 */
                if (cfg->probability) {                                /* svm-predict -b 1: svm-predict.c:111-118 */
                    label = hafo_probability(m, dec, row_prob + 2 * r);
                    row_label[r] = label;
                }
                grid[cells[r]] = (signed char)(label == m->label[0] ? gv0 : gv1);
                if (dbg && dbg->dec) dbg->dec[(size_t)roll * H * W + cells[r]] = dec;
            }
            free(xs_t);
        }
        n_evals += rows;
        int best[3];
        if (cfg->probability) {
            /* show_predicted_gps 815-816, 824-848: ONE getline before the loops, so the k-th masked cell (row-major) is filled
             * from line k of the output file -- and line 0 is the "labels" header (svm-predict.c:60-64): every cell holds the
             * prediction of the masked cell BEFORE it, the first one what the header parses to, the last row is never read */
            char line[160];
            snprintf(line, sizeof line, "labels %d %d", m->label[0], m->label[1]);
            for (int i = 0; i < H * W; i++) gridf[i] = -1;             /* 828-829 */
            if (dbg && dbg->prob) for (int i = 0; i < 2 * H * W; i++) dbg->prob[(size_t)roll * 2 * H * W + i] = NAN;
            for (long r = 0; r < rows; r++) {                          /* cells[] is row-major (a4) */
                gridf[cells[r]] = hafo_probability_gridval(line);
                char t0[32], t1[32], tl[32];
                snprintf(tl, sizeof tl, "%g", (double)row_label[r]);   /* svm-predict.c:114-117 */
                snprintf(t0, sizeof t0, "%g", row_prob[2 * r]);
                snprintf(t1, sizeof t1, "%g", row_prob[2 * r + 1]);
                snprintf(line, sizeof line, "%s %s %s", tl, t0, t1);
                if (dbg && dbg->prob) {
                    dbg->prob[((size_t)roll * H * W + cells[r]) * 2] = strtod(t0, NULL);
                    dbg->prob[((size_t)roll * H * W + cells[r]) * 2 + 1] = strtod(t1, NULL);
                }
            }
            hafo_vote_f(cfg, gridf, ev, best);
            if (dbg && dbg->graspsgrid) memcpy(dbg->graspsgrid + (size_t)roll * H * W, gridf, sizeof(float) * (size_t)H * W);
        } else {
            hafo_vote(cfg, grid, ev, best);
        }
        free(q4); free(cells); free(fmin); free(fmax); free(row_prob); free(row_label);

        if (best[2] > o_top) { o_row = best[0]; o_col = best[1]; o_roll = roll; o_top = best[2]; }   /* 953-960 */
        rolls_done++;

        if (dbg) {
            if (dbg->heights) memcpy(dbg->heights + (size_t)roll * H * W, h, sizeof(float) * (size_t)H * W);
            if (dbg->integral) memcpy(dbg->integral + (size_t)roll * (H + 1) * W1, ii, sizeof(float) * (size_t)(H + 1) * W1);
            if (dbg->mask) memcpy(dbg->mask + (size_t)roll * H * W, mask, (size_t)H * W);
            if (dbg->labels) memcpy(dbg->labels + (size_t)roll * H * W, grid, (size_t)H * W);
            if (dbg->graspseval) memcpy(dbg->graspseval + (size_t)roll * H * W, ev, sizeof(float) * (size_t)H * W);
            if (dbg->roll_best) memcpy(dbg->roll_best + (size_t)roll * 3, best, sizeof best);
            if (dbg->M) memcpy(dbg->M + (size_t)roll * 16, M, sizeof M);
        }
    }

    /* transform_gp_in_wcs_and_publish(best..., topval-20)   390, 1274-1401 */
    memset(out, 0, sizeof(*out));
    out->row = o_row; out->col = o_col; out->roll_idx = o_roll; out->top = o_top;
    out->eval = o_top - 20;
    out->n_evals = n_evals; out->rolls_done = rolls_done;
    if (o_roll >= 0) {
        float Mb[16], Minv[16];
        hafo_transform(cfg, in, o_roll, 1, Mb);
        float x_gp_roll = -((float)(H / 2 - o_row)) / 100;              /* 1339-1340 */
        float y_gp_roll = -((float)(W / 2 - o_col)) / 100;
        float h_locmax = -10;
        const float *hb = heights_all + (size_t)o_roll * H * W;
        for (int rz = -4; rz < 5; rz++)                                   /* 1343-1351 */
            for (int cz = -4; cz < 4; cz++) {
                int rr = o_row + rz, cc = o_col + cz;
                if (rr >= 0 && cc >= 0 && rr < H && cc < W && h_locmax < hb[rr * W + cc]) h_locmax = hb[rr * W + cc];
            }
        h_locmax = (float)(h_locmax - 0.01);                              /* 1354 */
        float z_gp_roll = h_locmax;
        float x_gp_dis = 0.03f;                                           /* 1360 */
        float gp1[4] = {x_gp_roll - x_gp_dis, y_gp_roll, z_gp_roll, 1.0f};
        float gp2[4] = {x_gp_roll + x_gp_dis, y_gp_roll, z_gp_roll, 1.0f};
        if (mat4_inverse(Mb, Minv) == 0) {
            float g1[3], g2[3];
            for (int i = 0; i < 3; i++) {                                 /* 1367-1368 */
                float s = Minv[i * 4] * gp1[0]; s = s + Minv[i * 4 + 1] * gp1[1]; s = s + Minv[i * 4 + 2] * gp1[2]; s = s + Minv[i * 4 + 3] * gp1[3];
                g1[i] = s;
                s = Minv[i * 4] * gp2[0]; s = s + Minv[i * 4 + 1] * gp2[1]; s = s + Minv[i * 4 + 2] * gp2[2]; s = s + Minv[i * 4 + 3] * gp2[3];
                g2[i] = s;
            }
            for (int i = 0; i < 3; i++) {
                out->gp1[i] = g1[i]; out->gp2[i] = g2[i];
                out->avg[i] = (g1[i] + g2[i]) / 2.0;                      /* 1395-1397 */
            }
        }
        out->av[0] = M_last[8]; out->av[1] = M_last[9]; out->av[2] = M_last[10];   /* 1370-1374, 1398-1400 */
        out->roll = (float)((o_roll * cfg->roll_step_deg * HAFO_PI) / 180);       /* 1401 */
    }
    free(heights_all); free(ii); free(mask); free(grid); free(ev); free(xs); free(gridf);
    return 0;
}

long hafo_dump_feature_file(const hafo_cfg *cfg, const hafo_features *ft, const float *xyz, size_t n, size_t stride,
                            const hafo_input *in, int roll, const char *path)
{
    const int H = cfg->H, W = cfg->W, W1 = W + 1, nf = ft->n;
    float *h = (float *)malloc(sizeof(float) * (size_t)H * W);
    float *ii = (float *)malloc(sizeof(float) * (size_t)(H + 1) * W1);
    unsigned char *mask = (unsigned char *)malloc((size_t)H * W);
    float *vals = (float *)malloc(sizeof(float) * (size_t)nf);
    char *line = (char *)malloc((size_t)nf * 32 + 16);
    float M[16];
    hafo_transform(cfg, in, roll, 0, M);
    hafo_height_grid(cfg, xyz, n, stride, M, h);
    hafo_integral(cfg, h, ii);
    hafo_mask(cfg, in, roll, ii, mask);
    FILE *fp = fopen(path, "w");
    long rows = -1;
    if (fp) {
        rows = 0;
        for (int row = 0; row < H - 14; row++)
            for (int col = 0; col < W - 14; col++) {
                if (!mask[(row + 7) * W + (col + 7)]) continue;
                hafo_feature_values(ft, cfg->nshaf, ii + row * W1 + col, W1, vals);
                int len = hafo_feature_line(vals, nf, line, (size_t)nf * 32 + 16);
                fwrite(line, 1, (size_t)len, fp);
                rows++;
            }
        fclose(fp);
    }
    free(h); free(ii); free(mask); free(vals); free(line);
    return rows;
}
