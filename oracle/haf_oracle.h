/*
 * haf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, scalar, CPU restatement of the haf_grasping sliding-window hot path
 * (point cloud -> height grid -> integral image -> mask -> 324 HAF/SHAF
 * features -> "%.4g" text -> svm-scale -> "%g" text -> libsvm RBF decision ->
 * 29-tap vote / argmax / run-centring -> grasp pose).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (haf_grasping_amd/)
 * never links, loads or calls it.
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - stages a7/a8 (svm-scale restore path, svm-predict/svm_predict_values)
 *     are PINNED: tests run the real reference binaries built from
 *     /root/reference/libsvm-3.12 into oracle/_ref/ on the oracle's own
 *     feature text and require identical scaled text and identical labels,
 *     and the committed fixtures in tests/golden/ hold their outputs.
 *   - stages a1-a6, a10-a12 (server.cpp, CIntImage_to_Featurevec.cpp) are
 *     restated from source; the reference holds no tests or golden vectors
 *     for them and they cannot be compiled in this image (ROS / PCL / Eigen /
 *     OpenCV absent): PARITY UNPINNED for those stages.
 *
 * All "server.cpp" citations are
 * /root/reference/src/calc_grasppoints_action_server.cpp, "fv.cpp" is
 * /root/reference/src/CIntImage_to_Featurevec.cpp.
 */
#ifndef HAF_ORACLE_H_
#define HAF_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- parsed inputs -------------------------------------------------- */

typedef struct {
    int    n;        /* number of feature rows incl. phantom rows (fv.cpp:58-82) */
    int   *reg;      /* n*16 ints: 4 regions x (x1,x2,y1,y2)                     */
    float *w;        /* n*4 effective weights; w[3] is always 0 (CHaarFeature.cpp:56-60) */
} hafo_features;

typedef struct {
    double lower, upper;      /* svm-scale.c:220 */
    int    max_index;         /* largest idx listed */
    double *fmin, *fmax;      /* [max_index+1] */
    unsigned char *present;   /* [max_index+1] listed in the file */
} hafo_range;

typedef struct {
    int    svm_type, kernel_type; /* indices into libsvm's tables; 0 = c_svc, 1 = nu_svc; kernels 0 linear, 1 polynomial, 2 rbf, 3 sigmoid */
    double gamma, rho;
    int    nr_class, l;
    int    nSV[2], label[2];
    int    D;                 /* max attribute index seen in the SVs */
    double *coef;             /* [l]                                  */
    double *sv;               /* dense [l][D], attribute k at column k-1 */
    int    has_prob;          /* probA and probB both present (svm.cpp:2811-2824): svm_check_probability_model, 3098-3104 */
    double probA, probB;      /* the pair (0,1) sigmoid of a 2-class model */
    int    degree;            /* polynomial kernel (svm.cpp:2742-2743); 0 in files that do not carry it */
    double coef0;             /* polynomial / sigmoid kernel (2746-2747) */
} hafo_model;

hafo_features *hafo_features_load(const char *path);
void           hafo_features_free(hafo_features *f);
hafo_range    *hafo_range_load(const char *path);
void           hafo_range_free(hafo_range *r);
hafo_model    *hafo_model_load(const char *path);
void           hafo_model_free(hafo_model *m);

/* ---- configuration / request --------------------------------------- */

typedef struct {
    int   H, W;              /* grid cells, 1 cm (server.cpp:92-93); H == W required (681-682, 705) */
    int   n_rolls;           /* ROLL_MAX_DEGREE/ROLL_STEPS_DEGREE (101, 345) */
    int   roll_step_deg;     /* ROLL_STEPS_DEGREE (95) */
    float z_shift;           /* trans_z_after_pc_transform (214) */
    int   graspval_top;      /* 119 (203) */
    int   nshaf;             /* nr_features_without_shaf = 302 (224) */
    int   skip_text;         /* test knob: 0 = go through both decimal text round trips (reference behaviour) */
    int   probability;       /* svm_with_probability (server.cpp:383, 791, 831-841): "svm-predict -b 1" and the float vote */
} hafo_cfg;

typedef struct {
    double center[3];        /* grasp_area_center (msg/GraspInput.msg)       */
    float  length_x, length_y; /* cm incl. +14 border; truncated to int (266-267) */
    double approach[3];      /* normalised inside (270-273)                  */
    int    show_only_best;   /* 284, 362-365                                 */
    int    gripper_width;    /* 281, 433                                     */
} hafo_input;

typedef struct {
    int    eval;             /* topval_overall - 20 (390, 1388) */
    double gp1[3], gp2[3], avg[3], av[3];
    float  roll;             /* radians (1401) */
    int    row, col, roll_idx, top;
    long   n_evals;          /* masked cells over all executed rolls */
    int    rolls_done;
} hafo_output;

/* per-roll debug record, all arrays caller-allocated or NULL */
typedef struct {
    float         *heights;  /* [R][H][W]      */
    float         *integral; /* [R][H+1][W+1]  */
    unsigned char *mask;     /* [R][H][W]      */
    signed char   *labels;   /* [R][H][W] grid value: -1 unmasked, else atoi(label text) */
    double        *dec;      /* [R][H][W] decision value, NaN where unmasked */
    float         *graspseval; /* [R][H][W]    */
    int           *roll_best;  /* [R][3] row, col, val after run-centring */
    float         *M;          /* [R][16] row-major transform */
    double        *sabs;       /* [R][H][W] sum_n |coef_n| K_n: the cancellation scale of the decision value */
    double        *prob;       /* [R][H][W][2] probability mode: the two "%g" probabilities svm-predict -b 1 prints for the
                                  cell's own row, as strtod reads them back; NaN where unmasked */
    float         *graspsgrid; /* [R][H][W] probability mode: the value show_predicted_gps stores for the cell (831-841) */
} hafo_debug;

/* ---- stage functions (each cites the reference lines it follows) ---- */

/* Alternative evaluation orders of the third-party arithmetic the reference leaves unpinned (haf_oracle.c: g_variant).  0 = the
 * definition of record.  hafo_set_variant() is for tests/test_oracle.py only: it measures how many results change. */
enum { HAFO_V_EIGEN_TREE = 1,         /* 4x4 products: (a0b0 + a1b1) + (a2b2 + a3b3)           server.cpp:483, 1334 */
       HAFO_V_CHAIN_RTL = 2,          /* the six-matrix product associated from the right       server.cpp:483      */
       HAFO_V_PCL_SSE = 4,            /* point transform (m0 x + m1 y) + (m2 z + m3)            server.cpp:488      */
       HAFO_V_FMA = 8,                /* point transform with a*b + c contracted to fma         server.cpp:488      */
       HAFO_V_INTEGRAL_COLFIRST = 16  /* summed-area table by running column sums               server.cpp:595      */ };
void hafo_set_variant(int flags);
int hafo_get_variant(void);
/* Test hook: hafo_run scores rolls [first, cfg->n_rolls) only (the reference's loop, server.cpp:345, always starts at 0; its rolls
 * are independent, 376-385).  For tests that compare ONE complete roll of a large request with the engine without paying for the
 * rolls in front of it; the debug arrays stay indexed by the absolute roll.  0 = the reference's behaviour. */
void hafo_set_roll_first(int first);

void hafo_transform(const hafo_cfg *cfg, const hafo_input *in, int roll,
                    int use_double_atan2, float M[16]);
void hafo_height_grid(const hafo_cfg *cfg, const float *xyz, size_t n, size_t stride_floats,
                      const float M[16], float *heights /* [H][W] */);
void hafo_integral(const hafo_cfg *cfg, const float *heights, float *integral);
void hafo_mask(const hafo_cfg *cfg, const hafo_input *in, int roll, const float *integral,
               unsigned char *mask);
/* window: pointer to II[row][col] of the 15x15 window's top-left, row stride ld floats */
void hafo_feature_values(const hafo_features *ft, int nshaf, const float *window, int ld,
                         float *out /* ft->n */);
/* the exact text line fv.cpp:125-135 appends (without the label's uninitialised sign: always "-1") */
int  hafo_feature_line(const float *vals, int n, char *buf, size_t cap);
/* svm-scale restore path on one row of already "%.4g"-quantised values; xs is dense [D] zero-filled;
 * skip[k] (1-based, size n+1) marks attributes svm-scale drops (feature_max==feature_min) */
void hafo_scale_row(const hafo_range *rg, const unsigned char *skip, const double *q4, int n,
                    int skip_text, double *xs, int D);
double hafo_q4(float v);   /* float -> "%.4g" -> strtod   (fv.cpp:133, svm-scale.c:270) */
double hafo_q6(double v);  /* double -> "%g"  -> strtod   (svm-scale.c:350, svm-predict.c:108) */
double hafo_decision(const hafo_model *m, const double *xs /* dense [m->D] */);
int    hafo_label_gridval(int label); /* atoi(first two chars of "%g" of label) (svm-predict.c:127, server.cpp:843) */
/* svm_predict_probability (svm.cpp:2550-2587) of a 2-class model with probA/probB on a decision value: the two
 * probability estimates (sigmoid_predict 1818-1826, clamp to [1e-7, 1-1e-7], multiclass_probability 1829-1888 for k = 2)
 * and the label it returns (the first maximum).  Returns 0 when the model has no probability information. */
int    hafo_probability(const hafo_model *m, double dec, double prob[2]);
/* the value show_predicted_gps (server.cpp:831-841) extracts from one line of "svm-predict -b 1" output */
float  hafo_probability_gridval(const char *line);
/* a10 with float cell values (probability mode): server.cpp:865-932 on res*prob instead of labels */
void   hafo_vote_f(const hafo_cfg *cfg, const float *grid, float *graspseval, int best[3]);
void hafo_vote(const hafo_cfg *cfg, const signed char *grid, float *graspseval, int best[3]);

int hafo_run(const hafo_cfg *cfg, const hafo_features *ft, const hafo_range *rg, const hafo_model *m,
             const float *xyz, size_t n, size_t stride_floats, const hafo_input *in,
             hafo_output *out, hafo_debug *dbg);

/* writes the per-roll /tmp/features.txt equivalent (server.cpp:632-655) for `roll` of a request */
long hafo_dump_feature_file(const hafo_cfg *cfg, const hafo_features *ft, const float *xyz, size_t n,
                            size_t stride_floats, const hafo_input *in, int roll, const char *path);

/* standalone SVM stage on caller-provided scaled rows: CPU-baseline kernel for bench.py */
void hafo_decision_rows(const hafo_model *m, const double *xs, long rows, double *dec);

#ifdef __cplusplus
}
#endif
#endif
