/*
 * hafgrasp.h -- C-ABI of the MI355X grasp-scoring engine (libhafgrasp.so).
 *
 * Drop-in boundary for the hot path of haf_grasping's CalcGraspPointsServer action server: the body of
 * CCalc_Grasppoints::loop_control() (reference src/calc_grasppoints_action_server.cpp:335-402) after
 * read_pc_cb() (250-329) has put the cloud into the base frame and before the result is published
 * (396-401).  Plain C types only; the caller owns every input/output buffer, the engine owns device
 * memory.  One engine handle may be used by one thread at a time (the reference runs one goal at a time on
 * actionlib's execute thread, server.cpp:182); several handles may coexist (no global state).
 *
 * Every entry point returns 0 on success or a negative HAF_E_* code; haf_last_error() gives the text.
 * The reference itself has no error convention on this path (system() failures are only logged,
 * server.cpp:778-796); a ROS shim maps a non-zero status to setAborted().
 *
 * The engine REQUIRES a HIP device.  There is no CPU fallback: haf_create() fails with HAF_E_DEVICE when
 * no gfx950 device is usable.
 */
#ifndef HAFGRASP_H_
#define HAFGRASP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAF_ABI_VERSION 2

enum {
    HAF_OK = 0,
    HAF_E_ARG = -1,        /* bad argument / unsupported configuration                         */
    HAF_E_IO = -2,         /* cannot read or parse Features.txt / range file / model file      */
    HAF_E_DEVICE = -3,     /* no usable HIP device, HIP runtime error                          */
    HAF_E_CAPACITY = -4,   /* request exceeds the capacity the engine was created with         */
    HAF_E_BUDGET = -5,     /* (not returned any more: a negative budget yields the reference's empty result, see haf_grasp_input) */
    HAF_E_INTERNAL = -6
};

/* Construction-time inputs: the four ROS params of the server (server.cpp:217-225) plus the reference's
 * compile-time constants generalised to fields (server.cpp:92-101, 202-214). */
typedef struct haf_config {
    const char *feature_file;        /* feature_file_path, default <pkg>/data/Features.txt (626-628)     */
    const char *range_file;          /* range_file_path, default <pkg>/data/range21062012_allfeatures (767-769) */
    const char *model_file;          /* svmmodel_file_path, default <pkg>/data/all_features.txt.scale.model (771-773) */
    int32_t nr_features_without_shaf;/* 302 (224)                                                         */
    int32_t grid_h, grid_w;          /* HEIGHT, WIDTH = 56 cells of 1 cm (92-93); must be equal (681-682)  */
    int32_t n_rolls;                 /* ROLL_MAX_DEGREE/ROLL_STEPS_DEGREE = 12 (101, 345)                 */
    int32_t roll_step_deg;           /* ROLL_STEPS_DEGREE = 15 (95)                                       */
    float   z_shift;                 /* trans_z_after_pc_transform = 0.15 (214)                           */
    int32_t graspval_top;            /* 119 (203): early-exit threshold with show_only_best_grasp         */
    int32_t device;                  /* HIP device ordinal                                                */
    int32_t max_clouds;              /* capacity: clouds per batch call                                   */
    int64_t max_points;              /* capacity: total points per batch call                             */
    uint32_t flags;                  /* HAF_FLAG_*                                                        */
    int32_t graspval_th;             /* 70 (202): a roll whose best vote exceeds it is published as its own hypothesis
                                        when show_only_best_grasp is off (962-969; haf_roll_pose)                          */
    int32_t max_rolls_per_call;      /* capacity: rolls per haf_score_rolls call; 0 = n_rolls.  A roll-sharded engine (one
                                        of N GPUs) only ever scores ceil(n_rolls / N) rolls at a time: its working buffers
                                        are sized for that, the roll geometry still uses the global roll index           */
} haf_config;

#define HAF_FLAG_KEEP_DEBUG 1u       /* keep per-roll intermediates for haf_debug_fetch()                 */
#define HAF_FLAG_PROFILE    2u       /* record HIP events per stage (haf_get_stage_ms)                    */
#define HAF_FLAG_FP32_MFMA  4u       /* RBF contraction as ONE fp32 MFMA pass for every evaluation: same labels, slowest   */
#define HAF_FLAG_SPLIT_F16  8u       /* three fp16 MFMA passes on the hi/lo halves of the fp32 operands for EVERY evaluation
                                        (fp32-grade decision values everywhere).  Default (neither flag): a single-pass fp16
                                        screening kernel decides every evaluation outside a rigorous guard band and only
                                        the rest goes through the three-pass kernel and the fp64 tiers: same labels, same
                                        grasps, about 2.5x the rate                                                       */

#define HAF_FLAG_PROBABILITY 16u     /* svm_with_probability (server.cpp:383 passes false; 791, 831-841): labels and cell values
                                        from "svm-predict -b 1" (svm_predict_probability, svm.cpp:2550-2587) as
                                        show_predicted_gps reads them -- each masked cell takes the prediction of the masked cell
                                        before it -- and the fp32 vote with an int topval.  Needs a model with probA/probB
                                        (svm-train -b 1).  Every decision value comes from the strict tier: complete, not fast */
#define HAF_FLAG_FULL_RANK  32u      /* the screening pass never runs in its low-rank form (haf_screen_low_rank): same labels; for A/B
                                      * measurements and for a deployment that prefers the ten-step kernels it has run so far          */

/* GraspInput (reference msg/GraspInput.msg:3-15) minus the cloud and the frame id: the cloud is passed
 * separately, already in the base frame (server.cpp:316). */
typedef struct haf_grasp_input {
    double  grasp_area_center[3];        /* geometry_msgs/Point, metres (258-260)                         */
    float   grasp_area_length_x;         /* "in m" in the .msg, used as integer cm incl. the +14 border    */
    float   grasp_area_length_y;         /*   (server.cpp:266-267 truncates to int; client.cpp:183-184)    */
    double  approach_vector[3];          /* normalised by the engine as server.cpp:270-273                 */
    double  max_calculation_time;        /* seconds (277), truncated to int like server.cpp:337.  The reference tests it
                                            at the START of every roll with time()'s 1 s resolution (367-374); this engine
                                            starts all rolls of a request together, so every roll sees 0 s elapsed and
                                            the budget only stops a request whose truncated value is negative: like the
                                            reference (break before roll 0, the goal still succeeds) the call returns HAF_OK
                                            with rolls_done = 0, best_roll = -1, eval = -1020 -- it never cuts the roll set
                                            of a request that has begun                                                   */
    int32_t show_only_best_grasp;        /* changes the result: early exit at >= graspval_top (362-365)    */
    int32_t threshold_grasp_evaluation;  /* carried for API parity; the reference server never reads it    */
    int32_t gripper_opening_width;       /* x-scale factor (281, 433)                                      */
} haf_grasp_input;

/* GraspOutput (reference msg/GraspOutput.msg:1-7) without the header, plus the grid-space winner. */
typedef struct haf_grasp_output {
    int32_t eval;                   /* best vote - 20; -20 = nothing found (390, 1388, 1418)              */
    double  grasp_point1[3];        /* 1389-1391 */
    double  grasp_point2[3];        /* 1392-1394 */
    double  averaged_grasp_point[3];/* 1395-1397 */
    double  approach_vector[3];     /* 1398-1400 */
    float   roll;                   /* radians (1401) */
    int32_t best_row, best_col, best_roll, best_vote;   /* id_row/col_top_overall, nr_roll_top_overall, topval_gp_overall */
    int32_t rolls_done;             /* rolls the sequential reference loop would have executed             */
    int64_t n_evals;                /* masked (cell, roll) pairs scored = SVM evaluations                  */
    int64_t n_rechecked;            /* evaluations re-done in fp64 (guard band of the fast contraction)     */
} haf_grasp_output;

/* One roll's outcome: what show_predicted_gps() leaves behind (server.cpp:865-932) plus the z estimate
 * transform_gp_in_wcs_and_publish() would read from that roll's height grid (1342-1351).  16 bytes: the
 * unit exchanged between GPUs when rolls are sharded. */
typedef struct haf_roll_record {
    int32_t vote;      /* topval_gp of the roll                                   */
    int16_t row, col;  /* after longest-run centring (904-932)                    */
    float   h_locmax;  /* max height in rows row-4..row+4, cols col-4..col+3      */
    int32_t n_evals;   /* masked cells of this roll                               */
} haf_roll_record;

typedef struct haf_cloud {
    const float *xyz;        /* x,y,z fp32 triples                                                        */
    size_t       n_points;
    size_t       stride_floats; /* 3 for packed xyz, 4 for pcl::PointXYZ                                  */
    int32_t      on_device;  /* 0: host memory (copied over PCIe inside the call); 1: HBM resident: the caller has
                                synchronised the stream that wrote it (the engine reads it on its own stream);
                                2: host memory inside a buffer registered with haf_register_host_cloud (packed xyz, stride 3):
                                the DMA engine reads it where it lies -- no staging copy on the host (a 1.2 MB cloud: 30 us
                                instead of 75); anything else about it as for 0                                        */
} haf_cloud;

typedef struct haf_engine haf_engine;

/* defaults of the reference: 56x56, 12 rolls of 15 deg, z_shift 0.15, nshaf 302, top 119 */
void haf_config_default(haf_config *cfg);
void haf_grasp_input_default(haf_grasp_input *in);   /* centre 0, 32x44, av (0,0,1), 50 s, width 1 (server.cpp:191-215) */

int  haf_create(const haf_config *cfg, haf_engine **out);
void haf_destroy(haf_engine *e);
const char *haf_last_error(const haf_engine *e);     /* e == NULL: error of the last failed haf_create in this thread */

/* Page-locks a host buffer the caller keeps reusing for its clouds (e.g. the PCL buffer of the action server's subscriber) so that
 * clouds inside it can be passed with on_device = 2.  The buffer must stay valid until haf_unregister_host_cloud or haf_destroy.
 * A cloud passed with on_device = 2 that does not lie inside a registered buffer is an error (HAF_E_ARG). */
int haf_register_host_cloud(haf_engine *e, const void *ptr, size_t bytes);
int haf_unregister_host_cloud(haf_engine *e, const void *ptr);

/* GraspInput -> GraspOutput for one cloud: replaces loop_control() (server.cpp:335-402). */
int haf_score(haf_engine *e, const haf_cloud *cloud, const haf_grasp_input *in, haf_grasp_output *out);
/* Batched clouds (BASELINE configs C4/C5): all clouds and all rolls go through the device together. */
int haf_score_batch(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in,
                    haf_grasp_output *out);

/* Roll-sharded form for multi-GPU: score rolls [roll_first, roll_first+roll_count) only and return their
 * records (records[c*roll_count + i]); no cross-roll rule applied.  Gather the records of all shards (one
 * all-gather of n_rolls*16 bytes per cloud) and call haf_finalize(). */
int haf_score_rolls(haf_engine *e, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in,
                    int32_t roll_first, int32_t roll_count, haf_roll_record *records);
/* Sequential cross-roll rule (strict '>' keeps the lowest roll, early exit at >= graspval_top when
 * show_only_best_grasp; server.cpp:362-365, 953-960) and the grasp pose (1274-1401) from n_rolls records. */
int haf_finalize(haf_engine *e, const haf_grasp_input *in, const haf_roll_record *records, haf_grasp_output *out);

/* One roll's own hypothesis: what show_predicted_gps() hands to transform_gp_in_wcs_and_publish() for that roll when
 * show_only_best_grasp is off and the roll's best vote exceeds graspval_th (server.cpp:962-969): pose from the roll's
 * record, eval = max(vote - 20, 10).  *published = 1 when the reference would publish it, 0 otherwise (out is filled
 * either way, with the clamped eval).  records = the n_rolls records of one cloud (haf_score_rolls / an all-gather). */
int haf_roll_pose(haf_engine *e, const haf_grasp_input *in, const haf_roll_record *records, int32_t roll,
                  haf_grasp_output *out, int32_t *published);

/* ---- several GPUs of one node in ONE process (csrc/multi.cpp) ---------------------------------------------------------
 * For a C++ host such as the action server: one engine, one host thread and one HIP stream per entry of devices[], one RCCL
 * communicator over the distinct devices (ncclCommInitAll), collectives over xGMI.  What is sharded is what the reference
 * leaves independent: the rolls of one request (the body of the roll loop, server.cpp:343-386) or the clouds of a batch.
 * A device may appear more than once in devices[] (several shards on one GPU share its rank); every device must then appear
 * the same number of times.  cfg->device is ignored; cfg->max_clouds / n_rolls are the capacity of the WHOLE handle. */
enum { HAF_SHARD_ROLLS = 0,    /* haf_score_sharded: rolls of one request split 5,5,5,5,4,4,4,4-style over the shards   */
       HAF_SHARD_CLOUDS = 1 }; /* haf_score_batch_sharded: cloud b of a batch goes to shard b % n                       */
typedef struct haf_multi haf_multi;
int  haf_create_multi(const haf_config *cfg, const int32_t *devices, int32_t n_devices, int32_t shard_mode, haf_multi **out);
void haf_destroy_multi(haf_multi *m);
const char *haf_multi_last_error(const haf_multi *m);   /* m == NULL: error of the last failed haf_create_multi in this thread */

/* GraspInput -> GraspOutput for one cloud with the rolls sharded: every shard scores its rolls (haf_score_rolls), ONE
 * ncclAllGather exchanges the 16-byte roll records so that every rank holds all n_rolls of them, then the sequential
 * cross-roll rule and the pose (haf_finalize; server.cpp:362-365, 953-960, 1274-1401).  Same result as haf_score.
 * A host cloud is copied to every GPU over that GPU's own PCIe link; a device-resident one (on_device = 1) must live on
 * devices[0] and reaches the other GPUs by one ncclBroadcast. */
int haf_score_sharded(haf_multi *m, const haf_cloud *cloud, const haf_grasp_input *in, haf_grasp_output *out);
/* Batch of host clouds, cloud b on shard b % n, no data-path exchange; ONE ncclAllReduce(max) of a packed 64-bit
 * (vote, cloud) key elects the best grasp of the batch: *best_cloud = its index (highest vote, then lowest index). */
int haf_score_batch_sharded(haf_multi *m, int32_t n_clouds, const haf_cloud *clouds, const haf_grasp_input *in,
                            haf_grasp_output *out, int32_t *best_cloud);
int haf_multi_info(const haf_multi *m, int32_t *n_shards, int32_t *n_ranks, int32_t *rccl_version);
/* The partition haf_create_multi would build for devices[], WITHOUT touching a device: per shard its RCCL rank (distinct devices in
 * order of first appearance) and its slot on that rank, and for HAF_SHARD_ROLLS its contiguous roll range (36 rolls over 8 shards:
 * 5,5,5,5,4,4,4,4).  Arrays of n_devices entries, any of them may be NULL.  Same argument checks, same error texts
 * (haf_multi_last_error(NULL)). */
int haf_multi_plan(const int32_t *devices, int32_t n_devices, int32_t shard_mode, int32_t n_rolls, int32_t *rank_of, int32_t *slot_of,
                   int32_t *roll_first, int32_t *roll_count, int32_t *n_ranks);
/* Host wall-clock of the parts of the last haf_score_sharded / haf_score_batch_sharded call: the whole call, the ncclBroadcast of a
 * device-resident cloud (0 for a host cloud), the collective (all-gather incl. every rank's copy of the records to the host, or the
 * all-reduce), and every shard's own haf_score_rolls / haf_score_batch (shard_ms: n_shards floats).  Any pointer may be NULL. */
int haf_multi_last_timing(const haf_multi *m, float *total_ms, float *bcast_us, float *collective_us, float *shard_ms);
haf_engine *haf_multi_engine(haf_multi *m, int32_t shard);      /* the shard's engine (stage timings, counters, roll grids) */
/* rank `rank`'s copy of the n_rolls gathered records of the last haf_score_sharded call (all ranks hold the same) */
int haf_multi_last_records(const haf_multi *m, int32_t rank, haf_roll_record *records);

/* Per-roll vote grid and mask of the LAST scored batch, for the marker grid the ROS shim publishes
 * (publish_grasp_grid, server.cpp:901-902, 979-1016).  eval_grid: H*W floats, mask: H*W bytes; either may be NULL.
 * (Integer-valued votes; with HAF_FLAG_PROBABILITY the fp32 votes of the probability branch.) */
int haf_get_roll_grid(haf_engine *e, int32_t cloud, int32_t roll, float *eval_grid, uint8_t *mask);

/* Intermediates of the last scored batch (needs HAF_FLAG_KEEP_DEBUG).  dst sizes per (cloud, roll):
 * HEIGHTS H*W f32, INTEGRAL (H+1)*(W+1) f32, MASK H*W u8, LABELS H*W i8 (-1 unmasked, else label text value),
 * DECISION H*W f64 (NaN unmasked), TRANSFORM 16 f32. */
enum { HAF_DBG_HEIGHTS = 0, HAF_DBG_INTEGRAL = 1, HAF_DBG_MASK = 2, HAF_DBG_LABELS = 3, HAF_DBG_DECISION = 4,
       HAF_DBG_TRANSFORM = 5,
       HAF_DBG_SCREEN_MARGIN = 6,    /* H*W f32, default mode: |dec^| / guard band for the cells the screening tier decided
                                        (> 1 by construction), 0 for the cells it handed on, NaN elsewhere */
       HAF_DBG_PROBABILITY = 7,      /* HAF_FLAG_PROBABILITY: H*W*2 f64, the two probabilities of the cell's own output line as
                                        atof reads them ("%g" text), NaN unmasked */
       HAF_DBG_GRASPSGRID = 8 };     /* HAF_FLAG_PROBABILITY: H*W f32, the grid show_predicted_gps builds (831-841) */
int haf_debug_fetch(haf_engine *e, int32_t what, int32_t cloud, int32_t roll, void *dst, size_t dst_bytes);

/* The attribute pipeline of the masked cells of one (cloud, roll) of the last scored batch, as the exact-form feature
 * kernels left it (HAF_FLAG_KEEP_DEBUG; engines of up to 2 GiB of records): per masked cell, in the row-major order of the
 * reference's feature file (server.cpp:637-643), 324 records of
 *   feature  the fp32 HAF/SHAF value (fv.cpp:141-199),
 *   q4       the double svm-scale reads back from its "%.4g" text (fv.cpp:133 -> svm-scale.c:270),
 *   scaled   the double svm-predict reads back from svm-scale's "%g" text (svm-scale.c:344-350 -> svm-predict.c:108);
 *            0 where svm-scale omits the attribute.
 * cells: row, col per masked cell; computed[i] = 1 when an exact-form feature kernel evaluated cell i in the last call
 * (every cell with HAF_FLAG_SPLIT_F16 / HAF_FLAG_FP32_MFMA; in the default mode only the cells the screening pass
 * handed on).  Returns the number of masked cells in *n_cells; fills at most max_cells entries. */
typedef struct haf_attr_record { float feature; float pad; double q4; double scaled; } haf_attr_record;
int haf_debug_fetch_attr(haf_engine *e, int32_t cloud, int32_t roll, int32_t max_cells, int32_t *cells /* [max_cells][2] */,
                         haf_attr_record *attr /* [max_cells][324] */, uint8_t *computed /* [max_cells] */, int32_t *n_cells);

/* Launch everything on this hipStream_t (default: a stream the engine creates).  The caller keeps ownership. */
int haf_set_stream(haf_engine *e, void *hip_stream);
void *haf_get_stream(haf_engine *e);

/* Stage timings of the last call in milliseconds (HAF_FLAG_PROFILE): HIP events on the engine's stream. */
enum { HAF_ST_UPLOAD = 0, HAF_ST_BIN, HAF_ST_INTEGRAL, HAF_ST_MASK, HAF_ST_FEATURES, HAF_ST_SVM, HAF_ST_REFINE,
       HAF_ST_RECHECK, HAF_ST_VOTE, HAF_ST_DOWNLOAD, HAF_ST_COUNT };   /* REFINE: three-pass kernel on the screened-out rest */
int haf_get_stage_ms(haf_engine *e, float *ms /* HAF_ST_COUNT */);

/* Counters of the last scored batch: masked (cell, roll) pairs; how many fell inside the guard band of the fast
 * contraction and were re-evaluated by the fp64 MFMA tier; how many of those were still too close to zero and were
 * re-evaluated in libsvm's strict fp64 summation order. */
int haf_last_counts(const haf_engine *e, int64_t *n_evals, int64_t *n_rechecked, int64_t *n_strict);

/* The same with the screening tier of the default mode: evaluations the single-pass screening kernel could not decide
 * (they went through the three-pass kernel; 0 in the other modes), then the fp64 MFMA tier, then the strict tier. */
int haf_last_tiers(const haf_engine *e, int64_t *n_evals, int64_t *n_refined, int64_t *n_rechecked, int64_t *n_strict);

/* The exact tiers of the last scored batch (both counted in n_rechecked of haf_last_tiers): evaluations that went through the
 * exact-integer tier (int8 digit planes on the matrix cores, no accumulation error; 0 when the model's support vectors do not
 * fit its fixed-point range), and those still inside its quantisation band that went on to the fp64 MFMA tier. */
int haf_last_exact_tiers(const haf_engine *e, int64_t *n_integer, int64_t *n_fp64);

/* Strict tier of the last scored batch: evaluations whose libsvm-order decision value lay within a last-bit exp error of zero
 * (2^-44 sum|coef|) and were therefore decided on the host with the C library's exp, the function svm-predict itself calls
 * (svm.cpp:364).  None in any run so far. */
int haf_last_strict_host(const haf_engine *e, int64_t *n_host);

/* Pre-stages of the last scored batch: (cloud, roll) grids whose integral image had to be summed in the reference's
 * sequential fp64 order because a parallel partial sum was not exact (normally 0; the result is bit-identical either way). */
int haf_last_prestage(const haf_engine *e, int64_t *n_inexact_grids);

/* Model facts for reporting: support vectors, attribute dimension, feature rows (incl. phantom rows). */
int haf_model_info(const haf_engine *e, int32_t *n_sv, int32_t *dim, int32_t *n_features);

/* Which form of the single-pass screening kernel serves this model in the default mode (chosen at haf_create on a synthetic scene,
 * re-chosen when a request leaves too much undecided): 0 plain, 1 with the measured |w|_2, 2 / 3 the centred-remainder form with the
 * exp / the polynomial epilogue (models with a large C, whose decisions are 1e-5..1e-8 of sum|coef|K); *active = 0 when no form can
 * decide enough and every evaluation takes the three-pass kernel.  Labels are identical in every case; for reporting only. */
int haf_screen_form(const haf_engine *e, int32_t *form, int32_t *active);

/* The low-rank form of the centred-remainder screening pass (round 4): the HAF attributes are linear functionals of the 15x15 window
 * (fv.cpp:141-199) spanning *rank dimensions (158 for the reference's Features.txt), so whole requests on large grids are swept on a
 * projected operand of rank + SHAF slots <= 192 instead of 320.  *available: the engine has the tables; *last_used: the last request's
 * screening pass ran in this form.  Labels are identical in every case; for reporting only. */
int haf_screen_low_rank(const haf_engine *e, int32_t *available, int32_t *rank, int32_t *last_used);

/* PCD v0.7 reader (ascii / binary / binary_compressed; pcl::io::loadPCDFile in client.cpp:141).
 * Returns a malloc'ed packed xyz array (free with haf_free) and the point count. */
int  haf_pcd_load(const char *path, float **xyz, size_t *n_points, char *err, size_t err_cap);
void haf_free(void *p);

int haf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HAFGRASP_H_ */
