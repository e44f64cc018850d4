/*
 * sf.h — C ABI of the StaticFusion coupled odometry + static/dynamic segmentation solver.
 *
 * This is the drop-in boundary for the ONE hot path of raluca-scona/staticfusion that this
 * repository rebuilds for MI355X (gfx950): the public surface of `class StaticFusion`
 * (reference StaticFusion.h:66-189) as exercised by the three drivers
 * (reference StaticFusion-datasets.cpp:79-94,148-199).  The reference has no FFI layer; its
 * "operator API" is: write public members, call four methods, read public members.  Each entry
 * point below names the reference member / method it replaces.
 *
 * Conventions (identical to the reference, Eigen::MatrixXf):
 *   - every image is column-major float32, element (v,u) at  v + u*rows,  rows x cols
 *   - depth in metres, 0 = invalid; intensity in [0,1]
 *   - labels are int32, SF_NUM_CLUSTERS (24) = invalid pixel
 *   - 4x4 transforms are column-major float32 (Eigen::Matrix4f storage order)
 *   - twists are (vx, vy, vz, wx, wy, wz)
 *
 * One handle owns `batch` independent streams (independent RGB-D sequences).  The MI355X build
 * runs one workgroup per stream; streams never exchange data.  All calls on one handle are
 * issued on one HIP stream in call order; handles are independent (thread-safe across handles).
 *
 * Every function returns SF_OK (0) or a negative error code; sf_last_error() returns a
 * human-readable description of the last failure on the calling thread.
 *
 * Two libraries implement this ABI with different symbol prefixes:
 *   libsf_hip.so     sf_*    the product: hand-written HIP for gfx950  (staticfusion_amd/csrc)
 *   liboracle.so     sfo_*   the CPU oracle used ONLY by tests / smoke / bench cpu_baseline (oracle/)
 * The product never links, loads or calls the oracle.
 */
#ifndef SF_H_
#define SF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef SF_PREFIX
#define SF_PREFIX sf_
#endif
#define SF_CAT2(a, b) a##b
#define SF_CAT(a, b) SF_CAT2(a, b)
#define SF_FN(name) SF_CAT(SF_PREFIX, name)

#define SF_NUM_CLUSTERS 24 /* reference StaticFusion.h:61 NUM_CLUSTERS */
#define SF_MAX_LEVELS 8
#define SF_HISTORY 5   /* reference StaticFusion.h:96 bufferLength */
#define SF_MAX_OUTER 32 /* trace capacity: ctf_levels * max_iter_per_level must not exceed it */

enum {
    SF_OK = 0,
    SF_ERR_ARG = -1,     /* bad argument (null pointer, index out of range, unsupported size) */
    SF_ERR_DEVICE = -2,  /* HIP runtime error / no usable device */
    SF_ERR_STATE = -3,   /* call sequence error */
    SF_ERR_NOMEM = -4
};

/* status bits in sf_frame_stats.status */
enum {
    SF_STATUS_EIG_SKIPPED = 1, /* non-finite covariance: pose update skipped (reference FrontEnd.cpp:720-724) */
    SF_STATUS_EMPTY_LEVEL = 2, /* a level had no valid pixel (reference divides by zero there, FrontEnd.cpp:505-509) */
    SF_STATUS_SYNC_TIMEOUT = 4 /* SF_VARIANT_CLUSTER: a workgroup waited too long for the others of its stream (they were not all
                                  resident: something else occupied the GPU). The images of this frame (labels, b image) are
                                  not valid; the stream's solver state (T_odometry, twists, b, covariance, per-cluster
                                  residuals, pose ring) is that of its last good frame. STICKY: every further frame of the
                                  stream reports it and does nothing until sf_clear_sync_timeout; after that, restart the
                                  frame counter like at start-up (the 5-frame ring took the images of the failed frame) */
};

/* The reference's parameter set: public members written by the drivers
 * (reference FrontEnd.cpp:57-76 ctor defaults, StaticFusion-datasets.cpp:79-94 driver values). */
typedef struct sf_params {
    int32_t ctf_levels;          /* 0 = log2(cols/40)+2 as in FrontEnd.cpp:61; 2..8, or 1 with segmentation_enabled = 0
                                    (K-means clusters image level 1, KMeans.cpp:145) */
    int32_t max_iter_per_level;  /* FrontEnd.cpp:69 / datasets.cpp:83 */
    int32_t max_iter_irls;       /* FrontEnd.cpp:68 / datasets.cpp:89 */
    int32_t use_motion_filter;   /* FrontEnd.cpp:76 / datasets.cpp:79 */
    int32_t segmentation_enabled; /* 1 = reference behaviour. 0 = "everything static": b_segm held at 1
                                     (the commented alternative at FrontEnd.cpp:606-607), K-means and the
                                     b-solve skipped: pure 6-DoF Cauchy IRLS (BASELINE.json configs[1]) */
    int32_t debug_planes;        /* 1 = also keep the Warped/Inter planes of every level for sf_get_plane */
    float fovh;                  /* FrontEnd.cpp:57 (radians); fovv is unused by the solver */
    float k_photometric_res;     /* FrontEnd.cpp:66 */
    float irls_delta_threshold;  /* FrontEnd.cpp:67 */
    float previous_speed_const_weight; /* FrontEnd.cpp:70 */
    float previous_speed_eig_weight;   /* FrontEnd.cpp:71 */
    float kc_Cauchy;             /* FrontEnd.cpp:72 */
    float kb;                    /* FrontEnd.cpp:73; per-stream override: sf_set_kb */
    float kz;                    /* FrontEnd.cpp:74 */
    float lambda_reg;            /* uninitialised in the ctor; datasets.cpp:90 */
    float lambda_prior;          /* datasets.cpp:91 */
} sf_params;

/* One record per executed outer ("linearisation") iteration, reference FrontEnd.cpp:1094-1132. */
typedef struct sf_outer_trace {
    int32_t level;      /* 0 = coarsest (reference `level`) */
    int32_t k;          /* iteration within the level */
    int32_t n_valid;    /* validPixels.size() */
    int32_t irls_iters; /* IRLS loop bodies executed (FrontEnd.cpp:611-684) */
    float aver_res;     /* after the last IRLS iteration */
    float var[6];       /* last IRLS solution, before the motion filter */
    float twist_level[6]; /* twist_level_odometry */
    float b_segm[SF_NUM_CLUSTERS];
    float T[16];        /* T_odometry after this iteration, column-major */
    float b_prior[SF_NUM_CLUSTERS];    /* computeSegPrior of this iteration (SegmentationBackground.cpp:53-103) */
    float lambda_t_w[SF_NUM_CLUSTERS];
    float AtA[36];      /* normal equations of the LAST IRLS iteration (FrontEnd.cpp:640-641), row-major 6x6 */
    float AtB[6];
    float delta_sol_max; /* |Var - previous Var|_inf of the LAST IRLS iteration: what the stopping test compared with
                            irls_delta_threshold (FrontEnd.cpp:676-679) */
} sf_outer_trace;

typedef struct sf_frame_stats {
    int32_t n_outer;      /* outer iterations executed by the last sf_run_solver */
    int32_t n_irls;       /* IRLS loop bodies executed ("solver iterations") */
    int64_t pixel_iters;  /* sum of n_valid over IRLS iterations */
    int32_t kmeans_iters; /* Lloyd iterations executed (KMeans.cpp:167-228) */
    int32_t status;       /* SF_STATUS_* bits */
    sf_outer_trace outer[SF_MAX_OUTER];
} sf_frame_stats;

typedef struct sf_handle sf_handle;

/* plane selectors for sf_get_plane: which pyramid, which channel */
enum { SF_SET_NEW = 0, SF_SET_PRED = 1, SF_SET_WARPED = 2, SF_SET_INTER = 3 };
enum { SF_CH_DEPTH = 0, SF_CH_INTENSITY = 1, SF_CH_XX = 2, SF_CH_YY = 3 };
/* linearisation planes of the LAST executed outer iteration (sf_get_lin_plane) */
enum {
    SF_LIN_DCU = 0, SF_LIN_DCV, SF_LIN_DCT, SF_LIN_DDU, SF_LIN_DDV, SF_LIN_DDT,
    SF_LIN_WC, SF_LIN_WD, /* weights_c / weights_d AFTER division by their maximum */
    SF_LIN_NULL,          /* Null mask as 0/1 float */
    SF_LIN_COUNT
};

/* ---- lifetime --------------------------------------------------------------------------- */

/* Driver parameter values (StaticFusion-datasets.cpp:79-94) with kb = 1.5. */
void SF_FN(default_params)(sf_params *p);
/* Constructor defaults (FrontEnd.cpp:57-76); lambda_reg / lambda_prior set to the driver values. */
void SF_FN(ctor_params)(sf_params *p);

/* Replaces StaticFusion::StaticFusion(res_factor) (FrontEnd.cpp:52-181) minus the GUI / map objects.
 * rows x cols is the solver resolution (240 x 320 for res_factor 2). device = HIP ordinal.
 * Sizes: rows, cols >= 8; every pyramid level a multiple of 4 pixels; the coarsest level at least 3 x 3. K-means needs EVEN
 * rows and cols: the reference starts the full-resolution search of pixel (v, u) at labels_lowres(v/2, u/2) (KMeans.cpp:267),
 * outside its rows/2 x cols/2 matrix for an odd size -- undefined there; here sf_kmeans, and with segmentation_enabled
 * sf_run_solver / sf_process_frame(s), return SF_ERR_ARG on such a handle (pure odometry, the input stage, prediction and
 * the map take any size). */
int SF_FN(create)(const sf_params *p, int rows, int cols, int batch, int device, sf_handle **out);
/* The same with an explicit choice of the frame-kernel build. The reference has one code path;
 * the MI355X library carries several builds of the same algorithm that differ in how many lanes / CUs serve one
 * stream, and sf_create picks by batch size (SF_VARIANT_AUTO). Tests and benchmarks name the build they measure:
 *   SF_VARIANT_THROUGHPUT  one 256-thread workgroup per stream, four per CU            (thousands of streams)
 *   SF_VARIANT_LATENCY     one 1024-thread workgroup per stream, one per CU             (up to ~2 streams per CU)
 *   SF_VARIANT_CLUSTER     several 1024-thread workgroups (CUs) per stream              (one live camera, a few streams)
 * The CPU oracle accepts and ignores the argument. */
enum { SF_VARIANT_AUTO = 0, SF_VARIANT_THROUGHPUT = 1, SF_VARIANT_LATENCY = 2, SF_VARIANT_CLUSTER = 3 };
int SF_FN(create_ex)(const sf_params *p, int rows, int cols, int batch, int device, int variant, sf_handle **out);
/* Which build the handle runs: *variant = SF_VARIANT_*, *threads = workgroup size, *workgroups_per_stream (1 except
 * for SF_VARIANT_CLUSTER). Any pointer may be NULL. */
int SF_FN(get_variant)(const sf_handle *h, int *variant, int *threads, int *workgroups_per_stream);
/* How many workgroups the next sf_process_frame / sf_run_solver launch keeps resident: per compute unit and in total
 * (the launch grid). The throughput build carries its frame kernel twice -- 4 workgroups per CU at <= 128 registers for
 * the pure-odometry configuration, 5 per CU at <= 96 registers for the full solver (segmentation_enabled) -- and picks per
 * launch; everything else has one answer. The CPU oracle reports 1 / 1. Any pointer may be NULL. */
int SF_FN(get_resident_workgroups)(const sf_handle *h, int *per_cu, int *total);
void SF_FN(destroy)(sf_handle *h);
int SF_FN(set_params)(sf_handle *h, const sf_params *p);
int SF_FN(get_params)(const sf_handle *h, sf_params *p);
/* Replaces `staticFusion.kb = ...` per frame (StaticFusion-datasets.cpp:158,163). stream = -1: all. */
int SF_FN(set_kb)(sf_handle *h, int stream, float kb);
/* Use an externally owned hipStream_t for all work of this handle (NULL = the handle's own stream). */
int SF_FN(set_hip_stream)(sf_handle *h, void *hip_stream);
int SF_FN(synchronize)(sf_handle *h);
const char *SF_FN(last_error)(void);
/* "hip:gfx950" for the product, "cpu-oracle" for the oracle. */
const char *SF_FN(backend)(void);

/* ---- inputs: the members the drivers write ------------------------------------------------ */

/* depthCurrent / intensityCurrent (StaticFusion.h:90). Host pointers, rows*cols floats each. */
int SF_FN(set_current)(sf_handle *h, int stream, const float *depth, const float *intensity);
/* depthPrediction / intensityPrediction (StaticFusion.h:91). */
int SF_FN(set_prediction)(sf_handle *h, int stream, const float *depth, const float *intensity);
/* Same, for the whole batch from DEVICE-resident buffers laid out [batch][cols][rows]
 * (what an MI355X-resident producer such as a HIP renderer / loader hands over). */
int SF_FN(set_current_device)(sf_handle *h, const void *d_depth, const void *d_intensity);
int SF_FN(set_prediction_device)(sf_handle *h, const void *d_depth, const void *d_intensity);
/* Frame-to-frame replay of many sequences that are resident in HBM (the dataset drivers' loop without the map:
 * prediction := the previous frame, StaticFusion-imagesequenceassoc.cpp:105-108 applied every frame): for every stream b
 *   depthPrediction / intensityPrediction := depthCurrent / intensityCurrent
 *   depthCurrent / intensityCurrent       := frame frame_index[b] of the pools
 * in ONE pass over the images. pool_depth / pool_intensity are DEVICE buffers of frames laid out [frame][cols][rows]
 * (column-major float32, rows*cols each) holding pool_frames frames each, 16-byte aligned; frame_index is a HOST array
 * of `batch` frame numbers in [0, pool_frames), a negative entry leaves that stream untouched (any other value outside
 * the pool: SF_ERR_ARG, nothing is launched). Asynchronous on the handle's stream (the index array is copied before the
 * call returns). */
int SF_FN(advance_sequences_device)(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index,
                                    int pool_frames);
/* n_frames consecutive frames of every stream in ONE launch: the result of
 *     for (k = 0; k < n_frames; k++) sf_process_frame(h, im_count0 + k);
 * on the inputs as they are (bit for bit, every stream), without a barrier over the batch between the frames: the launch
 * hands out (frame, stream) pairs and a stream's frame k starts as soon as ITS frame k - 1 is done, so streams that need
 * few iterations run ahead of the slow ones instead of waiting for them at every frame (the tail of a launch per frame).
 * T_out: NULL, or host memory for n_frames * batch * 16 floats that receives T_odometry (column-major) of every stream
 * after every frame, [frame][stream][16]; with T_out the call returns when the launch has finished, without it the call
 * is asynchronous like sf_process_frame. The getters report the state after the last frame. SF_VARIANT_CLUSTER handles
 * run the frames one launch at a time (their workgroups meet inside a frame). 1 <= n_frames <= 4096 (SF_ERR_ARG otherwise).
 * Device memory beyond the handle's own: 64 * batch * n_frames bytes for the trajectory when T_out is given, nothing else
 * (the sequence form below adds 4 * batch * n_frames bytes for the index table, on the device and in pinned host memory,
 * and waits for the previous call's copy of that table before it overwrites the staging block). */
int SF_FN(process_frames)(sf_handle *h, int im_count0, int n_frames, float *T_out);
/* The same for sequences resident in HBM: frame k of stream b is preceded by that stream's step of
 * sf_advance_sequences_device (prediction := current, current := pool frame frame_index[k * batch + b]; a negative entry
 * leaves the images alone). frame_index: HOST array [n_frames][batch]. The replay loop of the dataset drivers
 * (StaticFusion-imagesequenceassoc.cpp:140-191 without the map) for `batch` sequences and n_frames frames, one launch.
 * The pools are READ IN PLACE while the launch runs (from its second frame on a stream reads level 0 of its current and of
 * its predicted image in the pool; the last frame leaves both in the handle's own buffers again): they must stay valid and
 * unchanged until the launch has finished (sf_synchronize, any getter, or T_out).
 * Should a stream's frame never finish (cannot happen; every wait is bounded), the later frames of that stream are skipped
 * with SF_STATUS_SYNC_TIMEOUT, their rows of T_out are NaN, and the stream's images are undefined until
 * sf_clear_sync_timeout + fresh images. */
int SF_FN(process_sequence_frames_device)(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index,
                                          int pool_frames, int im_count0, int n_frames, float *T_out);
/* Overlapped upload for PCIe-fed deployments: sf_upload_current_async starts copying depthCurrent / intensityCurrent
 * of the WHOLE batch (host buffers laid out [batch][cols][rows]; page-locked memory from sf_alloc_pinned makes the
 * copy truly asynchronous) into a staging block on a second HIP stream and returns at once -- the solver launches
 * already queued on the handle's stream keep running meanwhile. sf_commit_upload makes the handle's stream wait (on
 * the device) for the copy and moves the frames into place; the host buffers may be refilled after it returns AND
 * the copy has finished (sf_synchronize, or the next sf_upload_current_async, which orders itself after it). */
int SF_FN(upload_current_async)(sf_handle *h, const float *depth_batch, const float *intensity_batch);
int SF_FN(commit_upload)(sf_handle *h);
int SF_FN(alloc_pinned)(size_t bytes, void **out);
int SF_FN(free_pinned)(void *p);
/* The bootstrap `depthCurrent.swap(depthPrediction)` (StaticFusion-imagesequenceassoc.cpp:105-108):
 * prediction := current, for every stream. */
int SF_FN(current_to_prediction)(sf_handle *h);
/* State injection for buildSegmImage (SegmentationBackground.cpp:176-197): clusterAllocation[0]
 * (rows*cols int32, column-major), b_segm (24), perClusterAverageResidual (24). Any pointer may be NULL. */
int SF_FN(set_segm_state)(sf_handle *h, int stream, const int32_t *labels0, const float *b_segm, const float *cluster_res);
/* twist_odometry_old (carried motion-filter state, FrontEnd.cpp:1141-1144); rarely needed. */
int SF_FN(set_twist_old)(sf_handle *h, int stream, const float twist[6]);

/* ---- input stage: what the drivers run before the four methods ---- */

/* StaticFusion::loadImageFromSequenceAssoc minus the file decoding (FrontEnd.cpp:216-254): from a decoded
 * full-resolution frame -- color_full: full_rows x full_cols x 3 uint8, interleaved, row-major, channels in
 * the decoder's memory order (the reference applies 0.299 / 0.587 / 0.114 to bytes 0 / 1 / 2 of cv::imread's
 * output, :231-236); depth_full: full_rows x full_cols uint16, row-major, millimetres -- take every
 * res_factor-th pixel of the vertically flipped image (row full_rows - res_factor*v - 1, column res_factor*u):
 *   intensityCurrent(v,u) = 0.299 c0/255 + 0.587 c1/255 + 0.114 c2/255           (:232-236)
 *   depthCurrent(v,u)     = float(mm) * float(1/1000)                              (:243,249)
 *   depth_mm(v,u)         = mm                                  rows x cols uint16, row-major (:244,250)
 *   color_full(v,u)       = (c0, c1, c2)                        rows x cols x 3 uint8, row-major (:237)
 * full_rows / res_factor and full_cols / res_factor must equal the handle's rows x cols. Host pointers. */
int SF_FN(load_frame)(sf_handle *h, int stream, const uint8_t *color_full, const uint16_t *depth_full, int full_rows,
                      int full_cols, int res_factor);
/* Same for the whole batch from DEVICE-resident buffers [batch][full_rows][full_cols][3] / [batch][full_rows][full_cols]. */
int SF_FN(load_frame_device)(sf_handle *h, const void *d_color_full, const void *d_depth_full, int full_rows, int full_cols,
                             int res_factor);
/* gui->depthCutoff / Reconstruction::depthCutoff, metres (FrontEnd.cpp:168,174: depth_max = 4.5). */
int SF_FN(set_depth_cutoff)(sf_handle *h, float max_depth_m);
/* Reconstruction::getFilteredDepth(depth_mm, depthCurrent) (Reconstruction.cpp:722-732), all streams:
 * bilateral filter of depth_mm (Shaders/depth_bilateral.frag:34-74: 13x13 window clipped to the image,
 * range gate 300 mm .. cutoff, weight exp(-(space2*0.024691358 + color2*0.000555556)) evaluated with
 * sf_exp_neg() of sf_detmath.h, result round()ed to integer mm), then metricise (Shaders/depth_metric.frag:32-39:
 * 0 outside 300 mm .. cutoff, else mm / 1000.0f) into depthCurrent. Also fills the unfiltered DEPTH_METRIC
 * image the fusion consumes (Reconstruction.cpp:337-346). Needs a previous sf_load_frame*. */
int SF_FN(filter_depth)(sf_handle *h);
/* depthCurrent / intensityCurrent as the solver will see them (column-major float, rows*cols each; either may be NULL). */
int SF_FN(get_current)(sf_handle *h, int stream, float *depth, float *intensity);
/* Intermediate images of the input stage, row-major rows x cols: which = SF_IN_* ; out type per selector. */
enum {
    SF_IN_DEPTH_MM = 0,      /* uint16: depth_mm (FrontEnd.cpp:250) */
    SF_IN_DEPTH_FILTERED_MM, /* uint16: DEPTH_FILTERED, output of depth_bilateral.frag */
    SF_IN_DEPTH_METRIC,      /* float : DEPTH_METRIC, unfiltered, metres */
    SF_IN_COLOR,             /* uint8 x 3: color_full */
    SF_IN_COUNT
};
int SF_FN(get_input_image)(sf_handle *h, int stream, int which, void *out);
/* Wall time of `calls` back-to-back sf_load_frame_device + sf_filter_depth pairs (HIP events). */
int SF_FN(timed_input_stage)(sf_handle *h, const void *d_color_full, const void *d_depth_full, int full_rows, int full_cols,
                             int res_factor, int calls, float *elapsed_ms);

/* ---- frame-to-model prediction without OpenGL ---------------------- */

/* Uniforms of Reconstruction::getPredictedImages (Reconstruction.cpp:628-720). */
typedef struct sf_model_params {
    float cx, cy, fx, fy;       /* Intrinsics: fx = 0.5 cols / tan(fovh/2), fy = 0.5 rows / tan(fovv/2), cx = cols/2, cy = rows/2 (FrontEnd.cpp:57-63,165) */
    float max_depth;            /* maxDepthProcessed = 20 (Reconstruction.cpp:36): surfel cull + depth-buffer scale */
    float conf_low, conf_high;  /* 0.13 and confidenceThreshold = 0.25 (Reconstruction.cpp:630-632, FrontEnd.cpp:167) */
    int32_t time, max_time;     /* tick, tick (Reconstruction.cpp:642-643) */
    int32_t time_delta;         /* INT_MAX in the drivers (FrontEnd.cpp:176) */
    float extract_max_depth;    /* 4.5 (FillIn.cpp:277) */
} sf_model_params;
/* The values the reference uses for a handle of this resolution (fovh from the handle's parameters, fovv = 48.5 deg). */
int SF_FN(default_model_params)(const sf_handle *h, sf_model_params *p);

/* Reconstruction::getPredictedImages(depthPrediction, intensityPrediction) for one stream, from a surfel
 * buffer in the layout of the reference's global model (Shaders/Vertex.cpp:40: 3 x vec4 per surfel =
 * position.xyz + confidence | encoded colour, -, init time, last time | normal.xyz + radius; host pointer,
 * count x 12 floats) and the camera pose `pose` (currPose, 4x4 column-major):
 *   two point-sprite renderings of the model at confidence >= conf_low / conf_high
 *       (IndexMap::combinedPredict IndexMap.cpp:221-300, Shaders/splat.vert, combo_splat.frag: per-fragment
 *        ray / surfel-disc intersection, nearest surface wins, surfels in buffer order on ties),
 *   the density test of the low-confidence image (Resize 1/40 + Reconstruction::denseEnough :218-233),
 *   fill-in from the stream's DEPTH_FILTERED / WEIGHT (= b_segm_perpixel) / RGB images where b > 0.6
 *       (Shaders/fill_vertex.frag, fill_vertex_from_texture.frag, fill_rgb.frag; FillIn.cpp),
 *   depth = vertex.z where 0 < z <= extract_max_depth (Shaders/extract_depth.frag), intensity =
 *       0.299 r + 0.587 g + 0.114 b of the 8-bit prediction (Reconstruction.cpp:684-692).
 * The fill-in reads what the stream holds at the time of the call: as in the reference's frame loop
 * (StaticFusion-imagesequenceassoc.cpp:164-165,183) that is the PREVIOUS frame's filtered depth, colour
 * and b image -- call this before sf_load_frame of the new frame. Writes depthPrediction / intensityPrediction. */
int SF_FN(predict_from_model)(sf_handle *h, int stream, const float *surfels, int count, const float pose[16],
                              const sf_model_params *p);
/* The same with the surfel buffer already in HBM (where a HIP-resident map keeps it); asynchronous on the handle's stream. */
int SF_FN(predict_from_model_device)(sf_handle *h, int stream, const void *d_surfels, int count, const float pose[16],
                                     const sf_model_params *p);
/* GlobalModel::initialise (GlobalModel.cpp:200-258) = Reconstruction::computeFeedbackBuffers (Reconstruction.cpp:205-216,
 * Shaders/vertex_feedback.vert/.geom) + Shaders/init_unstable.vert: the surfel model of the first fused frame, from
 * what the stream holds after a solve -- RGB and DEPTH_METRIC of the frame loaded last (input stage), its filtered
 * depth (depthCurrent) and the b image (WEIGHT): one surfel per pixel with 0 < depth <= max_depth, in the reference's
 * point order (x outer, y inner): world position from the RAW depth, colour bytes encoded into one float, normal
 * (central differences) and radius from the FILTERED depth, confidence = round(255 b) / 255, init time 1, last time =
 * `time`. The raw and the filtered point lists are paired by emission index, as the two transform-feedback buffers
 * are (:212-224). surfels_out: room for rows*cols*12 floats (host); *count = surfels written. The result is what
 * sf_predict_from_model renders. */
int SF_FN(init_model_from_frame)(sf_handle *h, int stream, const float pose[16], const sf_model_params *p, int time,
                                 float *surfels_out, int *count);
/* Reconstruction::denseEnough of the handle's LAST prediction (Reconstruction.cpp:218-233: more than a quarter of the
 * 1/40-resolution samples of the low-confidence rendering drawn). Reconstruction::checkIfDenseEnough (:762-776) returns
 * exactly this about the previous frame's getPredictedImages: it re-renders the HIGH-confidence target but reads the
 * low-confidence one. *dense = 0 before the first prediction. */
int SF_FN(get_prediction_dense)(sf_handle *h, int *dense);
/* The same per stream: *dense = Reconstruction::denseEnough of the last prediction rendered INTO `stream` (a batched
 * sf_map_predict_frames serves many sequences: each one's kb switch, StaticFusion-imagesequenceassoc.cpp:157-163, needs its
 * own flag); 0 for a stream the handle's LAST prediction call did not render into (the flags of a call are kept until
 * the next one). sf_get_prediction_dense reports the first job of the last call. */
int SF_FN(get_prediction_dense_stream)(sf_handle *h, int stream, int *dense);
/* depthPrediction / intensityPrediction (column-major float, rows*cols each; either may be NULL). */
int SF_FN(get_prediction)(sf_handle *h, int stream, float *depth, float *intensity);

/* ---- the surfel map: GlobalModel + the fusion half of Reconstruction::fuseFrame ---- */

/* One map = one GlobalModel (GlobalModel.cpp) plus Reconstruction's currPose / tick bookkeeping, resident in HBM.
 * A map belongs to the handle it was created from (its device, its HIP stream, its resolution); any stream of the
 * handle can be fused into it. capacity = most surfels it can hold (0 -> the reference's MAX_VERTICES = 3072 x 3072,
 * GlobalModel.cpp:21-22; 2 x 48 B per surfel of HBM); like the reference's transform feedback, a fuse that would
 * exceed it truncates the map and returns SF_ERR_STATE. */
typedef struct sf_map sf_map;
int SF_FN(map_create)(sf_handle *h, int capacity, sf_map **out);
void SF_FN(map_destroy)(sf_map *m);
/* Reconstruction::fuseFrame (Reconstruction.cpp:235-325) for the frame stream `stream` holds (sf_load_frame +
 * sf_filter_depth are the uploads and filterDepth / metriciseDepth of :242-249; the b image of the stream's last solve
 * is the weightedImage). in_pose = T_odometry of the solve, 4x4 column-major (may be NULL on the first call only).
 *   first call (tick == 1): currPose *= in_pose; GlobalModel::initialise (= sf_init_model_from_frame)  :255-262
 *   later calls: lastPose = currPose; currPose *= in_pose; velocity weighting (sf_fusion_weighting in sf_detmath.h)  :264-282
 *     IndexMap::predictIndices (IndexMap.cpp:117-184; index_map.vert/.frag): 4x-oversampled index image of the model  :284
 *     GlobalModel::fuse (GlobalModel.cpp:322-492): data association of every (x,y)%2 == tick%2 pixel within a window of
 *       the index image (data.vert) -> merge into the associated surfel (update.vert: confidence-weighted mean of
 *       position / colour / normal / radius, log-odds confidence update) or a new unstable surfel (confidence 0.08
 *       where b > 0.5, else 0)                                                                          :286-298
 *     IndexMap::predictIndices on the merged model                                                     :300
 *     GlobalModel::clean (GlobalModel.cpp:494-601; copy_unstable.vert/.geom): drops merged duplicates, free-space
 *       violations and stale unstable surfels, appends the surviving new ones                          :302-311
 *   tick++                                                                                             :323
 * p: intrinsics, max_depth (maxDepthProcessed), conf_high (confidenceThreshold), time_delta; p->time / max_time are
 * ignored (the map's tick is used). */
int SF_FN(map_fuse_frame)(sf_handle *h, int stream, sf_map *m, const float *in_pose, float weight_multiplier, const sf_model_params *p);
/* Reconstruction::getPredictedImages at the map's currPose and tick: sf_predict_from_model_device on the map's buffer. */
int SF_FN(map_predict)(sf_handle *h, int stream, sf_map *m, const sf_model_params *p);
/* The two calls above for n (stream, map) pairs at once -- the many-sequences-per-GPU form: every kernel of the pass is
 * launched ONCE for the whole batch (grid.y = pair) and the results come back in one read. streams[q] is fused into /
 * predicted from maps[q]; in_poses = n x 16 floats (column-major 4x4 each; NULL only if every map is at tick 1). The maps
 * of a batch must be distinct (predict: the streams too). Results are identical to n single calls. On overflow the other
 * maps of the batch are still fused; the error names the first truncated entry. */
int SF_FN(map_fuse_frames)(sf_handle *h, int n, const int *streams, sf_map *const *maps, const float *in_poses, float weight_multiplier,
                           const sf_model_params *p);
int SF_FN(map_predict_frames)(sf_handle *h, int n, const int *streams, sf_map *const *maps, const sf_model_params *p);
/* lastCount(), tick, currPose and the counters of the last fuse: stats[0] points emitted by the data pass, [1] of those
 * associated with a model surfel, [2] distinct surfels merged, [3] surfels after clean. Any pointer may be NULL. */
int SF_FN(map_info)(sf_map *m, int *count, int *tick, float pose[16], int stats[4]);
/* GlobalModel::downloadMap (GlobalModel.cpp:608-636): the first min(count, max_count) surfels, 12 floats each. */
int SF_FN(map_download)(sf_map *m, float *surfels, int max_count);
/* Replace the map's state (tests, checkpoints, teacher forcing): count x 12 floats, currPose, tick. */
int SF_FN(map_upload)(sf_map *m, const float *surfels, int count, const float pose[16], int tick);
/* The index texture of the last predictIndices (4 rows x 4 cols uint32, row-major; 0 = empty). Debug / tests. */
int SF_FN(map_get_index_map)(sf_map *m, uint32_t *out);

/* ---- the four methods the drivers call (all streams of the batch) ------------------------- */

/* StaticFusion::createImagePyramid(bool old_im)  FrontEnd.cpp:256-391 */
int SF_FN(build_pyramid)(sf_handle *h, int old_im);
/* StaticFusion::kMeans3DCoord() + createClustersPyramidUsingKMeans()  KMeans.cpp:137-391
 * (public in the reference, called only from runSolver; needs the NEW pyramid) */
int SF_FN(kmeans)(sf_handle *h);
/* StaticFusion::runSolver(bool create_image_pyr)  FrontEnd.cpp:1071-1146 */
int SF_FN(run_solver)(sf_handle *h, int create_image_pyr);
/* ring[im_count % 5] = (depthCurrent, intensityCurrent, T_odometry)  StaticFusion-datasets.cpp:182-184 */
int SF_FN(push_history)(sf_handle *h, int im_count);
/* StaticFusion::computeResidualsAgainstPreviousImage(int index)  FrontEnd.cpp:896-1069 */
int SF_FN(residuals_vs_history)(sf_handle *h, int index);
/* StaticFusion::buildSegmImage()  SegmentationBackground.cpp:176-197 */
int SF_FN(build_segm_image)(sf_handle *h);
/* The drivers' per-frame sequence in one call (StaticFusion-datasets.cpp:171-184):
 * createImagePyramid(true); runSolver(true); if (im_count >= 5) computeResiduals(im_count);
 * buildSegmImage(); push_history(im_count). */
int SF_FN(process_frame)(sf_handle *h, int im_count);

/* ---- outputs: the members the drivers read ------------------------------------------------ */

int SF_FN(get_T)(sf_handle *h, int stream, float T[16]);           /* T_odometry */
int SF_FN(get_twist)(sf_handle *h, int stream, float twist[6]);    /* twist_odometry */
int SF_FN(get_twist_old)(sf_handle *h, int stream, float twist[6]);/* twist_odometry_old */
int SF_FN(get_b)(sf_handle *h, int stream, float b[SF_NUM_CLUSTERS]); /* b_segm */
int SF_FN(get_b_image)(sf_handle *h, int stream, float *out);      /* b_segm_perpixel, rows*cols */
int SF_FN(get_labels)(sf_handle *h, int stream, int level, int32_t *out); /* clusterAllocation[level] */
int SF_FN(get_kmeans)(sf_handle *h, int stream, float centres[3 * SF_NUM_CLUSTERS]); /* kmeans, col-major 3x24 */
int SF_FN(get_connectivity)(sf_handle *h, int stream, uint8_t conn[SF_NUM_CLUSTERS * SF_NUM_CLUSTERS]);
int SF_FN(get_cluster_residuals)(sf_handle *h, int stream, float r[SF_NUM_CLUSTERS]); /* perClusterAverageResidual */
int SF_FN(get_stats)(sf_handle *h, int stream, sf_frame_stats *out);
/* Batched readback of the pose + iteration counters (what a throughput harness needs):
 * T is [batch][16], n_irls / n_outer are [batch]; any pointer may be NULL. */
int SF_FN(get_batch_results)(sf_handle *h, float *T, int32_t *n_irls, int32_t *n_outer, int64_t *pixel_iters);

/* Pyramid planes (depthPyr / intensityPredPyr / xxWarpedPyr ...). WARPED and INTER sets hold the
 * planes of the last outer iteration executed at that level and need params.debug_planes = 1. */
int SF_FN(get_plane)(sf_handle *h, int stream, int set, int channel, int level, float *out);
/* dcu..ddt, weights, Null of the last executed outer iteration (size of that iteration's level). */
int SF_FN(get_lin_plane)(sf_handle *h, int stream, int which, float *out, int *rows, int *cols);

/* The Jacobian of the last executed outer iteration (FrontEnd.cpp:539-586): A is 2N x 6 row-major, B has 2N entries,
 * N = validPixels.size(); rows 2i / 2i+1 are the intensity / depth row of the i-th valid pixel in the reference's
 * validPixels order (u outer, v inner). *n_rows = 2N; A and B may be NULL to query the size. Needs
 * params.debug_planes = 1. The MI355X kernels never store A: a debug kernel expands the rows from the factored
 * per-pixel form the IRLS passes evaluate, so this is the device arithmetic of those passes. */
int SF_FN(get_jacobian_rows)(sf_handle *h, int stream, float *A, float *B, int *n_rows);

int SF_FN(level_rows)(const sf_handle *h, int level);
int SF_FN(level_cols)(const sf_handle *h, int level);
int SF_FN(batch)(const sf_handle *h);

/* ---- measurement support ------------------------------------------------------------------ */

/* Time sf_process_frame(im_count), sf_process_frame(im_count + 1), ... (`calls` of them) with HIP
 * events recorded on the handle's stream (the events bracket the whole region; inputs must already
 * be resident). Returns the elapsed milliseconds of the region. */
int SF_FN(timed_process_frames)(sf_handle *h, int im_count, int calls, float *elapsed_ms);
/* Totals since sf_create over all streams (accumulated on the device, read without disturbing a
 * sequence of launches): frames solved, IRLS loop bodies, outer iterations, valid-pixel IRLS iterations. */
int SF_FN(get_counters)(sf_handle *h, int64_t *frames, int64_t *n_irls, int64_t *n_outer, int64_t *pixel_iters);
/* In-kernel stage timers: ticks (100 MHz wall clock, lane 0 of each workgroup) summed over all
 * streams since sf_create. Slots: 0 pyramid(old) 1 pyramid(new) 2 k-means 3 warp 4 linearise
 * 5 IRLS setup 6 IRLS pass 1 7 6x6 solve 8 IRLS pass 2 9 b-solve/convergence 10 filter/update
 * 11 residuals-vs-history 12 segm image + history push 13 total; 14..20 K-means sub-stages
 * (init, centre sort, assignment, stable partition, sequential sums, level-0 labels, connectivity +
 * label pyramid); 21..23 belong to the profiling builds; 24 is a counter, not a timer: warp tiles that were replayed
 * because some of their targets fell outside the tile's accumulation window (one-workgroup builds); 25 a counter:
 * levels whose ordered tile splat gave up and took the per-cell lists; 26 the shader clock's cycles over the intervals of
 * slot 13: 100 * [26] / [13] MHz is the clock the stream-frames ran at (it starts low at every launch and climbs for
 * several hundred ms: the package's power management, not the library). */
int SF_FN(get_stage_profile)(sf_handle *h, int64_t ticks[32]);
/* The IRLS streaming passes in isolation: `reps` executions of pass `which` (1 = weights + normal
 * equations, 2 = residuals + label sums) over the level-0 records of every stream left by the last
 * solve, one launch of sf_irls_pass_kernel. variant 0 = product code; 1 = loads only; 2 = no
 * accumulation (ablations); variant | (S << 8) splits every level into S pixel ranges walked by S different
 * workgroups (how fast the passes run when fewer streams are in flight and their records stay in the
 * Infinity Cache; an experiment, partial sums are not combined). Elapsed HIP-event milliseconds of the
 * launch. Not part of a solve. */
int SF_FN(microbench_pass)(sf_handle *h, int which, int variant, int reps, float *elapsed_ms);
/* Forget a timeout (SF_STATUS_SYNC_TIMEOUT) of every stream of the handle after the handle's stream has drained.
 * SF_VARIANT_CLUSTER: granules, epochs and the sticky flag of the rendezvous are reset; the solver state is left as it is.
 * Every build: a stream that a multi-frame launch gave up on gets back the image layout the host assumes (set its images
 * again before the next frame). */
int SF_FN(clear_sync_timeout)(sf_handle *h);
#ifdef SF_TESTING
/* Test support, declared only with -DSF_TESTING (SF_VARIANT_CLUSTER): from the next launch on, the workgroup of rank `rank` of
 * every stream idles stall_ms before its first stage -- a late workgroup, as a co-running kernel causes -- and every
 * rendezvous gives up after spin_limit polls (0 = the product's bound). rank < 0 switches it off. */
int SF_FN(debug_stall_rank)(sf_handle *h, int rank, float stall_ms, unsigned spin_limit);
#endif
/* The version of this header the LIBRARY was built from, and the sizes it assumes for what callers hand over: a binary built
 * against another header finds out before it passes a buffer that is too small (sf_get_stage_profile wrote 24 slots before
 * version 3 and writes 32 since; sf_outer_trace grew by delta_sol_max in version 3; sf_advance_sequences_device gained an
 * argument). Returns SF_ABI_VERSION; any pointer may be NULL. A caller checks
 *     sf_abi_version(&a, &b, &c) == SF_ABI_VERSION && a == sizeof(sf_params) && b == sizeof(sf_frame_stats) && c == 32. */
#define SF_ABI_VERSION 5 /* 5: sf_microbench_copy; sf_clear_sync_timeout acts on every build; pools of a sequence launch are read in place */
int SF_FN(abi_version)(int *sizeof_params, int *sizeof_frame_stats, int *stage_profile_slots);
/* Measurement support ("achieved / measured-peak copy bandwidth"): `reps` device-to-device copies of
 * `bytes` bytes (16-byte loads and stores, grid-stride, on the handle's stream; two scratch blocks of that size are allocated
 * and freed) -- what a plain streaming kernel reaches on THIS box at THIS moment; a copy moves 2 x bytes. Elapsed HIP-event
 * milliseconds of all repetitions. Not part of a solve. */
int SF_FN(microbench_copy)(sf_handle *h, size_t bytes, int reps, float *elapsed_ms);
/* Elapsed ms of the most recent solver kernel launch (HIP events around that launch; a launch of sf_process_frames
 * covers all its frames). */
int SF_FN(last_solver_kernel_ms)(sf_handle *h, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* SF_H_ */
