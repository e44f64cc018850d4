// sf_oracle_capi.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
// Exposes the scalar restatement (sf_oracle.hpp) through the SAME C ABI as the product
// (include/sf.h) with the symbol prefix `sfo_`, so that tests drive both through one harness.
// PARITY UNPINNED — see sf_oracle.hpp.
#define SF_PREFIX sfo_
#include "../include/sf.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "sf_oracle.hpp"
#include "sf_oracle_input.hpp"
#include "sf_oracle_predict.hpp"
#include "sf_oracle_fusion.hpp"
#include "../include/sf_detmath.h"

struct sf_handle {
    int rows, cols, batch;
    sf_params params;
    std::vector<std::unique_ptr<sfo::StaticFusion>> s;
    long long cum_frames = 0, cum_irls = 0, cum_outer = 0, cum_pix = 0;
    float last_ms = 0.f;
    // input stage (per stream, row-major rows x cols)
    float depth_cutoff = 4.5f;  // FrontEnd.cpp:168
    bool have_frame = false;
    bool pred_dense = false;
    std::vector<char> pred_dense_stream;  // per stream: denseEnough of the prediction the last call rendered into it
    bool in_predict_batch = false;
    std::vector<std::vector<uint16_t>> depth_mm, filtered_mm;
    std::vector<std::vector<float>> depth_metric;
    std::vector<std::vector<uint8_t>> color;
};

static thread_local std::string g_err;
static int fail(int code, const char *msg) {
    g_err = msg;
    return code;
}

static sfo::Params to_oracle(const sf_params &p) {
    sfo::Params o;
    o.ctf_levels = p.ctf_levels;
    o.max_iter_per_level = p.max_iter_per_level;
    o.max_iter_irls = p.max_iter_irls;
    o.use_motion_filter = p.use_motion_filter != 0;
    o.segmentation_enabled = p.segmentation_enabled != 0;
    o.fovh = p.fovh;
    o.k_photometric_res = p.k_photometric_res;
    o.irls_delta_threshold = p.irls_delta_threshold;
    o.previous_speed_const_weight = p.previous_speed_const_weight;
    o.previous_speed_eig_weight = p.previous_speed_eig_weight;
    o.kc_Cauchy = p.kc_Cauchy;
    o.kb = p.kb;
    o.kz = p.kz;
    o.lambda_reg = p.lambda_reg;
    o.lambda_prior = p.lambda_prior;
    o.keep_rows = p.debug_planes != 0;
    return o;
}

extern "C" {

void sfo_ctor_params(sf_params *p) {  // FrontEnd.cpp:57-76
    std::memset(p, 0, sizeof(*p));
    p->ctf_levels = 0;
    p->max_iter_per_level = 2;
    p->max_iter_irls = 10;
    p->use_motion_filter = 0;
    p->segmentation_enabled = 1;
    p->debug_planes = 0;
    p->fovh = float(M_PI * 62.5 / 180.0);
    p->k_photometric_res = 0.15f;
    p->irls_delta_threshold = 1e-6f;
    p->previous_speed_const_weight = 0.05f;
    p->previous_speed_eig_weight = 0.5f;
    p->kc_Cauchy = 0.5f;
    p->kb = 1.25f;
    p->kz = 1.5f;
    p->lambda_reg = 0.35f;
    p->lambda_prior = 0.5f;
}

void sfo_default_params(sf_params *p) {  // StaticFusion-datasets.cpp:79-94
    sfo_ctor_params(p);
    p->use_motion_filter = 1;
    p->max_iter_per_level = 3;
    p->previous_speed_const_weight = 0.1f;
    p->previous_speed_eig_weight = 2.f;
    p->k_photometric_res = 0.15f;
    p->irls_delta_threshold = 0.0015f;
    p->max_iter_irls = 6;
    p->lambda_reg = 0.35f;
    p->lambda_prior = 0.5f;
    p->kc_Cauchy = 0.5f;
    p->kb = 1.5f;
    p->kz = 1.5f;
}

const char *sfo_last_error(void) { return g_err.c_str(); }
const char *sfo_backend(void) { return "cpu-oracle"; }
int sfo_abi_version(int *sizeof_params, int *sizeof_frame_stats, int *stage_profile_slots) {
    if (sizeof_params) *sizeof_params = (int)sizeof(sf_params);
    if (sizeof_frame_stats) *sizeof_frame_stats = (int)sizeof(sf_frame_stats);
    if (stage_profile_slots) *stage_profile_slots = 32;
    return SF_ABI_VERSION;
}

int sfo_create(const sf_params *p, int rows, int cols, int batch, int device, sf_handle **out) {
    (void)device;
    if (!p || !out || rows < 8 || cols < 8 || batch < 1) return fail(SF_ERR_ARG, "bad argument");
    auto *h = new sf_handle;
    h->rows = rows;
    h->cols = cols;
    h->batch = batch;
    h->params = *p;
    sfo::Params op = to_oracle(*p);
    for (int i = 0; i < batch; i++) h->s.emplace_back(new sfo::StaticFusion(rows, cols, op));
    h->params.ctf_levels = int(h->s[0]->ctf_levels);
    if (h->params.ctf_levels < (p->segmentation_enabled ? 2 : 1) || h->params.ctf_levels > SF_MAX_LEVELS ||
        h->params.ctf_levels * h->params.max_iter_per_level > SF_MAX_OUTER ||
        (rows >> (h->params.ctf_levels - 1)) < 3 || (cols >> (h->params.ctf_levels - 1)) < 3) {
        delete h;
        return fail(SF_ERR_ARG, "unsupported ctf_levels for this resolution");
    }
    *out = h;
    return SF_OK;
}

// the oracle has one code path: the build selector of the MI355X library is accepted and ignored
int sfo_create_ex(const sf_params *p, int rows, int cols, int batch, int device, int variant, sf_handle **out) {
    if (variant < SF_VARIANT_AUTO || variant > SF_VARIANT_CLUSTER) return fail(SF_ERR_ARG, "unknown variant");
    return sfo_create(p, rows, cols, batch, device, out);
}
int sfo_get_variant(const sf_handle *h, int *variant, int *threads, int *workgroups_per_stream) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (variant) *variant = SF_VARIANT_AUTO;
    if (threads) *threads = 1;
    if (workgroups_per_stream) *workgroups_per_stream = 1;
    return SF_OK;
}

int sfo_get_resident_workgroups(const sf_handle *h, int *per_cu, int *total) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (per_cu) *per_cu = 1;
    if (total) *total = 1;
    return SF_OK;
}

void sfo_destroy(sf_handle *h) { delete h; }

int sfo_set_params(sf_handle *h, const sf_params *p) {
    if (!h || !p) return fail(SF_ERR_ARG, "null");
    if (p->ctf_levels > 0 && p->ctf_levels > int(h->s[0]->pyr_levels_alloc))
        return fail(SF_ERR_ARG, "ctf_levels exceeds the allocated pyramid");
    const int keep = h->params.ctf_levels;
    h->params = *p;
    if (p->ctf_levels <= 0) h->params.ctf_levels = keep;
    sfo::Params op = to_oracle(h->params);
    for (auto &s : h->s) s->setParams(op);
    return SF_OK;
}
int sfo_get_params(const sf_handle *h, sf_params *p) {
    if (!h || !p) return fail(SF_ERR_ARG, "null");
    *p = h->params;
    return SF_OK;
}
int sfo_set_kb(sf_handle *h, int stream, float kb) {
    if (!h || stream < -1 || stream >= h->batch) return fail(SF_ERR_ARG, "bad stream");
    for (int i = 0; i < h->batch; i++)
        if (stream < 0 || stream == i) h->s[i]->kb = kb;
    return SF_OK;
}
int sfo_set_hip_stream(sf_handle *, void *) { return SF_OK; }
int sfo_synchronize(sf_handle *) { return SF_OK; }

static int check_stream(sf_handle *h, int stream) {
    if (!h) return fail(SF_ERR_ARG, "null handle");
    if (stream < 0 || stream >= h->batch) return fail(SF_ERR_ARG, "stream out of range");
    return SF_OK;
}

int sfo_set_current(sf_handle *h, int stream, const float *depth, const float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    if (!depth || !intensity) return fail(SF_ERR_ARG, "null image");
    auto &s = *h->s[stream];
    std::memcpy(s.depthCurrent.d.data(), depth, sizeof(float) * size_t(h->rows) * h->cols);
    std::memcpy(s.intensityCurrent.d.data(), intensity, sizeof(float) * size_t(h->rows) * h->cols);
    return SF_OK;
}
int sfo_set_prediction(sf_handle *h, int stream, const float *depth, const float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    if (!depth || !intensity) return fail(SF_ERR_ARG, "null image");
    auto &s = *h->s[stream];
    std::memcpy(s.depthPrediction.d.data(), depth, sizeof(float) * size_t(h->rows) * h->cols);
    std::memcpy(s.intensityPrediction.d.data(), intensity, sizeof(float) * size_t(h->rows) * h->cols);
    return SF_OK;
}
int sfo_set_current_device(sf_handle *h, const void *d, const void *i) {
    // "device" buffers of the CPU oracle are host buffers [batch][cols][rows]
    if (!h || !d || !i) return fail(SF_ERR_ARG, "null");
    const size_t n = size_t(h->rows) * h->cols;
    for (int b = 0; b < h->batch; b++) sfo_set_current(h, b, (const float *)d + b * n, (const float *)i + b * n);
    return SF_OK;
}
int sfo_set_prediction_device(sf_handle *h, const void *d, const void *i) {
    if (!h || !d || !i) return fail(SF_ERR_ARG, "null");
    const size_t n = size_t(h->rows) * h->cols;
    for (int b = 0; b < h->batch; b++) sfo_set_prediction(h, b, (const float *)d + b * n, (const float *)i + b * n);
    return SF_OK;
}
int sfo_advance_sequences_device(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index, int pool_frames) {
    // "device" pools of the CPU oracle are host buffers [frame][cols][rows]
    if (!h || !pool_depth || !pool_intensity || !frame_index) return fail(SF_ERR_ARG, "null");
    if (pool_frames < 1) return fail(SF_ERR_ARG, "pool_frames < 1");
    for (int b = 0; b < h->batch; b++)
        if (frame_index[b] >= pool_frames) return fail(SF_ERR_ARG, "frame_index entry outside the pool");
    const size_t n = size_t(h->rows) * h->cols;
    for (int b = 0; b < h->batch; b++) {
        if (frame_index[b] < 0) continue;
        auto &s = *h->s[b];
        s.depthPrediction = s.depthCurrent;
        s.intensityPrediction = s.intensityCurrent;
        if (int e = sfo_set_current(h, b, (const float *)pool_depth + size_t(frame_index[b]) * n, (const float *)pool_intensity + size_t(frame_index[b]) * n)) return e;
    }
    return SF_OK;
}
int sfo_process_frames(sf_handle *h, int im_count0, int n_frames, float *T_out) {
    if (!h || im_count0 < 0 || n_frames < 1) return fail(SF_ERR_ARG, "bad argument");
    for (int k = 0; k < n_frames; k++) {
        if (int e = sfo_process_frame(h, im_count0 + k)) return e;
        if (T_out)
            for (int b = 0; b < h->batch; b++) std::memcpy(T_out + (size_t(k) * h->batch + b) * 16, h->s[b]->T_odometry.m, 16 * sizeof(float));
    }
    return SF_OK;
}
int sfo_process_sequence_frames_device(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index,
                                       int pool_frames, int im_count0, int n_frames, float *T_out) {
    if (!h || !pool_depth || !pool_intensity || !frame_index || im_count0 < 0 || n_frames < 1) return fail(SF_ERR_ARG, "bad argument");
    for (int k = 0; k < n_frames; k++) {
        if (int e = sfo_advance_sequences_device(h, pool_depth, pool_intensity, frame_index + size_t(k) * h->batch, pool_frames)) return e;
        if (int e = sfo_process_frame(h, im_count0 + k)) return e;
        if (T_out)
            for (int b = 0; b < h->batch; b++) std::memcpy(T_out + (size_t(k) * h->batch + b) * 16, h->s[b]->T_odometry.m, 16 * sizeof(float));
    }
    return SF_OK;
}
// the CPU oracle has no second stream: the "asynchronous" upload copies at commit time from the caller's buffers
static const float *g_up_d = nullptr, *g_up_i = nullptr;
int sfo_upload_current_async(sf_handle *h, const float *d, const float *i) {
    if (!h || !d || !i) return fail(SF_ERR_ARG, "null");
    if (g_up_d) return fail(SF_ERR_STATE, "an upload is already pending: call sf_commit_upload first");
    g_up_d = d;
    g_up_i = i;
    return SF_OK;
}
int sfo_commit_upload(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (!g_up_d) return fail(SF_ERR_STATE, "no upload pending");
    const int e = sfo_set_current_device(h, g_up_d, g_up_i);
    g_up_d = g_up_i = nullptr;
    return e;
}
int sfo_alloc_pinned(size_t bytes, void **out) {
    if (!out) return fail(SF_ERR_ARG, "null");
    *out = std::malloc(bytes ? bytes : 1);
    return *out ? SF_OK : fail(SF_ERR_NOMEM, "malloc");
}
int sfo_free_pinned(void *p) {
    std::free(p);
    return SF_OK;
}
int sfo_current_to_prediction(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    for (auto &s : h->s) {
        s->depthPrediction = s->depthCurrent;
        s->intensityPrediction = s->intensityCurrent;
    }
    return SF_OK;
}
int sfo_set_segm_state(sf_handle *h, int stream, const int32_t *labels0, const float *b_segm, const float *cluster_res) {
    if (int e = check_stream(h, stream)) return e;
    auto &s = *h->s[stream];
    if (labels0) {
        for (size_t q = 0; q < s.clusterAllocation[0].size(); q++) {
            if (labels0[q] < 0 || labels0[q] > SF_NUM_CLUSTERS) return fail(SF_ERR_ARG, "label out of range");
            s.clusterAllocation[0].d[q] = labels0[q];
        }
    }
    if (b_segm) std::memcpy(s.b_segm, b_segm, SF_NUM_CLUSTERS * sizeof(float));
    if (cluster_res) std::memcpy(s.perClusterAverageResidual, cluster_res, SF_NUM_CLUSTERS * sizeof(float));
    return SF_OK;
}
int sfo_set_twist_old(sf_handle *h, int stream, const float twist[6]) {
    if (int e = check_stream(h, stream)) return e;
    for (int i = 0; i < 6; i++) h->s[stream]->twist_odometry_old[i] = twist[i];
    return SF_OK;
}

int sfo_build_pyramid(sf_handle *h, int old_im) {
    if (!h) return fail(SF_ERR_ARG, "null");
    for (auto &s : h->s) s->createImagePyramid(old_im != 0);
    return SF_OK;
}
// KMeans.cpp:267 starts the full-resolution search of pixel (v, u) at labels_lowres(v/2, u/2): outside the rows/2 x cols/2 matrix
// for an odd image size (the reference, without bounds checks, reads whatever lies there and the labels depend on it). Undefined in
// the reference: every call that runs kMeans3DCoord refuses such a handle, on both sides of the ABI.
static int kmeans_size_ok(const sf_handle *h) {
    if ((h->rows | h->cols) & 1) return fail(SF_ERR_ARG, "K-means with segmentation_enabled needs even rows and cols (the reference reads labels_lowres(v/2, u/2) outside its matrix otherwise, KMeans.cpp:267)");
    return SF_OK;
}
int sfo_kmeans(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (int e = kmeans_size_ok(h)) return e;
    for (auto &s : h->s) {
        s->stats.kmeans_iters = 0;
        s->kMeans3DCoord();
        s->createClustersPyramidUsingKMeans();
    }
    return SF_OK;
}
int sfo_run_solver(sf_handle *h, int create_image_pyr) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (h->params.segmentation_enabled)
        if (int e = kmeans_size_ok(h)) return e;
    auto t0 = std::chrono::steady_clock::now();
    for (auto &s : h->s) {
        s->runSolver(create_image_pyr != 0);
        h->cum_frames += 1;
        h->cum_irls += s->stats.n_irls;
        h->cum_outer += s->stats.n_outer;
        h->cum_pix += s->stats.pixel_iters;
    }
    h->last_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SF_OK;
}
int sfo_push_history(sf_handle *h, int im_count) {
    if (!h || im_count < 0) return fail(SF_ERR_ARG, "bad argument");
    for (auto &s : h->s) s->pushHistory(im_count);
    return SF_OK;
}
int sfo_residuals_vs_history(sf_handle *h, int index) {
    if (!h || index < SF_HISTORY) return fail(SF_ERR_ARG, "index must be >= 5");
    for (auto &s : h->s) s->computeResidualsAgainstPreviousImage(index);
    return SF_OK;
}
int sfo_build_segm_image(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    for (auto &s : h->s) s->buildSegmImage();
    return SF_OK;
}
int sfo_process_frame(sf_handle *h, int im_count) {  // StaticFusion-datasets.cpp:171-184
    if (!h || im_count < 0) return fail(SF_ERR_ARG, "bad argument");
    if (h->params.segmentation_enabled)
        if (int e = kmeans_size_ok(h)) return e;
    sfo_build_pyramid(h, 1);
    sfo_run_solver(h, 1);
    if (im_count - SF_HISTORY >= 0) sfo_residuals_vs_history(h, im_count);
    sfo_build_segm_image(h);
    sfo_push_history(h, im_count);
    return SF_OK;
}

int sfo_get_T(sf_handle *h, int stream, float T[16]) {
    if (int e = check_stream(h, stream)) return e;
    std::memcpy(T, h->s[stream]->T_odometry.m, 16 * sizeof(float));
    return SF_OK;
}
int sfo_get_twist(sf_handle *h, int stream, float t[6]) {
    if (int e = check_stream(h, stream)) return e;
    std::memcpy(t, h->s[stream]->twist_odometry, 6 * sizeof(float));
    return SF_OK;
}
int sfo_get_twist_old(sf_handle *h, int stream, float t[6]) {
    if (int e = check_stream(h, stream)) return e;
    std::memcpy(t, h->s[stream]->twist_odometry_old, 6 * sizeof(float));
    return SF_OK;
}
int sfo_get_b(sf_handle *h, int stream, float b[SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    std::memcpy(b, h->s[stream]->b_segm, SF_NUM_CLUSTERS * sizeof(float));
    return SF_OK;
}
int sfo_get_b_image(sf_handle *h, int stream, float *out) {
    if (int e = check_stream(h, stream)) return e;
    std::memcpy(out, h->s[stream]->b_segm_perpixel.d.data(), sizeof(float) * size_t(h->rows) * h->cols);
    return SF_OK;
}
int sfo_get_labels(sf_handle *h, int stream, int level, int32_t *out) {
    if (int e = check_stream(h, stream)) return e;
    if (level < 0 || level >= h->params.ctf_levels) return fail(SF_ERR_ARG, "bad level");
    const auto &m = h->s[stream]->clusterAllocation[level];
    std::memcpy(out, m.d.data(), sizeof(int32_t) * m.size());
    return SF_OK;
}
int sfo_get_kmeans(sf_handle *h, int stream, float c[3 * SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    std::memcpy(c, h->s[stream]->kmeans, 3 * SF_NUM_CLUSTERS * sizeof(float));
    return SF_OK;
}
int sfo_get_connectivity(sf_handle *h, int stream, uint8_t conn[SF_NUM_CLUSTERS * SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    for (int i = 0; i < SF_NUM_CLUSTERS; i++)
        for (int j = 0; j < SF_NUM_CLUSTERS; j++) conn[i * SF_NUM_CLUSTERS + j] = h->s[stream]->connectivity[i][j];
    return SF_OK;
}
int sfo_get_cluster_residuals(sf_handle *h, int stream, float r[SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    std::memcpy(r, h->s[stream]->perClusterAverageResidual, SF_NUM_CLUSTERS * sizeof(float));
    return SF_OK;
}
int sfo_get_stats(sf_handle *h, int stream, sf_frame_stats *out) {
    if (int e = check_stream(h, stream)) return e;
    const sfo::FrameStats &st = h->s[stream]->stats;
    std::memset(out, 0, sizeof(*out));
    out->n_outer = st.n_outer;
    out->n_irls = st.n_irls;
    out->pixel_iters = st.pixel_iters;
    out->kmeans_iters = st.kmeans_iters;
    out->status = st.status;
    for (int i = 0; i < st.n_outer && i < SF_MAX_OUTER; i++) {
        const sfo::OuterTrace &a = st.outer[i];
        sf_outer_trace &b = out->outer[i];
        b.level = a.level; b.k = a.k; b.n_valid = a.n_valid; b.irls_iters = a.irls_iters;
        b.aver_res = a.aver_res;
        b.delta_sol_max = a.delta_sol_max;
        std::memcpy(b.var, a.var, sizeof(b.var));
        std::memcpy(b.twist_level, a.twist_level, sizeof(b.twist_level));
        std::memcpy(b.b_segm, a.b_segm, sizeof(b.b_segm));
        std::memcpy(b.T, a.T, sizeof(b.T));
        std::memcpy(b.b_prior, a.b_prior, sizeof(b.b_prior));
        std::memcpy(b.lambda_t_w, a.lambda_t_w, sizeof(b.lambda_t_w));
        std::memcpy(b.AtA, a.AtA, sizeof(b.AtA));
        std::memcpy(b.AtB, a.AtB, sizeof(b.AtB));
    }
    return SF_OK;
}
int sfo_get_batch_results(sf_handle *h, float *T, int32_t *n_irls, int32_t *n_outer, int64_t *pixel_iters) {
    if (!h) return fail(SF_ERR_ARG, "null");
    for (int b = 0; b < h->batch; b++) {
        if (T) std::memcpy(T + 16 * b, h->s[b]->T_odometry.m, 16 * sizeof(float));
        if (n_irls) n_irls[b] = h->s[b]->stats.n_irls;
        if (n_outer) n_outer[b] = h->s[b]->stats.n_outer;
        if (pixel_iters) pixel_iters[b] = h->s[b]->stats.pixel_iters;
    }
    return SF_OK;
}

int sfo_get_plane(sf_handle *h, int stream, int set, int channel, int level, float *out) {
    if (int e = check_stream(h, stream)) return e;
    if (level < 0 || level >= h->params.ctf_levels || set < 0 || set > 3 || channel < 0 || channel > 3)
        return fail(SF_ERR_ARG, "bad selector");
    auto &s = *h->s[stream];
    std::vector<sfo::MatF> *tab[4][4] = {
        {&s.depthPyr, &s.intensityPyr, &s.xxPyr, &s.yyPyr},
        {&s.depthPredPyr, &s.intensityPredPyr, &s.xxPredPyr, &s.yyPredPyr},
        {&s.depthWarpedPyr, &s.intensityWarpedPyr, &s.xxWarpedPyr, &s.yyWarpedPyr},
        {&s.depthInterPyr, &s.intensityInterPyr, &s.xxInterPyr, &s.yyInterPyr}};
    const sfo::MatF &m = (*tab[set][channel])[level];
    std::memcpy(out, m.d.data(), sizeof(float) * m.size());
    return SF_OK;
}

int sfo_get_lin_plane(sf_handle *h, int stream, int which, float *out, int *rows, int *cols) {
    if (int e = check_stream(h, stream)) return e;
    if (which < 0 || which >= SF_LIN_COUNT) return fail(SF_ERR_ARG, "bad selector");
    auto &s = *h->s[stream];
    const int r = int(s.rows_i), c = int(s.cols_i);
    if (rows) *rows = r;
    if (cols) *cols = c;
    if (!out) return SF_OK;
    for (int u = 0; u < c; u++)
        for (int v = 0; v < r; v++) {
            float val = 0.f;
            const bool valid = !s.Null(v, u) && u != 0 && v != 0 && u != c - 1 && v != r - 1;
            switch (which) {
                // dcu/dcv/ddu/ddv hold stale values outside validPixels in the reference: report 0 there
                case SF_LIN_DCU: val = valid ? s.dcu(v, u) : 0.f; break;
                case SF_LIN_DCV: val = valid ? s.dcv(v, u) : 0.f; break;
                case SF_LIN_DCT: val = s.dct(v, u); break;
                case SF_LIN_DDU: val = valid ? s.ddu(v, u) : 0.f; break;
                case SF_LIN_DDV: val = valid ? s.ddv(v, u) : 0.f; break;
                case SF_LIN_DDT: val = s.ddt(v, u); break;
                case SF_LIN_WC: val = s.weights_c(v, u); break;
                case SF_LIN_WD: val = s.weights_d(v, u); break;
                case SF_LIN_NULL: val = s.Null(v, u) ? 1.f : 0.f; break;
            }
            out[v + size_t(u) * r] = val;
        }
    return SF_OK;
}

int sfo_get_jacobian_rows(sf_handle *h, int stream, float *A, float *B, int *n_rows) {
    if (int e = check_stream(h, stream)) return e;
    if (!n_rows) return fail(SF_ERR_ARG, "null");
    if (!h->params.debug_planes) return fail(SF_ERR_STATE, "the Jacobian rows need params.debug_planes = 1");
    auto &s = *h->s[stream];
    const size_t M = s.dbg_B.size();
    *n_rows = int(M);
    if (A)
        for (size_t r = 0; r < M; r++)
            for (int c = 0; c < 6; c++) A[r * 6 + c] = s.dbg_A[r + size_t(c) * M];  // the oracle keeps A column-major
    if (B) std::memcpy(B, s.dbg_B.data(), M * sizeof(float));
    return SF_OK;
}

int sfo_level_rows(const sf_handle *h, int level) { return h ? (h->rows >> level) : 0; }
int sfo_level_cols(const sf_handle *h, int level) { return h ? (h->cols >> level) : 0; }
int sfo_batch(const sf_handle *h) { return h ? h->batch : 0; }

// test hook (not part of include/sf.h): accumulate the reference's sequential fp32 per-cluster sums in fp64 instead
int sfo_test_set_exact_sums(sf_handle *h, int on) {
    if (!h) return fail(SF_ERR_ARG, "null");
    for (auto &s : h->s) s->exact_sums = on != 0;
    return SF_OK;
}
// test hooks (not part of include/sf.h): summation convention of AtA / AtB and the HIP build's behind-the-camera rule
// (sf_oracle.hpp: gemm_mode, hip_behind_camera_rule); the count of validPixels with a negative warped depth
int sfo_test_set_gemm_mode(sf_handle *h, int mode) {
    if (!h || mode < 0 || mode > 3) return fail(SF_ERR_ARG, "gemm mode 0..3");
    for (auto &s : h->s) s->gemm_mode = mode;
    return SF_OK;
}
int sfo_test_set_exact_warp(sf_handle *h, int on) {
    if (!h) return fail(SF_ERR_ARG, "null");
    for (auto &s : h->s) s->exact_warp = on != 0;
    return SF_OK;
}
int sfo_test_set_hip_behind_camera_rule(sf_handle *h, int on) {
    if (!h) return fail(SF_ERR_ARG, "null");
    for (auto &s : h->s) s->hip_behind_camera_rule = on != 0;
    return SF_OK;
}
long long sfo_test_behind_camera_valid(sf_handle *h, int stream) {
    if (!h || stream < 0 || stream >= h->batch) return -1;
    return h->s[stream]->behind_camera_valid;
}
// test hook (not part of include/sf.h): the weight function of include/sf_detmath.h, evaluated by this library
void sfo_test_exp_neg(const float *a, int n, float *out) {
    for (int i = 0; i < n; i++) out[i] = sf_exp_neg(a[i]);
}
void sfo_test_exp_det(const float *a, int n, float *out) {
    for (int i = 0; i < n; i++) out[i] = sf_exp_det(a[i]);
}
void sfo_test_log_det(const float *a, int n, float *out) {
    for (int i = 0; i < n; i++) out[i] = sf_log_det(a[i]);
}
float sfo_test_fusion_weighting(const float *last_pose, const float *curr_pose, float multiplier) {
    return sf_fusion_weighting(last_pose, curr_pose, multiplier);
}

// ---- input stage (sf_oracle_input.cpp) ----------------------------------------------------
static int input_alloc(sf_handle *h) {
    const size_t n = size_t(h->rows) * h->cols;
    if (h->depth_mm.size() == size_t(h->batch)) return SF_OK;
    h->depth_mm.assign(h->batch, std::vector<uint16_t>(n, 0));
    h->filtered_mm.assign(h->batch, std::vector<uint16_t>(n, 0));
    h->depth_metric.assign(h->batch, std::vector<float>(n, 0.f));
    h->color.assign(h->batch, std::vector<uint8_t>(n * 3, 0));
    return SF_OK;
}
static int check_full(sf_handle *h, int full_rows, int full_cols, int res) {
    if (res < 1 || full_rows != h->rows * res || full_cols != h->cols * res) return fail(SF_ERR_ARG, "full resolution / res_factor do not match the handle");
    return SF_OK;
}
int sfo_load_frame(sf_handle *h, int stream, const uint8_t *color_full, const uint16_t *depth_full, int full_rows, int full_cols,
                   int res_factor) {
    if (int e = check_stream(h, stream)) return e;
    if (!color_full || !depth_full) return fail(SF_ERR_ARG, "null image");
    if (int e = check_full(h, full_rows, full_cols, res_factor)) return e;
    input_alloc(h);
    auto &s = *h->s[stream];
    sfo::load_frame(color_full, depth_full, full_rows, full_cols, res_factor, h->rows, h->cols, s.depthCurrent.d.data(),
                    s.intensityCurrent.d.data(), h->depth_mm[stream].data(), h->color[stream].data());
    h->have_frame = true;
    return SF_OK;
}
int sfo_load_frame_device(sf_handle *h, const void *c, const void *d, int full_rows, int full_cols, int res_factor) {
    if (!h || !c || !d) return fail(SF_ERR_ARG, "null");
    const size_t n = size_t(full_rows) * full_cols;
    for (int b = 0; b < h->batch; b++)
        if (int e = sfo_load_frame(h, b, (const uint8_t *)c + b * n * 3, (const uint16_t *)d + b * n, full_rows, full_cols, res_factor)) return e;
    return SF_OK;
}
int sfo_set_depth_cutoff(sf_handle *h, float m) {
    if (!h || !(m > 0.f)) return fail(SF_ERR_ARG, "bad cutoff");
    h->depth_cutoff = m;
    return SF_OK;
}
int sfo_filter_depth(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_filter_depth needs sf_load_frame first");
    const int n = h->rows * h->cols;
    for (int b = 0; b < h->batch; b++) {
        sfo::bilateral_mm(h->depth_mm[b].data(), h->rows, h->cols, h->depth_cutoff, h->filtered_mm[b].data());  // filterDepth()
        sfo::metricise(h->depth_mm[b].data(), n, h->depth_cutoff, h->depth_metric[b].data());                    // METRIC
        std::vector<float> mf(n);
        sfo::metricise(h->filtered_mm[b].data(), n, h->depth_cutoff, mf.data());                                 // METRIC_FILTERED
        auto &dc = h->s[b]->depthCurrent;  // cv::cv2eigen (Reconstruction.cpp:730)
        for (int v = 0; v < h->rows; v++)
            for (int u = 0; u < h->cols; u++) dc.d[v + size_t(u) * h->rows] = mf[size_t(v) * h->cols + u];
    }
    return SF_OK;
}
int sfo_get_current(sf_handle *h, int stream, float *depth, float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    auto &s = *h->s[stream];
    const size_t bytes = sizeof(float) * size_t(h->rows) * h->cols;
    if (depth) std::memcpy(depth, s.depthCurrent.d.data(), bytes);
    if (intensity) std::memcpy(intensity, s.intensityCurrent.d.data(), bytes);
    return SF_OK;
}
int sfo_get_input_image(sf_handle *h, int stream, int which, void *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "no frame loaded");
    const size_t n = size_t(h->rows) * h->cols;
    switch (which) {
        case SF_IN_DEPTH_MM: std::memcpy(out, h->depth_mm[stream].data(), n * 2); return SF_OK;
        case SF_IN_DEPTH_FILTERED_MM: std::memcpy(out, h->filtered_mm[stream].data(), n * 2); return SF_OK;
        case SF_IN_DEPTH_METRIC: std::memcpy(out, h->depth_metric[stream].data(), n * 4); return SF_OK;
        case SF_IN_COLOR: std::memcpy(out, h->color[stream].data(), n * 3); return SF_OK;
        default: return fail(SF_ERR_ARG, "bad selector");
    }
}
int sfo_timed_input_stage(sf_handle *h, const void *c, const void *d, int full_rows, int full_cols, int res_factor, int calls,
                          float *elapsed_ms) {
    if (!h || calls < 1) return fail(SF_ERR_ARG, "bad argument");
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < calls; k++) {
        if (int e = sfo_load_frame_device(h, c, d, full_rows, full_cols, res_factor)) return e;
        if (int e = sfo_filter_depth(h)) return e;
    }
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SF_OK;
}

// ---- frame-to-model prediction (sf_oracle_predict.cpp) --------------------------------------
int sfo_default_model_params(const sf_handle *h, sf_model_params *p) {
    if (!h || !p) return fail(SF_ERR_ARG, "null");
    const float fovv = float(M_PI * 48.5 / 180.0);  // FrontEnd.cpp:58
    p->fx = float(0.5 * h->cols / std::tan(h->params.fovh * 0.5));  // :62 (double arithmetic, then float)
    p->fy = float(0.5 * h->rows / std::tan(fovv * 0.5));             // :63
    p->cx = float(h->cols / 2);                                       // :165 (integer division)
    p->cy = float(h->rows / 2);
    p->max_depth = 20.0f;
    p->conf_low = 0.13f;
    p->conf_high = 0.25f;
    p->time = p->max_time = 0;
    p->time_delta = 2147483647;
    p->extract_max_depth = 4.5f;
    return SF_OK;
}
int sfo_predict_from_model(sf_handle *h, int stream, const float *surfels, int count, const float pose[16], const sf_model_params *p) {
    if (int e = check_stream(h, stream)) return e;
    if ((!surfels && count > 0) || count < 0 || !pose || !p) return fail(SF_ERR_ARG, "bad argument");
    input_alloc(h);
    double A[16], Ai[16];  // t_inv = pose.inverse() (IndexMap.cpp:251), [C5]: double Gauss-Jordan, rounded to float
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) A[r * 4 + c] = double(pose[r + 4 * c]);
    sfo::inverse_double(A, Ai, 4);
    float t_inv[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) t_inv[r + 4 * c] = float(Ai[r * 4 + c]);
    sfo::ModelParams mp{p->cx, p->cy, p->fx, p->fy, p->max_depth, p->conf_low, p->conf_high, p->time, p->max_time, p->time_delta, p->extract_max_depth};
    auto &s = *h->s[stream];
    h->pred_dense = sfo::predict_from_model(surfels, count, t_inv, mp, h->rows, h->cols, h->filtered_mm[stream].data(), h->color[stream].data(),
                            s.b_segm_perpixel.d.data(), s.depthPrediction.d.data(), s.intensityPrediction.d.data());
    if (!h->in_predict_batch) h->pred_dense_stream.assign(size_t(h->batch), 0);  // the flags of a call are kept until the next one
    h->pred_dense_stream.resize(size_t(h->batch), 0);
    h->pred_dense_stream[size_t(stream)] = h->pred_dense ? 1 : 0;
    return SF_OK;
}
int sfo_predict_from_model_device(sf_handle *h, int stream, const void *s, int count, const float pose[16], const sf_model_params *p) {
    return sfo_predict_from_model(h, stream, (const float *)s, count, pose, p);
}
int sfo_init_model_from_frame(sf_handle *h, int stream, const float pose[16], const sf_model_params *p, int time, float *out, int *count) {
    if (int e = check_stream(h, stream)) return e;
    if (!pose || !p || !out || !count) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_init_model_from_frame needs a loaded frame (sf_load_frame + sf_filter_depth)");
    sfo::ModelParams mp{p->cx, p->cy, p->fx, p->fy, p->max_depth, p->conf_low, p->conf_high, p->time, p->max_time, p->time_delta, p->extract_max_depth};
    auto &s = *h->s[stream];
    *count = sfo::init_model_from_frame(h->depth_metric[stream].data(), s.depthCurrent.d.data(), h->color[stream].data(),
                                        s.b_segm_perpixel.d.data(), h->rows, h->cols, pose, mp, time, out);
    return SF_OK;
}
int sfo_get_prediction_dense(sf_handle *h, int *dense) {
    if (!h || !dense) return fail(SF_ERR_ARG, "null");
    *dense = h->pred_dense ? 1 : 0;
    return SF_OK;
}
int sfo_get_prediction_dense_stream(sf_handle *h, int stream, int *dense) {
    if (int e = check_stream(h, stream)) return e;
    if (!dense) return fail(SF_ERR_ARG, "null");
    *dense = (size_t(stream) < h->pred_dense_stream.size() && h->pred_dense_stream[size_t(stream)]) ? 1 : 0;
    return SF_OK;
}
int sfo_get_prediction(sf_handle *h, int stream, float *depth, float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    auto &s = *h->s[stream];
    const size_t bytes = sizeof(float) * size_t(h->rows) * h->cols;
    if (depth) std::memcpy(depth, s.depthPrediction.d.data(), bytes);
    if (intensity) std::memcpy(intensity, s.intensityPrediction.d.data(), bytes);
    return SF_OK;
}

// ---- the surfel map (sf_oracle_fusion.cpp) ----------------------------------------------------
struct sf_map {
    sf_handle *h;
    sfo::SurfelMap m;
};
static sfo::ModelParams model_params(const sf_model_params *p, int time) {
    return sfo::ModelParams{p->cx, p->cy, p->fx, p->fy, p->max_depth, p->conf_low, p->conf_high, time, time, p->time_delta, p->extract_max_depth};
}
int sfo_map_create(sf_handle *h, int capacity, sf_map **out) {
    if (!h || !out || capacity < 0) return fail(SF_ERR_ARG, "bad argument");
    if (capacity && capacity < h->rows * h->cols) return fail(SF_ERR_ARG, "capacity below rows * cols (the first frame alone can need that many surfels)");
    sf_map *m = new sf_map{h, {}};
    m->m.capacity = capacity ? capacity : 3072 * 3072;  // GlobalModel.cpp:21-22
    *out = m;
    return SF_OK;
}
void sfo_map_destroy(sf_map *m) { delete m; }
int sfo_map_fuse_frame(sf_handle *h, int stream, sf_map *m, const float *in_pose, float weight_multiplier, const sf_model_params *p) {
    if (int e = check_stream(h, stream)) return e;
    if (!m || m->h != h || !p) return fail(SF_ERR_ARG, "bad argument");
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_map_fuse_frame needs a loaded frame (sf_load_frame + sf_filter_depth)");
    if (!in_pose && m->m.tick != 1) return fail(SF_ERR_ARG, "in_pose may be NULL on the first fuse only");
    auto &s = *h->s[stream];
    const sfo::FrameImages f{h->depth_metric[stream].data(), s.depthCurrent.d.data(), h->color[stream].data(), s.b_segm_perpixel.d.data(), h->rows, h->cols};
    if (sfo::fuse_frame(m->m, f, in_pose, weight_multiplier, model_params(p, m->m.tick))) return fail(SF_ERR_STATE, "surfel map capacity exceeded (truncated)");
    return SF_OK;
}
int sfo_map_predict(sf_handle *h, int stream, sf_map *m, const sf_model_params *p) {
    if (!m || m->h != h || !p) return fail(SF_ERR_ARG, "bad argument");
    sf_model_params q = *p;
    q.time = q.max_time = m->m.tick;
    return sfo_predict_from_model(h, stream, m->m.surfels.data(), m->m.count, m->m.pose, &q);
}
int sfo_map_fuse_frames(sf_handle *h, int n, const int *streams, sf_map *const *maps, const float *in_poses, float weight_multiplier,
                        const sf_model_params *p) {
    if (!h || n < 0 || (n && (!streams || !maps)) || !p) return fail(SF_ERR_ARG, "bad argument");
    for (int q = 0; q < n; q++) {
        if (!maps[q] || maps[q]->h != h) return fail(SF_ERR_ARG, "a map belongs to the handle it was created from");
        if (!in_poses && maps[q]->m.tick != 1) return fail(SF_ERR_ARG, "in_pose may be NULL on the first fuse only");
        for (int r = 0; r < q; r++)
            if (maps[r] == maps[q]) return fail(SF_ERR_ARG, "the same map twice in one batch");
    }
    int first_error = SF_OK;
    for (int q = 0; q < n; q++) {
        const int e = sfo_map_fuse_frame(h, streams[q], maps[q], in_poses ? in_poses + size_t(q) * 16 : nullptr, weight_multiplier, p);
        if (e == SF_ERR_STATE && first_error == SF_OK) first_error = e;  // truncated: the rest of the batch is still fused
        else if (e != SF_OK && e != SF_ERR_STATE) return e;
    }
    return first_error == SF_OK ? SF_OK : fail(first_error, "surfel map capacity exceeded (truncated)");
}
int sfo_map_predict_frames(sf_handle *h, int n, const int *streams, sf_map *const *maps, const sf_model_params *p) {
    if (!h || n < 0 || (n && (!streams || !maps)) || !p) return fail(SF_ERR_ARG, "bad argument");
    for (int q = 0; q < n; q++)
        for (int r = 0; r < q; r++)
            if (streams[r] == streams[q]) return fail(SF_ERR_ARG, "the same stream twice in one batch (its prediction would be written twice)");
    h->pred_dense_stream.assign(size_t(h->batch), 0);
    h->in_predict_batch = true;
    int err = SF_OK;
    bool first_dense = false;
    for (int q = 0; q < n && !err; q++) {
        err = sfo_map_predict(h, streams[q], maps[q], p);
        if (q == 0) first_dense = h->pred_dense;
    }
    if (n > 0) h->pred_dense = first_dense;  // sf_get_prediction_dense: the first job of the last call (as the MI355X library)
    h->in_predict_batch = false;
    return err;
}
int sfo_map_info(sf_map *m, int *count, int *tick, float pose[16], int stats[4]) {
    if (!m) return fail(SF_ERR_ARG, "null");
    if (count) *count = m->m.count;
    if (tick) *tick = m->m.tick;
    if (pose) std::memcpy(pose, m->m.pose, sizeof m->m.pose);
    if (stats) std::memcpy(stats, m->m.stats, sizeof m->m.stats);
    return SF_OK;
}
int sfo_map_download(sf_map *m, float *surfels, int max_count) {
    if (!m || (!surfels && max_count > 0) || max_count < 0) return fail(SF_ERR_ARG, "bad argument");
    std::memcpy(surfels, m->m.surfels.data(), size_t(std::min(max_count, m->m.count)) * 12 * sizeof(float));
    return SF_OK;
}
int sfo_map_upload(sf_map *m, const float *surfels, int count, const float pose[16], int tick) {
    if (!m || (!surfels && count > 0) || count < 0 || !pose || tick < 1) return fail(SF_ERR_ARG, "bad argument");
    if (count > m->m.capacity) return fail(SF_ERR_ARG, "count exceeds the map's capacity");
    m->m.surfels.assign(surfels, surfels + size_t(count) * 12);
    m->m.count = count;
    std::memcpy(m->m.pose, pose, sizeof m->m.pose);
    m->m.tick = tick;
    return SF_OK;
}
int sfo_map_get_index_map(sf_map *m, uint32_t *out) {
    if (!m || !out) return fail(SF_ERR_ARG, "null");
    const size_t n = size_t(m->h->rows) * 4 * m->h->cols * 4;
    if (m->m.index_map.size() != n) return fail(SF_ERR_STATE, "no index map yet (sf_map_fuse_frame with tick > 1 renders it)");
    std::memcpy(out, m->m.index_map.data(), n * sizeof(uint32_t));
    return SF_OK;
}

int sfo_timed_process_frames(sf_handle *h, int im_count, int calls, float *elapsed_ms) {
    if (!h || calls < 1) return fail(SF_ERR_ARG, "bad argument");
    auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < calls; c++) sfo_process_frame(h, im_count + c);
    if (elapsed_ms)
        *elapsed_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SF_OK;
}
int sfo_get_counters(sf_handle *h, int64_t *frames, int64_t *n_irls, int64_t *n_outer, int64_t *pixel_iters) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (frames) *frames = h->cum_frames;
    if (n_irls) *n_irls = h->cum_irls;
    if (n_outer) *n_outer = h->cum_outer;
    if (pixel_iters) *pixel_iters = h->cum_pix;
    return SF_OK;
}
int sfo_get_stage_profile(sf_handle *h, int64_t ticks[32]) {
    if (!h || !ticks) return fail(SF_ERR_ARG, "null");
    for (int q = 0; q < 32; q++) ticks[q] = 0;  // the oracle keeps no stage timers
    return SF_OK;
}
int sfo_microbench_pass(sf_handle *, int, int, int, float *) { return fail(SF_ERR_STATE, "not available in the CPU oracle"); }
int sfo_microbench_copy(sf_handle *, size_t, int, float *) { return fail(SF_ERR_STATE, "not available in the CPU oracle"); }
int sfo_clear_sync_timeout(sf_handle *h) { return h ? SF_OK : fail(SF_ERR_ARG, "null"); }  // one thread: nothing to time out
int sfo_debug_stall_rank(sf_handle *, int, float, unsigned) { return fail(SF_ERR_STATE, "not available in the CPU oracle"); }
int sfo_last_solver_kernel_ms(sf_handle *h, float *ms) {
    if (!h || !ms) return fail(SF_ERR_ARG, "null");
    *ms = h->last_ms;
    return SF_OK;
}

}  // extern "C"
