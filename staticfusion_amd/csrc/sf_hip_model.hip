// sf_hip_model.hip — libsf_hip.so, the map side of include/sf.h without OpenGL: the
// frame-to-model prediction (sf_predict.h) and the surfel map -- index images, data association, fusion, cleaning (sf_fusion.h).
#include "sf_host.h"
#include "sf_predict.h"
#include "sf_fusion.h"

extern "C" {

// ---- frame-to-model prediction -------------------------------------------------------------------
int sf_default_model_params(const sf_handle *h, sf_model_params *p) {
    if (!h || !p) return fail(SF_ERR_ARG, "null");
    const float fovv = float(M_PI * 48.5 / 180.0);                      // FrontEnd.cpp:58
    p->fx = float(0.5 * h->k.cols / std::tan(h->k.p.fovh * 0.5));       // :62 (double arithmetic, then float)
    p->fy = float(0.5 * h->k.rows / std::tan(fovv * 0.5));              // :63
    p->cx = float(h->k.cols / 2);                                        // :165 (integer division)
    p->cy = float(h->k.rows / 2);
    p->max_depth = 20.0f;
    p->conf_low = 0.13f;
    p->conf_high = 0.25f;
    p->time = p->max_time = 0;
    p->time_delta = 2147483647;
    p->extract_max_depth = 4.5f;
    return SF_OK;
}
// 4x4 inverse, double Gauss-Jordan with partial pivoting, rounded to float (the [C5] convention of the solver)
static void invert_pose(const float pose[16], float out[16]) {
    double A[16], Ai[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            A[r * 4 + c] = double(pose[r + 4 * c]);
            Ai[r * 4 + c] = (r == c) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        double pv = std::fabs(A[c * 4 + c]);
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(A[r * 4 + c]) > pv) {
                pv = std::fabs(A[r * 4 + c]);
                piv = r;
            }
        if (piv != c)
            for (int j = 0; j < 4; j++) {
                std::swap(A[c * 4 + j], A[piv * 4 + j]);
                std::swap(Ai[c * 4 + j], Ai[piv * 4 + j]);
            }
        const double inv = 1.0 / A[c * 4 + c];
        for (int j = 0; j < 4; j++) {
            A[c * 4 + j] *= inv;
            Ai[c * 4 + j] *= inv;
        }
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            const double f = A[r * 4 + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 4; j++) {
                A[r * 4 + j] -= f * A[c * 4 + j];
                Ai[r * 4 + j] -= f * Ai[c * 4 + j];
            }
        }
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[r + 4 * c] = float(Ai[r * 4 + c]);
}
// device table for a batched launch: grows on demand, filled from a host copy that lives in the handle
static int upload_table(sf_handle *h, const void *src, size_t bytes, void **dev) {
    if (h->tab_bytes < bytes) {
        unsigned char *q = nullptr;
        if (int e = dev_alloc(h, &q, bytes * 2)) return e;  // the old block is freed with the handle
        h->tab_dev = q;
        h->tab_bytes = bytes * 2;
    }
    h->tab_host.assign((const unsigned char *)src, (const unsigned char *)src + bytes);
    HIP_TRY(hipMemcpyAsync(h->tab_dev, h->tab_host.data(), bytes, hipMemcpyHostToDevice, h->stream));
    *dev = h->tab_dev;
    return SF_OK;
}
static int predict_scratch(sf_handle *h, size_t n_maps) {
    if (h->pr_maps >= n_maps) return SF_OK;
    if (int e = dev_alloc(h, &h->pr_keys, n_maps * 2 * h->k.n0)) return e;
    if (int e = dev_alloc(h, &h->pr_dense, n_maps * 2)) return e;
    h->pr_maps = n_maps;
    return SF_OK;
}
static int results_scratch(sf_handle *h, size_t n_maps) {
    if (h->res_maps >= n_maps) return SF_OK;
    if (int e = dev_alloc(h, &h->res_dev, n_maps * 8)) return e;
    h->res_maps = n_maps;
    return SF_OK;
}
struct PredictJob {
    int stream;
    const float *d_surfels;
    int count;
    const float *pose;
    int time, max_time;
};
// Reconstruction::getPredictedImages for n (stream, surfel buffer, pose) triples in four launches
static int predict_batch(sf_handle *h, const std::vector<PredictJob> &jobs, const sf_model_params *p) {
    const size_t n = h->k.n0;
    if (!(p->conf_low <= p->conf_high)) return fail(SF_ERR_ARG, "conf_low must not exceed conf_high");
    if (jobs.empty()) return SF_OK;
    if (int e = predict_scratch(h, jobs.size())) return e;
    if (!h->pr_rays)
        if (int e = dev_alloc(h, &h->pr_rays, n)) return e;
    if (h->pr_rays_for[0] != p->cx || h->pr_rays_for[1] != p->cy || h->pr_rays_for[2] != p->fx || h->pr_rays_for[3] != p->fy) {
        hipLaunchKernelGGL(sf_predict_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->pr_rays, h->k.rows, h->k.cols, p->cx, p->cy,
                           p->fx, p->fy);
        h->pr_rays_for[0] = p->cx; h->pr_rays_for[1] = p->cy; h->pr_rays_for[2] = p->fx; h->pr_rays_for[3] = p->fy;
    }
    std::vector<PredictArgs> tab(jobs.size());
    int max_count = 0;
    for (size_t q = 0; q < jobs.size(); q++) {
        const PredictJob &j = jobs[q];
        PredictArgs &a = tab[q];
        a.surfels = j.d_surfels;
        a.count = j.count;
        max_count = std::max(max_count, j.count);
        invert_pose(j.pose, a.t_inv);  // t_inv = pose.inverse() (IndexMap.cpp:251)
        a.cx = p->cx; a.cy = p->cy; a.fx = p->fx; a.fy = p->fy;
        a.max_depth = p->max_depth; a.conf_low = p->conf_low; a.conf_high = p->conf_high; a.extract_max_depth = p->extract_max_depth;
        a.time = j.time; a.max_time = j.max_time; a.time_delta = p->time_delta;
        a.rows = h->k.rows; a.cols = h->k.cols;
        a.key_low = h->pr_keys + q * 2 * n; a.key_high = a.key_low + n; a.dense_count = h->pr_dense + q * 2;
        a.filtered_mm = h->in_filtered_mm + (size_t)j.stream * n;
        a.color = h->in_color + (size_t)j.stream * n * 3;
        a.b_img = h->k.b_img + (size_t)j.stream * n;
        a.depth_pred = h->k.pyr_pred[0] + (size_t)j.stream * h->k.n_tot;
        a.inten_pred = h->k.pyr_pred[1] + (size_t)j.stream * h->k.n_tot;
        a.rays = h->pr_rays;
    }
    void *dev = nullptr;
    if (int e = upload_table(h, tab.data(), tab.size() * sizeof(PredictArgs), &dev)) return e;
    const PredictArgs *d_tab = (const PredictArgs *)dev;
    const unsigned nm = (unsigned)jobs.size();
    const unsigned pix_blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(sf_predict_clear_kernel, dim3(pix_blocks, nm), dim3(256), 0, h->stream, d_tab);
    if (max_count) hipLaunchKernelGGL(sf_predict_splat_kernel, dim3((max_count + SF_SPLAT_NT - 1) / SF_SPLAT_NT, nm), dim3(SF_SPLAT_NT), 0, h->stream, d_tab);
    hipLaunchKernelGGL(sf_predict_dense_kernel, dim3(nm), dim3(64), 0, h->stream, d_tab);
    hipLaunchKernelGGL(sf_predict_resolve_kernel, dim3(pix_blocks, nm), dim3(256), 0, h->stream, d_tab);
    HIP_TRY(hipGetLastError());
    h->pr_rendered = true;
    // the density sums of this batch live in pr_dense[2 q] until the next prediction call: which job served which stream
    h->pr_job_of_stream.assign((size_t)h->k.batch, -1);
    for (size_t q = 0; q < jobs.size(); q++) h->pr_job_of_stream[(size_t)jobs[q].stream] = (int)q;
    return SF_OK;
}
static int predict_launch(sf_handle *h, int stream, const float *d_surfels, int count, const float pose[16], const sf_model_params *p) {
    return predict_batch(h, std::vector<PredictJob>{PredictJob{stream, d_surfels, count, pose, p->time, p->max_time}}, p);
}
int sf_predict_from_model(sf_handle *h, int stream, const float *surfels, int count, const float pose[16], const sf_model_params *p) {
    if (int e = check_stream(h, stream)) return e;
    if ((!surfels && count > 0) || count < 0 || !pose || !p) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    if (int e = dev_grow(h, &h->pr_surfels, &h->pr_floats, (size_t)count * 12)) return e;
    if (count) HIP_TRY(hipMemcpyAsync(h->pr_surfels, surfels, (size_t)count * 12 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (int e = predict_launch(h, stream, h->pr_surfels, count, pose, p)) return e;
    HIP_TRY(hipStreamSynchronize(h->stream));  // the host surfel buffer is free again; the staging block may be reused
    return SF_OK;
}
int sf_predict_from_model_device(sf_handle *h, int stream, const void *d_surfels, int count, const float pose[16],
                                 const sf_model_params *p) {
    if (int e = check_stream(h, stream)) return e;
    if ((!d_surfels && count > 0) || count < 0 || !pose || !p) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    return predict_launch(h, stream, (const float *)d_surfels, count, pose, p);
}
// GlobalModel::initialise for n maps: zero-fill (the feedback buffers start zero-filled), the two ordered compactions, trim
static void init_model_launch(sf_handle *h, const InitModelArgs *d_tab, int n_maps);
static int init_model_batch(sf_handle *h, const InitModelArgs *args, int n_maps) {
    void *dev = nullptr;
    if (int e = upload_table(h, args, (size_t)n_maps * sizeof(InitModelArgs), &dev)) return e;
    init_model_launch(h, (const InitModelArgs *)dev, n_maps);
    HIP_TRY(hipGetLastError());
    return SF_OK;
}
static void init_model_launch(sf_handle *h, const InitModelArgs *d_tab, int n_maps) {
    const unsigned blocks = (unsigned)((h->k.n0 * 12 + 255) / 256);
    hipLaunchKernelGGL(sf_init_model_zero_kernel, dim3(blocks, n_maps), dim3(256), 0, h->stream, d_tab);
    hipLaunchKernelGGL(sf_init_model_kernel, dim3(n_maps), dim3(1024), 0, h->stream, d_tab);
    hipLaunchKernelGGL(sf_init_model_trim_kernel, dim3(blocks, n_maps), dim3(256), 0, h->stream, d_tab);
}
int sf_init_model_from_frame(sf_handle *h, int stream, const float pose[16], const sf_model_params *p, int time, float *surfels_out,
                             int *count) {
    if (int e = check_stream(h, stream)) return e;
    if (!pose || !p || !surfels_out || !count) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_init_model_from_frame needs a loaded frame (sf_load_frame + sf_filter_depth)");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = h->k.n0;
    if (int e = dev_grow(h, &h->pr_surfels, &h->pr_floats, n * 12)) return e;
    if (int e = results_scratch(h, 1)) return e;
    InitModelArgs a;
    a.depth_metric = h->in_depth_metric + (size_t)stream * n;
    a.depth_filtered = h->k.pyr_new[0] + (size_t)stream * h->k.n_tot;
    a.color = h->in_color + (size_t)stream * n * 3;
    a.b_img = h->k.b_img + (size_t)stream * n;
    a.rows = h->k.rows; a.cols = h->k.cols; a.time = time;
    for (int q = 0; q < 16; q++) a.pose[q] = pose[q];
    a.cx = p->cx; a.cy = p->cy; a.fx = p->fx; a.fy = p->fy; a.max_depth = p->max_depth;
    a.out = h->pr_surfels;
    a.count = h->res_dev;
    if (int e = init_model_batch(h, &a, 1)) return e;
    int counts[2] = {0, 0};
    if (int e = d2h(h, counts, h->res_dev, sizeof counts)) return e;
    if (int e = d2h(h, surfels_out, h->pr_surfels, n * 12 * sizeof(float))) return e;
    *count = counts[0];
    return SF_OK;
}
int sf_get_prediction_dense(sf_handle *h, int *dense) {
    if (!h || !dense) return fail(SF_ERR_ARG, "null");
    *dense = 0;
    if (!h->pr_rendered) return SF_OK;  // nothing rendered yet
    int sum = 0;
    if (int e = d2h(h, &sum, h->pr_dense, sizeof sum)) return e;
    const int rw = h->k.cols / 40, rh = h->k.rows / 40;
    *dense = (rw * rh > 0) && (float(sum) / float(rh * rw) > 0.25f);
    return SF_OK;
}
int sf_get_prediction_dense_stream(sf_handle *h, int stream, int *dense) {
    if (int e = check_stream(h, stream)) return e;
    if (!dense) return fail(SF_ERR_ARG, "null");
    *dense = 0;
    if (!h->pr_rendered || h->pr_job_of_stream.empty()) return SF_OK;
    const int q = h->pr_job_of_stream[(size_t)stream];
    if (q < 0) return SF_OK;  // not part of the last prediction call
    int sum = 0;
    if (int e = d2h(h, &sum, h->pr_dense + (size_t)q * 2, sizeof sum)) return e;
    const int rw = h->k.cols / 40, rh = h->k.rows / 40;
    *dense = (rw * rh > 0) && (float(sum) / float(rh * rw) > 0.25f);
    return SF_OK;
}
int sf_get_prediction(sf_handle *h, int stream, float *depth, float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    const size_t bytes = sizeof(float) * h->k.n0;
    if (depth)
        if (int e = d2h(h, depth, h->k.pyr_pred[0] + (size_t)stream * h->k.n_tot, bytes)) return e;
    if (intensity)
        if (int e = d2h(h, intensity, h->k.pyr_pred[1] + (size_t)stream * h->k.n_tot, bytes)) return e;
    return SF_OK;
}

// ---- the surfel map (sf_fusion.h) ------------------------------------------------------------------
struct sf_map {
    sf_handle *h = nullptr;
    int capacity = 0;
    float *buf[2] = {nullptr, nullptr};  // the model lives in buf[0] between calls; buf[1] holds the merged model inside a fuse
    int count = 0, tick = 1;
    float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int stats[4] = {0, 0, 0, 0};
    unsigned long long *keys = nullptr, *occ = nullptr;
    unsigned *winner = nullptr, *meta = nullptr, *index_export = nullptr;
    float *rec = nullptr;
    unsigned char *flags = nullptr;
    int *block_counts = nullptr;
    bool have_index = false;
    int epoch = 0;  // index images rendered since the key image was last filled with ones; tag = 255 - epoch
    std::vector<void *> allocs;
};
static int map_alloc_bytes(sf_map *m, void **p, size_t bytes) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(SF_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    m->allocs.push_back(q);
    *p = q;
    return SF_OK;
}
#define map_alloc(m, p, count) map_alloc_bytes(m, (void **)(p), (size_t)(count) * sizeof(**(p)))
int sf_map_create(sf_handle *h, int capacity, sf_map **out) {
    if (!h || !out || capacity < 0) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n0 = h->k.n0;
    const size_t cap = capacity ? (size_t)capacity : (size_t)3072 * 3072;  // GlobalModel.cpp:21-22
    if (cap < n0) return fail(SF_ERR_ARG, "capacity below rows * cols (the first frame alone can need that many surfels)");
    sf_map *m = new sf_map;
    m->h = h;
    m->capacity = (int)cap;
    const size_t n_cand_max = (size_t)((h->k.rows + 1) / 2) * ((h->k.cols + 1) / 2);
    int e = SF_OK;
    if (!e) e = map_alloc(m, &m->buf[0], cap * 12);
    if (!e) e = map_alloc(m, &m->buf[1], cap * 12);
    if (!e) e = map_alloc(m, &m->keys, n0 * 16);
    if (!e && hipMemset(m->keys, 0xff, n0 * 16 * sizeof(unsigned long long)) != hipSuccess) e = fail(SF_ERR_DEVICE, "hipMemset");
    if (!e) e = map_alloc(m, &m->occ, (size_t)h->k.cols * 4 * ((h->k.rows * 4 + 63) / 64));
    if (!e) e = map_alloc(m, &m->index_export, n0 * 16);
    if (!e) e = map_alloc(m, &m->winner, cap);
    if (!e) e = map_alloc(m, &m->rec, n_cand_max * 12);
    if (!e) e = map_alloc(m, &m->meta, n_cand_max * 2);
    if (!e) e = map_alloc(m, &m->flags, cap + n_cand_max);
    if (!e) e = map_alloc(m, &m->block_counts, (cap + n_cand_max + SF_CLEAN_BLOCK - 1) / SF_CLEAN_BLOCK + 1);
    if (e) {
        sf_map_destroy(m);
        return e;
    }
    h->maps.push_back(m);
    *out = m;
    return SF_OK;
}
static void map_release(sf_map *m) {  // device memory of a map (its handle's device is current, its stream drained)
    for (void *q : m->allocs) (void)hipFree(q);
    m->allocs.clear();
}
void orphan_maps(sf_handle *h) {
    for (sf_map *m : h->maps) {
        map_release(m);
        m->h = nullptr;
    }
    h->maps.clear();
}
void sf_map_destroy(sf_map *m) {
    if (!m) return;
    if (sf_handle *h = m->h) {  // the handle is alive: its device, after its queued work
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        for (auto it = h->maps.begin(); it != h->maps.end(); ++it)
            if (*it == m) {
                h->maps.erase(it);
                break;
            }
        map_release(m);
    }  // else: sf_destroy of the handle already released the memory and left the map as an empty shell
    delete m;
}
static void pose_compose(const float *a, const float *b, float *out) {  // Eigen::Matrix4f product, column-major
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) {
            float acc = a[rr] * b[4 * c];
            for (int k = 1; k < 4; k++) acc = acc + a[rr + 4 * k] * b[k + 4 * c];
            r[rr + 4 * c] = acc;
        }
    std::memcpy(out, r, sizeof r);
}
// Reconstruction::fuseFrame for n (stream, map) pairs: at most 3 + 9 launches and one read-back for the whole batch
int sf_map_fuse_frames(sf_handle *h, int n, const int *streams, sf_map *const *maps, const float *in_poses, float weight_multiplier,
                       const sf_model_params *p) {
    if (!h || n < 0 || (n && (!streams || !maps)) || !p) return fail(SF_ERR_ARG, "bad argument");
    if (n == 0) return SF_OK;
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_map_fuse_frame needs a loaded frame (sf_load_frame + sf_filter_depth)");
    for (int q = 0; q < n; q++) {
        if (int e = check_stream(h, streams[q])) return e;
        if (!maps[q] || maps[q]->h != h) return fail(SF_ERR_ARG, "a map belongs to the handle it was created from");
        if (!in_poses && maps[q]->tick != 1) return fail(SF_ERR_ARG, "in_pose may be NULL on the first fuse only");
        for (int r = 0; r < q; r++)
            if (maps[r] == maps[q]) return fail(SF_ERR_ARG, "the same map twice in one batch");
    }
    HIP_TRY(hipSetDevice(h->device));
    if (int e = results_scratch(h, (size_t)n)) return e;
    const size_t npx = h->k.n0;
    std::vector<InitModelArgs> init;
    std::vector<FuseArgs> fuse;
    std::vector<int> init_of, fuse_of;  // batch index of each table entry
    // the maps' new poses / epochs are held here and committed together with count and tick only after the results have
    // been read back: a failed upload, launch or copy leaves every map as it was (a retry must not compose in_pose twice)
    std::vector<float> new_pose((size_t)n * 16);
    std::vector<int> new_epoch((size_t)n);
    int max_count = 0, max_cand = 0, max_elems = 0;
    for (int q = 0; q < n; q++) {
        sf_map *m = maps[q];
        const int stream = streams[q];
        const float *in_pose = in_poses ? in_poses + (size_t)q * 16 : nullptr;
        const float *depth_metric = h->in_depth_metric + (size_t)stream * npx;
        const float *depth_filtered = h->k.pyr_new[0] + (size_t)stream * h->k.n_tot;
        const uint8_t *color = h->in_color + (size_t)stream * npx * 3;
        const float *b_img = h->k.b_img + (size_t)stream * npx;
        float *pose_q = new_pose.data() + (size_t)q * 16;
        std::memcpy(pose_q, m->pose, sizeof m->pose);
        new_epoch[q] = m->epoch;
        if (m->tick == 1) {  // Reconstruction.cpp:255-262
            if (in_pose) pose_compose(m->pose, in_pose, pose_q);
            InitModelArgs a;
            a.depth_metric = depth_metric; a.depth_filtered = depth_filtered; a.color = color; a.b_img = b_img;
            a.rows = h->k.rows; a.cols = h->k.cols; a.time = m->tick;
            for (int k = 0; k < 16; k++) a.pose[k] = pose_q[k];
            a.cx = p->cx; a.cy = p->cy; a.fx = p->fx; a.fy = p->fy; a.max_depth = p->max_depth;
            a.out = m->buf[0];
            a.count = h->res_dev + (size_t)q * 8;
            init.push_back(a);
            init_of.push_back(q);
            continue;
        }
        float last_pose[16];
        std::memcpy(last_pose, m->pose, sizeof last_pose);
        pose_compose(m->pose, in_pose, pose_q);                                         // :268
        FuseArgs a;
        a.depth_metric = depth_metric; a.depth_filtered = depth_filtered; a.color = color; a.b_img = b_img;
        a.rows = h->k.rows; a.cols = h->k.cols;
        for (int k = 0; k < 16; k++) a.pose[k] = pose_q[k];
        invert_pose(pose_q, a.t_inv);
        a.cx = p->cx; a.cy = p->cy; a.fx = p->fx; a.fy = p->fy;
        a.camz = float(1.0 / double(p->fx)); a.camw = float(1.0 / double(p->fy));       // GlobalModel.cpp:365-368
        a.max_depth = p->max_depth; a.conf_threshold = p->conf_high;
        a.weighting = sf_fusion_weighting(last_pose, pose_q, weight_multiplier);         // :270-282
        a.time = m->tick; a.time_delta = p->time_delta;
        a.src = m->buf[0]; a.dst = m->buf[1]; a.out = m->buf[0];
        a.count = m->count; a.capacity = m->capacity;
        a.keys = m->keys; a.winner = m->winner;
        a.occ = m->occ; a.occ_words = (a.rows * 4 + 63) / 64;
        if (new_epoch[q] + 2 > 255) {  // the 8-bit tag is used up: one real clear, then count again
            HIP_TRY(hipMemsetAsync(m->keys, 0xff, npx * 16 * sizeof(unsigned long long), h->stream));
            m->epoch = new_epoch[q] = 0;  // the key image IS cleared from here on, whatever happens next
        }
        a.tag_first = 255u - (unsigned)(new_epoch[q] + 1); a.tag_merged = 255u - (unsigned)(new_epoch[q] + 2);
        new_epoch[q] += 2;
        a.par = m->tick % 2;
        a.cand_rows = (a.rows - a.par + 1) / 2; a.cand_cols = (a.cols - a.par + 1) / 2;
        a.n_cand = a.cand_rows * a.cand_cols;
        a.rec = m->rec; a.meta = m->meta; a.flags = m->flags; a.block_counts = m->block_counts;
        a.result = h->res_dev + (size_t)q * 8;
        max_count = std::max(max_count, a.count);
        max_cand = std::max(max_cand, a.n_cand);
        max_elems = std::max(max_elems, a.count + a.n_cand);
        fuse.push_back(a);
        fuse_of.push_back(q);
    }
    // one upload: [init table | fuse table]
    const size_t init_bytes = (init.size() * sizeof(InitModelArgs) + 255) / 256 * 256;
    std::vector<unsigned char> blob(init_bytes + fuse.size() * sizeof(FuseArgs));
    if (!init.empty()) std::memcpy(blob.data(), init.data(), init.size() * sizeof(InitModelArgs));
    if (!fuse.empty()) std::memcpy(blob.data() + init_bytes, fuse.data(), fuse.size() * sizeof(FuseArgs));
    void *dev = nullptr;
    if (int e = upload_table(h, blob.data(), blob.size(), &dev)) return e;
    if (!init.empty()) init_model_launch(h, (const InitModelArgs *)dev, (int)init.size());
    if (!fuse.empty()) {
        const FuseArgs *tab = (const FuseArgs *)((const unsigned char *)dev + init_bytes);
        const unsigned nm = (unsigned)fuse.size();
        const unsigned surfel_blocks = (unsigned)((max_count + 255) / 256);
        const unsigned occ_blocks = (unsigned)(((size_t)h->k.cols * 4 * ((h->k.rows * 4 + 63) / 64) + 255) / 256);
        const unsigned begin_blocks = std::max(occ_blocks, surfel_blocks);
        const unsigned clean_blocks = (unsigned)((max_elems + SF_CLEAN_BLOCK - 1) / SF_CLEAN_BLOCK);
        hipLaunchKernelGGL(sf_fuse_begin_kernel, dim3(begin_blocks, nm), dim3(256), 0, h->stream, tab);                        // :284
        if (max_count) hipLaunchKernelGGL(sf_index_splat_kernel, dim3(surfel_blocks, nm), dim3(256), 0, h->stream, tab);
        if (max_cand) hipLaunchKernelGGL(sf_fuse_data_kernel, dim3((max_cand + 63) / 64, nm), dim3(64), 0, h->stream, tab);   // :286-298
        hipLaunchKernelGGL(sf_index_clear_kernel, dim3(occ_blocks, nm), dim3(256), 0, h->stream, tab);                        // :300
        if (max_count) hipLaunchKernelGGL(sf_fuse_update_kernel, dim3(surfel_blocks, nm), dim3(256), 0, h->stream, tab);       // merge + index image of the result
        if (clean_blocks) {                                                                                                    // :302-311
            hipLaunchKernelGGL(sf_clean_flag_kernel, dim3(clean_blocks, nm), dim3(SF_CLEAN_BLOCK), 0, h->stream, tab);
            hipLaunchKernelGGL(sf_clean_scan_kernel, dim3(nm), dim3(1024), 0, h->stream, tab);
            hipLaunchKernelGGL(sf_clean_write_kernel, dim3(clean_blocks, nm), dim3(SF_CLEAN_BLOCK), 0, h->stream, tab);
        }
    }
    HIP_TRY(hipGetLastError());
    std::vector<int> res((size_t)n * 8);
    if (int e = d2h(h, res.data(), h->res_dev, res.size() * sizeof(int))) return e;
    int overflow = -1;
    for (int q = 0; q < n; q++) {  // commit
        std::memcpy(maps[q]->pose, new_pose.data() + (size_t)q * 16, sizeof maps[q]->pose);
        maps[q]->epoch = new_epoch[q];
    }
    for (int q : init_of) {
        sf_map *m = maps[q];
        m->count = res[(size_t)q * 8];
        m->stats[0] = m->stats[1] = m->stats[2] = 0;
        m->stats[3] = m->count;
        m->tick++;
    }
    for (int q : fuse_of) {
        sf_map *m = maps[q];
        const int *r = res.data() + (size_t)q * 8;
        m->count = r[0];
        m->stats[0] = r[2]; m->stats[1] = r[3]; m->stats[2] = r[4]; m->stats[3] = r[0];
        m->have_index = true;
        m->tick++;
        if (r[1] > m->capacity && overflow < 0) overflow = q;
    }
    if (overflow >= 0) return fail(SF_ERR_STATE, "surfel map capacity exceeded (truncated): batch entry " + std::to_string(overflow));
    return SF_OK;
}
int sf_map_fuse_frame(sf_handle *h, int stream, sf_map *m, const float *in_pose, float weight_multiplier, const sf_model_params *p) {
    return sf_map_fuse_frames(h, 1, &stream, &m, in_pose, weight_multiplier, p);
}
// Reconstruction::getPredictedImages for n (stream, map) pairs at each map's currPose and tick, in four launches
int sf_map_predict_frames(sf_handle *h, int n, const int *streams, sf_map *const *maps, const sf_model_params *p) {
    if (!h || n < 0 || (n && (!streams || !maps)) || !p) return fail(SF_ERR_ARG, "bad argument");
    std::vector<PredictJob> jobs((size_t)n);
    for (int q = 0; q < n; q++) {
        if (int e = check_stream(h, streams[q])) return e;
        if (!maps[q] || maps[q]->h != h) return fail(SF_ERR_ARG, "a map belongs to the handle it was created from");
        for (int r = 0; r < q; r++)
            if (streams[r] == streams[q]) return fail(SF_ERR_ARG, "the same stream twice in one batch (its prediction would be written twice)");
        jobs[(size_t)q] = PredictJob{streams[q], maps[q]->buf[0], maps[q]->count, maps[q]->pose, maps[q]->tick, maps[q]->tick};
    }
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    return predict_batch(h, jobs, p);
}
int sf_map_predict(sf_handle *h, int stream, sf_map *m, const sf_model_params *p) {
    return sf_map_predict_frames(h, 1, &stream, &m, p);
}
int sf_map_info(sf_map *m, int *count, int *tick, float pose[16], int stats[4]) {
    if (!m) return fail(SF_ERR_ARG, "null");
    if (count) *count = m->count;
    if (tick) *tick = m->tick;
    if (pose) std::memcpy(pose, m->pose, sizeof m->pose);
    if (stats) std::memcpy(stats, m->stats, sizeof m->stats);
    return SF_OK;
}
int sf_map_download(sf_map *m, float *surfels, int max_count) {
    if (!m || (!surfels && max_count > 0) || max_count < 0) return fail(SF_ERR_ARG, "bad argument");
    if (!m->h) return fail(SF_ERR_STATE, "the handle this map was created from has been destroyed");
    const size_t k = (size_t)std::min(max_count, m->count);
    if (k) return d2h(m->h, surfels, m->buf[0], k * 12 * sizeof(float));
    return SF_OK;
}
int sf_map_upload(sf_map *m, const float *surfels, int count, const float pose[16], int tick) {
    if (!m || (!surfels && count > 0) || count < 0 || !pose || tick < 1) return fail(SF_ERR_ARG, "bad argument");
    if (count > m->capacity) return fail(SF_ERR_ARG, "count exceeds the map's capacity");
    if (!m->h) return fail(SF_ERR_STATE, "the handle this map was created from has been destroyed");
    HIP_TRY(hipSetDevice(m->h->device));
    if (count) {
        HIP_TRY(hipMemcpyAsync(m->buf[0], surfels, (size_t)count * 12 * sizeof(float), hipMemcpyHostToDevice, m->h->stream));
        HIP_TRY(hipStreamSynchronize(m->h->stream));
    }
    m->count = count;
    std::memcpy(m->pose, pose, sizeof m->pose);
    m->tick = tick;
    return SF_OK;
}
int sf_map_get_index_map(sf_map *m, uint32_t *out) {
    if (!m || !out) return fail(SF_ERR_ARG, "null");
    if (!m->h) return fail(SF_ERR_STATE, "the handle this map was created from has been destroyed");
    if (!m->have_index) return fail(SF_ERR_STATE, "no index map yet (sf_map_fuse_frame with tick > 1 renders it)");
    HIP_TRY(hipSetDevice(m->h->device));
    const size_t n = m->h->k.n0 * 16;
    hipLaunchKernelGGL(sf_index_export_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->h->stream, m->keys, m->occ,
                       (m->h->k.rows * 4 + 63) / 64, m->index_export, m->h->k.cols * 4, m->h->k.rows * 4);
    HIP_TRY(hipGetLastError());
    return d2h(m->h, out, m->index_export, n * sizeof(uint32_t));
}

}  // extern "C"
