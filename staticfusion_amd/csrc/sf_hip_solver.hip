// sf_hip_solver.hip — libsf_hip.so, the solver's entry points of include/sf.h: images in (host, device, the frame pool of
// resident sequences), frames (one or several per launch of the frame kernel), results and debug planes out, measurement support.
#include "sf_host.h"

// prediction := current; current := pool[frame_index[stream]] for every stream, 16 bytes per lane and plane
// (sf_advance_sequences_device). grid = (slices, batch).
__global__ __launch_bounds__(256) void sf_advance_kernel(float *cur_d, float *cur_i, float *pred_d, float *pred_i, const float *pool_d,
                                                         const float *pool_i, const int *frame_index, int n0, int n_tot) {
    const int b = blockIdx.y;
    const int f = frame_index[b];
    if (f < 0) return;
    typedef float __attribute__((ext_vector_type(4))) f4;
    const size_t so = (size_t)b * n_tot, po = (size_t)f * n0;
    for (int q = (blockIdx.x * 256 + threadIdx.x) * 4; q < n0; q += gridDim.x * 256 * 4) {
        const f4 cd = *(const f4 *)(cur_d + so + q), ci = *(const f4 *)(cur_i + so + q);
        const f4 nd = *(const f4 *)(pool_d + po + q), ni = *(const f4 *)(pool_i + po + q);
        *(f4 *)(pred_d + so + q) = cd;
        *(f4 *)(pred_i + so + q) = ci;
        *(f4 *)(cur_d + so + q) = nd;
        *(f4 *)(cur_i + so + q) = ni;
    }
}

// sf_microbench_copy: what a plain streaming kernel reaches (16-byte loads and stores, grid-stride)
__global__ __launch_bounds__(256) void sf_copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n16) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += (size_t)gridDim.x * 256) dst[q] = src[q];
}

extern "C" {

static int upload_pair(sf_handle *h, float *const *set, int stream, const float *depth, const float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    if (!depth || !intensity) return fail(SF_ERR_ARG, "null image");
    HIP_TRY(hipSetDevice(h->device));
    const size_t bytes = sizeof(float) * h->k.n0, o = (size_t)stream * h->k.n_tot;
    HIP_TRY(hipMemcpyAsync(set[0] + o, depth, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(set[1] + o, intensity, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // the caller may reuse its buffers on return
    return SF_OK;
}
// depthCurrent / intensityCurrent ARE level 0 of the new pyramid (createImagePyramid copies them there)
int sf_set_current(sf_handle *h, int stream, const float *depth, const float *intensity) {
    return h ? upload_pair(h, h->k.pyr_new, stream, depth, intensity) : fail(SF_ERR_ARG, "null");
}
int sf_set_prediction(sf_handle *h, int stream, const float *depth, const float *intensity) {
    return h ? upload_pair(h, h->k.pyr_pred, stream, depth, intensity) : fail(SF_ERR_ARG, "null");
}
static int copy_batch_device(sf_handle *h, float *const *set, const void *d, const void *i) {
    if (!h || !d || !i) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipSetDevice(h->device));
    const size_t w = sizeof(float) * h->k.n0;
    HIP_TRY(hipMemcpy2DAsync(set[0], sizeof(float) * h->k.n_tot, d, w, w, h->k.batch, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpy2DAsync(set[1], sizeof(float) * h->k.n_tot, i, w, w, h->k.batch, hipMemcpyDeviceToDevice, h->stream));
    return SF_OK;
}
int sf_set_current_device(sf_handle *h, const void *d, const void *i) {
    return h ? copy_batch_device(h, h->k.pyr_new, d, i) : fail(SF_ERR_ARG, "null");
}
int sf_set_prediction_device(sf_handle *h, const void *d, const void *i) {
    return h ? copy_batch_device(h, h->k.pyr_pred, d, i) : fail(SF_ERR_ARG, "null");
}
int sf_advance_sequences_device(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index, int pool_frames) {
    if (!h || !pool_depth || !pool_intensity || !frame_index) return fail(SF_ERR_ARG, "null");
    if (pool_frames < 1) return fail(SF_ERR_ARG, "pool_frames < 1");
    if (h->k.n0 % 4 || h->k.n_tot % 4) return fail(SF_ERR_ARG, "level sizes must be multiples of 4 pixels");
    if (((uintptr_t)pool_depth | (uintptr_t)pool_intensity) & 15u) return fail(SF_ERR_ARG, "the frame pools must be 16-byte aligned (16-byte loads)");
    const size_t B = (size_t)h->k.batch;
    for (size_t b = 0; b < B; b++)
        if (frame_index[b] >= pool_frames) return fail(SF_ERR_ARG, "frame_index entry outside the pool");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->seq_ready) {  // all or nothing: a failure leaves nothing half-initialised behind (the next call starts over)
        if (!h->seq_index)
            if (int e = dev_alloc(h, &h->seq_index, B * sf_handle::SEQ_SLOTS)) return e;
        if (!h->seq_index_host) HIP_TRY(hipHostMalloc((void **)&h->seq_index_host, sizeof(int) * B * sf_handle::SEQ_SLOTS, hipHostMallocDefault));
        for (auto &e : h->seq_done)
            if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->seq_ready = true;
    }
    const unsigned slot = h->seq_calls % sf_handle::SEQ_SLOTS;
    if (h->seq_calls >= (unsigned)sf_handle::SEQ_SLOTS) HIP_TRY(hipEventSynchronize(h->seq_done[slot]));  // eight calls ago
    h->seq_calls++;
    int *host = h->seq_index_host + slot * B, *dev = h->seq_index + slot * B;
    std::memcpy(host, frame_index, sizeof(int) * B);
    HIP_TRY(hipMemcpyAsync(dev, host, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    const dim3 grid((unsigned)std::min(16, (h->k.n0 / 4 + 255) / 256), (unsigned)h->k.batch);
    hipLaunchKernelGGL(sf_advance_kernel, grid, dim3(256), 0, h->stream, h->k.pyr_new[0], h->k.pyr_new[1], h->k.pyr_pred[0], h->k.pyr_pred[1],
                       (const float *)pool_depth, (const float *)pool_intensity, (const int *)dev, h->k.n0, h->k.n_tot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->seq_done[slot], h->stream));
    return SF_OK;
}

// ---- overlapped upload: the next batch of frames crosses PCIe on a second HIP stream while the solver runs ----
int sf_upload_current_async(sf_handle *h, const float *depth_batch, const float *intensity_batch) {
    if (!h || !depth_batch || !intensity_batch) return fail(SF_ERR_ARG, "null");
    if (h->upload_pending) return fail(SF_ERR_STATE, "an upload is already pending: call sf_commit_upload first");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->k.n0 * h->k.batch;
    if (!h->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->copy_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->compute_done, hipEventDisableTiming));
        if (int e = dev_alloc(h, &h->up_depth, n)) return e;
        if (int e = dev_alloc(h, &h->up_inten, n)) return e;
    }
    // the staging block may still be read by the previous commit's copy on the compute stream
    HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->compute_done, 0));
    HIP_TRY(hipMemcpyAsync(h->up_depth, depth_batch, n * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipMemcpyAsync(h->up_inten, intensity_batch, n * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipEventRecord(h->copy_done, h->copy_stream));
    h->upload_pending = true;
    return SF_OK;
}
int sf_commit_upload(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (!h->upload_pending) return fail(SF_ERR_STATE, "no upload pending");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->copy_done, 0));  // device-side dependency: the host does not block
    if (int e = copy_batch_device(h, h->k.pyr_new, h->up_depth, h->up_inten)) return e;
    HIP_TRY(hipEventRecord(h->compute_done, h->stream));
    h->upload_pending = false;
    return SF_OK;
}
int sf_alloc_pinned(size_t bytes, void **out) {
    if (!out) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return SF_OK;
}
int sf_free_pinned(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return SF_OK;
}

int sf_current_to_prediction(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipSetDevice(h->device));
    const size_t w = sizeof(float) * h->k.n0, pitch = sizeof(float) * h->k.n_tot;
    for (int c = 0; c < 2; c++)
        HIP_TRY(hipMemcpy2DAsync(h->k.pyr_pred[c], pitch, h->k.pyr_new[c], pitch, w, h->k.batch, hipMemcpyDeviceToDevice,
                                 h->stream));
    return SF_OK;
}
int sf_set_segm_state(sf_handle *h, int stream, const int32_t *labels0, const float *b_segm, const float *cluster_res) {
    if (int e = check_stream(h, stream)) return e;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<uint8_t> lab;
    if (labels0) {
        lab.resize(h->k.n0);
        for (int q = 0; q < h->k.n0; q++) {
            if (labels0[q] < 0 || labels0[q] > SF_NC) return fail(SF_ERR_ARG, "label out of range");
            lab[q] = (uint8_t)labels0[q];
        }
        HIP_TRY(hipMemcpyAsync(h->k.labels + (size_t)stream * h->k.n_tot, lab.data(), lab.size(), hipMemcpyHostToDevice, h->stream));
    }
    if (b_segm) HIP_TRY(hipMemcpyAsync(h->k.state[stream].b_segm, b_segm, SF_NC * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (cluster_res)
        HIP_TRY(hipMemcpyAsync(h->k.state[stream].cluster_res, cluster_res, SF_NC * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SF_OK;
}
int sf_set_twist_old(sf_handle *h, int stream, const float twist[6]) {
    if (int e = check_stream(h, stream)) return e;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->k.state[stream].twist_old, twist, 6 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SF_OK;
}

int sf_build_pyramid(sf_handle *h, int old_im) {
    if (!h) return fail(SF_ERR_ARG, "null");
    return launch(h, old_im ? ST_PYR_OLD : ST_PYR_NEW, 0);
}
int sf_kmeans(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    return launch(h, ST_KMEANS, 0);
}
int sf_run_solver(sf_handle *h, int create_image_pyr) {
    if (!h) return fail(SF_ERR_ARG, "null");
    return launch(h, solve_mask(h, create_image_pyr), 0);
}
int sf_push_history(sf_handle *h, int im_count) {
    if (!h || im_count < 0) return fail(SF_ERR_ARG, "bad argument");
    return launch(h, ST_PUSH_HISTORY, im_count);
}
int sf_residuals_vs_history(sf_handle *h, int index) {
    if (!h || index < SF_HISTORY) return fail(SF_ERR_ARG, "index must be >= 5");
    return launch(h, ST_RESIDUALS, index);
}
int sf_build_segm_image(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    return launch(h, ST_SEGM_IMAGE, 0);
}
int sf_process_frame(sf_handle *h, int im_count) {
    if (!h || im_count < 0) return fail(SF_ERR_ARG, "bad argument");
    int m = ST_PYR_OLD | solve_mask(h, 1) | ST_SEGM_IMAGE | ST_PUSH_HISTORY;
    if (im_count - SF_HISTORY >= 0) m |= ST_RESIDUALS;
    return launch(h, m, im_count);
}

// ---- several frames per launch ----------------------------------------------------------------
static int multi_buffers(sf_handle *h, int n_frames, bool want_index, bool want_traj) {
    const size_t B = (size_t)h->k.batch;
    if (!h->d_frame_done)
        if (int e = dev_alloc(h, &h->d_frame_done, B)) return e;
    // Each buffer only when this call needs it, each with its own capacity (at 16 384 streams and 4096 frames the three
    // together would be 4.3 GB of HBM + 268 MB of pinned memory; sf_process_frames without T_out needs none of them).
    // Growing: the old block may still be in use by a queued launch.
    if (want_index && n_frames > h->multi_capacity) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->d_multi_index) (void)hipFree(h->d_multi_index);
        if (h->h_multi_index) (void)hipHostFree(h->h_multi_index);
        h->d_multi_index = nullptr; h->h_multi_index = nullptr; h->multi_capacity = 0;
        HIP_TRY(hipMalloc((void **)&h->d_multi_index, sizeof(int) * B * n_frames));
        HIP_TRY(hipHostMalloc((void **)&h->h_multi_index, sizeof(int) * B * n_frames, hipHostMallocDefault));
        h->multi_capacity = n_frames;
    }
    if (want_traj && n_frames > h->traj_capacity) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->d_traj) (void)hipFree(h->d_traj);
        h->d_traj = nullptr; h->traj_capacity = 0;
        HIP_TRY(hipMalloc((void **)&h->d_traj, sizeof(float) * 16 * B * n_frames));
        h->traj_capacity = n_frames;
    }
    return SF_OK;
}
static int process_frames(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index, int pool_frames,
                          int im_count0, int n_frames, float *T_out) {
    if (!h || im_count0 < 0 || n_frames < 1 || n_frames > 4096) return fail(SF_ERR_ARG, "bad argument");
    const size_t B = (size_t)h->k.batch;
    const bool seq = pool_depth || pool_intensity || frame_index;
    if (seq) {
        if (!pool_depth || !pool_intensity || !frame_index) return fail(SF_ERR_ARG, "null");
        if (pool_frames < 1) return fail(SF_ERR_ARG, "pool_frames < 1");
        if (h->k.n0 % 4 || h->k.n_tot % 4) return fail(SF_ERR_ARG, "level sizes must be multiples of 4 pixels");
        if (((uintptr_t)pool_depth | (uintptr_t)pool_intensity) & 15u) return fail(SF_ERR_ARG, "the frame pools must be 16-byte aligned (16-byte loads)");
        for (size_t q = 0; q < B * n_frames; q++)
            if (frame_index[q] >= pool_frames) return fail(SF_ERR_ARG, "frame_index entry outside the pool");
    }
    if (h->cluster_grid || n_frames == 1) {
        // the cluster build keeps all workgroups of a stream resident together, one frame per launch: the same calls one by one
        for (int k = 0; k < n_frames; k++) {
            if (seq)
                if (int e = sf_advance_sequences_device(h, pool_depth, pool_intensity, frame_index + (size_t)k * B, pool_frames)) return e;
            if (int e = sf_process_frame(h, im_count0 + k)) return e;
            if (T_out) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                HIP_TRY(hipMemcpy2D(T_out + (size_t)k * B * 16, 16 * sizeof(float), h->k.state, sizeof(StreamState), 16 * sizeof(float), B, hipMemcpyDeviceToHost));
            }
        }
        return SF_OK;
    }
    HIP_TRY(hipSetDevice(h->device));
    if (int e = multi_buffers(h, n_frames, seq, T_out != nullptr)) return e;
    HIP_TRY(hipMemsetAsync(h->d_frame_done, 0, sizeof(int) * B, h->stream));
    FrameLaunch ml{};
    ml.frame_done = h->d_frame_done;
    if (seq) {
        HIP_TRY(hipStreamSynchronize(h->stream));  // the staging block of the previous call has been consumed
        std::memcpy(h->h_multi_index, frame_index, sizeof(int) * B * n_frames);
        HIP_TRY(hipMemcpyAsync(h->d_multi_index, h->h_multi_index, sizeof(int) * B * n_frames, hipMemcpyHostToDevice, h->stream));
        ml.seq_index = h->d_multi_index;
        ml.pool_d = (const float *)pool_depth;
        ml.pool_i = (const float *)pool_intensity;
        // 2: swap the pyramid buffers and read level 0 of both images in the pool; 1: swap, copy the new frame in (round 4's form);
        // 0: copy both images, rebuild the prediction's pyramid (A/B switches; the results are identical in all three)
        ml.flip_ok = std::getenv("SF_NO_PYRAMID_FLIP") ? 0 : std::getenv("SF_NO_POOL_IN_PLACE") ? 1 : 2;
    }
    if (T_out) {
        ml.traj = h->d_traj;
        // a frame the launch skips (its stream's previous frame never finished: SF_STATUS_SYNC_TIMEOUT) leaves its row NaN
        HIP_TRY(hipMemsetAsync(h->d_traj, 0xff, sizeof(float) * 16 * B * n_frames, h->stream));
    }
    const int m = ST_PYR_OLD | solve_mask(h, 1) | ST_SEGM_IMAGE | ST_PUSH_HISTORY | ST_AUTO_RESIDUALS;
    if (int e = launch(h, m, im_count0, n_frames, &ml)) return e;
    if (T_out) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(T_out, h->d_traj, sizeof(float) * 16 * B * n_frames, hipMemcpyDeviceToHost));
    }
    return SF_OK;
}
int sf_process_frames(sf_handle *h, int im_count0, int n_frames, float *T_out) {
    return process_frames(h, nullptr, nullptr, nullptr, 0, im_count0, n_frames, T_out);
}
int sf_process_sequence_frames_device(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index, int pool_frames,
                                      int im_count0, int n_frames, float *T_out) {
    if (!pool_depth || !pool_intensity || !frame_index) return fail(SF_ERR_ARG, "null");
    return process_frames(h, pool_depth, pool_intensity, frame_index, pool_frames, im_count0, n_frames, T_out);
}

// ---- getters (synchronise the handle's stream, then copy) ----------------------------------
int d2h(sf_handle *h, void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SF_OK;
}
int sf_get_T(sf_handle *h, int stream, float T[16]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, T, h->k.state[stream].T, 16 * sizeof(float));
}
int sf_get_twist(sf_handle *h, int stream, float t[6]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, t, h->k.state[stream].twist, 6 * sizeof(float));
}
int sf_get_twist_old(sf_handle *h, int stream, float t[6]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, t, h->k.state[stream].twist_old, 6 * sizeof(float));
}
int sf_get_b(sf_handle *h, int stream, float b[SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, b, h->k.state[stream].b_segm, SF_NC * sizeof(float));
}
int sf_get_b_image(sf_handle *h, int stream, float *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out) return fail(SF_ERR_ARG, "null");
    return d2h(h, out, h->k.b_img + (size_t)stream * h->k.n0, sizeof(float) * h->k.n0);
}
int sf_get_labels(sf_handle *h, int stream, int level, int32_t *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out || level < 0 || level >= h->k.levels) return fail(SF_ERR_ARG, "bad level");
    std::vector<uint8_t> tmp(h->k.ln[level]);
    if (int e = d2h(h, tmp.data(), h->k.labels + (size_t)stream * h->k.n_tot + h->k.loff[level], tmp.size())) return e;
    for (size_t q = 0; q < tmp.size(); q++) out[q] = tmp[q];
    return SF_OK;
}
int sf_get_kmeans(sf_handle *h, int stream, float c[3 * SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, c, h->k.state[stream].kmeans, 3 * SF_NC * sizeof(float));
}
int sf_get_connectivity(sf_handle *h, int stream, uint8_t conn[SF_NUM_CLUSTERS * SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    uint32_t rows[SF_NC];
    if (int e = d2h(h, rows, h->k.state[stream].conn, sizeof(rows))) return e;
    for (int i = 0; i < SF_NC; i++)
        for (int j = 0; j < SF_NC; j++) conn[i * SF_NC + j] = (rows[i] >> j) & 1u;
    return SF_OK;
}
int sf_get_cluster_residuals(sf_handle *h, int stream, float r[SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, r, h->k.state[stream].cluster_res, SF_NC * sizeof(float));
}
int sf_get_stats(sf_handle *h, int stream, sf_frame_stats *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out) return fail(SF_ERR_ARG, "null");
    return d2h(h, out, &h->k.stats[stream], sizeof(sf_frame_stats));
}
int sf_get_batch_results(sf_handle *h, float *T, int32_t *n_irls, int32_t *n_outer, int64_t *pixel_iters) {
    if (!h) return fail(SF_ERR_ARG, "null");
    const int B = h->k.batch;
    if (T) {
        std::vector<StreamState> st(B);
        if (int e = d2h(h, st.data(), h->k.state, B * sizeof(StreamState))) return e;
        for (int b = 0; b < B; b++) std::memcpy(T + 16 * b, st[b].T, 16 * sizeof(float));
    }
    if (n_irls || n_outer || pixel_iters) {
        std::vector<sf_frame_stats> fs(B);
        if (int e = d2h(h, fs.data(), h->k.stats, B * sizeof(sf_frame_stats))) return e;
        for (int b = 0; b < B; b++) {
            if (n_irls) n_irls[b] = fs[b].n_irls;
            if (n_outer) n_outer[b] = fs[b].n_outer;
            if (pixel_iters) pixel_iters[b] = fs[b].pixel_iters;
        }
    }
    return SF_OK;
}

int sf_get_plane(sf_handle *h, int stream, int set, int channel, int level, float *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out || level < 0 || level >= h->k.levels || set < 0 || set > 3 || channel < 0 || channel > 3)
        return fail(SF_ERR_ARG, "bad selector");
    float *const *tab[4] = {h->k.pyr_new, h->k.pyr_pred, h->k.dbg_warped, h->k.dbg_inter};
    const size_t off = (size_t)stream * h->k.n_tot + h->k.loff[level], n = h->k.ln[level];
    if (set <= SF_SET_PRED && channel >= SF_CH_XX) {
        // xx / yy of the pyramids are not stored on the device (every kernel recomputes them from the depth):
        // the same float expression (reference FrontEnd.cpp:385-386) evaluated here
        if (int e = d2h(h, out, tab[set][SF_CH_DEPTH] + off, sizeof(float) * n)) return e;
        const int rows_i = h->k.lrows[level], cols_i = h->k.lcols[level];
        const float inv_f_i = 2.f * h->k.tan_half_fovh / float(cols_i);
        const float disp = (channel == SF_CH_XX) ? 0.5f * (cols_i - 1) : 0.5f * (rows_i - 1);
        for (int u = 0; u < cols_i; u++)
            for (int v = 0; v < rows_i; v++) {
                float &d = out[v + (size_t)u * rows_i];
                d = (inv_f_i * (float(channel == SF_CH_XX ? u : v) - disp)) * d;
            }
        return SF_OK;
    }
    const float *base = tab[set][channel];
    if (!base) return fail(SF_ERR_STATE, "WARPED / INTER planes need params.debug_planes = 1 at sf_create");
    return d2h(h, out, base + off, sizeof(float) * n);
}

int sf_get_jacobian_rows(sf_handle *h, int stream, float *A, float *B, int *n_rows) {
    if (int e = check_stream(h, stream)) return e;
    if (!n_rows) return fail(SF_ERR_ARG, "null");
    if (!h->k.p.debug_planes) return fail(SF_ERR_STATE, "the Jacobian rows need params.debug_planes = 1");
    HIP_TRY(hipSetDevice(h->device));
    StreamState st;
    if (int e = d2h(h, &st, &h->k.state[stream], sizeof(st))) return e;
    const int L = st.last_level;
    if (L < 0 || L >= h->k.levels || st.cum_frames == 0) return fail(SF_ERR_STATE, "no outer iteration executed yet");
    const size_t n = h->k.ln[L];
    float *dev = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, 14 * n * sizeof(float)));
    std::vector<float> planes(14 * n);
    if (h->args_dirty) {
        HIP_TRY(hipMemcpyAsync(h->d_args, &h->k, sizeof(KArgs), hipMemcpyHostToDevice, h->stream));
        h->args_dirty = false;
    }
    h->fv->launch_debug_rows(int((n + 1023) / 1024), h->stream, (const KArgs *)h->d_args, stream, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(planes.data(), dev, planes.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(SF_ERR_DEVICE, std::string("sf_get_jacobian_rows: ") + hipGetErrorString(e));
    int rows = 0;  // validPixels order of the reference = ascending column-major index (u outer, v inner)
    for (size_t q = 0; q < n; q++) {
        if (std::isnan(planes[q])) continue;
        for (int half = 0; half < 2; half++) {
            if (A)
                for (int c = 0; c < 6; c++) A[(size_t)rows * 6 + c] = planes[(size_t)(7 * half + c) * n + q];
            if (B) B[rows] = planes[(size_t)(7 * half + 6) * n + q];
            rows++;
        }
    }
    *n_rows = rows;
    return SF_OK;
}

int sf_get_lin_plane(sf_handle *h, int stream, int which, float *out, int *rows, int *cols) {
    if (int e = check_stream(h, stream)) return e;
    if (which < 0 || which >= SF_LIN_COUNT) return fail(SF_ERR_ARG, "bad selector");
    StreamState st;
    if (int e = d2h(h, &st, &h->k.state[stream], sizeof(st))) return e;
    const int L = st.last_level;
    if (L < 0 || L >= h->k.levels) return fail(SF_ERR_STATE, "no outer iteration executed yet");
    if (rows) *rows = h->k.lrows[L];
    if (cols) *cols = h->k.lcols[L];
    if (!out) return SF_OK;
    const size_t n = h->k.ln[L], o = (size_t)st.last_slot * h->k.n0;  // the slot the last outer iteration ran on
    if (which == SF_LIN_NULL) {
        if (!h->k.p.debug_planes) return fail(SF_ERR_STATE, "the Null plane needs params.debug_planes = 1");
        std::vector<uint8_t> tmp(n);
        if (int e = d2h(h, tmp.data(), h->k.rec_null + o, n)) return e;
        for (size_t q = 0; q < n; q++) out[q] = tmp[q] ? 1.f : 0.f;
        return SF_OK;
    }
    // dcu..ddv and dct are stored; ddt and the pre-weights are recomputed exactly as the kernels do
    std::vector<float> dn(n), dw(n);
    if (int e = d2h(h, dn.data(), h->k.pyr_new[0] + (size_t)stream * h->k.n_tot + h->k.loff[L], sizeof(float) * n)) return e;
    if (int e = d2h(h, dw.data(), h->k.rec[R_DW] + o, sizeof(float) * n)) return e;
    auto fetch = [&](int plane, std::vector<float> &v) { v.resize(n); return d2h(h, v.data(), h->k.rec[plane] + o, sizeof(float) * n); };
    std::vector<uint8_t> lab(n);  // validPixels: the sign of the stored warped depth (sf_solver.h, linearise)
    if (h->reforder) {  // ... or, in the reference-order build, the label plane (the sign there is the warp's own)
        if (int e = d2h(h, lab.data(), h->k.rec_lab + o, n)) return e;
    } else
    for (size_t q = 0; q < n; q++) {
        lab[q] = (dw[q] > 0.f) ? 0 : SF_INVALID_LABEL;
        dw[q] = std::fabs(dw[q]);
    }
    switch (which) {
        case SF_LIN_DCU: return d2h(h, out, h->k.rec[R_DCU] + o, sizeof(float) * n);
        case SF_LIN_DCV: return d2h(h, out, h->k.rec[R_DCV] + o, sizeof(float) * n);
        case SF_LIN_DCT: return d2h(h, out, h->k.rec[R_DCT] + o, sizeof(float) * n);
        case SF_LIN_DDU: return d2h(h, out, h->k.rec[R_DDU] + o, sizeof(float) * n);
        case SF_LIN_DDV: return d2h(h, out, h->k.rec[R_DDV] + o, sizeof(float) * n);
        case SF_LIN_DDT:
            for (size_t q = 0; q < n; q++) out[q] = dn[q] - dw[q];
            return SF_OK;
        default: break;
    }
    std::vector<float> t, gu, gv;
    const bool colour = (which == SF_LIN_WC);
    if (int e = fetch(colour ? R_DCU : R_DDU, gu)) return e;
    if (int e = fetch(colour ? R_DCV : R_DDV, gv)) return e;
    if (colour) {
        if (int e = fetch(R_DCT, t)) return e;
    } else {
        t.resize(n);
        for (size_t q = 0; q < n; q++) t[q] = dn[q] - dw[q];
    }
    for (size_t q = 0; q < n; q++) {
        float w = 0.f;
        if (lab[q] != SF_INVALID_LABEL) {  // weights are 0 outside validPixels (reference :483-484)
            const float err = (colour ? 10.f : 200.f) * (std::fabs(t[q]) + std::fabs(gu[q]) + std::fabs(gv[q]));
            w = std::sqrt(1.f / ((colour ? 1.f : 0.01f) + err));
            w = (colour ? st.inv_max_c : st.inv_max_d) * w;
        }
        out[q] = w;
    }
    return SF_OK;
}

int sf_level_rows(const sf_handle *h, int level) { return (h && level >= 0 && level < h->k.levels) ? h->k.lrows[level] : 0; }
int sf_level_cols(const sf_handle *h, int level) { return (h && level >= 0 && level < h->k.levels) ? h->k.lcols[level] : 0; }
int sf_batch(const sf_handle *h) { return h ? h->k.batch : 0; }

int sf_timed_process_frames(sf_handle *h, int im_count, int calls, float *elapsed_ms) {
    if (!h || calls < 1) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if (std::getenv("SF_TIMED_LAUNCH_PER_FRAME")) {  // A/B: one launch per frame, as before multi-frame launches existed
        for (int c = 0; c < calls; c++)
            if (int e = sf_process_frame(h, im_count + c)) return e;
    } else if (int e = sf_process_frames(h, im_count, calls, nullptr)) {
        return e;
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return SF_OK;
}
int sf_get_counters(sf_handle *h, int64_t *frames, int64_t *n_irls, int64_t *n_outer, int64_t *pixel_iters) {
    if (!h) return fail(SF_ERR_ARG, "null");
    std::vector<StreamState> st(h->k.batch);
    if (int e = d2h(h, st.data(), h->k.state, st.size() * sizeof(StreamState))) return e;
    long long f = 0, i = 0, o = 0, p = 0;
    for (auto &s : st) {
        f += s.cum_frames;
        i += s.cum_irls;
        o += s.cum_outer;
        p += s.cum_pixel_iters;
    }
    if (frames) *frames = f;
    if (n_irls) *n_irls = i;
    if (n_outer) *n_outer = o;
    if (pixel_iters) *pixel_iters = p;
    return SF_OK;
}
int sf_get_stage_profile(sf_handle *h, int64_t ticks[32]) {
    if (!h || !ticks) return fail(SF_ERR_ARG, "null");
    std::vector<StreamState> st(h->k.batch);
    if (int e = d2h(h, st.data(), h->k.state, st.size() * sizeof(StreamState))) return e;
    for (int q = 0; q < SF_PROF_SLOTS; q++) ticks[q] = 0;
    for (auto &s : st)
        for (int q = 0; q < SF_PROF_SLOTS; q++) ticks[q] += s.prof[q];
    return SF_OK;
}
int sf_microbench_pass(sf_handle *h, int which, int variant, int reps, float *elapsed_ms) {
    const int slices = (variant >> 8) ? (variant >> 8) : 1;  // bits 8.. of `variant`: workgroups per stream (experiment)
    variant &= 255;
    if (!h || (which != 1 && which != 2) || variant < 0 || variant > 2 || reps < 1 || slices > 64) return fail(SF_ERR_ARG, "bad argument");
    if (h->fv->id == SF_VARIANT_CLUSTER) return fail(SF_ERR_STATE, "the isolated passes are not built for the cluster variant");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(h->k.queue, 0, sizeof(int), h->stream));
    const int grid = std::min(h->k.batch * slices, h->max_blocks);
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    h->fv->launch_irls_pass(grid, h->stream, (const KArgs *)h->d_args, which, variant, reps, slices);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return SF_OK;
}
int sf_microbench_copy(sf_handle *h, size_t bytes, int reps, float *elapsed_ms) {
    if (!h || !elapsed_ms || bytes < 4096 || (bytes & 15u) || bytes > ((size_t)1 << 36) || reps < 1 || reps > 1000) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    void *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (src) (void)hipFree(src);
        return fail(SF_ERR_NOMEM, "sf_microbench_copy: scratch blocks");
    }
    int e = SF_OK;
    float ms = 0.f;
    // events of its own: the handle's ev0 / ev1 belong to the timed_* entry points, evk0 / evk1 to the solver launches
    // (sf_last_solver_kernel_ms goes on reporting the last of those)
    hipEvent_t c0 = nullptr, c1 = nullptr;
    if (hipEventCreate(&c0) != hipSuccess || hipEventCreate(&c1) != hipSuccess) {
        (void)hipGetLastError();
        if (c0) (void)hipEventDestroy(c0);
        (void)hipFree(src);
        (void)hipFree(dst);
        return fail(SF_ERR_DEVICE, "sf_microbench_copy: events");
    }
    const size_t n16 = bytes / 16;
    const int grid = (int)std::min<size_t>((n16 + 255) / 256, (size_t)std::max(1, h->max_blocks / std::max(1, h->wg_per_cu)) * 8);
    auto body = [&]() -> int {
        HIP_TRY(hipMemsetAsync(src, 0x3c, bytes, h->stream));
        hipLaunchKernelGGL(sf_copy_kernel, dim3(grid), dim3(256), 0, h->stream, (const float4 *)src, (float4 *)dst, n16);  // warm-up: page tables, clocks
        HIP_TRY(hipEventRecord(c0, h->stream));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(sf_copy_kernel, dim3(grid), dim3(256), 0, h->stream, (const float4 *)src, (float4 *)dst, n16);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c1, h->stream));
        HIP_TRY(hipEventSynchronize(c1));
        HIP_TRY(hipEventElapsedTime(&ms, c0, c1));
        return SF_OK;
    };
    e = body();
    (void)hipEventDestroy(c0);
    (void)hipEventDestroy(c1);
    (void)hipFree(src);
    (void)hipFree(dst);
    if (e) return e;
    *elapsed_ms = ms;
    return SF_OK;
}
int sf_last_solver_kernel_ms(sf_handle *h, float *ms) {
    if (!h || !ms) return fail(SF_ERR_ARG, "null");
    if (!h->solver_timed) return fail(SF_ERR_STATE, "no solver launch yet");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(h->evk1));
    HIP_TRY(hipEventElapsedTime(ms, h->evk0, h->evk1));
    return SF_OK;
}

}  // extern "C"
