// sf_hip_input.hip — libsf_hip.so, the input stage of include/sf.h: loader decimation / flip /
// RGB -> intensity and the bilateral depth filter + metricise on the GPU (kernels: sf_input.h).
#include "sf_host.h"
#include "sf_input.h"

extern "C" {

// ---- input stage ------------------------------------------------------------------------------
int input_alloc(sf_handle *h) {
    if (h->in_depth_mm) return SF_OK;
    const size_t n = (size_t)h->k.n0 * h->k.batch;
    if (int e = dev_alloc(h, &h->in_depth_mm, n)) return e;
    if (int e = dev_alloc(h, &h->in_filtered_mm, n)) return e;
    if (int e = dev_alloc(h, &h->in_depth_metric, n)) return e;
    if (int e = dev_alloc(h, &h->in_color, n * 3)) return e;
    return SF_OK;
}
static int check_full(sf_handle *h, int full_rows, int full_cols, int res) {
    if (res < 1 || full_rows != h->k.rows * res || full_cols != h->k.cols * res)
        return fail(SF_ERR_ARG, "full resolution / res_factor do not match the handle");
    return SF_OK;
}
static int launch_load(sf_handle *h, const uint8_t *d_color, const uint16_t *d_depth, int full_rows, int full_cols, int res,
                       int stream0, int count) {
    const dim3 grid((h->k.cols + LD_T - 1) / LD_T, (h->k.rows + LD_T - 1) / LD_T, count);
    hipLaunchKernelGGL(sf_load_frame_kernel, grid, dim3(256), 0, h->stream, d_color, d_depth, full_cols, (size_t)full_rows * full_cols,
                       res, h->k.rows, h->k.cols, h->k.pyr_new[0], h->k.pyr_new[1], (size_t)h->k.n_tot, h->in_depth_mm, h->in_color,
                       stream0);
    HIP_TRY(hipGetLastError());
    h->have_frame = true;
    return SF_OK;
}
int sf_load_frame(sf_handle *h, int stream, const uint8_t *color_full, const uint16_t *depth_full, int full_rows, int full_cols,
                  int res_factor) {
    if (int e = check_stream(h, stream)) return e;
    if (!color_full || !depth_full) return fail(SF_ERR_ARG, "null image");
    if (int e = check_full(h, full_rows, full_cols, res_factor)) return e;
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    const size_t px = (size_t)full_rows * full_cols;
    if (h->stage_px < px) {
        if (int e = dev_alloc(h, &h->stage_color, px * 3)) return e;
        if (int e = dev_alloc(h, &h->stage_depth, px)) return e;
        h->stage_px = px;
    }
    HIP_TRY(hipMemcpyAsync(h->stage_color, color_full, px * 3, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->stage_depth, depth_full, px * 2, hipMemcpyHostToDevice, h->stream));
    if (int e = launch_load(h, h->stage_color, h->stage_depth, full_rows, full_cols, res_factor, stream, 1)) return e;
    HIP_TRY(hipStreamSynchronize(h->stream));  // the staging frame is reused by the next call; the host buffers are free again
    return SF_OK;
}
int sf_load_frame_device(sf_handle *h, const void *d_color_full, const void *d_depth_full, int full_rows, int full_cols,
                         int res_factor) {
    if (!h || !d_color_full || !d_depth_full) return fail(SF_ERR_ARG, "null");
    if (int e = check_full(h, full_rows, full_cols, res_factor)) return e;
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    return launch_load(h, (const uint8_t *)d_color_full, (const uint16_t *)d_depth_full, full_rows, full_cols, res_factor, 0,
                       h->k.batch);
}
int sf_set_depth_cutoff(sf_handle *h, float m) {
    if (!h || !(m > 0.f)) return fail(SF_ERR_ARG, "bad cutoff");
    h->depth_cutoff = m;
    return SF_OK;
}
int sf_filter_depth(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_filter_depth needs sf_load_frame first");
    HIP_TRY(hipSetDevice(h->device));
    const dim3 grid((h->k.cols + BF_TX - 1) / BF_TX, (h->k.rows + BF_TY - 1) / BF_TY, h->k.batch);
    hipLaunchKernelGGL(sf_bilateral_kernel, grid, dim3(256), 0, h->stream, h->in_depth_mm, h->k.rows, h->k.cols, h->depth_cutoff,
                       h->in_filtered_mm, h->in_depth_metric, h->k.pyr_new[0], (size_t)h->k.n_tot);
    HIP_TRY(hipGetLastError());
    return SF_OK;
}
int sf_get_current(sf_handle *h, int stream, float *depth, float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    const size_t bytes = sizeof(float) * h->k.n0;
    if (depth)
        if (int e = d2h(h, depth, h->k.pyr_new[0] + (size_t)stream * h->k.n_tot, bytes)) return e;
    if (intensity)
        if (int e = d2h(h, intensity, h->k.pyr_new[1] + (size_t)stream * h->k.n_tot, bytes)) return e;
    return SF_OK;
}
int sf_get_input_image(sf_handle *h, int stream, int which, void *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "no frame loaded");
    const size_t n = h->k.n0, o = (size_t)stream * n;
    switch (which) {
        case SF_IN_DEPTH_MM: return d2h(h, out, h->in_depth_mm + o, n * 2);
        case SF_IN_DEPTH_FILTERED_MM: return d2h(h, out, h->in_filtered_mm + o, n * 2);
        case SF_IN_DEPTH_METRIC: return d2h(h, out, h->in_depth_metric + o, n * 4);
        case SF_IN_COLOR: return d2h(h, out, h->in_color + o * 3, n * 3);
        default: return fail(SF_ERR_ARG, "bad selector");
    }
}
int sf_timed_input_stage(sf_handle *h, const void *d_color_full, const void *d_depth_full, int full_rows, int full_cols,
                         int res_factor, int calls, float *elapsed_ms) {
    if (!h || calls < 1) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    for (int c = 0; c < calls; c++) {
        if (int e = sf_load_frame_device(h, d_color_full, d_depth_full, full_rows, full_cols, res_factor)) return e;
        if (int e = sf_filter_depth(h)) return e;
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return SF_OK;
}

}  // extern "C"
