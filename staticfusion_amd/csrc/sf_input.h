// sf_input.h — input stage on gfx950: what the reference drivers run before
// createImagePyramid: loadImageFromSequenceAssoc's decimation (reference FrontEnd.cpp:216-254) and
// Reconstruction::getFilteredDepth (Reconstruction.cpp:722-732) = bilateral filter
// (Shaders/depth_bilateral.frag:34-74) + metricise (Shaders/depth_metric.frag:32-39).
//
// These are plain grid kernels (no cross-pixel reductions): grid = tiles x streams.
// Every float operation is written out with the association of the shader; the build has no FMA
// contraction, the weight is sf_exp_neg() (include/sf_detmath.h), the per-pixel sums run in the
// shader's loop order -> the integer millimetre output is bit-identical to the CPU oracle.
#pragma once
#include "../../include/sf_detmath.h"
#include "sf_device_common.h"

// ---- loader: decimate + vertical flip -----------------------------------------------------------
// one lane per output pixel; lanes run along v (the column-major fast axis of the Eigen images)
__global__ __launch_bounds__(256) void sf_load_frame_kernel(const uint8_t *__restrict__ color_full, const uint16_t *__restrict__ depth_full,
                                                            int full_cols, size_t full_px, int res, int rows, int cols, float *depth_cur,
                                                            float *inten_cur, size_t plane_stride, uint16_t *depth_mm, uint8_t *color,
                                                            int stream0) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;  // v + u * rows
    if (idx >= rows * cols) return;
    const int u = idx / rows, v = idx - u * rows;
    const int sr = rows * res - res * v - 1, sc = res * u;  // FrontEnd.cpp:231
    const size_t src = (size_t)b * full_px + (size_t)sr * full_cols + sc;
    const uint8_t *px = color_full + src * 3;
    const uint8_t c0 = px[0], c1 = px[1], c2 = px[2];
    const float norm_factor = 1.f / 255.f;
    const float r = norm_factor * float(c0), g = norm_factor * float(c1), bl = norm_factor * float(c2);
    const size_t s = (size_t)(stream0 + b);
    inten_cur[s * plane_stride + idx] = 0.299f * r + 0.587f * g + 0.114f * bl;  // :236
    const uint16_t mm = depth_full[src];
    depth_cur[s * plane_stride + idx] = float(mm) * float(1.0 / 1000.0);  // :243,249
    const size_t o = s * (size_t)(rows * cols) + (size_t)v * cols + u;   // cv::Mat, row-major
    depth_mm[o] = mm;                                                    // :250
    const float back[3] = {r * 255.f, g * 255.f, bl * 255.f};            // :237, cv::saturate_cast<uchar>
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t = rintf(back[k]);
        t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
        color[o * 3 + k] = (uint8_t)t;
    }
}

// ---- bilateral filter + metricise -----------------------------------------------------------------
#define BF_R 6
#define BF_D (2 * BF_R + 1)
#define BF_TX 32                 // output tile: 32 columns (x) x 32 rows (y)
#define BF_TY 32
#define BF_LX (BF_TX + 2 * BF_R)  // LDS tile with halo
#define BF_LY (BF_TY + 2 * BF_R)

// One workgroup = one 32 x 32 tile of one stream. Lane t: y = t % 32 (consecutive lanes walk down a
// column: the column-major depthCurrent store is coalesced), x = t / 32 + 8 k, k = 0..3.
__global__ __launch_bounds__(256) void sf_bilateral_kernel(const uint16_t *__restrict__ depth_mm, int rows, int cols, float maxD,
                                                           uint16_t *filtered_mm, float *depth_metric, float *depth_cur,
                                                           size_t plane_stride) {
    __shared__ float tile[BF_LY][BF_LX + 1];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * BF_TX, y0 = blockIdx.y * BF_TY;
    const size_t n = (size_t)rows * cols;
    const uint16_t *in = depth_mm + (size_t)b * n;
    for (int e = threadIdx.x; e < BF_LX * BF_LY; e += 256) {
        const int ly = e / BF_LX, lx = e - ly * BF_LX;
        const int gx = x0 - BF_R + lx, gy = y0 - BF_R + ly;
        const bool in_img = gx >= 0 && gx < cols && gy >= 0 && gy < rows;
        tile[ly][lx] = in_img ? float(in[(size_t)gy * cols + gx]) : 0.f;
    }
    __syncthreads();
    const unsigned gate_hi = (unsigned)(maxD * 1000.0f);    // depth_bilateral.frag:36
    const float sigma_space2_inv_half = 0.024691358f;       // :45
    const float sigma_color2_inv_half = 0.000555556f;       // :46
    const int ly = threadIdx.x & 31, y = y0 + ly;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const int lx = (threadIdx.x >> 5) + 8 * k, x = x0 + lx;
        if (x >= cols || y >= rows) continue;
        const float value = tile[ly + BF_R][lx + BF_R];
        const unsigned uvalue = (unsigned)value;
        unsigned filt = 0;
        if (!(uvalue > gate_hi || uvalue < 300u)) {
            float sum1 = 0.f, sum2 = 0.f;
            // the shader's loops (:57-74) run cy, then cx, over the window clipped to the image
#pragma unroll 1
            for (int dy = -BF_R; dy <= BF_R; dy++) {
                const int cy = y + dy;
                if (cy < 0 || cy >= rows) continue;
                const float fdy = float(y) - float(cy);
                const float dy2 = fdy * fdy;
#pragma unroll
                for (int dx = -BF_R; dx <= BF_R; dx++) {
                    const int cx = x + dx;
                    if (cx < 0 || cx >= cols) continue;
                    const float tmp = tile[ly + BF_R + dy][lx + BF_R + dx];
                    const float fdx = float(x) - float(cx);
                    const float space2 = fdx * fdx + dy2;                       // :66
                    const float dc = value - tmp;
                    const float color2 = dc * dc;                                // :67
                    const float weight = sf_exp_neg(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half);  // :69
                    sum1 += tmp * weight;                                        // :71
                    sum2 += weight;                                              // :72
                }
            }
            filt = (unsigned)roundf(sum1 / sum2);                                // :76
        }
        const size_t o = (size_t)b * n + (size_t)y * cols + x;
        filtered_mm[o] = (uint16_t)filt;
        // metricise (depth_metric.frag:32-39): raw -> DEPTH_METRIC, filtered -> depthCurrent (cv2eigen: column-major)
        depth_metric[o] = (uvalue > gate_hi || uvalue < 300u) ? 0.f : value / 1000.0f;
        depth_cur[(size_t)b * plane_stride + (size_t)y + (size_t)x * rows] = (filt > gate_hi || filt < 300u) ? 0.f : float(filt) / 1000.0f;
    }
}
