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
// One workgroup = a 32 (u) x 32 (v) tile of one stream. Phase 1 runs lanes along u: the strided source
// reads (6 / 4 bytes apart) and the row-major depth_mm / color stores are contiguous per half-wave.
// The two float images are column-major (Eigen): they go through an LDS transpose and are stored with
// lanes along v.
#define LD_T 32
__global__ __launch_bounds__(256) void sf_load_frame_kernel(const uint8_t *__restrict__ color_full, const uint16_t *__restrict__ depth_full,
                                                            int full_cols, size_t full_px, int res, int rows, int cols, float *depth_cur,
                                                            float *inten_cur, size_t plane_stride, uint16_t *depth_mm, uint8_t *color,
                                                            int stream0) {
    __shared__ float t_d[LD_T][LD_T + 1], t_i[LD_T][LD_T + 1];
    const int b = blockIdx.z;
    const size_t s = (size_t)(stream0 + b);
    const int u0 = blockIdx.x * LD_T, v0 = blockIdx.y * LD_T;
    const float norm_factor = 1.f / 255.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ul = threadIdx.x & 31, vl = (threadIdx.x >> 5) + 8 * k;
        const int u = u0 + ul, v = v0 + vl;
        if (u >= cols || v >= rows) continue;
        const int sr = rows * res - res * v - 1, sc = res * u;  // FrontEnd.cpp:231
        const size_t src = (size_t)b * full_px + (size_t)sr * full_cols + sc;
        const uint8_t *px = color_full + src * 3;
        const uint8_t c0 = px[0], c1 = px[1], c2 = px[2];
        const float r = norm_factor * float(c0), g = norm_factor * float(c1), bl = norm_factor * float(c2);
        t_i[vl][ul] = 0.299f * r + 0.587f * g + 0.114f * bl;  // :236
        const uint16_t mm = depth_full[src];
        t_d[vl][ul] = float(mm) * float(1.0 / 1000.0);  // :243,249
        const size_t o = s * (size_t)(rows * cols) + (size_t)v * cols + u;  // cv::Mat, row-major
        depth_mm[o] = mm;                                                   // :250
        const float back[3] = {r * 255.f, g * 255.f, bl * 255.f};           // :237, cv::saturate_cast<uchar>
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float t = rintf(back[q]);
            t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
            color[o * 3 + q] = (uint8_t)t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int vl = threadIdx.x & 31, ul = (threadIdx.x >> 5) + 8 * k;
        const int u = u0 + ul, v = v0 + vl;
        if (u >= cols || v >= rows) continue;
        const size_t o = s * plane_stride + (size_t)v + (size_t)u * rows;
        depth_cur[o] = t_d[vl][ul];
        inten_cur[o] = t_i[vl][ul];
    }
}

// ---- bilateral filter + metricise -----------------------------------------------------------------
#define BF_R 6
#define BF_D (2 * BF_R + 1)
#define BF_TX 32                 // output tile: 32 columns (x) x 32 rows (y)
#define BF_TY 32
#define BF_LX (BF_TX + 2 * BF_R)  // LDS tile with halo
#define BF_LY (BF_TY + 2 * BF_R)
// A tap outside the image is SKIPPED by the shader's clipped loops (:51-59). The halo stores this
// value there instead: color2 = (value - 1e9)^2 makes the exponent > 87, sf_exp_neg() returns exactly
// 0, and adding tmp * 0 = 0 and 0 changes neither sum -- the same result as skipping, without
// per-tap bounds tests.
#define BF_OUTSIDE 1.0e9f

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
        tile[ly][lx] = in_img ? float(in[(size_t)gy * cols + gx]) : BF_OUTSIDE;
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
            // the shader's loops (:57-74): cy ascending, then cx ascending; offsets are compile-time constants
#pragma unroll
            for (int dy = -BF_R; dy <= BF_R; dy++) {
#pragma unroll
                for (int dx = -BF_R; dx <= BF_R; dx++) {
                    const float tmp = tile[ly + BF_R + dy][lx + BF_R + dx];
                    const float space2 = float(dx * dx) + float(dy * dy);       // :66 (exact small integers)
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
