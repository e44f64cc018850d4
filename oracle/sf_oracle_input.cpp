// sf_oracle_input.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for the input stage
// (SURVEY.md §8(f) rank 1). Scalar restatement of
//   StaticFusion::loadImageFromSequenceAssoc   reference FrontEnd.cpp:216-254 (minus cv::imread)
//   Reconstruction::getFilteredDepth           reference Reconstruction.cpp:722-732
//     filterDepth    -> Shaders/depth_bilateral.frag:34-74
//     metriciseDepth -> Shaders/depth_metric.frag:32-39   (Reconstruction.cpp:337-346)
// PARITY UNPINNED: the reference has no golden vectors for these stages, its filter runs in an
// OpenGL fragment shader (implementation-defined exp() / round() / texel addressing) and OpenCV's
// convertTo is not in this container. Choices made here, all documented in DESIGN.md §10:
//   * texture(gSampler, vec2(cx/cols, cy/rows)) is read as the texel (cx, cy) it names;
//   * exp() is sf_exp_neg() of include/sf_detmath.h (<= 2 ulp, every operation specified);
//   * GLSL round() is roundf() (halves away from zero; the value is non-negative);
//   * uint16 -> float conversion with scale 1/1000 is float(mm) * float(1.0/1000.0) (OpenCV's
//     16u -> 32f cvtScale computes in float), cv::saturate_cast<uchar>(float) is rint + clamp.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../include/sf_detmath.h"
#include "sf_oracle_input.hpp"

namespace sfo {

void load_frame(const uint8_t *color_full, const uint16_t *depth_full, int full_rows, int full_cols, int res, int rows, int cols,
                float *depthCurrent, float *intensityCurrent, uint16_t *depth_mm, uint8_t *color) {
    const float norm_factor = 1.f / 255.f;                 // :218
    const float mm_to_m = float(1.0 / 1000.0);             // :243 convertTo(CV_32FC1, 1.0 / 1000.0)
    (void)full_rows;
    for (int v = 0; v < rows; v++)                         // :228-238 (height == rows, width == cols)
        for (int u = 0; u < cols; u++) {
            const int sr = rows * res - res * v - 1, sc = res * u;  // :231 vertical flip + decimation
            const uint8_t *px = color_full + (size_t(sr) * full_cols + sc) * 3;
            const float r = norm_factor * float(px[0]);
            const float g = norm_factor * float(px[1]);
            const float b = norm_factor * float(px[2]);
            intensityCurrent[v + size_t(u) * rows] = 0.299f * r + 0.587f * g + 0.114f * b;  // :236
            const float back[3] = {r * 255.f, g * 255.f, b * 255.f};                         // :237
            for (int k = 0; k < 3; k++) {
                float t = std::rint(back[k]);
                t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
                color[(size_t(v) * cols + u) * 3 + k] = uint8_t(t);
            }
            const uint16_t mm = depth_full[size_t(sr) * full_cols + sc];
            depthCurrent[v + size_t(u) * rows] = float(mm) * mm_to_m;  // :249
            depth_mm[size_t(v) * cols + u] = mm;                       // :250
        }
}

// Shaders/depth_bilateral.frag:34-74; in / out are rows x cols, row-major (texel (x, y) = column x of row y)
void bilateral_mm(const uint16_t *in, int rows, int cols, float maxD, uint16_t *out) {
    const unsigned gate_hi = unsigned(maxD * 1000.0f);  // :36
    const float sigma_space2_inv_half = 0.024691358f;   // :45
    const float sigma_color2_inv_half = 0.000555556f;   // :46
    const int R = 6, D = R * 2 + 1;                     // :48-49
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const unsigned value = in[size_t(y) * cols + x];
            if (value > gate_hi || value < 300u) {      // :36
                out[size_t(y) * cols + x] = 0;
                continue;
            }
            const int tx = std::min(x - D / 2 + D, cols);  // :51
            const int ty = std::min(y - D / 2 + D, rows);  // :52
            float sum1 = 0.f, sum2 = 0.f;
            for (int cy = std::max(y - D / 2, 0); cy < ty; ++cy)      // :57
                for (int cx = std::max(x - D / 2, 0); cx < tx; ++cx) {  // :59
                    const unsigned tmp = in[size_t(cy) * cols + cx];
                    const float dx = float(x) - float(cx), dy = float(y) - float(cy);
                    const float space2 = dx * dx + dy * dy;            // :66
                    const float dc = float(value) - float(tmp);
                    const float color2 = dc * dc;                      // :67
                    const float weight = sf_exp_neg(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half);  // :69
                    sum1 += float(tmp) * weight;                       // :71
                    sum2 += weight;                                    // :72
                }
            out[size_t(y) * cols + x] = uint16_t(unsigned(std::round(sum1 / sum2)));  // :76
        }
}

// Shaders/depth_metric.frag:32-39; row-major in, float metres out (row-major)
void metricise(const uint16_t *in, int n, float maxD, float *out) {
    const unsigned gate_hi = unsigned(maxD * 1000.0f);
    for (int i = 0; i < n; i++) {
        const unsigned value = in[i];
        out[i] = (value > gate_hi || value < 300u) ? 0.f : float(value) / 1000.0f;
    }
}

}  // namespace sfo
