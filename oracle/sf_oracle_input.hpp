// sf_oracle_input.hpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): input stage, see the .cpp.
#pragma once
#include <cstdint>
namespace sfo {
void load_frame(const uint8_t *color_full, const uint16_t *depth_full, int full_rows, int full_cols, int res, int rows, int cols,
                float *depthCurrent, float *intensityCurrent, uint16_t *depth_mm, uint8_t *color);
void bilateral_mm(const uint16_t *in, int rows, int cols, float maxD, uint16_t *out);
void metricise(const uint16_t *in, int n, float maxD, float *out);
}  // namespace sfo
