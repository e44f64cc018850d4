// sf_oracle_predict.hpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): model prediction, see the .cpp.
#pragma once
#include <cstdint>
namespace sfo {
struct ModelParams {
    float cx, cy, fx, fy, max_depth, conf_low, conf_high;
    int time, max_time, time_delta;
    float extract_max_depth;
};
// surfels: count x 12 floats; t_inv: pose.inverse(), column-major; filtered_mm / color: rows x cols row-major;
// b_img: rows x cols column-major; outputs column-major; returns denseEnough of the low-confidence image
bool predict_from_model(const float *surfels, int count, const float t_inv[16], const ModelParams &p, int rows, int cols,
                        const uint16_t *filtered_mm, const uint8_t *color, const float *b_img, float *depth_pred, float *inten_pred);
}  // namespace sfo
namespace sfo {
// depth_metric / color: rows x cols row-major; depth_filtered / b_img: column-major; pose column-major; returns the count
int init_model_from_frame(const float *depth_metric, const float *depth_filtered, const uint8_t *color, const float *b_img, int rows, int cols,
                          const float pose[16], const ModelParams &p, int time, float *surfels);
}  // namespace sfo
