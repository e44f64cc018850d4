// sf_oracle_fusion.hpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): surfel-map fusion, see the .cpp.
#pragma once
#include <cstdint>
#include <vector>

#include "sf_oracle_predict.hpp"
namespace sfo {
struct FrameImages {           // what the stream holds after sf_load_frame + sf_filter_depth + a solve
    const float *depth_metric;    // rows x cols row-major    (DEPTH_METRIC)
    const float *depth_filtered;  // column-major             (DEPTH_METRIC_FILTERED = depthCurrent)
    const uint8_t *color;         // rows x cols x 3 row-major (RGB)
    const float *b_img;           // column-major             (WEIGHT = b_segm_perpixel)
    int rows, cols;
};
struct SurfelMap {  // GlobalModel + the pose / tick bookkeeping of Reconstruction
    int capacity = 0;
    std::vector<float> surfels;  // count x 12
    int count = 0;
    float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // currPose, column-major
    int tick = 1;
    std::vector<uint32_t> index_map;  // 4 rows x 4 cols, row-major: the index texture of the LAST predictIndices
    int stats[4] = {0, 0, 0, 0};      // emitted by the data pass, of those merged (updateId 1), surfels updated, count after clean
};
// IndexMap::predictIndices: index texture only (0 = empty or surfel 0, as in the reference)
void predict_indices(const float *surfels, int count, const float t_inv[16], const ModelParams &p, int rows, int cols, int time,
                     uint32_t *index_map);
// Reconstruction::fuseFrame after the uploads and depth filters; returns 0, or 1 when the map overflowed its capacity (truncated)
int fuse_frame(SurfelMap &m, const FrameImages &f, const float *in_pose, float weight_multiplier, const ModelParams &p);
}  // namespace sfo
