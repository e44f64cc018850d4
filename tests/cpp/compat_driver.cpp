// The reference drivers' frame loop (StaticFusion-imagesequenceassoc.cpp:102-191) written against
// StaticFusionCompat, in frame-to-frame mode (prediction := previous frame, as in the bootstrap).
// Reads two column-major float32 QVGA pairs from a file written by the test, prints T_odometry.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "StaticFusionCompat.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    StaticFusionCompat staticFusion(2);
    // parameters exactly as the drivers set them (StaticFusion-datasets.cpp:79-94)
    staticFusion.use_motion_filter = true;
    staticFusion.max_iter_per_level = 3;
    staticFusion.previous_speed_const_weight = 0.1f;
    staticFusion.previous_speed_eig_weight = 2.f;
    staticFusion.k_photometric_res = 0.15f;
    staticFusion.irls_delta_threshold = 0.0015f;
    staticFusion.max_iter_irls = 6;
    staticFusion.lambda_reg = 0.35f;
    staticFusion.lambda_prior = 0.5f;
    staticFusion.kc_Cauchy = 0.5f;
    staticFusion.kb = 1.05f;
    staticFusion.kz = 1.5f;

    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    const size_t n = size_t(staticFusion.rows) * staticFusion.cols;
    auto rd = [&](sf::MatrixXf &m) { return std::fread(m.data(), sizeof(float), n, f) == n; };
    if (!rd(staticFusion.depthPrediction) || !rd(staticFusion.intensityPrediction) || !rd(staticFusion.depthCurrent) ||
        !rd(staticFusion.intensityCurrent))
        return 4;
    std::fclose(f);

    staticFusion.createImagePyramid(true);
    staticFusion.runSolver(true);
    staticFusion.buildSegmImage();
    staticFusion.pushBuffers(1);

    for (int r = 0; r < 4; r++)
        std::printf("%.9g %.9g %.9g %.9g\n", staticFusion.T_odometry(r, 0), staticFusion.T_odometry(r, 1),
                    staticFusion.T_odometry(r, 2), staticFusion.T_odometry(r, 3));
    double dyn = 0;
    for (size_t q = 0; q < n; q++) dyn += staticFusion.b_segm_perpixel.data()[q] < 0.5f;
    std::printf("dynamic_fraction %.6f\n", dyn / double(n));

    if (argc >= 3) {  // the drivers' input stage (StaticFusion-imagesequenceassoc.cpp:149,165) on a decoded VGA frame
        FILE *g = std::fopen(argv[2], "rb");
        if (!g) return 5;
        const size_t full = size_t(480) * 640;
        std::vector<uint8_t> color(full * 3);
        std::vector<uint16_t> depth(full);
        if (std::fread(color.data(), 1, full * 3, g) != full * 3 || std::fread(depth.data(), 2, full, g) != full) return 6;
        std::fclose(g);
        if (staticFusion.loadImageFromDecoded(color.data(), depth.data(), 2)) return 7;
        staticFusion.getFilteredDepth();
        double sum_d = 0, sum_i = 0;
        unsigned long long sum_mm = 0;
        for (size_t q = 0; q < n; q++) {
            sum_d += staticFusion.depthCurrent.data()[q];
            sum_i += staticFusion.intensityCurrent.data()[q];
            sum_mm += staticFusion.depth_mm[q];
        }
        std::printf("input_stage %.9f %.9f %llu\n", sum_d, sum_i, sum_mm);
    }
    return 0;
}
