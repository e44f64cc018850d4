// imagesequence_driver.cpp — the reference's StaticFusion-imagesequenceassoc.cpp (main loop :57-191) written
// against this repository's C++ mirror of the class surface (include/StaticFusionCompat.hpp) and the I/O
// library (include/sf_io.h), without the GUI and without the OpenGL map: FRAME-TO-FRAME mode, i.e. the
// prediction is the previous filtered frame (the reference renders it from the surfel map; with a surfel
// buffer at hand staticFusion.getPredictedImages(surfels, n, currPose) takes that place).
//
//   g++ -std=c++17 -O2 -Iinclude examples/imagesequence_driver.cpp -o imagesequence_driver
//       -Lstaticfusion_amd/csrc -lsf_hip -lsf_io -Wl,-rpath,$PWD/staticfusion_amd/csrc -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib   (one line)
//   ./imagesequence_driver <dataset dir>/ [trajectory.freiburg]
//
// dataset layout (reference README.md:67-89): rgb/*.png, depth/*.png (16 bit, millimetres), rgbd_assoc.txt
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "StaticFusionCompat.hpp"
#include "sf_io.h"

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <dataset dir>/ [trajectory file]\n", argv[0]);
        return 2;
    }
    std::string dir = argv[1];
    if (dir.empty() || dir.back() != '/') dir += '/';
    const std::string out_path = argc > 2 ? argv[2] : "trajectory.freiburg";
    const unsigned int res_factor = 2;  // :57

    StaticFusionCompat staticFusion(res_factor);  // :59
    // flags and parameters exactly as the driver sets them (:62-79)
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
    staticFusion.kb = 1.5f;
    staticFusion.kz = 1.5f;

    sf_io_assoc *assoc = nullptr;  // loadAssoc (:92-96)
    if (sf_io_assoc_load(dir.c_str(), "rgbd_assoc.txt", &assoc) != SF_IO_OK) {
        std::fprintf(stderr, "dataset absent: %s\n", sf_io_last_error());
        return 3;
    }
    const int n_frames = sf_io_assoc_count(assoc);
    FILE *traj = std::fopen(out_path.c_str(), "w");
    if (!traj || n_frames == 0) return 4;

    auto load = [&](int k) -> bool {  // loadImageFromSequenceAssoc (:105,119,149)
        const char *fd, *fc;
        double ts;
        sf_io_assoc_entry(assoc, k, &ts, &fd, &fc);
        uint8_t *bgr = nullptr;
        uint16_t *depth = nullptr;
        int r, c, r2, c2;
        if (sf_io_imread_color(fc, &bgr, &r, &c) != SF_IO_OK || sf_io_imread_depth16(fd, &depth, &r2, &c2) != SF_IO_OK || r != r2 || c != c2) {
            std::fprintf(stderr, "End of sequence (or image not readable): %s\n", sf_io_last_error());
            sf_io_free(bgr);
            sf_io_free(depth);
            return true;
        }
        const bool end = staticFusion.loadImageFromDecoded(bgr, depth, res_factor);
        sf_io_free(bgr);
        sf_io_free(depth);
        return end;
    };
    auto write_pose = [&](int k, const float pose[16]) {
        double ts;
        sf_io_assoc_entry(assoc, k, &ts, nullptr, nullptr);
        char line[256];
        if (sf_io_trajectory_line(ts, pose, 0, line, sizeof line) > 0) std::fputs(line, traj);
    };

    float currPose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    // bootstrap (:102-137): the first frame becomes the prediction
    if (load(0)) return 5;
    staticFusion.depthPrediction.swap(staticFusion.depthCurrent);          // :107
    staticFusion.intensityPrediction.swap(staticFusion.intensityCurrent);  // :108
    staticFusion.depthCurrent = staticFusion.depthPrediction;              // keep a copy as "current" for the ring
    staticFusion.intensityCurrent = staticFusion.intensityPrediction;
    staticFusion.createImagePyramid(false);                                // uploads depthCurrent for the ring push
    staticFusion.pushBuffers(0);                                           // :129-131
    write_pose(0, currPose);

    int im_count = 0;
    for (int k = 1; k < n_frames; k++) {  // :140-191
        im_count = k;
        if (load(im_count)) break;
        staticFusion.kb = 1.05f;  // without a map the model is never "dense" (:152-163)
        // frame-to-frame: depthPrediction / intensityPrediction are the previous filtered frame (set below)
        staticFusion.getFilteredDepth();          // reconstruction->getFilteredDepth(depth_mm, depthCurrent) :165
        staticFusion.createImagePyramid(true);    // :167
        staticFusion.runSolver(true);             // :169
        if (im_count - staticFusion.bufferLength >= 0) staticFusion.computeResidualsAgainstPreviousImage(im_count);  // :171-173
        staticFusion.buildSegmImage();            // :175
        staticFusion.pushBuffers(im_count);       // :177-179
        float next[16];
        sf_io_pose_compose(currPose, staticFusion.T_odometry.m, next);  // currPose = currPose * T_odometry, Reconstruction.cpp:265
        for (int q = 0; q < 16; q++) currPose[q] = next[q];
        write_pose(im_count, currPose);
        staticFusion.depthPrediction = staticFusion.depthCurrent;       // next prediction := this (filtered) frame
        staticFusion.intensityPrediction = staticFusion.intensityCurrent;
    }
    std::fclose(traj);
    sf_io_assoc_free(assoc);
    std::printf("%d frames -> %s\n", im_count + 1, out_path.c_str());
    return 0;
}
