// staticfusion_headless.cpp — the reference's StaticFusion-imagesequenceassoc.cpp in full (main loop :57-191:
// solve -> fuseFrame -> getPredictedImages from the growing surfel map) without the GUI and without OpenGL, written
// against this repository's C++ mirrors of the two classes the driver uses (include/StaticFusionCompat.hpp:
// StaticFusionCompat = class StaticFusion, ReconstructionCompat = class Reconstruction) and the I/O library.
//
//   g++ -std=c++17 -O2 -Iinclude examples/staticfusion_headless.cpp -o staticfusion_headless
//       -Lstaticfusion_amd/csrc -lsf_hip -lsf_io -Wl,-rpath,$PWD/staticfusion_amd/csrc -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib   (one line)
//   ./staticfusion_headless <dataset dir>/ [output prefix]      -> <prefix>.freiburg (trajectory), <prefix>.ply (map)
//
// Two deliberate differences from the reference's main(): it starts at the association file's FIRST entry (the reference sets
// im_count = 1, :83, and never reads entry 0), and it writes a trajectory line for the bootstrap frame too (identity), so
// that the file has one line per frame read.
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

#include "StaticFusionCompat.hpp"
#include "sf_io.h"

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <dataset dir>/ [output prefix]\n", argv[0]);
        return 2;
    }
    std::string dir = argv[1];
    if (dir.empty() || dir.back() != '/') dir += '/';
    const std::string prefix = argc > 2 ? argv[2] : "sf-mesh";  // FrontEnd.cpp:169
    const unsigned int res_factor = 2;                          // :57

    StaticFusionCompat staticFusion(res_factor);                // :59
    ReconstructionCompat reconstruction(staticFusion, std::numeric_limits<int>::max(), 0.25f, 4.5f);  // FrontEnd.cpp:165-180
    staticFusion.use_motion_filter = true;                      // :62-79
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
    bool denseModel = false, modelInitialised = false;          // :78-79

    sf_io_assoc *assoc = nullptr;  // loadAssoc (:92-96)
    if (sf_io_assoc_load(dir.c_str(), "rgbd_assoc.txt", &assoc) != SF_IO_OK) {
        std::fprintf(stderr, "dataset absent: %s\n", sf_io_last_error());
        return 3;
    }
    const int n_frames = sf_io_assoc_count(assoc);
    if (n_frames < 2) return 4;
    FILE *traj = std::fopen((prefix + ".freiburg").c_str(), "w");
    if (!traj) return 4;

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
    auto fuse = [&](int im_count) {  // :133-135 / :181-183
        reconstruction.fuseFrame(staticFusion.color_full.data(), staticFusion.depth_mm.data(), staticFusion.b_segm_perpixel.data(), im_count,
                                 &staticFusion.T_odometry, nullptr, 1.f);
        write_pose(im_count, reconstruction.getCurrPose().m);
    };

    // ---- bootstrap (:102-137) ----
    int im_count = 0;
    if (load(im_count)) return 5;
    const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    write_pose(0, identity);
    staticFusion.depthPrediction.swap(staticFusion.depthCurrent);          // :107-108
    staticFusion.intensityPrediction.swap(staticFusion.intensityCurrent);
    staticFusion.depthCurrent = staticFusion.depthPrediction;              // the ring slot of frame 0 holds the prediction (:111-113)
    staticFusion.intensityCurrent = staticFusion.intensityPrediction;
    staticFusion.createImagePyramid(false);
    staticFusion.pushBuffers(im_count);
    im_count += 1;
    if (load(im_count)) return 5;                                          // :119
    staticFusion.createImagePyramid(true);                                 // :121
    staticFusion.kb = 1.05f;                                               // :123
    staticFusion.runSolver(true);                                          // :125
    staticFusion.buildSegmImage();                                         // :127
    staticFusion.pushBuffers(im_count);                                    // :129-131
    fuse(im_count);                                                        // :135: tick 1 -> GlobalModel::initialise

    // ---- main loop (:140-191) ----
    while (im_count + 1 < n_frames) {
        im_count += 1;
        denseModel = reconstruction.checkIfDenseEnough();                  // :151 (about the previous prediction)
        if (!denseModel && !modelInitialised) {                            // :153-163
            staticFusion.kb = 1.05f;
            modelInitialised = true;
        } else {
            staticFusion.kb = 1.5f;
            modelInitialised = true;
        }
        // getPredictedImages reads the PREVIOUS frame's filtered depth / colour / b for its fill-in: before the load.
        // (the reference loads first, :149, but its GL textures still hold the previous frame until fuseFrame uploads)
        reconstruction.getPredictedImages(staticFusion.depthPrediction, staticFusion.intensityPrediction);  // :164
        if (load(im_count)) break;                                         // :149
        reconstruction.getFilteredDepth(staticFusion.depth_mm, staticFusion.depthCurrent);                 // :165
        staticFusion.createImagePyramid(true);                             // :167
        staticFusion.runSolver(true);                                      // :169
        if (im_count - staticFusion.bufferLength >= 0) staticFusion.computeResidualsAgainstPreviousImage(im_count);  // :171-173
        staticFusion.buildSegmImage();                                     // :175
        staticFusion.pushBuffers(im_count);                                // :177-179
        fuse(im_count);                                                    // :183
    }
    std::fclose(traj);
    sf_io_assoc_free(assoc);
    const std::vector<float> map = reconstruction.downloadMap();           // savePly (FrontEnd.cpp:1290, Reconstruction.cpp:358-455)
    const int vertices = sf_io_save_ply((prefix + ".ply").c_str(), map.data(), int(map.size() / 12), reconstruction.getConfidenceThreshold());
    std::printf("%d frames, %u surfels (%d above the confidence threshold) -> %s.freiburg, %s.ply\n", im_count + 1, reconstruction.lastCount(), vertices,
                prefix.c_str(), prefix.c_str());
    return vertices < 0 ? 6 : 0;
}
