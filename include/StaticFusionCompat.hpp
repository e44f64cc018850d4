// StaticFusionCompat.hpp — the reference's `class StaticFusion` surface over the C ABI (include/sf.h).
//
// Host-side mirror of the ONE path this repository replaces.  Member and method names, argument
// meaning and call order are the reference's (StaticFusion.h:66-189), so that the reference drivers
// (StaticFusion-datasets.cpp:79-199) and the untouched OpenGL map (`Reconstruction`) can be pointed
// at this class instead of the CPU solver: write depthCurrent / intensityCurrent / depthPrediction /
// intensityPrediction and the parameter members, call createImagePyramid(true), runSolver(true),
// computeResidualsAgainstPreviousImage(i), buildSegmImage(), read T_odometry / b_segm_perpixel /
// clusterAllocation[0].  What is NOT here, on purpose: the GUI / Reconstruction members and
// updateGUI (the solver object must not own a GL context), the image loaders (OpenCV), dead members.
//
// Matrices: `sf::Matrix<T>` is a minimal column-major matrix with Eigen's (row, col) indexing and
// `.data()`; where Eigen is available, `Eigen::Map<Eigen::MatrixXf>(m.data(), m.rows(), m.cols())`
// views it without a copy (same storage order).
//
// Unlike the reference this class is a batch of ONE stream per object; a throughput harness uses the
// C ABI directly with batch > 1.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sf.h"

namespace sf {

template <class T>
class Matrix {  // column-major, like Eigen::Matrix<T, Dynamic, Dynamic>
   public:
    Matrix() = default;
    Matrix(int r, int c, T v = T()) : rows_(r), cols_(c), d_(size_t(r) * c, v) {}
    void resize(int r, int c) { rows_ = r; cols_ = c; d_.resize(size_t(r) * c); }
    void fill(T v) { for (auto &x : d_) x = v; }
    T &operator()(int v, int u) { return d_[size_t(v) + size_t(u) * rows_]; }
    const T &operator()(int v, int u) const { return d_[size_t(v) + size_t(u) * rows_]; }
    T *data() { return d_.data(); }
    const T *data() const { return d_.data(); }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    void swap(Matrix &o) { std::swap(rows_, o.rows_); std::swap(cols_, o.cols_); d_.swap(o.d_); }
   private:
    int rows_ = 0, cols_ = 0;
    std::vector<T> d_;
};
using MatrixXf = Matrix<float>;
using MatrixXi = Matrix<int32_t>;

struct Matrix4f {  // column-major 4x4 like Eigen::Matrix4f
    float m[16];
    float &operator()(int r, int c) { return m[r + 4 * c]; }
    float operator()(int r, int c) const { return m[r + 4 * c]; }
    const float *data() const { return m; }
};

}  // namespace sf

class StaticFusionCompat {
   public:
    // ---- images the drivers write (StaticFusion.h:90-91) ----
    sf::MatrixXf depthCurrent, intensityCurrent;
    sf::MatrixXf depthPrediction, intensityPrediction;

    // ---- outputs the drivers read ----
    sf::Matrix4f T_odometry;                       // StaticFusion.h:106
    float twist_odometry[6], twist_odometry_old[6];
    float b_segm[SF_NUM_CLUSTERS];                 // StaticFusion.h:164
    sf::MatrixXf b_segm_perpixel;                  // StaticFusion.h:165
    std::vector<sf::MatrixXi> clusterAllocation;   // StaticFusion.h:146
    float perClusterAverageResidual[SF_NUM_CLUSTERS];

    // ---- parameters (StaticFusion.h:116-140,168-170; written by the drivers) ----
    unsigned int rows, cols, width, height, ctf_levels;
    bool use_motion_filter;
    float previous_speed_const_weight, previous_speed_eig_weight;
    unsigned int max_iter_irls, max_iter_per_level;
    float k_photometric_res, irls_delta_threshold, kc_Cauchy, kb;
    float lambda_reg, lambda_prior, kz;
    float fovh;
    int bufferLength = SF_HISTORY;

    // reference: StaticFusion(unsigned int res_factor), FrontEnd.cpp:52-181
    explicit StaticFusionCompat(unsigned int res_factor = 2, int device = 0) {
        rows = height = 480 / res_factor;
        cols = width = 640 / res_factor;
        {   // the library this binary is linked against must have been built from this header (sf_abi_version, sf.h)
            int sp = 0, ss = 0, slots = 0;
            if (sf_abi_version(&sp, &ss, &slots) != SF_ABI_VERSION || sp != int(sizeof(sf_params)) || ss != int(sizeof(sf_frame_stats)))
                throw std::runtime_error("libsf_hip.so was built from another version of include/sf.h");
        }
        sf_params p;
        sf_ctor_params(&p);
        check(sf_create(&p, int(rows), int(cols), 1, device, &h_), "sf_create");
        sf_get_params(h_, &p);
        pull_params(p);
        depthCurrent.resize(rows, cols); intensityCurrent.resize(rows, cols);
        depthPrediction.resize(rows, cols); intensityPrediction.resize(rows, cols);
        depthCurrent.fill(0.f); intensityCurrent.fill(0.f); depthPrediction.fill(0.f); intensityPrediction.fill(0.f);
        b_segm_perpixel.resize(rows, cols);
        b_segm_perpixel.fill(0.5f);
        clusterAllocation.resize(ctf_levels);
        for (unsigned L = 0; L < ctf_levels; L++) clusterAllocation[L].resize(sf_level_rows(h_, L), sf_level_cols(h_, L));
        for (int i = 0; i < 16; i++) T_odometry.m[i] = (i % 5 == 0) ? 1.f : 0.f;
        for (int i = 0; i < 6; i++) twist_odometry[i] = twist_odometry_old[i] = 0.f;
        for (int l = 0; l < SF_NUM_CLUSTERS; l++) b_segm[l] = 0.5f;
    }
    ~StaticFusionCompat() { sf_destroy(h_); }
    StaticFusionCompat(const StaticFusionCompat &) = delete;
    StaticFusionCompat &operator=(const StaticFusionCompat &) = delete;

    // reference: createImagePyramid(bool old_im), FrontEnd.cpp:256-391
    void createImagePyramid(bool old_im) {
        push_params();
        if (old_im)
            check(sf_set_prediction(h_, 0, depthPrediction.data(), intensityPrediction.data()), "set_prediction");
        else
            check(sf_set_current(h_, 0, depthCurrent.data(), intensityCurrent.data()), "set_current");
        check(sf_build_pyramid(h_, old_im ? 1 : 0), "build_pyramid");
    }
    // reference: runSolver(bool create_image_pyr), FrontEnd.cpp:1071-1146
    void runSolver(bool create_image_pyr) {
        push_params();
        if (create_image_pyr) check(sf_set_current(h_, 0, depthCurrent.data(), intensityCurrent.data()), "set_current");
        check(sf_run_solver(h_, create_image_pyr ? 1 : 0), "run_solver");
        check(sf_get_T(h_, 0, T_odometry.m), "get_T");
        check(sf_get_twist(h_, 0, twist_odometry), "get_twist");
        check(sf_get_twist_old(h_, 0, twist_odometry_old), "get_twist_old");
        check(sf_get_b(h_, 0, b_segm), "get_b");
        check(sf_get_labels(h_, 0, 0, clusterAllocation[0].data()), "get_labels");
    }
    // reference: kMeans3DCoord(), KMeans.cpp:137-295 (+ createClustersPyramidUsingKMeans)
    void kMeans3DCoord() {
        check(sf_kmeans(h_), "kmeans");
        for (unsigned L = 0; L < ctf_levels; L++) check(sf_get_labels(h_, 0, int(L), clusterAllocation[L].data()), "get_labels");
    }
    // reference: computeResidualsAgainstPreviousImage(int index), FrontEnd.cpp:896-1069
    void computeResidualsAgainstPreviousImage(int index) {
        check(sf_residuals_vs_history(h_, index), "residuals");
        check(sf_get_cluster_residuals(h_, 0, perClusterAverageResidual), "get_cluster_residuals");
    }
    // reference: buildSegmImage(), SegmentationBackground.cpp:176-197
    void buildSegmImage() {
        check(sf_build_segm_image(h_), "build_segm_image");
        check(sf_get_b_image(h_, 0, b_segm_perpixel.data()), "get_b_image");
    }
    // reference: loadImageFromSequenceAssoc(depthFile, rgbFile, res_factor), FrontEnd.cpp:216-254, after its two
    // cv::imread calls: `color` is the decoded full-resolution 3-channel image (decoder byte order), `depth`
    // the decoded 16-bit image in millimetres, both (height*res_factor) x (width*res_factor), row-major.
    // Fills intensityCurrent, depthCurrent, depth_mm and color_full like the reference does.
    bool loadImageFromDecoded(const uint8_t *color, const uint16_t *depth, unsigned int res_factor) {
        if (!color || !depth) return true;  // "End of sequence (or color image not found...)" (:222-226)
        check(sf_load_frame(h_, 0, color, depth, int(height * res_factor), int(width * res_factor), int(res_factor)), "load_frame");
        depth_mm.resize(size_t(rows) * cols);
        color_full.resize(size_t(rows) * cols * 3);
        check(sf_get_current(h_, 0, depthCurrent.data(), intensityCurrent.data()), "get_current");
        check(sf_get_input_image(h_, 0, SF_IN_DEPTH_MM, depth_mm.data()), "get_input_image");
        check(sf_get_input_image(h_, 0, SF_IN_COLOR, color_full.data()), "get_input_image");
        return false;
    }
    // reference: reconstruction->getFilteredDepth(depth_mm, depthCurrent), Reconstruction.cpp:722-732
    // (bilateral filter + metricise of the frame loaded last); depthCurrent := the filtered depth in metres
    void getFilteredDepth() {
        check(sf_set_depth_cutoff(h_, depth_max), "set_depth_cutoff");
        check(sf_filter_depth(h_), "filter_depth");
        check(sf_get_current(h_, 0, depthCurrent.data(), nullptr), "get_current");
    }
    // reference: reconstruction->getPredictedImages(depthPrediction, intensityPrediction), Reconstruction.cpp:628-720,
    // without OpenGL: `surfels` is the global model in the reference's vertex layout (count x 12 floats),
    // `currPose` the camera pose. Call it where the reference does: before the new frame is loaded.
    void getPredictedImages(const float *surfels, int count, const sf::Matrix4f &currPose, int tick = 0) {
        sf_model_params mp;
        check(sf_default_model_params(h_, &mp), "default_model_params");
        mp.time = mp.max_time = tick;
        check(sf_predict_from_model(h_, 0, surfels, count, currPose.m, &mp), "predict_from_model");
        check(sf_get_prediction(h_, 0, depthPrediction.data(), intensityPrediction.data()), "get_prediction");
    }
    std::vector<uint16_t> depth_mm;   // cv::Mat depth_mm  (StaticFusion.h:71), rows x cols, row-major
    std::vector<uint8_t> color_full;  // cv::Mat color_full (StaticFusion.h:71), rows x cols x 3
    float depth_max = 4.5f;           // StaticFusion.h:77, FrontEnd.cpp:168

    // the drivers' ring buffer writes: depthBuffer[i%5] = depthCurrent; intensityBuffer[i%5] = ...;
    // odomBuffer[i%5] = T_odometry   (StaticFusion-datasets.cpp:182-184)
    void pushBuffers(int im_count) { check(sf_push_history(h_, im_count), "push_history"); }
    // no counterpart in the reference: with SF_VARIANT=cluster (24 workgroups share the one stream) a frame whose workgroups were not
    // all resident reports SF_STATUS_SYNC_TIMEOUT and the solver keeps the state of its last good frame; this puts it back into service
    void clearSyncTimeout() { check(sf_clear_sync_timeout(h_), "clear_sync_timeout"); }

    sf_handle *handle() { return h_; }

   private:
    sf_handle *h_ = nullptr;

    static void check(int rc, const char *what) {
        if (rc != SF_OK) throw std::runtime_error(std::string(what) + ": " + sf_last_error());
    }
    void pull_params(const sf_params &p) {
        ctf_levels = p.ctf_levels; use_motion_filter = p.use_motion_filter != 0;
        previous_speed_const_weight = p.previous_speed_const_weight; previous_speed_eig_weight = p.previous_speed_eig_weight;
        max_iter_irls = p.max_iter_irls; max_iter_per_level = p.max_iter_per_level;
        k_photometric_res = p.k_photometric_res; irls_delta_threshold = p.irls_delta_threshold;
        kc_Cauchy = p.kc_Cauchy; kb = p.kb; lambda_reg = p.lambda_reg; lambda_prior = p.lambda_prior; kz = p.kz; fovh = p.fovh;
    }
    void push_params() {  // the drivers set public members at any time; forward them before each call
        sf_params p;
        sf_get_params(h_, &p);
        p.ctf_levels = int(ctf_levels); p.use_motion_filter = use_motion_filter ? 1 : 0;
        p.previous_speed_const_weight = previous_speed_const_weight; p.previous_speed_eig_weight = previous_speed_eig_weight;
        p.max_iter_irls = int(max_iter_irls); p.max_iter_per_level = int(max_iter_per_level);
        p.k_photometric_res = k_photometric_res; p.irls_delta_threshold = irls_delta_threshold;
        p.kc_Cauchy = kc_Cauchy; p.kb = kb; p.lambda_reg = lambda_reg; p.lambda_prior = lambda_prior; p.kz = kz; p.fovh = fovh;
        check(sf_set_params(h_, &p), "set_params");
    }
};

// ReconstructionCompat — the reference's `class Reconstruction` (Reconstruction.h:44-235) as the three drivers use it
// (staticFusion.reconstruction->...), over the HIP surfel map (sf_map_*, include/sf.h) instead of OpenGL. It works on
// the frame the front end holds: StaticFusionCompat::loadImageFromDecoded + buildSegmImage have already put RGB, depth
// and the weighted image into the handle, so fuseFrame's three image pointers are accepted for signature compatibility
// and not read (all three drivers pass exactly the front end's color_full / depth_mm / b_segm_perpixel).
class ReconstructionCompat {
   public:
    // reference: Reconstruction(timeDelta, confidence, depthCut, fileName, clusters), Reconstruction.cpp:21-51; the front
    // end constructs it with (INT_MAX, 0.25, 4.5, "sf-mesh", 24) (FrontEnd.cpp:165-180). capacity: surfels the map can
    // hold (0 = the reference's 3072 x 3072).
    explicit ReconstructionCompat(StaticFusionCompat &front, int timeDelta = 200, float confidence = 10.f, float depthCut = 3.f, int capacity = 0)
        : front_(front), timeDelta_(timeDelta), confidenceThreshold_(confidence), depthCutoff_(depthCut) {
        check(sf_map_create(front_.handle(), capacity, &map_), "map_create");
    }
    ~ReconstructionCompat() { sf_map_destroy(map_); }
    ReconstructionCompat(const ReconstructionCompat &) = delete;
    ReconstructionCompat &operator=(const ReconstructionCompat &) = delete;

    // reference: fuseFrame(rgb, depth, weightedImage, timestamp, inPose, gtPose, weightMultiplier), Reconstruction.cpp:235-325.
    // Call it where the drivers do: after buildSegmImage and the ring-buffer writes of the frame.
    void fuseFrame(const unsigned char * /*rgb*/, const unsigned short * /*depth*/, const float * /*weightedImage*/, const int64_t &timestamp,
                   const sf::Matrix4f *inPose, const sf::Matrix4f *gtPose = nullptr, const float weightMultiplier = 1.f) {
        check(sf_set_depth_cutoff(front_.handle(), depthCutoff_), "set_depth_cutoff");
        check(sf_filter_depth(front_.handle()), "filter_depth");  // filterDepth(); metriciseDepth(); (:246-247)
        const sf_model_params mp = params();
        check(sf_map_fuse_frame(front_.handle(), 0, map_, inPose ? inPose->m : nullptr, weightMultiplier, &mp), "map_fuse_frame");
        int tick = 0;
        sf::Matrix4f pose;
        check(sf_map_info(map_, nullptr, &tick, pose.m, nullptr), "map_info");
        poseGraph.push_back(std::make_pair((unsigned long long)(tick - 1), pose));  // :315
        if (gtPose) gtPoseGraph.push_back(std::make_pair((unsigned long long)(tick - 1), *gtPose));
        poseLogTimes.push_back(timestamp);                                          // :321
    }
    // reference: getPredictedImages(depthPrediction, intensityPrediction), Reconstruction.cpp:628-720
    void getPredictedImages(sf::MatrixXf &depthPrediction, sf::MatrixXf &intensityPrediction) {
        const sf_model_params mp = params();
        check(sf_map_predict(front_.handle(), 0, map_, &mp), "map_predict");
        check(sf_get_prediction(front_.handle(), 0, depthPrediction.data(), intensityPrediction.data()), "get_prediction");
    }
    // reference: getFilteredDepth(depth_mm, depthCurrent), Reconstruction.cpp:722-732 (of the frame the front end loaded last)
    void getFilteredDepth(const std::vector<uint16_t> & /*depth_mm*/, sf::MatrixXf &depthCurrent) {
        check(sf_set_depth_cutoff(front_.handle(), depthCutoff_), "set_depth_cutoff");
        check(sf_filter_depth(front_.handle()), "filter_depth");
        check(sf_get_current(front_.handle(), 0, depthCurrent.data(), nullptr), "get_current");
    }
    // reference: checkIfDenseEnough(), Reconstruction.cpp:762-776 -- the density of the low-confidence image the LAST
    // getPredictedImages rendered (the call re-renders only the high-confidence target)
    bool checkIfDenseEnough() {
        int dense = 0;
        check(sf_get_prediction_dense(front_.handle(), &dense), "get_prediction_dense");
        return dense != 0;
    }
    sf::Matrix4f getCurrPose() const {
        sf::Matrix4f pose;
        check(sf_map_info(map_, nullptr, nullptr, pose.m, nullptr), "map_info");
        return pose;
    }
    int getTick() const {
        int tick = 0;
        check(sf_map_info(map_, nullptr, &tick, nullptr, nullptr), "map_info");
        return tick;
    }
    unsigned int lastCount() const {  // getGlobalModel().lastCount()
        int n = 0;
        check(sf_map_info(map_, &n, nullptr, nullptr, nullptr), "map_info");
        return (unsigned int)n;
    }
    // GlobalModel::downloadMap (GlobalModel.cpp:608-636): lastCount() x 12 floats
    std::vector<float> downloadMap() const {
        std::vector<float> out(size_t(lastCount()) * 12);
        check(sf_map_download(map_, out.data(), int(out.size() / 12)), "map_download");
        return out;
    }
    const float &getConfidenceThreshold() const { return confidenceThreshold_; }
    void setConfidenceThreshold(const float &val) { confidenceThreshold_ = val; }
    void setDepthCutoff(const float &val) { depthCutoff_ = val; }
    const int &getTimeDelta() const { return timeDelta_; }
    sf_map *map() { return map_; }

    std::vector<std::pair<unsigned long long, sf::Matrix4f>> poseGraph, gtPoseGraph;  // Reconstruction.h:190-191
    std::vector<int64_t> poseLogTimes;

   private:
    StaticFusionCompat &front_;
    sf_map *map_ = nullptr;
    int timeDelta_;
    float confidenceThreshold_, depthCutoff_;

    sf_model_params params() const {
        sf_model_params mp;
        check(sf_default_model_params(front_.handle(), &mp), "default_model_params");
        mp.conf_high = confidenceThreshold_;
        mp.time_delta = timeDelta_;
        return mp;
    }
    static void check(int rc, const char *what) {
        if (rc != SF_OK) throw std::runtime_error(std::string(what) + ": " + sf_last_error());
    }
};
