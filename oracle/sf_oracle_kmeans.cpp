// sf_oracle_kmeans.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See sf_oracle.hpp.
// Restatement of the geometric clustering of the reference: KMeans.cpp:52-391.
// PARITY UNPINNED (no reference tests / golden vectors exist; reference not buildable here).
#include "sf_oracle.hpp"

namespace sfo {

namespace {
struct IndexAndDistance {  // KMeans.cpp:52-61
    int idx;
    float distance;
    bool operator<(const IndexAndDistance &o) const { return distance < o.distance; }
};

// (a - b).squaredNorm() for 3-vectors  [C3]
inline float sqdist3(const float *a, const float *b) {
    const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return (d0 * d0 + d1 * d1) + d2 * d2;
}
}  // namespace

// =============================================================================================
//  initializeKMeans — KMeans.cpp:63-135
// =============================================================================================
void StaticFusion::initializeKMeans() {
    rows_km = rows / 2;
    cols_km = cols / 2;
    image_level_km = 1;  // round(log2(width/cols_km))
    const MatF &depth_ref = depthPyr[image_level_km];
    MatI &labels_ref = clusterAllocation[image_level_km];
    labels_ref.assign(NUM_CLUSTERS);

    unsigned int u_label[NUM_CLUSTERS], v_label[NUM_CLUSTERS];
    const unsigned int vert_div = (unsigned int)std::ceil(std::sqrt(double(NUM_CLUSTERS)));
    const float u_div = float(cols_km) / float(NUM_CLUSTERS + 1);
    const float v_div = float(rows_km) / float(vert_div + 1);
    for (unsigned int i = 0; i < NUM_CLUSTERS; i++) {
        u_label[i] = (unsigned int)std::round((i + 1) * u_div);
        v_label[i] = (unsigned int)std::round((i % vert_div + 1) * v_div);
    }

    for (unsigned int u = 0; u < cols_km; u++)
        for (unsigned int v = 0; v < rows_km; v++)
            if (depth_ref(v, u) != 0.f) {
                unsigned int min_dist = 1000000.f, quad_dist;
                unsigned int ini_label = NUM_CLUSTERS;
                for (unsigned int l = 0; l < NUM_CLUSTERS; l++) {
                    const unsigned int dv = v - v_label[l], du = u - u_label[l];  // unsigned wrap-around, as in the reference
                    if ((quad_dist = dv * dv + du * du) < min_dist) {
                        ini_label = l;
                        min_dist = quad_dist;
                    }
                }
                labels_ref(v, u) = ini_label;
            }

    std::vector<float> depth_sorted[NUM_CLUSTERS];
    for (unsigned int u = 0; u < cols_km; u++)
        for (unsigned int v = 0; v < rows_km; v++)
            if (depth_ref(v, u) != 0.f) depth_sorted[labels_ref(v, u)].push_back(depth_ref(v, u));

    const float inv_f_i = 2.f * std::tan(0.5f * fovh) / float(cols_km);
    const float disp_u_i = 0.5f * (cols_km - 1);
    const float disp_v_i = 0.5f * (rows_km - 1);
    for (unsigned int l = 0; l < NUM_CLUSTERS; l++) {
        const unsigned int size_label = depth_sorted[l].size();
        const unsigned int med_pos = size_label / 2;
        if (size_label > 0) {
            std::nth_element(depth_sorted[l].begin(), depth_sorted[l].begin() + med_pos, depth_sorted[l].end());
            kmeans[0 + 3 * l] = depth_sorted[l].at(med_pos);
            kmeans[1 + 3 * l] = (u_label[l] - disp_u_i) * kmeans[0 + 3 * l] * inv_f_i;
            kmeans[2 + 3 * l] = (v_label[l] - disp_v_i) * kmeans[0 + 3 * l] * inv_f_i;
        } else {
            kmeans[0 + 3 * l] = kmeans[1 + 3 * l] = kmeans[2 + 3 * l] = 0.f;
        }
    }
}

// =============================================================================================
//  kMeans3DCoord — KMeans.cpp:137-295
// =============================================================================================
void StaticFusion::kMeans3DCoord() {
    const unsigned int max_level = 0;  // round(log2(width/cols))
    const unsigned int lower_level = max_level + 1;
    const unsigned int iter_kmeans = 10;

    const MatF &depth_ref = depthPyr[lower_level];
    const MatF &xx_ref = xxPyr[lower_level];
    const MatF &yy_ref = yyPyr[lower_level];
    MatI &labels_lowres = clusterAllocation[lower_level];

    initializeKMeans();

    std::vector<std::vector<IndexAndDistance>> cluster_distances(NUM_CLUSTERS,
                                                                 std::vector<IndexAndDistance>(NUM_CLUSTERS));
    float centers_a[3 * NUM_CLUSTERS], centers_b[3 * NUM_CLUSTERS];  // 3 x 24 column-major
    int count[NUM_CLUSTERS];
    for (unsigned int c = 0; c < NUM_CLUSTERS; c++)
        for (unsigned int r = 0; r < 3; r++) centers_a[r + 3 * c] = kmeans[r + 3 * c];

    for (unsigned int i = 0; i < iter_kmeans - 1; i++) {
        stats.kmeans_iters++;
        for (auto &c : centers_b) c = 0.f;

        for (unsigned int l = 0; l < NUM_CLUSTERS; l++) {
            count[l] = 0;
            std::vector<IndexAndDistance> &distances = cluster_distances.at(l);
            for (unsigned int li = 0; li < NUM_CLUSTERS; li++) {
                distances.at(li).idx = li;
                distances.at(li).distance = sqdist3(&centers_a[3 * l], &centers_a[3 * li]);
            }
            std::sort(distances.begin(), distances.end());
        }

        for (unsigned int u = 0; u < cols_km; u++)
            for (unsigned int v = 0; v < rows_km; v++)
                if (depth_ref(v, u) != 0.f) {
                    const int last_label = labels_lowres(v, u);
                    int best_label = last_label;
                    std::vector<IndexAndDistance> &distances = cluster_distances.at(last_label);

                    const float p[3] = {depth_ref(v, u), xx_ref(v, u), yy_ref(v, u)};
                    const float distance_to_last_label = sqdist3(&centers_a[3 * last_label], p);
                    float best_distance = distance_to_last_label;

                    for (size_t li = 1; li < distances.size(); ++li) {
                        const IndexAndDistance &idx_and_distance = distances.at(li);
                        if (idx_and_distance.distance > 4.f * distance_to_last_label) break;
                        const float distance_to_label = sqdist3(&centers_a[3 * idx_and_distance.idx], p);
                        if (distance_to_label < best_distance) {
                            best_distance = distance_to_label;
                            best_label = idx_and_distance.idx;
                        }
                    }

                    labels_lowres(v, u) = best_label;
                    for (int r = 0; r < 3; r++) centers_b[r + 3 * best_label] += p[r];
                    count[best_label] += 1;
                }

        for (unsigned int l = 0; l < NUM_CLUSTERS; l++)
            if (count[l] > 0)
                for (int r = 0; r < 3; r++) centers_b[r + 3 * l] /= count[l];

        float max_diff = 0.f;  // (centers_a - centers_b).lpNorm<Infinity>()
        for (int q = 0; q < 3 * NUM_CLUSTERS; q++) max_diff = std::max(max_diff, std::fabs(centers_a[q] - centers_b[q]));
        for (int q = 0; q < 3 * NUM_CLUSTERS; q++) std::swap(centers_a[q], centers_b[q]);

        if (max_diff < 1e-2f) break;
    }

    for (unsigned int c = 0; c < NUM_CLUSTERS; c++)
        for (unsigned int r = 0; r < 3; r++) kmeans[r + 3 * c] = centers_a[r + 3 * c];

    // labelling at the max resolution (:238-291)
    const MatF &depth_highres = depthPyr[max_level];
    const MatF &xx_highres = xxPyr[max_level];
    const MatF &yy_highres = yyPyr[max_level];
    MatI &labels_ref = clusterAllocation[max_level];
    labels_ref.assign(NUM_CLUSTERS);

    for (unsigned int l = 0; l < NUM_CLUSTERS; l++) {
        std::vector<IndexAndDistance> &distances = cluster_distances.at(l);
        for (unsigned int li = 0; li < NUM_CLUSTERS; li++) {
            distances.at(li).idx = li;
            distances.at(li).distance = sqdist3(&centers_a[3 * l], &centers_a[3 * li]);
        }
        std::sort(distances.begin(), distances.end());
    }

    for (unsigned int u = 0; u < cols; u++)
        for (unsigned int v = 0; v < rows; v++)
            if (depth_highres(v, u) != 0.f) {
                const int label_lowres_here = labels_lowres(v / 2, u / 2);
                const int last_label = (label_lowres_here == NUM_CLUSTERS) ? 0 : label_lowres_here;

                int best_label = last_label;
                std::vector<IndexAndDistance> &distances = cluster_distances.at(last_label);
                const float p[3] = {depth_highres(v, u), xx_highres(v, u), yy_highres(v, u)};

                const float distance_to_last_label = sqdist3(&centers_a[3 * last_label], p);
                float best_distance = distance_to_last_label;

                for (size_t li = 1; li < distances.size(); ++li) {
                    const IndexAndDistance &idx_and_distance = distances.at(li);
                    if (idx_and_distance.distance > 4.f * distance_to_last_label) break;
                    const float distance_to_label = sqdist3(&centers_a[3 * idx_and_distance.idx], p);
                    if (distance_to_label < best_distance) {
                        best_distance = distance_to_label;
                        best_label = idx_and_distance.idx;
                    }
                }
                labels_ref(v, u) = best_label;
            }

    computeRegionConnectivity();
}

// =============================================================================================
//  computeRegionConnectivity — KMeans.cpp:297-341
// =============================================================================================
void StaticFusion::computeRegionConnectivity() {
    const unsigned int max_level = 0;
    const float dist2_threshold = sq(0.03f * 120.f / float(rows));

    const MatI &labels_ref = clusterAllocation[max_level];
    const MatF &depth_ref = depthPyr[max_level];
    const MatF &xx_ref = xxPyr[max_level];
    const MatF &yy_ref = yyPyr[max_level];

    for (unsigned int i = 0; i < NUM_CLUSTERS; i++)
        for (unsigned int j = 0; j < NUM_CLUSTERS; j++) connectivity[i][j] = (i == j);

    for (unsigned int u = 0; u < cols - 1; u++)
        for (unsigned int v = 0; v < rows - 1; v++)
            if (depth_ref(v, u) != 0.f) {
                if ((labels_ref(v, u) != labels_ref(v + 1, u)) && (labels_ref(v + 1, u) != NUM_CLUSTERS)) {
                    const float disty = sq(depth_ref(v, u) - depth_ref(v + 1, u)) + sq(yy_ref(v, u) - yy_ref(v + 1, u));
                    if (disty < dist2_threshold) {
                        connectivity[labels_ref(v, u)][labels_ref(v + 1, u)] = true;
                        connectivity[labels_ref(v + 1, u)][labels_ref(v, u)] = true;
                    }
                }
                if ((labels_ref(v, u) != labels_ref(v, u + 1)) && (labels_ref(v, u + 1) != NUM_CLUSTERS)) {
                    const float distx = sq(depth_ref(v, u) - depth_ref(v, u + 1)) + sq(xx_ref(v, u) - xx_ref(v, u + 1));
                    if (distx < dist2_threshold) {
                        connectivity[labels_ref(v, u)][labels_ref(v, u + 1)] = true;
                        connectivity[labels_ref(v, u + 1)][labels_ref(v, u)] = true;
                    }
                }
            }
}

// =============================================================================================
//  createClustersPyramidUsingKMeans — KMeans.cpp:343-391
// =============================================================================================
void StaticFusion::createClustersPyramidUsingKMeans() {
    float kmeans_dist[NUM_CLUSTERS][NUM_CLUSTERS];
    for (unsigned int la = 0; la < NUM_CLUSTERS; la++)
        for (unsigned int lb = la + 1; lb < NUM_CLUSTERS; lb++)
            kmeans_dist[la][lb] = sqdist3(&kmeans[3 * la], &kmeans[3 * lb]);

    for (unsigned int i = 2; i < ctf_levels; i++) {
        unsigned int s = (unsigned int)std::pow(2.f, int(i));
        cols_km = cols / s;
        rows_km = rows / s;
        image_level_km = i;

        MatI &labels_ref = clusterAllocation[image_level_km];
        const MatF &depth_old_ref = depthPyr[image_level_km];
        const MatF &xx_old_ref = xxPyr[image_level_km];
        const MatF &yy_old_ref = yyPyr[image_level_km];

        labels_ref.assign(NUM_CLUSTERS);

        for (unsigned int u = 0; u < cols_km; u++)
            for (unsigned int v = 0; v < rows_km; v++)
                if (depth_old_ref(v, u) != 0.f) {
                    unsigned int label = 0;
                    const float p[3] = {depth_old_ref(v, u), xx_old_ref(v, u), yy_old_ref(v, u)};
                    float min_dist = sqdist3(&kmeans[0], p);
                    float dist_here;
                    for (unsigned int l = 1; l < NUM_CLUSTERS; l++) {
                        if (kmeans_dist[label][l] > 4.f * min_dist)
                            continue;
                        else if ((dist_here = sqdist3(&kmeans[3 * l], p)) < min_dist) {
                            label = l;
                            min_dist = dist_here;
                        }
                    }
                    labels_ref(v, u) = label;
                }
    }
}

}  // namespace sfo
