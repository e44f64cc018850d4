// sf_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See sf_oracle.hpp.
// PARITY UNPINNED (no reference tests / golden vectors exist; reference not buildable here).
#include "sf_oracle.hpp"

#include <cassert>
#include <cstdio>

namespace sfo {

// =============================================================================================
//  Small linear algebra
// =============================================================================================

// [C4] Eigen 3.3 LDLT: Eigen/src/Cholesky/LDLT.h ldlt_inplace<Lower>::unblocked + _solve_impl.
// Call sites in the reference: FrontEnd.cpp:642 (6x6), SegmentationBackground.cpp:168 (24x24).
void ldlt_solve(const float *Ain, const float *b, float *x, int n) {
    std::vector<float> A(Ain, Ain + size_t(n) * n);  // row-major, lower triangle used
    auto M = [&](int r, int c) -> float & { return A[size_t(r) * n + c]; };
    std::vector<int> transp(n);
    std::vector<float> temp(n);
    bool all_zero = false;

    for (int k = 0; k < n; k++) {
        // largest |diagonal| in the trailing block, first maximum wins (maxCoeff)
        int big = k;
        float bigv = std::fabs(M(k, k));
        for (int i = k + 1; i < n; i++) {
            const float a = std::fabs(M(i, i));
            if (a > bigv) {
                bigv = a;
                big = i;
            }
        }
        transp[k] = big;
        if (big != k) {
            // symmetric row/column swap on the lower triangle
            const int s = n - big - 1;
            for (int j = 0; j < k; j++) std::swap(M(k, j), M(big, j));
            for (int i = 0; i < s; i++) std::swap(M(big + 1 + i, k), M(big + 1 + i, big));
            std::swap(M(k, k), M(big, big));
            for (int i = k + 1; i < big; i++) std::swap(M(i, k), M(big, i));
        }
        const int rs = n - k - 1;
        if (k > 0) {
            for (int j = 0; j < k; j++) temp[j] = M(j, j) * M(k, j);
            float acc = 0.f;
            for (int j = 0; j < k; j++) acc += M(k, j) * temp[j];
            M(k, k) -= acc;
            for (int i = k + 1; i < n; i++) {
                float a2 = 0.f;
                for (int j = 0; j < k; j++) a2 += M(i, j) * temp[j];
                M(i, k) -= a2;
            }
        }
        const float akk = M(k, k);
        const bool pivot_valid = std::fabs(akk) > 0.f;
        if (k == 0 && !pivot_valid) {
            // the whole diagonal is zero: Eigen stops, D = 0, identity transpositions
            for (int j = 0; j < n; j++) transp[j] = j;
            all_zero = true;
            break;
        }
        if (rs > 0 && pivot_valid)
            for (int i = k + 1; i < n; i++) M(i, k) /= akk;
    }

    std::vector<float> y(b, b + n);
    // dst = P b
    for (int k = 0; k < n; k++)
        if (transp[k] != k) std::swap(y[k], y[transp[k]]);
    if (!all_zero) {
        // dst = L^-1 (P b)   (unit lower)
        for (int i = 0; i < n; i++) {
            float acc = y[i];
            for (int j = 0; j < i; j++) acc -= M(i, j) * y[j];
            y[i] = acc;
        }
    }
    // dst = D^-1 dst with Eigen's tolerance = numeric_limits<float>::min()
    const float tol = std::numeric_limits<float>::min();
    for (int i = 0; i < n; i++) {
        const float di = all_zero ? 0.f : M(i, i);
        if (std::fabs(di) > tol)
            y[i] /= di;
        else
            y[i] = 0.f;
    }
    if (!all_zero) {
        // dst = L^-T dst
        for (int i = n - 1; i >= 0; i--) {
            float acc = y[i];
            for (int j = n - 1; j > i; j--) acc -= M(j, i) * y[j];
            y[i] = acc;
        }
    }
    // dst = P^T dst
    for (int k = n - 1; k >= 0; k--)
        if (transp[k] != k) std::swap(y[k], y[transp[k]]);
    for (int i = 0; i < n; i++) x[i] = y[i];
}

bool inverse_double(const double *Ain, double *Ainv, int n) {
    std::vector<double> A(Ain, Ain + size_t(n) * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ainv[i * n + j] = (i == j) ? 1.0 : 0.0;
    bool ok = true;
    for (int c = 0; c < n; c++) {
        int p = c;
        double pv = std::fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++)
            if (std::fabs(A[r * n + c]) > pv) {
                pv = std::fabs(A[r * n + c]);
                p = r;
            }
        if (pv == 0.0) ok = false;
        if (p != c)
            for (int j = 0; j < n; j++) {
                std::swap(A[c * n + j], A[p * n + j]);
                std::swap(Ainv[c * n + j], Ainv[p * n + j]);
            }
        const double inv = 1.0 / A[c * n + c];
        for (int j = 0; j < n; j++) {
            A[c * n + j] *= inv;
            Ainv[c * n + j] *= inv;
        }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const double f = A[r * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < n; j++) {
                A[r * n + j] -= f * A[c * n + j];
                Ainv[r * n + j] -= f * Ainv[c * n + j];
            }
        }
    }
    return ok;
}

void jacobi_eig6(const double S[36], double evals[6], double V[36]) {
    const int n = 6;
    double A[36];
    std::memcpy(A, S, sizeof(A));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-60 || off <= 1e-34 * diag) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) evals[i] = A[i * n + i];
}

static void skew_sq(const double w[3], double K[9], double K2[9]) {
    K[0] = 0;      K[1] = -w[2]; K[2] = w[1];
    K[3] = w[2];   K[4] = 0;     K[5] = -w[0];
    K[6] = -w[1];  K[7] = w[0];  K[8] = 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[i * 3 + k] * K[k * 3 + j];
            K2[i * 3 + j] = s;
        }
}

void se3_exp(const double xi[6], double T[16]) {
    const double *v = xi, *w = xi + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = std::sqrt(th2);
    double a, b, c;  // sin(th)/th, (1-cos th)/th^2, (th - sin th)/th^3
    if (th < 1e-5) {
        a = 1.0 - th2 / 6.0;
        b = 0.5 - th2 / 24.0;
        c = 1.0 / 6.0 - th2 / 120.0;
    } else {
        a = std::sin(th) / th;
        b = (1.0 - std::cos(th)) / th2;
        c = (th - std::sin(th)) / (th2 * th);
    }
    double K[9], K2[9];
    skew_sq(w, K, K2);
    for (int i = 0; i < 3; i++) {
        double t = 0;
        for (int j = 0; j < 3; j++) {
            const double I = (i == j) ? 1.0 : 0.0;
            T[i * 4 + j] = I + a * K[i * 3 + j] + b * K2[i * 3 + j];
            t += (I + b * K[i * 3 + j] + c * K2[i * 3 + j]) * v[j];
        }
        T[i * 4 + 3] = t;
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}

void se3_log(const double T[16], double xi[6]) {
    // rotation part
    const double rx = 0.5 * (T[2 * 4 + 1] - T[1 * 4 + 2]);
    const double ry = 0.5 * (T[0 * 4 + 2] - T[2 * 4 + 0]);
    const double rz = 0.5 * (T[1 * 4 + 0] - T[0 * 4 + 1]);
    const double s = std::sqrt(rx * rx + ry * ry + rz * rz);        // sin(theta)
    const double cth = 0.5 * (T[0] + T[5] + T[10] - 1.0);           // cos(theta)
    const double th = std::atan2(s, cth);
    double w[3];
    if (s < 1e-9) {
        // theta ~ 0 (theta ~ pi is outside the solver's operating range: inter-frame motion)
        const double k = 1.0 + th * th / 6.0;
        w[0] = k * rx; w[1] = k * ry; w[2] = k * rz;
    } else {
        const double k = th / s;
        w[0] = k * rx; w[1] = k * ry; w[2] = k * rz;
    }
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double tha = std::sqrt(th2);
    double d;  // coefficient of K^2 in V^-1
    if (tha < 1e-4)
        d = 1.0 / 12.0 + th2 / 720.0;
    else
        d = (1.0 - (tha * std::sin(tha)) / (2.0 * (1.0 - std::cos(tha)))) / th2;
    double K[9], K2[9];
    skew_sq(w, K, K2);
    const double t[3] = {T[3], T[7], T[11]};
    for (int i = 0; i < 3; i++) {
        double vv = 0;
        for (int j = 0; j < 3; j++) {
            const double I = (i == j) ? 1.0 : 0.0;
            vv += (I - 0.5 * K[i * 3 + j] + d * K2[i * 3 + j]) * t[j];
        }
        xi[i] = vv;
    }
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

// Matrix4f -> double row-major and back
static void to_double_rm(const Mat4f &T, double out[16]) {
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[r * 4 + c] = double(T(r, c));
}

// [C5] Matrix4f::inverse() (FrontEnd.cpp:800,909): double Gauss-Jordan, rounded to float.
static Mat4f inverse4(const Mat4f &T) {
    double A[16], Ai[16];
    to_double_rm(T, A);
    inverse_double(A, Ai, 4);
    Mat4f R;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) R(r, c) = float(Ai[r * 4 + c]);
    return R;
}

// Matrix4f * Matrix4f in float, inner sum left-to-right (FrontEnd.cpp:766,903-907).
static Mat4f mul4(const Mat4f &A, const Mat4f &B) {
    Mat4f C;
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float s = A(r, 0) * B(0, c);
            s += A(r, 1) * B(1, c);
            s += A(r, 2) * B(2, c);
            s += A(r, 3) * B(3, c);
            C(r, c) = s;
        }
    return C;
}

// twist_odometry = vee(log(T_odometry)) (FrontEnd.cpp:769-771). [C5]
static void log_twist(const Mat4f &T, float out[6]) {
    double Td[16], xi[6];
    to_double_rm(T, Td);
    se3_log(Td, xi);
    for (int i = 0; i < 6; i++) out[i] = float(xi[i]);
}

// =============================================================================================
//  Constructor — FrontEnd.cpp:52-181 (GUI / Reconstruction / cv::Mat members are not part of the path)
// =============================================================================================
StaticFusion::StaticFusion(unsigned int rows_, unsigned int cols_, const Params &p) {
    rows = rows_;
    cols = cols_;
    width = cols;
    height = rows;  // all drivers: width == cols (res_factor applied to both), FrontEnd.cpp:55-60
    setParams(p);
    if (p.ctf_levels <= 0) ctf_levels = (unsigned int)(std::log2(double(cols / 40)) + 2);  // FrontEnd.cpp:61

    for (int i = 0; i < 6; i++) twist_odometry_old[i] = twist_odometry[i] = twist_level_odometry[i] = 0.f;
    T_odometry = Mat4f::Identity();
    std::memset(est_cov, 0, sizeof(est_cov));

    depthCurrent.resize(height, width); depthCurrent.assign(0.f);
    depthPrediction.resize(height, width); depthPrediction.assign(0.f);
    intensityCurrent.resize(height, width); intensityCurrent.assign(0.f);
    intensityPrediction.resize(height, width); intensityPrediction.assign(0.f);

    dct.resize(rows, cols); ddt.resize(rows, cols);
    dcu.resize(rows, cols); ddu.resize(rows, cols);
    dcv.resize(rows, cols); ddv.resize(rows, cols);
    dct.assign(0.f); ddt.assign(0.f); dcu.assign(0.f); ddu.assign(0.f); dcv.assign(0.f); ddv.assign(0.f);
    Null.resize(rows, cols); Null.assign(0);
    weights_c.resize(rows, cols); weights_c.assign(0.f);
    weights_d.resize(rows, cols); weights_d.assign(0.f);

    intensityBuffer.resize(bufferLength);
    depthBuffer.resize(bufferLength);
    odomBuffer.resize(bufferLength, Mat4f::Identity());
    for (int i = 0; i < bufferLength; i++) {
        intensityBuffer[i].resize(rows, cols); intensityBuffer[i].assign(0.f);
        depthBuffer[i].resize(rows, cols); depthBuffer[i].assign(0.f);
    }
    for (int l = 0; l < NUM_CLUSTERS; l++) perClusterAverageResidual[l] = std::numeric_limits<float>::quiet_NaN();

    const unsigned int pyr_levels = ctf_levels;  // round(log2(width/cols)) == 0
    pyr_levels_alloc = pyr_levels;
    auto rs = [&](std::vector<MatF> &v) { v.resize(pyr_levels); };
    rs(intensityPyr); rs(intensityPredPyr); rs(intensityInterPyr); rs(intensityWarpedPyr);
    rs(depthPyr); rs(depthPredPyr); rs(depthInterPyr); rs(depthWarpedPyr);
    rs(xxPyr); rs(xxPredPyr); rs(xxInterPyr); rs(xxWarpedPyr);
    rs(yyPyr); rs(yyPredPyr); rs(yyInterPyr); rs(yyWarpedPyr);
    clusterAllocation.resize(pyr_levels);
    xxBuffer.resize(height, width); xxBuffer.assign(0.f);
    yyBuffer.resize(height, width); yyBuffer.assign(0.f);

    for (unsigned int i = 0; i < pyr_levels; i++) {
        const unsigned int s = (unsigned int)std::pow(2.f, int(i));
        cols_i = width / s;
        rows_i = height / s;
        std::vector<MatF> *all[] = {&intensityPyr, &intensityPredPyr, &intensityInterPyr, &intensityWarpedPyr,
                                    &depthPyr, &depthPredPyr, &depthInterPyr, &depthWarpedPyr,
                                    &xxPyr, &xxPredPyr, &xxInterPyr, &xxWarpedPyr,
                                    &yyPyr, &yyPredPyr, &yyInterPyr, &yyWarpedPyr};
        for (auto *v : all) {
            (*v)[i].resize(rows_i, cols_i);
            (*v)[i].assign(0.f);
        }
        clusterAllocation[i].resize(rows_i, cols_i);
        clusterAllocation[i].assign(0);
    }

    // convMask = [1 2 2 1]^T [1 2 2 1] / 36   (FrontEnd.cpp:146-149)
    const float v_mask[4] = {1.f, 2.f, 2.f, 1.f};
    for (unsigned int i = 0; i < 4; i++)
        for (unsigned int j = 0; j < 4; j++) convMask[i + 4 * j] = v_mask[i] * v_mask[j] / 36.f;

    b_segm_perpixel.resize(rows, cols);
    b_segm_perpixel.assign(0.5f);
    for (int l = 0; l < NUM_CLUSTERS; l++) {
        b_segm[l] = 0.5f;
        b_prior[l] = 0.f;
        lambda_t_w[l] = 0.f;
        B_seg[l] = 0.f;
    }
    for (int c = 0; c < 3 * NUM_CLUSTERS; c++) kmeans[c] = 0.f;
    for (int i = 0; i < NUM_CLUSTERS; i++)
        for (int j = 0; j < NUM_CLUSTERS; j++) connectivity[i][j] = (i == j);

    depthWarpedRefference.resize(rows, cols); depthWarpedRefference.assign(0.f);
    intensityWarpedRefference.resize(rows, cols); intensityWarpedRefference.assign(0.f);
}

void StaticFusion::setParams(const Params &p) {
    if (p.ctf_levels > 0) ctf_levels = p.ctf_levels;
    max_iter_per_level = p.max_iter_per_level;
    max_iter_irls = p.max_iter_irls;
    use_motion_filter = p.use_motion_filter;
    segmentation_enabled = p.segmentation_enabled;
    keep_rows = p.keep_rows;
    fovh = p.fovh;
    k_photometric_res = p.k_photometric_res;
    irls_delta_threshold = p.irls_delta_threshold;
    previous_speed_const_weight = p.previous_speed_const_weight;
    previous_speed_eig_weight = p.previous_speed_eig_weight;
    kc_Cauchy = p.kc_Cauchy;
    kb = p.kb;
    kz = p.kz;
    lambda_reg = p.lambda_reg;
    lambda_prior = p.lambda_prior;
}

// =============================================================================================
//  createImagePyramid — FrontEnd.cpp:256-391
// =============================================================================================
void StaticFusion::createImagePyramid(bool old_im) {
    const float max_depth_dif = 0.1f;  // :259
    const unsigned int pyr_levels = ctf_levels;  // :263 with width == cols

    for (unsigned int i = 0; i < pyr_levels; i++) {
        const unsigned int s = (unsigned int)std::pow(2.f, int(i));
        cols_i = width / s;
        rows_i = height / s;

        MatF &depth_here = old_im ? depthPredPyr[i] : depthPyr[i];
        MatF &intensity_here = old_im ? intensityPredPyr[i] : intensityPyr[i];
        MatF &xx_here = old_im ? xxPredPyr[i] : xxPyr[i];
        MatF &yy_here = old_im ? yyPredPyr[i] : yyPyr[i];

        if (i == 0 && !old_im) {  // :282-286
            depth_here = depthCurrent;
            intensity_here = intensityCurrent;
        } else if (i == 0 && old_im) {  // :287-291
            depth_here = depthPrediction;
            intensity_here = intensityPrediction;
        } else {
            const MatF &depth_prev = old_im ? depthPredPyr[i - 1] : depthPyr[i - 1];
            const MatF &intensity_prev = old_im ? intensityPredPyr[i - 1] : intensityPyr[i - 1];
            for (unsigned int u = 0; u < cols_i; u++)
                for (unsigned int v = 0; v < rows_i; v++) {
                    const int u2 = 2 * u;
                    const int v2 = 2 * v;

                    if ((v > 0) && (v < rows_i - 1) && (u > 0) && (u < cols_i - 1)) {  // inner pixels :305
                        // 4x4 blocks, column-major index k = r + 4c  (:308-309)
                        float depth_block[16], intensity_block[16];
                        for (int c = 0; c < 4; c++)
                            for (int r = 0; r < 4; r++) {
                                depth_block[r + 4 * c] = depth_prev(v2 - 1 + r, u2 - 1 + c);
                                intensity_block[r + 4 * c] = intensity_prev(v2 - 1 + r, u2 - 1 + c);
                            }
                        float depths[4] = {depth_block[5], depth_block[6], depth_block[9], depth_block[10]};  // :311

                        // "second maximum" of the central block  (:315-317)
                        if (depths[1] < depths[0]) std::swap(depths[1], depths[0]);
                        if (depths[3] < depths[2]) std::swap(depths[3], depths[2]);
                        const float dcenter =
                            (depths[3] < depths[1]) ? std::max(depths[3], depths[0]) : std::max(depths[1], depths[2]);

                        if (dcenter != 0.f) {  // :319-337
                            float sum_d = 0.f, sum_c = 0.f, weight = 0.f;
                            for (unsigned char k = 0; k < 16; k++) {
                                const float abs_dif = std::abs(depth_block[k] - dcenter);
                                if (abs_dif < max_depth_dif) {
                                    const float aux_w = convMask[k] * (max_depth_dif - abs_dif);
                                    weight += aux_w;
                                    sum_d += aux_w * depth_block[k];
                                    sum_c += aux_w * intensity_block[k];
                                }
                            }
                            depth_here(v, u) = sum_d / weight;
                            intensity_here(v, u) = sum_c / weight;
                        } else {  // :339-343
                            // (convMask*intensity_block.array()).sum()   [C2]: packets = block columns,
                            // (p0+p1)+(p2+p3) lane-wise, then (l0+l2)+(l1+l3).
                            float lane[4];
                            for (int j = 0; j < 4; j++) {
                                const float m0 = convMask[j] * intensity_block[j];
                                const float m1 = convMask[4 + j] * intensity_block[4 + j];
                                const float m2 = convMask[8 + j] * intensity_block[8 + j];
                                const float m3 = convMask[12 + j] * intensity_block[12 + j];
                                lane[j] = (m0 + m1) + (m2 + m3);
                            }
                            intensity_here(v, u) = (lane[0] + lane[2]) + (lane[1] + lane[3]);
                            depth_here(v, u) = 0.f;
                        }
                    } else {  // boundary :347-373
                        float db[4], ib[4];  // Matrix2f column-major k = r + 2c
                        for (int c = 0; c < 2; c++)
                            for (int r = 0; r < 2; r++) {
                                db[r + 2 * c] = depth_prev(v2 + r, u2 + c);
                                ib[r + 2 * c] = intensity_prev(v2 + r, u2 + c);
                            }
                        // 0.25f*intensity_block.sumAll()   [C2]: one packet, (a0+a2)+(a1+a3)
                        intensity_here(v, u) = 0.25f * ((ib[0] + ib[2]) + (ib[1] + ib[3]));

                        float new_d = 0.f;
                        unsigned int cont = 0;
                        for (unsigned int k = 0; k < 4; k++)
                            if (db[k] != 0.f) {
                                new_d += db[k];
                                cont++;
                            }
                        if (cont != 0)
                            depth_here(v, u) = new_d / float(cont);
                        else
                            depth_here(v, u) = 0.f;
                    }
                }
        }

        // coordinates "xy" of the points  (:378-388)
        const float inv_f_i = 2.f * tan_half_fovh() / float(cols_i);
        const float disp_u_i = 0.5f * (cols_i - 1);
        const float disp_v_i = 0.5f * (rows_i - 1);
        for (unsigned int u = 0; u != cols_i; u++) {
            const float xs = inv_f_i * (float(u) - disp_u_i);
            for (unsigned int v = 0; v != rows_i; v++) {
                yy_here(v, u) = (inv_f_i * (float(v) - disp_v_i)) * depth_here(v, u);
                xx_here(v, u) = xs * depth_here(v, u);
            }
        }
    }
}

// =============================================================================================
//  calculateCoord — FrontEnd.cpp:393-430
// =============================================================================================
void StaticFusion::calculateCoord() {
    validPixels.clear();
    validPixels.reserve(size_t(rows_i) * cols_i);
    Null.assign(0);

    MatF &depth_inter_ref = depthInterPyr[image_level];
    MatF &xx_inter_ref = xxInterPyr[image_level];
    MatF &yy_inter_ref = yyInterPyr[image_level];
    MatF &intensity_inter_ref = intensityInterPyr[image_level];
    const MatF &depth_ref = depthPyr[image_level];
    const MatF &depth_warped_ref = depthWarpedPyr[image_level];

    for (unsigned int u = 0; u != cols_i; u++)
        for (unsigned int v = 0; v != rows_i; v++) {
            if ((depth_ref(v, u) != 0.f) && (depth_warped_ref(v, u) != 0.f)) {
                depth_inter_ref(v, u) = 0.5f * (depth_ref(v, u) + depth_warped_ref(v, u));
                xx_inter_ref(v, u) = 0.5f * (xxPyr[image_level](v, u) + xxWarpedPyr[image_level](v, u));
                yy_inter_ref(v, u) = 0.5f * (yyPyr[image_level](v, u) + yyWarpedPyr[image_level](v, u));
                if ((u != 0) && (v != 0) && (u != cols_i - 1) && (v != rows_i - 1)) {
                    const bool behind = depth_warped_ref(v, u) < 0.f;  // not in the reference: see hip_behind_camera_rule
                    if (behind && !hip_behind_camera_rule) behind_camera_valid++;
                    if (!(behind && hip_behind_camera_rule)) validPixels.push_back(std::make_pair(int(v), int(u)));
                }
            } else {
                Null(v, u) = 1;
                depth_inter_ref(v, u) = 0.f;
                xx_inter_ref(v, u) = 0.f;
                yy_inter_ref(v, u) = 0.f;
            }
            intensity_inter_ref(v, u) =
                0.5f * (intensityPyr[image_level](v, u) + intensityWarpedPyr[image_level](v, u));
        }
}

// =============================================================================================
//  calculateDerivatives — FrontEnd.cpp:432-479
// =============================================================================================
void StaticFusion::calculateDerivatives() {
    MatF rx, ry, rx_intensity, ry_intensity;
    rx.resize(rows_i, cols_i); rx.assign(1.f);
    ry.resize(rows_i, cols_i); ry.assign(1.f);
    rx_intensity.resize(rows_i, cols_i); rx_intensity.assign(1.f);
    ry_intensity.resize(rows_i, cols_i); ry_intensity.assign(1.f);

    const MatF &depth_ref = depthInterPyr[image_level];
    const MatF &intensity_ref = intensityInterPyr[image_level];

    const float epsilon_intensity = 1e-6f;
    const float epsilon_depth = 0.005f;

    for (unsigned int u = 0; u < cols_i - 1; u++)
        for (unsigned int v = 0; v < rows_i; v++)
            if (Null(v, u) == 0) {
                rx(v, u) = std::abs(depth_ref(v, u + 1) - depth_ref(v, u)) + epsilon_depth;
                rx_intensity(v, u) = std::abs(intensity_ref(v, u + 1) - intensity_ref(v, u)) + epsilon_intensity;
            }

    for (unsigned int u = 0; u < cols_i; u++)
        for (unsigned int v = 0; v < rows_i - 1; v++)
            if (Null(v, u) == 0) {
                ry(v, u) = std::abs(depth_ref(v + 1, u) - depth_ref(v, u)) + epsilon_depth;
                ry_intensity(v, u) = std::abs(intensity_ref(v + 1, u) - intensity_ref(v, u)) + epsilon_intensity;
            }

    // spatial derivatives (:464-474)
    for (unsigned int v = 1; v < rows_i - 1; v++)
        for (unsigned int u = 1; u < cols_i - 1; u++)
            if (Null(v, u) == 0) {
                dcu(v, u) = (rx_intensity(v, u - 1) * (intensity_ref(v, u + 1) - intensity_ref(v, u)) +
                             rx_intensity(v, u) * (intensity_ref(v, u) - intensity_ref(v, u - 1))) /
                            (rx_intensity(v, u) + rx_intensity(v, u - 1));
                ddu(v, u) = (rx(v, u - 1) * (depth_ref(v, u + 1) - depth_ref(v, u)) +
                             rx(v, u) * (depth_ref(v, u) - depth_ref(v, u - 1))) /
                            (rx(v, u) + rx(v, u - 1));
                dcv(v, u) = (ry_intensity(v - 1, u) * (intensity_ref(v + 1, u) - intensity_ref(v, u)) +
                             ry_intensity(v, u) * (intensity_ref(v, u) - intensity_ref(v - 1, u))) /
                            (ry_intensity(v, u) + ry_intensity(v - 1, u));
                ddv(v, u) = (ry(v - 1, u) * (depth_ref(v + 1, u) - depth_ref(v, u)) +
                             ry(v, u) * (depth_ref(v, u) - depth_ref(v - 1, u))) /
                            (ry(v, u) + ry(v - 1, u));
            }

    // temporal derivative (:477-478) — these assignments resize dct/ddt to the level size
    dct.resize(rows_i, cols_i);
    ddt.resize(rows_i, cols_i);
    for (unsigned int u = 0; u < cols_i; u++)
        for (unsigned int v = 0; v < rows_i; v++) {
            dct(v, u) = intensityPyr[image_level](v, u) - intensityWarpedPyr[image_level](v, u);
            ddt(v, u) = depthPyr[image_level](v, u) - depthWarpedPyr[image_level](v, u);
        }
}

// =============================================================================================
//  computeWeights — FrontEnd.cpp:481-510
// =============================================================================================
void StaticFusion::computeWeights() {
    weights_c.assign(0.f);
    weights_d.assign(0.f);

    const float kduvt_c = 10.f;
    const float kduvt_d = 200.f;
    const float error_m_c = 1.f;
    const float error_m_d = 0.01f;

    for (auto i : validPixels) {
        const int &v = i.first;
        const int &u = i.second;
        const float error_l_c = kduvt_c * (std::abs(dct(v, u)) + std::abs(dcu(v, u)) + std::abs(dcv(v, u)));
        const float error_l_d = kduvt_d * (std::abs(ddt(v, u)) + std::abs(ddu(v, u)) + std::abs(ddv(v, u)));
        weights_c(v, u) = sqrtf(1.f / (error_m_c + error_l_c));
        weights_d(v, u) = sqrtf(1.f / (error_m_d + error_l_d));
    }

    float max_c = weights_c.d[0], max_d = weights_d.d[0];  // .maximum() over the full buffer
    for (size_t k = 1; k < weights_c.size(); k++) {
        if (weights_c.d[k] > max_c) max_c = weights_c.d[k];
        if (weights_d.d[k] > max_d) max_d = weights_d.d[k];
    }
    if (validPixels.empty()) {
        // The reference computes 1/0 here and fills both planes with NaN (no row is ever read
        // afterwards because validPixels is empty). Keep the planes at 0 and flag it.
        stats.status |= 2;
        return;
    }
    const float inv_max_c = 1.f / max_c;
    for (auto &w : weights_c.d) w = inv_max_c * w;
    const float inv_max_d = 1.f / max_d;
    for (auto &w : weights_d.d) w = inv_max_d * w;
}

// =============================================================================================
//  computeSegPrior — SegmentationBackground.cpp:53-103
// =============================================================================================
void StaticFusion::computeSegPrior() {
    int cluster_size[NUM_CLUSTERS], cluster_nonnull[NUM_CLUSTERS];
    const MatI &labels_ref = clusterAllocation[image_level];

    double exact[NUM_CLUSTERS];  // exact_sums (test hook): the same terms accumulated in fp64
    for (int l = 0; l < NUM_CLUSTERS; l++) {
        b_prior[l] = 0.f;
        exact[l] = 0.0;
        cluster_size[l] = 0;
        cluster_nonnull[l] = 0;
        lambda_t_w[l] = 0.f;
    }

    for (unsigned int u = 0; u < cols_i; u++)
        for (unsigned int v = 0; v < rows_i; v++) {
            const int l = labels_ref(v, u);
            if (l != NUM_CLUSTERS) {
                if (Null(v, u) == 0) {
                    cluster_nonnull[l]++;
                    float ddt_here = ddt(v, u);
                    if (hip_behind_camera_rule)  // the HIP build keeps |depthWarped| for a pixel outside validPixels
                        ddt_here = depthPyr[image_level](v, u) - std::abs(depthWarpedPyr[image_level](v, u));
                    b_prior[l] += 1.f - kz * std::abs(ddt_here);
                    exact[l] += double(1.f - kz * std::abs(ddt_here));
                }
                cluster_size[labels_ref(v, u)]++;
            }
        }
    if (exact_sums)
        for (int l = 0; l < NUM_CLUSTERS; l++) b_prior[l] = float(exact[l]);

    for (unsigned int l = 0; l < NUM_CLUSTERS; l++) {
        if (cluster_size[l] != 0) {
            const float ratio = float(cluster_nonnull[l]) / float(cluster_size[l]);
            if (ratio < 0.1f) {
                lambda_t_w[l] = 0.1f;
                b_prior[l] = -1.f;
            } else {
                lambda_t_w[l] = ratio;
                b_prior[l] = std::max(-1.f, std::min(2.f, b_prior[l] / cluster_nonnull[l]));
            }
        }
    }
}

// =============================================================================================
//  buildSystemSegm — SegmentationBackground.cpp:105-130
//  A_seg is (24 + nconn) x 24 with one diagonal entry per data row and (+w, -w) per
//  regularisation row; stored sparsely (same numbers).
// =============================================================================================
void StaticFusion::buildSystemSegm() {
    A_seg_diag.assign(NUM_CLUSTERS, 0.f);
    seg_edges.clear();
    for (int l = 0; l < NUM_CLUSTERS; l++) B_seg[l] = 0.f;
    for (unsigned int l = 0; l < NUM_CLUSTERS; l++)
        for (unsigned int lc = l + 1; lc < NUM_CLUSTERS; lc++)
            if (connectivity[l][lc] == true) seg_edges.push_back(std::make_pair(int(l), int(lc)));
}

// =============================================================================================
//  solveSegmIteration — SegmentationBackground.cpp:133-174
// =============================================================================================
void StaticFusion::solveSegmIteration(const float aver_res[NUM_CLUSTERS], float aver_res_overall, float kc) {
    const float repr_res = std::max(0.001f, aver_res_overall);  // :137

    // [C6] float log evaluated as float(log(double))
    auto logf_c6 = [](float x) { return float(std::log(double(x))); };
    const float fixed_term = logf_c6(1.f + sq(kb * repr_res / (kc * aver_res_overall)));
    const float mult_res = 1.f / (kc * aver_res_overall);
    for (unsigned int l = 0; l < NUM_CLUSTERS; l++) {
        if (lambda_t_w[l] > 0.1f) {
            const float dataterm = fixed_term - logf_c6(1.f + sq(aver_res[l] * mult_res));
            A_seg_diag[l] = 2.f * lambda_t_w[l] * lambda_prior;
            B_seg[l] = dataterm + 2.f * lambda_prior * lambda_t_w[l] * b_prior[l];
        } else {
            A_seg_diag[l] = 2.f * lambda_t_w[l];
            B_seg[l] = 2.f * lambda_t_w[l] * b_prior[l];
        }
    }

    // AtA_seg = A_seg^T A_seg, AtB_seg = A_seg^T B_seg  (:164-165)  [C1]: float products, double sums
    const float weight_reg = 2.f * lambda_reg;  // :125
    const float w2 = weight_reg * weight_reg;
    const float nw2 = weight_reg * (-weight_reg);
    double AtA_d[NUM_CLUSTERS * NUM_CLUSTERS];
    for (auto &a : AtA_d) a = 0.0;
    for (int l = 0; l < NUM_CLUSTERS; l++) AtA_d[l * NUM_CLUSTERS + l] += double(A_seg_diag[l] * A_seg_diag[l]);
    for (auto &e : seg_edges) {
        const int l = e.first, lc = e.second;
        AtA_d[l * NUM_CLUSTERS + l] += double(w2);
        AtA_d[lc * NUM_CLUSTERS + lc] += double(w2);
        AtA_d[l * NUM_CLUSTERS + lc] += double(nw2);
        AtA_d[lc * NUM_CLUSTERS + l] += double(nw2);
    }
    float AtA[NUM_CLUSTERS * NUM_CLUSTERS], AtB[NUM_CLUSTERS];
    for (int i = 0; i < NUM_CLUSTERS * NUM_CLUSTERS; i++) AtA[i] = float(AtA_d[i]);
    for (int l = 0; l < NUM_CLUSTERS; l++) AtB[l] = A_seg_diag[l] * B_seg[l];  // regularisation rows have B = 0

    ldlt_solve(AtA, AtB, b_segm, NUM_CLUSTERS);  // :168

    for (unsigned int l = 0; l < NUM_CLUSTERS; l++) b_segm[l] = std::max(-1.f, std::min(2.f, b_segm[l]));
}

// =============================================================================================
//  solveOdometryAndSegmJoint — FrontEnd.cpp:513-692
// =============================================================================================
void StaticFusion::solveOdometryAndSegmJoint() {
    buildSystemSegm();

    const MatI &labels_ref = clusterAllocation[image_level];
    const MatF &depth_inter_ref = depthInterPyr[image_level];
    const MatF &xx_inter_ref = xxInterPyr[image_level];
    const MatF &yy_inter_ref = yyInterPyr[image_level];

    const size_t N = validPixels.size();
    const size_t M = 2 * N;
    std::vector<float> A(M * 6, 0.f), B(M, 0.f), Aw(M * 6, 0.f), Bw(M, 0.f);  // column-major M x 6
    auto A_ = [&](size_t r, int c) -> float & { return A[r + size_t(c) * M]; };
    auto Aw_ = [&](size_t r, int c) -> float & { return Aw[r + size_t(c) * M]; };
    float Var[6] = {0, 0, 0, 0, 0, 0};

    size_t cont = 0;
    const float f_inv = float(cols_i) / (2.f * tan_half_fovh());  // :537 (it is f, not 1/f)

    for (auto i : validPixels) {
        const int &v = i.first;
        const int &u = i.second;

        const float d = depth_inter_ref(v, u);
        const float inv_d = 1.f / d;
        const float x = xx_inter_ref(v, u);
        const float y = yy_inter_ref(v, u);

        // colour (:552-566)
        const float dycomp_c = dcu(v, u) * f_inv * inv_d;
        const float dzcomp_c = dcv(v, u) * f_inv * inv_d;
        const float twc = weights_c(v, u) * k_photometric_res;

        A_(cont, 0) = twc * (-dycomp_c);
        A_(cont, 1) = twc * (-dzcomp_c);
        A_(cont, 2) = twc * (dycomp_c * x * inv_d + dzcomp_c * y * inv_d);
        A_(cont, 3) = twc * (dycomp_c * inv_d * y * x + dzcomp_c * (y * y * inv_d + d));
        A_(cont, 4) = twc * (-dycomp_c * (x * x * inv_d + d) - dzcomp_c * inv_d * y * x);
        A_(cont, 5) = twc * (dycomp_c * y - dzcomp_c * x);
        B[cont] = twc * (-dct(v, u));
        cont++;

        // geometry (:570-585)
        const float dycomp_d = ddu(v, u) * f_inv * inv_d;
        const float dzcomp_d = ddv(v, u) * f_inv * inv_d;
        const float twd = weights_d(v, u);

        A_(cont, 0) = twd * (-dycomp_d);
        A_(cont, 1) = twd * (-dzcomp_d);
        A_(cont, 2) = twd * (1.f + dycomp_d * x * inv_d + dzcomp_d * y * inv_d);
        A_(cont, 3) = twd * (y + dycomp_d * inv_d * y * x + dzcomp_d * (y * y * inv_d + d));
        A_(cont, 4) = twd * (-x - dycomp_d * (x * x * inv_d + d) - dzcomp_d * inv_d * y * x);
        A_(cont, 5) = twd * (dycomp_d * y - dzcomp_d * x);
        B[cont] = twd * (-ddt(v, u));
        cont++;
    }

    float AtA[36], AtB[6];
    for (auto &a : AtA) a = 0.f;
    for (auto &a : AtB) a = 0.f;
    std::vector<float> res(M);
    for (size_t r = 0; r < M; r++) res[r] = -B[r];
    // aver_res = res.cwiseAbs().sumAll() / res.size()   (:590)  [C1]
    double sabs = 0.0;
    for (size_t r = 0; r < M; r++) sabs += double(std::fabs(res[r]));
    float aver_res = float(sabs) / float(M);

    float prev_sol[6] = {0, 0, 0, 0, 0, 0};

    if (!segmentation_enabled) {
        for (int l = 0; l < NUM_CLUSTERS; l++) b_segm[l] = 1.f;  // :606-607 alternative
    } else if (level == 0) {
        for (int l = 0; l < NUM_CLUSTERS; l++) b_segm[l] = b_prior[l];  // :603-604
    }

    if (N == 0) {
        // Not in the reference: a level without a single valid pixel makes the reference divide by
        // zero (aver_res = 0/0, est_cov = inf*0).  Defined behaviour here and in the HIP build: flag
        // it, leave T_odometry and b_segm untouched, report a zero level twist.
        stats.status |= 2;
        for (int c = 0; c < 6; c++) twist_level_odometry[c] = 0.f;
        if (stats.n_outer < MAX_OUTER) {
            OuterTrace &t0 = stats.outer[stats.n_outer];
            t0.n_valid = 0;
            t0.irls_iters = 0;
            t0.aver_res = 0.f;
            t0.delta_sol_max = 0.f;
            for (int c = 0; c < 6; c++) t0.var[c] = t0.twist_level[c] = 0.f;
            for (int l = 0; l < NUM_CLUSTERS; l++) t0.b_segm[l] = b_segm[l];
            std::memcpy(t0.T, T_odometry.m, sizeof(t0.T));
            for (int l = 0; l < NUM_CLUSTERS; l++) {
                t0.b_prior[l] = b_prior[l];
                t0.lambda_t_w[l] = lambda_t_w[l];
            }
            std::memset(t0.AtA, 0, sizeof(t0.AtA));
            std::memset(t0.AtB, 0, sizeof(t0.AtB));
        }
        return;
    }

    unsigned int iters_done = 0;
    float last_delta = 0.f;
    for (unsigned int k = 1; k <= max_iter_irls; k++) {
        iters_done = k;
        const float inv_c_Cauchy = 1.f / (kc_Cauchy * aver_res);  // :615

        cont = 0;
        for (auto i : validPixels) {  // :619-637
            const int &v = i.first;
            const int &u = i.second;
            // labels_ref is 24 only where the NEW depth is 0, which is never a valid pixel
            const float b_weight = std::max(0.f, std::min(1.f, b_segm[segmentation_enabled ? labels_ref(v, u) : 0]));

            const float res_weight_color = b_weight * sqrtf(1.f / (1.f + sq(res[cont] * inv_c_Cauchy)));
            for (int c = 0; c < 6; c++) Aw_(cont, c) = res_weight_color * A_(cont, c);
            Bw[cont] = res_weight_color * B[cont];
            cont++;

            const float res_weight_depth = b_weight * sqrtf(1.f / (1.f + sq(res[cont] * inv_c_Cauchy)));
            for (int c = 0; c < 6; c++) Aw_(cont, c) = res_weight_depth * A_(cont, c);
            Bw[cont] = res_weight_depth * B[cont];
            cont++;
        }

        // AtA.multiply_AtA(Aw); AtB.multiply_AtB(Aw,Bw);   (:640-641)  [C1]
        {
            double acc[27];
            for (auto &a : acc) a = 0.0;
            if (gemm_mode == 0 || gemm_mode == 3) {
                for (size_t rr = 0; rr < M; rr++) {
                    const size_t r = (gemm_mode == 3) ? M - 1 - rr : rr;
                    int q = 0;
                    for (int i = 0; i < 6; i++)
                        for (int j = i; j < 6; j++) acc[q++] += double(Aw_(r, i)) * double(Aw_(r, j));
                    for (int i = 0; i < 6; i++) acc[21 + i] += double(Aw_(r, i)) * double(Bw[r]);
                }
            } else {  // test hook: float accumulators (see gemm_mode in sf_oracle.hpp)
                const int P = (gemm_mode == 2) ? 4 : 1;
                float part[4][27];
                for (auto &pp : part)
                    for (auto &a : pp) a = 0.f;
                for (size_t r = 0; r < M; r++) {
                    float *pa = part[r % size_t(P)];
                    int q = 0;
                    for (int i = 0; i < 6; i++)
                        for (int j = i; j < 6; j++) pa[q++] += Aw_(r, i) * Aw_(r, j);
                    for (int i = 0; i < 6; i++) pa[21 + i] += Aw_(r, i) * Bw[r];
                }
                for (int q = 0; q < 27; q++) acc[q] = double((part[0][q] + part[1][q]) + (part[2][q] + part[3][q]));
            }
            int q = 0;
            for (int i = 0; i < 6; i++)
                for (int j = i; j < 6; j++) {
                    AtA[i * 6 + j] = AtA[j * 6 + i] = float(acc[q]);
                    q++;
                }
            for (int i = 0; i < 6; i++) AtB[i] = float(acc[21 + i]);
        }
        ldlt_solve(AtA, AtB, Var, 6);  // :642

        // res = -B; res += Var(k)*A.col(k)   (:644-646)
        for (size_t r = 0; r < M; r++) res[r] = -B[r];
        for (unsigned int kk = 0; kk < 6; kk++)
            for (size_t r = 0; r < M; r++) res[r] += Var[kk] * A_(r, kk);

        // residuals, overall and cluster-wise (:650-667)
        float aver_res_label[NUM_CLUSTERS];
        int num_pix_label[NUM_CLUSTERS];
        for (int l = 0; l < NUM_CLUSTERS; l++) {
            aver_res_label[l] = 0.f;
            num_pix_label[l] = 1;  // "to avoid division by zero"
        }
        const float aver_res_old = aver_res;

        double exact_label[NUM_CLUSTERS];  // exact_sums (test hook): the same terms accumulated in fp64
        for (int l = 0; l < NUM_CLUSTERS; l++) exact_label[l] = 0.0;
        for (size_t i = 0; i < N; ++i) {
            const std::pair<int, int> &vu = validPixels[i];
            const float ress_here = std::abs(res[2 * i]) + std::abs(res[2 * i + 1]);
            const int lab = segmentation_enabled ? labels_ref(vu.first, vu.second) : 0;
            aver_res_label[lab] += ress_here;
            exact_label[lab] += double(ress_here);
            num_pix_label[lab]++;
        }
        if (exact_sums)
            for (int l = 0; l < NUM_CLUSTERS; l++) aver_res_label[l] = float(exact_label[l]);
        // aver_res_label.matrix().sumAll()  [C1]
        double ssum = 0.0;
        for (int l = 0; l < NUM_CLUSTERS; l++) ssum += double(aver_res_label[l]);
        aver_res = float(ssum) / float(2 * N);
        for (int l = 0; l < NUM_CLUSTERS; l++) aver_res_label[l] /= float(2 * num_pix_label[l]);

        if (segmentation_enabled) solveSegmIteration(aver_res_label, aver_res_old, kc_Cauchy);  // :672

        stats.n_irls++;
        stats.pixel_iters += (long long)N;

        // convergence (:676-683)
        float delta_sol_max = 0.f;
        for (int c = 0; c < 6; c++) delta_sol_max = std::max(delta_sol_max, std::fabs(prev_sol[c] - Var[c]));
        for (int c = 0; c < 6; c++) prev_sol[c] = Var[c];
        last_delta = delta_sol_max;
        if ((delta_sol_max < irls_delta_threshold) || (k == max_iter_irls)) break;
    }

    // est_cov = AtA.inverse()*res.squaredNorm()   (:689)  [C5]/[C1]
    {
        double Ad[36], Ai[36];
        for (int i = 0; i < 36; i++) Ad[i] = double(AtA[i]);
        inverse_double(Ad, Ai, 6);
        double sqn = 0.0;
        for (size_t r = 0; r < M; r++) sqn += double(res[r]) * double(res[r]);
        const float sqnf = float(sqn);
        for (int i = 0; i < 36; i++) est_cov[i] = float(Ai[i]) * sqnf;
    }

    // trace (not in the reference)
    OuterTrace *tr = (stats.n_outer < MAX_OUTER) ? &stats.outer[stats.n_outer] : nullptr;
    if (tr) {
        tr->n_valid = int(N);
        tr->irls_iters = int(iters_done);
        tr->aver_res = aver_res;
        tr->delta_sol_max = last_delta;
        for (int c = 0; c < 6; c++) tr->var[c] = Var[c];
        for (int l = 0; l < NUM_CLUSTERS; l++) {
            tr->b_prior[l] = b_prior[l];
            tr->lambda_t_w[l] = lambda_t_w[l];
        }
        std::memcpy(tr->AtA, AtA, sizeof(tr->AtA));
        std::memcpy(tr->AtB, AtB, sizeof(tr->AtB));
    }
    if (keep_rows) {
        dbg_A = A;
        dbg_B = B;
    }

    filterEstimateAndComputeT(Var);  // :690

    if (tr) {
        for (int c = 0; c < 6; c++) tr->twist_level[c] = twist_level_odometry[c];
        for (int l = 0; l < NUM_CLUSTERS; l++) tr->b_segm[l] = b_segm[l];
        std::memcpy(tr->T, T_odometry.m, sizeof(tr->T));
    }
}

// =============================================================================================
//  filterEstimateAndComputeT — FrontEnd.cpp:713-772   [C5]
// =============================================================================================
void StaticFusion::filterEstimateAndComputeT(float twist[6]) {
    if (use_motion_filter) {
        // SelfAdjointEigenSolver reads the lower triangle of est_cov (:719)
        bool finite = true;
        double S[36];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j <= i; j++) {
                const double a = double(est_cov[i * 6 + j]);
                if (!std::isfinite(a)) finite = false;
                S[i * 6 + j] = S[j * 6 + i] = a;
            }
        if (!finite) {
            // "Eigensolver couldn't find a solution. Pose is not updated" (:720-724)
            stats.status |= 1;
            return;
        }
        double evals[6], V[36];
        jacobi_eig6(S, evals, V);

        // kai_b = Bii^-1 twist = Bii^T twist (:729)
        double kai_b[6], kai_b_old[6];
        float kai_loc_sub[6];
        for (int i = 0; i < 6; i++) kai_loc_sub[i] = twist_odometry_old[i];
        // subtract the previous levels' solution: kai_loc_sub -= vee(log(T_odometry)) (:735-738)
        float lt[6];
        log_twist(T_odometry, lt);
        for (int i = 0; i < 6; i++) kai_loc_sub[i] -= lt[i];
        for (int i = 0; i < 6; i++) {
            double s = 0, so = 0;
            for (int r = 0; r < 6; r++) {
                s += V[r * 6 + i] * double(twist[r]);
                so += V[r * 6 + i] * double(kai_loc_sub[r]);
            }
            kai_b[i] = s;
            kai_b_old[i] = so;
        }

        // filter (:745-752); expf(-int(level)) as float(exp(double))  [C6]
        const float e_l = float(std::exp(-double(int(level))));
        const float cf = previous_speed_eig_weight * e_l, df = previous_speed_const_weight * e_l;
        double kai_b_fil[6];
        for (unsigned int i = 0; i < 6; i++) {
            const double wgt = double(cf) * evals[i] + double(df);
            kai_b_fil[i] = (kai_b[i] + wgt * kai_b_old[i]) / (1.0 + wgt);
        }
        // twist = Bii * kai_b_fil (:755)
        for (int r = 0; r < 6; r++) {
            double s = 0;
            for (int i = 0; i < 6; i++) s += V[r * 6 + i] * kai_b_fil[i];
            twist[r] = float(s);
        }
    }

    // rigid transformation associated to the twist (:759-766)
    double xi[6], E[16];
    for (int i = 0; i < 6; i++) xi[i] = double(twist[i]);
    se3_exp(xi, E);
    Mat4f Ef;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) Ef(r, c) = float(E[r * 4 + c]);

    for (int i = 0; i < 6; i++) twist_level_odometry[i] = twist[i];
    T_odometry = mul4(Ef, T_odometry);

    log_twist(T_odometry, twist_odometry);  // :769-771
}

// =============================================================================================
//  warpImagesAccurateInverse — FrontEnd.cpp:775-892
// =============================================================================================
void StaticFusion::warpImagesAccurateInverse() {
    const float f = float(cols_i) / (2.f * tan_half_fovh());
    const float disp_u_i = 0.5f * float(cols_i - 1);
    const float disp_v_i = 0.5f * float(rows_i - 1);

    MatF &depth_warped_ref = depthWarpedPyr[image_level];
    MatF &intensity_warped_ref = intensityWarpedPyr[image_level];
    MatF &xx_warped_ref = xxWarpedPyr[image_level];
    MatF &yy_warped_ref = yyWarpedPyr[image_level];
    const MatF &depth_ref = depthPredPyr[image_level];
    const MatF &intensity_ref = intensityPredPyr[image_level];
    const MatF &xx_ref = xxPredPyr[image_level];
    const MatF &yy_ref = yyPredPyr[image_level];
    depth_warped_ref.assign(0.f);
    intensity_warped_ref.assign(0.f);

    MatF wacu;
    wacu.resize(rows_i, cols_i);
    wacu.assign(0.f);
    std::vector<double> ex_d, ex_i;  // exact_warp (test hook): fp64 sums of the exact products
    if (exact_warp) {
        ex_d.assign(size_t(rows_i) * cols_i, 0.0);
        ex_i.assign(size_t(rows_i) * cols_i, 0.0);
    }
    auto ex_add = [&](int v, int u, int w, float dw, float iw) {
        if (!exact_warp) return;
        ex_d[size_t(v) + size_t(u) * rows_i] += double(w) * double(dw);
        ex_i[size_t(v) + size_t(u) * rows_i] += double(w) * double(iw);
    };
    const int cols_lim = 100 * (cols_i - 1);
    const int rows_lim = 100 * (rows_i - 1);

    const Mat4f T = inverse4(T_odometry);  // :800

    for (unsigned int j = 0; j < cols_i; j++)
        for (unsigned int i = 0; i < rows_i; i++) {
            const float z = depth_ref(i, j);
            if (z != 0.f) {
                const float intensity_w = intensity_ref(i, j);
                const float x_w = T(0, 0) * xx_ref(i, j) + T(0, 1) * yy_ref(i, j) + T(0, 2) * z + T(0, 3);
                const float y_w = T(1, 0) * xx_ref(i, j) + T(1, 1) * yy_ref(i, j) + T(1, 2) * z + T(1, 3);
                const float depth_w = T(2, 0) * xx_ref(i, j) + T(2, 1) * yy_ref(i, j) + T(2, 2) * z + T(2, 3);

                const int uwarp = cvt_trunc_x86(100.f * (f * x_w / depth_w + disp_u_i));
                const int vwarp = cvt_trunc_x86(100.f * (f * y_w / depth_w + disp_v_i));

                if ((uwarp >= 0) && (uwarp < cols_lim) && (vwarp >= 0) && (vwarp < rows_lim)) {
                    const int uwarp_l = uwarp - uwarp % 100;
                    const int uwarp_r = uwarp_l + 100;
                    const int vwarp_d = vwarp - vwarp % 100;
                    const int vwarp_u = vwarp_d + 100;
                    const int delta_r = uwarp_r - uwarp;
                    const int delta_l = 100 - delta_r;
                    const int delta_u = vwarp_u - vwarp;
                    const int delta_d = 100 - delta_u;

                    if (std::min(delta_r, delta_l) + std::min(delta_u, delta_d) < 5) {  // :835
                        const int ind_u = delta_r > delta_l ? uwarp_l / 100 : uwarp_r / 100;
                        const int ind_v = delta_u > delta_d ? vwarp_d / 100 : vwarp_u / 100;
                        depth_warped_ref(ind_v, ind_u) += 200.f * depth_w;
                        intensity_warped_ref(ind_v, ind_u) += 200.f * intensity_w;
                        wacu(ind_v, ind_u) += 200;
                        ex_add(ind_v, ind_u, 200, depth_w, intensity_w);
                    } else {
                        const int v_d = vwarp_d / 100, u_l = uwarp_l / 100;
                        const int v_u = v_d + 1, u_r = u_l + 1;

                        const int w_ur = delta_l + delta_d;
                        depth_warped_ref(v_u, u_r) += w_ur * depth_w;
                        intensity_warped_ref(v_u, u_r) += w_ur * intensity_w;
                        wacu(v_u, u_r) += w_ur;
                        ex_add(v_u, u_r, w_ur, depth_w, intensity_w);

                        const int w_ul = delta_r + delta_d;
                        depth_warped_ref(v_u, u_l) += w_ul * depth_w;
                        intensity_warped_ref(v_u, u_l) += w_ul * intensity_w;
                        wacu(v_u, u_l) += w_ul;
                        ex_add(v_u, u_l, w_ul, depth_w, intensity_w);

                        const int w_dr = delta_l + delta_u;
                        depth_warped_ref(v_d, u_r) += w_dr * depth_w;
                        intensity_warped_ref(v_d, u_r) += w_dr * intensity_w;
                        wacu(v_d, u_r) += w_dr;
                        ex_add(v_d, u_r, w_dr, depth_w, intensity_w);

                        const int w_dl = delta_r + delta_u;
                        depth_warped_ref(v_d, u_l) += w_dl * depth_w;
                        intensity_warped_ref(v_d, u_l) += w_dl * intensity_w;
                        wacu(v_d, u_l) += w_dl;
                        ex_add(v_d, u_l, w_dl, depth_w, intensity_w);
                    }
                }
            }
        }

    const float inv_f_i = 1.f / f;  // :874
    for (unsigned int u = 0; u < cols_i; u++)
        for (unsigned int v = 0; v < rows_i; v++) {
            if (wacu(v, u) != 0) {
                intensity_warped_ref(v, u) /= float(wacu(v, u));
                depth_warped_ref(v, u) /= float(wacu(v, u));
                if (exact_warp) {
                    intensity_warped_ref(v, u) = float(ex_i[size_t(v) + size_t(u) * rows_i] / double(wacu(v, u)));
                    depth_warped_ref(v, u) = float(ex_d[size_t(v) + size_t(u) * rows_i] / double(wacu(v, u)));
                }
                xx_warped_ref(v, u) = (u - disp_u_i) * depth_warped_ref(v, u) * inv_f_i;
                yy_warped_ref(v, u) = (v - disp_v_i) * depth_warped_ref(v, u) * inv_f_i;
            } else {
                xx_warped_ref(v, u) = 0.f;
                yy_warped_ref(v, u) = 0.f;
            }
        }
}

// =============================================================================================
//  computeResidualsAgainstPreviousImage — FrontEnd.cpp:896-1069
// =============================================================================================
void StaticFusion::computeResidualsAgainstPreviousImage(int index) {
    int idx_to_warp = (index - bufferLength) % bufferLength;
    int trans_start = (index - bufferLength + 1);

    Mat4f T = Mat4f::Identity();
    for (int i = trans_start; i < index; i++) {
        int idx = i % bufferLength;
        T = mul4(T, odomBuffer[idx]);
    }
    T = mul4(T, T_odometry);
    T = inverse4(T);

    const float inv_f_i = 2.f * tan_half_fovh() / float(cols);
    const float disp_u_i = 0.5f * (cols - 1);
    const float disp_v_i = 0.5f * (rows - 1);

    for (unsigned int u = 0; u != cols; u++) {
        const float xs = inv_f_i * (float(u) - disp_u_i);
        for (unsigned int v = 0; v != rows; v++) {
            yyBuffer(v, u) = (inv_f_i * (float(v) - disp_v_i)) * depthBuffer[idx_to_warp](v, u);
            xxBuffer(v, u) = xs * depthBuffer[idx_to_warp](v, u);
        }
    }

    const float f = float(cols) / (2.f * tan_half_fovh());

    depthWarpedRefference.assign(0.f);
    intensityWarpedRefference.assign(0.f);

    MatF intensity_diff = intensityCurrent;
    const MatI &labels_ref = clusterAllocation[0];

    MatF wacu;
    wacu.resize(rows, cols);
    wacu.assign(0.f);
    const int cols_lim = 100 * (cols - 1);
    const int rows_lim = 100 * (rows - 1);

    for (unsigned int j = 0; j < cols; j++)
        for (unsigned int i = 0; i < rows; i++) {
            const float z = depthBuffer[idx_to_warp](i, j);

            if (z != 0.f && depthCurrent(i, j) != 0.f) {
                const float intensity_w = intensityBuffer[idx_to_warp](i, j);
                const float x_w = T(0, 0) * xxBuffer(i, j) + T(0, 1) * yyBuffer(i, j) + T(0, 2) * z + T(0, 3);
                const float y_w = T(1, 0) * xxBuffer(i, j) + T(1, 1) * yyBuffer(i, j) + T(1, 2) * z + T(1, 3);
                const float depth_w = T(2, 0) * xxBuffer(i, j) + T(2, 1) * yyBuffer(i, j) + T(2, 2) * z + T(2, 3);

                const int uwarp = cvt_trunc_x86(100.f * (f * x_w / depth_w + disp_u_i));
                const int vwarp = cvt_trunc_x86(100.f * (f * y_w / depth_w + disp_v_i));

                if ((uwarp >= 0) && (uwarp < cols_lim) && (vwarp >= 0) && (vwarp < rows_lim)) {
                    const int uwarp_l = uwarp - uwarp % 100;
                    const int uwarp_r = uwarp_l + 100;
                    const int vwarp_d = vwarp - vwarp % 100;
                    const int vwarp_u = vwarp_d + 100;
                    const int delta_r = uwarp_r - uwarp;
                    const int delta_l = 100 - delta_r;
                    const int delta_u = vwarp_u - vwarp;
                    const int delta_d = 100 - delta_u;

                    if (std::min(delta_r, delta_l) + std::min(delta_u, delta_d) < 5) {
                        const int ind_u = delta_r > delta_l ? uwarp_l / 100 : uwarp_r / 100;
                        const int ind_v = delta_u > delta_d ? vwarp_d / 100 : vwarp_u / 100;
                        depthWarpedRefference(ind_v, ind_u) += 200.f * depth_w;
                        intensityWarpedRefference(ind_v, ind_u) += 200.f * intensity_w;
                        wacu(ind_v, ind_u) += 200;
                    } else {
                        const int v_d = vwarp_d / 100, u_l = uwarp_l / 100;
                        const int v_u = v_d + 1, u_r = u_l + 1;

                        const int w_ur = delta_l + delta_d;
                        depthWarpedRefference(v_u, u_r) += w_ur * depth_w;
                        intensityWarpedRefference(v_u, u_r) += w_ur * intensity_w;
                        wacu(v_u, u_r) += w_ur;

                        const int w_ul = delta_r + delta_d;
                        depthWarpedRefference(v_u, u_l) += w_ul * depth_w;
                        intensityWarpedRefference(v_u, u_l) += w_ul * intensity_w;
                        wacu(v_u, u_l) += w_ul;

                        const int w_dr = delta_l + delta_u;
                        depthWarpedRefference(v_d, u_r) += w_dr * depth_w;
                        intensityWarpedRefference(v_d, u_r) += w_dr * intensity_w;
                        wacu(v_d, u_r) += w_dr;

                        const int w_dl = delta_r + delta_u;
                        depthWarpedRefference(v_d, u_l) += w_dl * depth_w;
                        intensityWarpedRefference(v_d, u_l) += w_dl * intensity_w;
                        wacu(v_d, u_l) += w_dl;
                    }
                }
            } else {
                intensity_diff(i, j) = 0;
            }
        }

    for (unsigned int u = 0; u < cols; u++)
        for (unsigned int v = 0; v < rows; v++)
            if (wacu(v, u) != 0) {
                intensityWarpedRefference(v, u) /= float(wacu(v, u));
                depthWarpedRefference(v, u) /= float(wacu(v, u));
            }

    // residuals, overall and cluster-wise (:1036-1068)
    for (int l = 0; l < NUM_CLUSTERS; l++) perClusterAverageResidual[l] = std::numeric_limits<float>::quiet_NaN();
    int num_pix_label[NUM_CLUSTERS];
    double exact[NUM_CLUSTERS];  // exact_sums (test hook)
    for (int l = 0; l < NUM_CLUSTERS; l++) {
        num_pix_label[l] = 1;
        exact[l] = 0.0;
    }

    for (unsigned int j = 0; j < cols; j++)
        for (unsigned int i = 0; i < rows; i++) {
            if (depthWarpedRefference(i, j) != 0 && depthCurrent(i, j) != 0) {
                const float depthRes = depthCurrent(i, j) - depthWarpedRefference(i, j);
                const float intRes = intensity_diff(i, j) - intensityWarpedRefference(i, j);
                const float cumulative = std::abs(depthRes) + k_photometric_res * std::abs(intRes);
                const int lab = labels_ref(i, j);
                if (std::isnan(perClusterAverageResidual[lab]))
                    perClusterAverageResidual[lab] = cumulative;
                else
                    perClusterAverageResidual[lab] += cumulative;
                exact[lab] += double(cumulative);
                num_pix_label[lab]++;
            }
        }
    if (exact_sums)
        for (int l = 0; l < NUM_CLUSTERS; l++)
            if (!std::isnan(perClusterAverageResidual[l])) perClusterAverageResidual[l] = float(exact[l]);
    for (int l = 0; l < NUM_CLUSTERS; l++) perClusterAverageResidual[l] /= float(2 * num_pix_label[l]);
}

// =============================================================================================
//  runSolver — FrontEnd.cpp:1071-1146
// =============================================================================================
void StaticFusion::runSolver(bool create_image_pyr) {
    stats.n_outer = 0;
    stats.n_irls = 0;
    stats.pixel_iters = 0;
    stats.kmeans_iters = 0;
    stats.status = 0;
    behind_camera_valid = 0;

    if (create_image_pyr) createImagePyramid(false);

    if (segmentation_enabled) {
        kMeans3DCoord();
        createClustersPyramidUsingKMeans();
    }

    T_odometry = Mat4f::Identity();

    for (unsigned int i = 0; i < ctf_levels; i++)
        for (unsigned int k = 0; k < max_iter_per_level; k++) {
            level = i;
            unsigned int s = (unsigned int)std::pow(2.f, int(ctf_levels - (i + 1)));
            cols_i = cols / s;
            rows_i = rows / s;
            image_level = ctf_levels - i - 1;

            // 1. warping (:1103-1112)
            if ((i == 0) && (k == 0)) {
                depthWarpedPyr[image_level] = depthPredPyr[image_level];
                intensityWarpedPyr[image_level] = intensityPredPyr[image_level];
                xxWarpedPyr[image_level] = xxPredPyr[image_level];
                yyWarpedPyr[image_level] = yyPredPyr[image_level];
            } else
                warpImagesAccurateInverse();

            calculateCoord();        // 2.
            calculateDerivatives();  // 3.
            computeWeights();        // 4.
            if (segmentation_enabled) computeSegPrior();  // 5.

            if (stats.n_outer < MAX_OUTER) {
                stats.outer[stats.n_outer].level = int(i);
                stats.outer[stats.n_outer].k = int(k);
            }
            solveOdometryAndSegmJoint();  // 6.
            stats.n_outer++;

            float nrm2 = 0.f;  // twist_level_odometry.norm()  [C1]-style: double sum of float squares
            {
                double s2 = 0.0;
                for (int c = 0; c < 6; c++) s2 += double(twist_level_odometry[c] * twist_level_odometry[c]);
                nrm2 = float(std::sqrt(double(float(s2))));
            }
            if (nrm2 < 0.04f) break;
        }

    // Transform the local velocity to the new reference frame after motion (:1139-1144)  [C5]
    double R[9], Ri[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R[r * 3 + c] = double(T_odometry(r, c));
    inverse_double(R, Ri, 3);
    float Rif[9];
    for (int q = 0; q < 9; q++) Rif[q] = float(Ri[q]);
    for (int half = 0; half < 2; half++)
        for (int r = 0; r < 3; r++) {
            float s2 = Rif[r * 3 + 0] * twist_odometry[half * 3 + 0];
            s2 += Rif[r * 3 + 1] * twist_odometry[half * 3 + 1];
            s2 += Rif[r * 3 + 2] * twist_odometry[half * 3 + 2];
            twist_odometry_old[half * 3 + r] = s2;
        }
}

void StaticFusion::pushHistory(int im_count) {  // StaticFusion-datasets.cpp:182-184
    const int slot = im_count % bufferLength;
    depthBuffer[slot] = depthCurrent;
    intensityBuffer[slot] = intensityCurrent;
    odomBuffer[slot] = T_odometry;
}

// =============================================================================================
//  buildSegmImage — SegmentationBackground.cpp:176-197
// =============================================================================================
void StaticFusion::buildSegmImage() {
    const MatI &labels_maxres = clusterAllocation[0];
    for (unsigned int u = 0; u < cols; u++)
        for (unsigned int v = 0; v < rows; v++) {
            if (labels_maxres(v, u) == NUM_CLUSTERS) {
                b_segm_perpixel(v, u) = 1;
                continue;
            }
            b_segm_perpixel(v, u) = std::max(0.f, std::min(1.f, b_segm[labels_maxres(v, u)]));
            if (perClusterAverageResidual[labels_maxres(v, u)] < 0.017)
                b_segm_perpixel(v, u) = std::max(b_segm_perpixel(v, u), 1.0f - b_segm_perpixel(v, u));
        }
}

}  // namespace sfo
