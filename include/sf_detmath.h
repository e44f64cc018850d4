/*
 * sf_detmath.h — exp / log defined operation by operation (bilateral depth filter, surfel fusion), and the
 * host-side velocity weighting of Reconstruction::fuseFrame.
 *
 * The reference evaluates the filter weight exp(-(space2*s + color2*c)) in a GLSL fragment shader
 * (reference Shaders/depth_bilateral.frag:64), where exp() has an implementation-defined error of a
 * few ulp, and then ROUNDS the filtered depth to integer millimetres (:71). To make that integer
 * output a well-defined function of the input -- the same on the CPU oracle, on gfx950 and in the
 * NumPy derivation that generated tests/golden/ -- both implementations of the ABI evaluate the
 * weight with the function below: a fixed sequence of IEEE-754 binary32 operations (multiply,
 * round-to-nearest-even integer, correctly rounded fused multiply-add). fmaf() is exact on every
 * platform (one rounding), so the result does not depend on compiler contraction settings.
 * Error vs the exact exp: < 1.5 ulp (tests/test_input_stage.py checks it against libm), i.e. inside
 * what the GL specification allows the reference's own exp().
 *
 * Usable from C, C++ and HIP device code (v_fma_f32 / v_rndne_f32 on gfx950).
 */
#ifndef SF_DETMATH_H_
#define SF_DETMATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SF_DETMATH_FN __host__ __device__ static inline
#else
#define SF_DETMATH_FN static inline
#endif

/* exp(-a) for a >= 0; exactly 0 for a > 87 and for NaN (results below FLT_MIN are flushed, as GPUs do) */
SF_DETMATH_FN float sf_exp_neg(float a) {
    const int live = a <= 87.0f;
    const float aa = live ? a : 0.0f;
    const float log2e = 1.44269502162933349609375f;
    const float ln2_hi = 0.693145751953125f;        /* ln2 rounded to 15 significant bits */
    const float ln2_lo = 1.42860676533018589e-06f;  /* ln2 - ln2_hi */
    const float n = rintf(aa * log2e); /* ties to even (default rounding mode; v_rndne_f32 on gfx950) */
    float r = fmaf(-n, ln2_hi, aa);
    r = fmaf(-n, ln2_lo, r); /* r = a - n ln2 in [-0.35, 0.35]; we need exp(-r) */
    const float x = -r;
    /* exp(x) ~ 1 + x + x^2/2 + ... + x^7/5040, Horner */
    float p = 1.0f / 5040.0f;
    p = fmaf(p, x, 1.0f / 720.0f);
    p = fmaf(p, x, 1.0f / 120.0f);
    p = fmaf(p, x, 1.0f / 24.0f);
    p = fmaf(p, x, 1.0f / 6.0f);
    p = fmaf(p, x, 0.5f);
    p = fmaf(p, x, 1.0f);
    p = fmaf(p, x, 1.0f);
    /* scale by 2^-n (n in [0, 126]): an exact multiplication by a power of two */
    const uint32_t bits = (uint32_t)(127 - (int)n) << 23;
    float s;
    memcpy(&s, &bits, sizeof s);
    return live ? p * s : 0.0f;
}

/* exp(x) for |x| <= 87 (the log-odds update of Shaders/update.vert:58-60 stays within +-12): the same reduction and
 * polynomial as sf_exp_neg with a signed exponent. Outside the range: 0 below, +inf above, NaN for NaN. */
SF_DETMATH_FN float sf_exp_det(float x) {
    if (!(x >= -87.0f)) return x < -87.0f ? 0.0f : x; /* NaN passes through */
    if (x > 87.0f) return INFINITY;
    const float log2e = 1.44269502162933349609375f;
    const float ln2_hi = 0.693145751953125f;
    const float ln2_lo = 1.42860676533018589e-06f;
    const float n = rintf(x * log2e);
    float r = fmaf(-n, ln2_hi, x);
    r = fmaf(-n, ln2_lo, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const uint32_t bits = (uint32_t)(127 + (int)n) << 23; /* n in [-126, 126] */
    float s;
    memcpy(&s, &bits, sizeof s);
    return p * s;
}

/* log(x) for normal positive x (Shaders/update.vert:58-59). x = m 2^e with m in [sqrt(1/2), sqrt(2));
 * log m = 2 atanh(s), s = (m - 1) / (m + 1), odd series to s^9 (|s| <= 0.1716: truncation 2e-9 relative);
 * result e ln2_hi + (e ln2_lo + log m). < 2 ulp vs libm on [1e-3, 1e3] (tests/test_map_fusion.py).
 * x == 0 -> -inf, x < 0 or NaN -> NaN, +inf -> +inf; subnormals are treated as 0 (GPUs flush them). */
SF_DETMATH_FN float sf_log_det(float x) {
    uint32_t bits;
    memcpy(&bits, &x, sizeof bits);
    if (bits & 0x80000000u) return (bits << 1) == 0u ? -INFINITY : NAN;
    if (bits >= 0x7f800000u) return x; /* +inf, NaN */
    if (bits < 0x00800000u) return -INFINITY;
    int e = (int)(bits >> 23) - 127;
    uint32_t mb = (bits & 0x007fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &mb, sizeof m);
    if (m > 1.41421356237f) {
        m *= 0.5f;
        e += 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float s2 = s * s;
    float p = 1.0f / 9.0f;
    p = fmaf(p, s2, 1.0f / 7.0f);
    p = fmaf(p, s2, 1.0f / 5.0f);
    p = fmaf(p, s2, 1.0f / 3.0f);
    p = p * s2;
    const float logm = 2.0f * fmaf(s, p, s);
    const float ef = (float)e;
    const float ln2_hi = 0.693145751953125f;
    const float ln2_lo = 1.42860676533018589e-06f;
    return fmaf(ef, ln2_hi, fmaf(ef, ln2_lo, logm));
}

/* "Weight by velocity" of Reconstruction::fuseFrame (reference Reconstruction.cpp:263-282): diff = currPose^-1 lastPose,
 * weighting = max(|diff.t|, |rodrigues2(diff.R)|) clamped to 0.15, then max(1 - weighting / 0.15, 0.5) * multiplier.
 * Poses are 4x4 column-major floats. rodrigues2 (:487-532) re-orthonormalises R through an SVD (U V^T = the polar
 * factor) before taking the rotation vector; here the polar factor comes from the Newton iteration X <- (X + X^-T)/2
 * in double, which converges to the same matrix. Host code; both implementations of the ABI call THIS function, so the
 * scalar they feed to the fusion kernels is the same. */
static inline float sf_fusion_weighting(const float last_pose[16], const float curr_pose[16], float weight_multiplier) {
    double A[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            A[r][c] = (double)curr_pose[r + 4 * c];
            A[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) { /* Gauss-Jordan, partial pivoting */
        int piv = c;
        for (int r = c + 1; r < 4; r++)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        for (int j = 0; j < 8; j++) {
            const double t = A[c][j];
            A[c][j] = A[piv][j];
            A[piv][j] = t;
        }
        const double inv = 1.0 / A[c][c];
        for (int j = 0; j < 8; j++) A[c][j] *= inv;
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            const double f = A[r][c];
            for (int j = 0; j < 8; j++) A[r][j] -= f * A[c][j];
        }
    }
    float diff[4][4]; /* Eigen::Matrix4f product: float */
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += (double)(float)A[r][4 + k] * (double)last_pose[k + 4 * c];
            diff[r][c] = (float)acc;
        }
    const double tn = sqrt((double)diff[0][3] * diff[0][3] + (double)diff[1][3] * diff[1][3] + (double)diff[2][3] * diff[2][3]);
    double X[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) X[r][c] = (double)diff[r][c];
    for (int it = 0; it < 12; it++) {
        const double det = X[0][0] * (X[1][1] * X[2][2] - X[1][2] * X[2][1]) - X[0][1] * (X[1][0] * X[2][2] - X[1][2] * X[2][0]) +
                           X[0][2] * (X[1][0] * X[2][1] - X[1][1] * X[2][0]);
        if (!(fabs(det) > 1e-300)) break;
        double C[3][3]; /* cofactors: X^-T = C / det */
        C[0][0] = X[1][1] * X[2][2] - X[1][2] * X[2][1];
        C[0][1] = X[1][2] * X[2][0] - X[1][0] * X[2][2];
        C[0][2] = X[1][0] * X[2][1] - X[1][1] * X[2][0];
        C[1][0] = X[0][2] * X[2][1] - X[0][1] * X[2][2];
        C[1][1] = X[0][0] * X[2][2] - X[0][2] * X[2][0];
        C[1][2] = X[0][1] * X[2][0] - X[0][0] * X[2][1];
        C[2][0] = X[0][1] * X[1][2] - X[0][2] * X[1][1];
        C[2][1] = X[0][2] * X[1][0] - X[0][0] * X[1][2];
        C[2][2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) X[r][c] = 0.5 * (X[r][c] + C[r][c] / det);
    }
    float R[3][3]; /* Eigen::Matrix3f R = U V^T */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R[r][c] = (float)X[r][c];
    double rx = (double)(R[2][1] - R[1][2]), ry = (double)(R[0][2] - R[2][0]), rz = (double)(R[1][0] - R[0][1]); /* :492-494 */
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (double)((R[0][0] + R[1][1] + R[2][2]) - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) { /* :501-522 */
        if (c > 0)
            rx = ry = rz = 0;
        else {
            double t = ((double)R[0][0] + 1) * 0.5;
            rx = sqrt(t > 0.0 ? t : 0.0);
            t = ((double)R[1][1] + 1) * 0.5;
            ry = sqrt(t > 0.0 ? t : 0.0) * (R[0][1] < 0 ? -1.0 : 1.0);
            t = ((double)R[2][2] + 1) * 0.5;
            rz = sqrt(t > 0.0 ? t : 0.0) * (R[0][2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[1][2] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    const float frx = (float)rx, fry = (float)ry, frz = (float)rz; /* .cast<float>(), then .norm() in float */
    const float rn = sqrtf(frx * frx + fry * fry + frz * frz);
    float weighting = (float)tn > rn ? (float)tn : rn; /* :273 */
    const float largest = 0.15f, min_weight = 0.5f;
    if (weighting > largest) weighting = largest;
    const float w = 1.0f - (weighting / largest);
    return (w > min_weight ? w : min_weight) * weight_multiplier; /* :282 */
}

#endif /* SF_DETMATH_H_ */
