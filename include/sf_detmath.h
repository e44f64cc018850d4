/*
 * sf_detmath.h — exp(-a) for the bilateral depth filter, defined operation by operation.
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

#endif /* SF_DETMATH_H_ */
