// VALU issue-rate probe for gfx950: how many cycles does a wave64 spend per v_fma_f32 / v_pk_fma_f32 /
// v_fma_f64 / v_rsq_f32 when the SIMD is saturated?  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vfloat2 __attribute__((ext_vector_type(2)));
#define UNROLL 16
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed) {
    float a[UNROLL];
    vfloat2 p[UNROLL];
    double d[UNROLL];
    for (int k = 0; k < UNROLL; k++) {
        a[k] = seed + k + threadIdx.x;
        p[k] = vfloat2{a[k], a[k] + 1.f};
        d[k] = a[k];
    }
    const float m = seed * 0.999f;
    const vfloat2 pm{m, m};
    const double dm = m;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            if (MODE == 0) a[k] = __builtin_fmaf(a[k], m, 0.5f);
            if (MODE == 1) p[k] = __builtin_elementwise_fma(p[k], pm, vfloat2{0.5f, 0.5f});
            if (MODE == 2) d[k] = __builtin_fma(d[k], dm, 0.5);
            if (MODE == 3) a[k] = __builtin_amdgcn_rsqf(a[k]);
            if (MODE == 4) a[k] = a[k] * m;
            if (MODE == 5) p[k] = p[k] * pm;
            if (MODE == 6) p[k] = __builtin_elementwise_fma(p[k], p[(k + 1) % UNROLL], p[(k + 2) % UNROLL]);
        }
    }
    float r = 0;
    for (int k = 0; k < UNROLL; k++) r += a[k] + p[k].x + p[k].y + (float)d[k];
    if (r == 12345.f) out[0] = r;
}
template <int MODE>
void run(const char *name, float *out) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 100, 1.0f);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = blocks*4 / (256 CU * 4 SIMD); instructions per wave = iters*UNROLL
    const double wave_instr_per_simd = (double)blocks * 4 / 1024.0 * iters * UNROLL;
    printf("%-22s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, ms, ms * 1e6 / wave_instr_per_simd);
}
int main() {
    float *out;
    hipMalloc(&out, 4);
    run<0>("v_fma_f32", out);
    run<1>("v_pk_fma_f32 (const)", out);
    run<6>("v_pk_fma_f32 (3 vgpr)", out);
    run<2>("v_fma_f64", out);
    run<3>("v_rsq_f32", out);
    run<4>("v_mul_f32", out);
    run<5>("v_pk_mul_f32", out);
    return 0;
}
