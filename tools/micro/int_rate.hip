// Integer VALU issue-rate probe for gfx950: cycles a saturated SIMD spends per wave64 instruction for the integer
// operations the warp splat is made of (32-bit multiplies, 64 x 32-bit multiply-add, 64-bit shifts / adds, conversions).
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/int_rate tools/micro/int_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define UNROLL 16
template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned *out, int iters, unsigned seed) {
    unsigned a[UNROLL];
    unsigned long long q[UNROLL];
    float f[UNROLL];
    for (int k = 0; k < UNROLL; k++) {
        a[k] = seed + k * 77u + threadIdx.x;
        q[k] = ((unsigned long long)a[k] << 20) | k;
        f[k] = (float)a[k];
    }
    const unsigned m = seed * 2654435761u | 1u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            if (MODE == 0) a[k] = a[k] + m;                                                    // v_add_u32
            if (MODE == 1) a[k] = a[k] * m;                                                    // v_mul_lo_u32
            if (MODE == 2) a[k] = __umulhi(a[k], m);                                           // v_mul_hi_u32
            if (MODE == 3) q[k] = (unsigned long long)(unsigned)q[k] * m + q[k];               // v_mad_u64_u32
            if (MODE == 4) q[k] = q[k] + ((unsigned long long)a[k] << 3);                      // v_lshl_add_u64
            if (MODE == 5) a[k] = (unsigned)((int)a[k] / 100);                                 // signed division by a constant
            if (MODE == 6) a[k] = (unsigned)(int)f[k], f[k] = f[k] + 1.5f;                     // v_cvt_i32_f32 + v_add_f32
            if (MODE == 7) a[k] = ((a[k] & 0xffffffu) * (m & 0xffffffu));                           // v_mul_u32_u24
            if (MODE == 8) a[k] = ((a[k] & 0xffffffu) * (m & 0xffffffu)) + a[k];                    // v_mad_u32_u24
            if (MODE == 9) q[k] = q[k] * (unsigned long long)m;                                // 64 x 32 -> 64 product
            if (MODE == 10) a[k] = (a[k] > m) ? a[k] - m : a[k] + 3u;                           // v_cmp + v_cndmask + 2 adds
            if (MODE == 11) q[k] = (unsigned long long)((long long)q[k] >> 7) + 1ull;           // v_ashrrev_i64 + 64-bit add
            if (MODE == 12) f[k] = f[k] / (float)(a[k] | 1u), a[k] += 1u;                       // IEEE fp32 division
        }
    }
    unsigned r = 0;
    for (int k = 0; k < UNROLL; k++) r += a[k] + (unsigned)q[k] + (unsigned)(q[k] >> 32) + (unsigned)f[k];
    if (r == 12345u) out[0] = r;
}
template <int MODE>
void run(const char *name, unsigned *out) {
    const int iters = 5000, blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 100, 1u);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 / 1024.0 * iters * UNROLL;  // source-level operations per SIMD
    printf("%-44s %8.3f ms  -> %6.2f ns per wave-operation per SIMD\n", name, ms, ms * 1e6 / per_simd);
}
int main() {
    unsigned *out;
    hipMalloc(&out, 4);
    run<0>("v_add_u32", out);
    run<1>("v_mul_lo_u32", out);
    run<2>("v_mul_hi_u32", out);
    run<3>("v_mad_u64_u32", out);
    run<4>("v_lshl_add_u64", out);
    run<5>("int / 100 (mul_hi + shifts)", out);
    run<6>("v_cvt_i32_f32 + v_add_f32", out);
    run<7>("v_mul_u32_u24", out);
    run<8>("v_mad_u32_u24", out);
    run<9>("u64 * u32 (mad_u64_u32 + mul_lo + add)", out);
    run<10>("cmp + cndmask + 2 add", out);
    run<11>("v_ashrrev_i64 + 64-bit add", out);
    run<12>("IEEE f32 division + add", out);
    return 0;
}
