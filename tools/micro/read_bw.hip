// Read-only HBM bandwidth probe for gfx950: what does a streaming READ reach, as a function of the bytes a wave keeps in flight
// (loads issued before the first is consumed) and of the waves per CU? The IRLS passes keep one 8-plane record pair per lane in
// flight (3.6 KB per wave, 16 waves per CU) and stop at 6.45 TB/s streamed; is that the chip or the depth?
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int U, class V>
__global__ __launch_bounds__(256) void rd(const V *__restrict__ p, size_t n, float *out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        V v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) acc += v[k].x;
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int U, class V>
void run(const void *buf, size_t bytes, int blocks_per_cu, float *out) {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount * blocks_per_cu;
    const size_t n = bytes / sizeof(V);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rd<U, V><<<blocks, 256>>>((const V *)buf, n, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) rd<U, V><<<blocks, 256>>>((const V *)buf, n, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%2zu-byte loads, %d in flight per lane (%5zu B per wave), %2d blocks/CU (%2d waves/CU): %7.1f GB/s\n", sizeof(V), U, sizeof(V) * U * 64,
           blocks_per_cu, blocks_per_cu * 4, 5.0 * bytes / (ms * 1e-3) / 1e9);
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    void *buf;
    float *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    for (int bpc : {4, 5, 8}) {
        run<1, f2>(buf, bytes, bpc, out);
        run<4, f2>(buf, bytes, bpc, out);
        run<8, f2>(buf, bytes, bpc, out);   // = one record pair of the passes: 8 planes x 8 bytes
        run<16, f2>(buf, bytes, bpc, out);  // = two of them
        run<2, f4>(buf, bytes, bpc, out);
        run<4, f4>(buf, bytes, bpc, out);
        run<8, f4>(buf, bytes, bpc, out);
    }
    return 0;
}
