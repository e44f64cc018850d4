// Microbenchmark: what a store that does not fill its 32-byte sectors costs on MI355X (write-through L2).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/partial_writes tools/micro/partial_writes.hip && tools/micro/partial_writes
// Every wave stores runs of 64 consecutive elements of ELEM bytes; the runs tile the buffer; MIS shifts every run start by
// `mis` bytes. Reported: useful GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class T>
__global__ void k_store(T *p, size_t n, int mis_elems) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + mis_elems < n; i += stride) p[i + mis_elems] = (T)i;
}
// holes: every lane whose index has (i & mask) == 0 skips its store
template <class T>
__global__ void k_store_holes(T *p, size_t n, int mask) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) if ((i & mask) != 0) p[i] = (T)i;
}
// column runs: every wave stores one run of 64 elements per column, at c * pitch + k * 64 (k-th run of the column, k fixed per wave
// group): the pattern of a 64-row tile over a column-major image whose column pitch is not a multiple of the line
template <class T>
__global__ void k_store_cols(T *p, int pitch, int cols, int runs_per_col) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int waves = (gridDim.x * blockDim.x) >> 6;
    for (int k = 0; k < runs_per_col; k++)
        for (int c = wave; c < cols; c += waves) p[(size_t)c * pitch + k * 64 + lane] = (T)c;
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 5; r++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    const size_t bytes = 1ull << 30;
    void *buf; hipMalloc(&buf, bytes + 4096);
    const int grid = 256 * 8, block = 256;
    for (int mis : {0, 4, 8, 16, 32, 64, 128}) {
        float ms = timeit([&] { k_store<float><<<grid, block>>>((float *)buf, bytes / 4, mis / 4); });
        printf("float runs of 256 B, start shifted by %2d B: %7.1f GB/s\n", mis, bytes / ms / 1e6);
    }
    for (int mis : {0, 8, 16, 32}) {
        float ms = timeit([&] { k_store<uint8_t><<<grid, block>>>((uint8_t *)buf, bytes / 4, mis); });
        printf("u8 runs of 64 B, start shifted by %2d B: %7.1f GB/s\n", mis, bytes / 4 / ms / 1e6);
    }
    for (int mask : {0x7fffffff, 63, 15, 3}) {
        float ms = timeit([&] { k_store_holes<float><<<grid, block>>>((float *)buf, bytes / 4, mask); });
        printf("float, one lane in %d skipped: %7.1f GB/s\n", mask == 0x7fffffff ? 0 : mask + 1, bytes / ms / 1e6);
    }
    for (int pitch : {256, 240, 224, 208, 192}) {  // float: 64-row runs start at 4 * pitch * c: multiples of 1024 / 960 / 896 / 832 / 768 B
        const int cols = (int)(bytes / 4 / pitch), rpc = 3;
        float ms = timeit([&] { k_store_cols<float><<<grid, block>>>((float *)buf, pitch, cols, rpc); });
        printf("float column runs of 256 B, pitch %d elements (%d B): %7.1f GB/s\n", pitch, pitch * 4, (double)cols * rpc * 256 / ms / 1e6);
    }
    for (int pitch : {256, 240, 248}) {
        const int cols = (int)(bytes / 8 / pitch), rpc = 3;
        float ms = timeit([&] { k_store_cols<long long><<<grid, block>>>((long long *)buf, pitch, cols, rpc); });
        printf("int64 column runs of 512 B, pitch %d elements (%d B): %7.1f GB/s\n", pitch, pitch * 8, (double)cols * rpc * 512 / ms / 1e6);
    }
    return 0;
}
